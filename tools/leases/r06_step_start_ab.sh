#!/bin/bash
# Round 6: the host side of the training step -- (a) pose gather and rigidity flag queued after the encoder's launches (the flag read
# through pinned memory + an event instead of .item()), Module.train() only when the module is not in training mode; (b) the launch
# wrappers take the stream from torch's raw getter instead of torch.cuda.current_stream() -- against the commit before
# (tools/ab/pyold/*.py swapped in), interleaved in one lease -> gpurun_out/r06/step_start_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O /tmp/pynew; cd $R
for f in train ops train_ops; do cp disconet_amd/$f.py /tmp/pynew/$f.py; done
: > $O/step_start_ab.txt
for rep in 1 2 3; do
  for v in old new; do
    for f in train ops train_ops; do [ $v = old ] && cp tools/ab/pyold/$f.py disconet_amd/$f.py || cp /tmp/pynew/$f.py disconet_amd/$f.py; done
    echo -n "host=$v " >> $O/step_start_ab.txt
    timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], repr(d['loss_last']), d['range_flags'])" >> $O/step_start_ab.txt
  done
done
for f in train ops train_ops; do cp /tmp/pynew/$f.py disconet_amd/$f.py; done
cat $O/step_start_ab.txt
