#!/bin/bash
# Round 6: dn_warp_backward's gathers with one candidate pixel per lane against the scalar candidate loops (DN_WARP_GATHER_LEGACY=1):
# parity (autograd, bit for bit against the scalar kernels), then the training step interleaved in one lease
# -> gpurun_out/r06/warp_gather_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train_ops.py -q -m gpu -k "warp" 2>&1 | tail -3 > $O/warp_gather_ab.txt
for rep in 1 2 3; do
  for v in 1 0; do
    echo -n "legacy=$v " >> $O/warp_gather_ab.txt
    DN_WARP_GATHER_LEGACY=$v timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], repr(d['loss_last']), d['range_flags'])" >> $O/warp_gather_ab.txt
  done
done
cat $O/warp_gather_ab.txt
