#!/bin/bash
# Round 6: every known weight form of the step packed by ONE launch at the start of the forward (dn_spconv_pack_weights_multi,
# TrainEngine._pack_multi) against the per-layer launches (DN_TRAIN_PACK_MULTI=0): byte-for-byte / bit-for-bit tests, then the
# training step interleaved in one lease -> gpurun_out/r06/pack_multi_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py -q -m gpu -k "one_launch" 2>&1 | tail -3 > $O/pack_multi_ab.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "multi=$v " >> $O/pack_multi_ab.txt
    DN_TRAIN_PACK_MULTI=$v timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], repr(d['loss_last']), d['range_flags'])" >> $O/pack_multi_ab.txt
  done
done
cat $O/pack_multi_ab.txt
