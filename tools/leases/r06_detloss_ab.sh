#!/bin/bash
# Round 6: the detection loss on flat float4 streams (default) against the per-anchor kernel (DN_DET_LOSS_LEGACY=1), the step timed
# alone, interleaved in ONE lease -> gpurun_out/r06/detloss_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/detloss_ab.txt
for rep in 1 2 3; do
  for m in 1 0; do
    echo -n "legacy=$m " >> $O/detloss_ab.txt
    DN_DET_LOSS_LEGACY=$m timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], d['loss_last'], d['range_flags'])" >> $O/detloss_ab.txt
  done
done
cat $O/detloss_ab.txt
