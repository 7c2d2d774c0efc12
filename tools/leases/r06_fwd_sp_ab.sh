#!/bin/bash
# Round 6: the training forward on the SP engine (DISCONET_FWD_MATH=sp, the default) against rounds 2-5's fp32-NHWC engine
# (DISCONET_FWD_MATH=nhwc), the step timed alone, interleaved in ONE lease -> gpurun_out/r06/fwd_sp_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/fwd_sp_ab.txt
for rep in 1 2 3; do
  for m in nhwc sp; do
    echo -n "$m " >> $O/fwd_sp_ab.txt
    DISCONET_FWD_MATH=$m timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], d['loss_last'], d['range_flags'])" >> $O/fwd_sp_ab.txt
  done
done
cat $O/fwd_sp_ab.txt
