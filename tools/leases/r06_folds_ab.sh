#!/bin/bash
# Round 6: fold + finish in one launch (statistics, parameter gradients, bias sums: unconditional in this build), the running-statistics
# update inside the statistics' finish launch (DN_BN_FUSED_RUNNING=1, default) and one foreach add for the call counters, against
# DN_BN_FUSED_RUNNING=0; and the previous build (git stash of csrc is not possible on the box: its figure is the last A/B file's,
# profiles/r06_detloss_ab.txt legacy=0 rows) -> gpurun_out/r06/folds_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/folds_ab.txt
for rep in 1 2 3; do
  for m in 0 1; do
    echo -n "fused_running=$m " >> $O/folds_ab.txt
    DN_BN_FUSED_RUNNING=$m timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], d['loss_last'], d['range_flags'])" >> $O/folds_ab.txt
  done
done
cat $O/folds_ab.txt
