#!/bin/bash
# Round 6: the conv bias gradients fused into the BatchNorm backward's apply launch (DN_BN_FUSED_BIAS=1) against a
# dn_channel_sum pass per layer (=0), the step timed alone, interleaved in ONE lease -> gpurun_out/r06/bias_ab.txt
# DN_BN_BIAS_BLOCKS = workgroups of the fused launch (= rows of partials the fold walks)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/bias_ab.txt
for rep in 1 2 3; do
  for m in "0 2048" "1 512" "1 1024" "1 2048" "1 4096"; do
    set -- $m
    echo -n "fused_bias=$1 blocks=$2 " >> $O/bias_ab.txt
    DN_BN_FUSED_BIAS=$1 DN_BN_BIAS_BLOCKS=$2 timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], d['loss_last'], d['range_flags'])" >> $O/bias_ab.txt
  done
done
cat $O/bias_ab.txt
