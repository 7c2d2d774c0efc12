#!/bin/bash
# per-channel reductions: 2048 / 1024 / 512 workgroups (DN_REDUCE_BLOCKS) -- the step timed alone + per-kernel totals, one lease
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r05rb; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do for b in 2048 1024 512; do
  echo -n "$b " >> $OUT/ab.txt; DN_REDUCE_BLOCKS=$b timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp 2>> $OUT/ab.err | cut -c1-90 >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
cd /tmp
for b in 2048 1024 512; do
  DN_REDUCE_BLOCKS=$b timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rb_$b -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp > $OUT/prof_$b.log 2>&1
  p=$(find /tmp/rb_$b -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/stats_$b.csv
  echo "== $b"; grep -E "fold_partials|bn_bwd_reduce_v4|bn_stats_v4|channel_sum_v4|bn_bwd_apply" $OUT/stats_$b.csv | cut -d, -f1-4 | cut -c1-60,150-
done
