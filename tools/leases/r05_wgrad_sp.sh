#!/bin/bash
# round 5: the split-f16 weight gradient (dn_conv_wgrad_sp; TrainEngine(wgrad_math="sp")) -- unit + step tests, A/B of the
# training step in one lease (interleaved), per-kernel totals of the f32 and sp forms.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r05wg; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -m gpu -k "wgrad" -x > $OUT/pytest_ops.log 2>&1
echo "pytest ops rc $?" >> $OUT/pytest_ops.log; tail -30 $OUT/pytest_ops.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_train_step.py -q -m gpu -k "split_f16 or train_step_matches" > $OUT/pytest_step.log 2>&1
echo "pytest step rc $?" >> $OUT/pytest_step.log; tail -15 $OUT/pytest_step.log | cut -c1-600
for i in 1 2; do
  for w in f32 sp; do timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad $w >> $OUT/ab.txt 2>> $OUT/ab.err; done
done
cut -c1-140 $OUT/ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg_sp -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp > $OUT/prof_sp.log 2>&1
p=$(find /tmp/wg_sp -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/train_step_kernel_stats_sp_sp.csv
head -12 $OUT/train_step_kernel_stats_sp_sp.csv | cut -c1-160
