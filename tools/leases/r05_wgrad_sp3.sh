#!/bin/bash
# round 5: stride-2 weight gradients on the split-f16 kernel + the four-wave slice sum -- unit tests, two step tests, the step timed alone
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r05wg3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -m gpu -k "wgrad" > $OUT/pytest_ops.log 2>&1
echo "pytest ops rc $?" >> $OUT/pytest_ops.log; tail -25 $OUT/pytest_ops.log | cut -c1-700
timeout 600 python -m pytest tests/test_gpu_train_step.py -q -m gpu -k "split_f16 and cfg1" > $OUT/pytest_step.log 2>&1
echo "pytest step rc $?" >> $OUT/pytest_step.log; tail -6 $OUT/pytest_step.log | cut -c1-400
for i in 1 2; do timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp >> $OUT/ab.txt 2>> $OUT/ab.err; done
cut -c1-120 $OUT/ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg_sp -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp > $OUT/prof_sp.log 2>&1
p=$(find /tmp/wg_sp -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/train_step_kernel_stats_sp_sp.csv
grep -E "wgrad" $OUT/train_step_kernel_stats_sp_sp.csv | cut -c1-150
