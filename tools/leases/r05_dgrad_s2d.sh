#!/bin/bash
# round 5: the stride-2 data gradients as ONE split-f16 launch per layer (space-to-depth output, read in place by the next BatchNorm
# backward) -- unit tests, step tests, the step timed alone against the four-launch fp32 form (DN_DGRAD_S2D=0), per-kernel totals
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r05s2d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -m gpu -k "bn_ or dgrad" > $OUT/pytest_ops.log 2>&1
echo "pytest ops rc $?" >> $OUT/pytest_ops.log; tail -12 $OUT/pytest_ops.log | cut -c1-500
timeout 600 python -m pytest tests/test_gpu_train_step.py -q -m gpu -k "split_f16 or (train_step_matches and cfg1)" > $OUT/pytest_step.log 2>&1
echo "pytest step rc $?" >> $OUT/pytest_step.log; tail -8 $OUT/pytest_step.log | cut -c1-500
for i in 1 2; do
  echo -n "s2d=0 " >> $OUT/ab.txt; DN_DGRAD_S2D=0 timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp 2>> $OUT/ab.err | cut -c1-90 >> $OUT/ab.txt
  echo -n "s2d=1 " >> $OUT/ab.txt; timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp 2>> $OUT/ab.err | cut -c1-90 >> $OUT/ab.txt
done
cat $OUT/ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s2d -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp > $OUT/prof.log 2>&1
p=$(find /tmp/s2d -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/train_step_kernel_stats.csv
head -14 $OUT/train_step_kernel_stats.csv | cut -d, -f1-5 | cut -c1-70,140-
