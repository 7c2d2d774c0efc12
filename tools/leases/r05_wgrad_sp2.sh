#!/bin/bash
# round 5: ReLU byte mask in the BatchNorm backward, the 13-channel layer on the split-f16 weight gradient, the stream-ordered
# range check of the training step -- unit + step tests, the training step timed alone and inside bench.py (one lease).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r05wg2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -m gpu -k "wgrad or bn_" -x > $OUT/pytest_ops.log 2>&1
echo "pytest ops rc $?" >> $OUT/pytest_ops.log; tail -30 $OUT/pytest_ops.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_seg_train.py tests/test_gpu_sharded_train.py -q -m gpu > $OUT/pytest_step.log 2>&1
echo "pytest step rc $?" >> $OUT/pytest_step.log; tail -15 $OUT/pytest_step.log | cut -c1-600
for i in 1 2; do
  timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp >> $OUT/ab.txt 2>> $OUT/ab.err
done
cut -c1-140 $OUT/ab.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r05wg2/bench_default.json'))
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d.get('train_step'))[:900])
P
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg_sp -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp > $OUT/prof_sp.log 2>&1
p=$(find /tmp/wg_sp -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/train_step_kernel_stats_sp_sp.csv
head -8 $OUT/train_step_kernel_stats_sp_sp.csv | cut -c1-160
