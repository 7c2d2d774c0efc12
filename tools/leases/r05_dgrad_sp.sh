#!/bin/bash
# round 5: the split-f16 data gradient (TrainEngine(dgrad_math="sp")) -- unit + step tests, A/B of the training step in one
# lease (interleaved), per-kernel totals of both forms, and the default bench line (the conv epilogue gained a uniform branch).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r05dg; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py -q -m gpu -k "sp_copy or split_f16 or train_step_matches or dgrad" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for i in 1 2 3; do
  for m in f32 sp; do timeout 300 python tools/train_step_probe.py --dgrad $m >> $OUT/ab.txt 2>> $OUT/ab.err; done
done
cat $OUT/ab.txt | cut -c1-120
cd /tmp
for m in f32 sp; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dg_$m -o t -- python $R/tools/train_step_probe.py --dgrad $m > $OUT/prof_$m.log 2>&1
  p=$(find /tmp/dg_$m -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/train_step_kernel_stats_$m.csv
done
cd $R
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r05dg/bench_default.json'))
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('train_step'))
P
