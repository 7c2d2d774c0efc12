#!/bin/bash
# Is the default line's value sensitive to what ran on the box just before it?  (round 5: 1973 scenes/s right behind two minutes of
# back-to-back training steps, 2295-2410 on every other lease.)  bench first on a fresh box, then training steps, then bench again.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r05heat; mkdir -p $OUT
B="python bench.py --no-alt-math --no-cpu-baseline --train-steps 0 --no-agent-leg --no-voxelize"
smi() { rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "Temperature|Power|sclk|mclk" | head -8; }
smi > $OUT/smi_0.txt
$B > $OUT/bench_1.json 2> $OUT/bench_1.err
smi > $OUT/smi_1.txt
for i in 1 2 3; do python tools/train_step_probe.py --dgrad sp --steps 20 >> $OUT/train.txt 2>> $OUT/train.err; done
smi > $OUT/smi_2.txt
$B > $OUT/bench_2.json 2> $OUT/bench_2.err
smi > $OUT/smi_3.txt
sleep 30
$B > $OUT/bench_3.json 2> $OUT/bench_3.err
python - <<'P'
import json
for i in (1,2,3):
    d=json.load(open('gpurun_out/r05heat/bench_%d.json'%i))
    r=d['roofline']
    print(i, d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_step'], r['other_kernels_ms_per_step'])
P
cat $OUT/smi_*.txt | cut -c1-120
