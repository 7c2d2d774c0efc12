#!/bin/bash
# last validation of round 5's tree: the whole GPU suite, smoke(), the training step's kernel totals, the default line and the seg line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r05final2; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log )
tail -16 $O/pytest_gpu.log | cut -c1-200
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log ); tail -2 $O/smoke.log | cut -c1-300
for i in 1 2; do timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp 2>> $O/probe.err | cut -c1-100 >> $O/probe.txt; done
timeout 300 python tools/train_step_probe.py --dgrad f32 --wgrad f32 2>> $O/probe.err | cut -c1-100 >> $O/probe.txt
cat $O/probe.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/f2_tr -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp > $O/prof_train.log 2>&1
p=$(find /tmp/f2_tr -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $O/train_step_kernel_stats.csv
cd $R
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
timeout 500 python bench.py --task seg 2> $O/bench_seg.err | tail -1 > $O/bench_seg.json
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json")); r = d["roofline"]
print("default", d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_per_step"], (r.get("rocprof") or {}).get("conv_ms_per_step"), json.dumps(d.get("train_step"))[:700])
s = json.load(open("$O/bench_seg.json")); print("seg", s["value"], s["ms_per_step"], s["config"]["launch"], json.dumps(s.get("train_step"))[:300])
PY
