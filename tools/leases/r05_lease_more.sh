#!/bin/bash
# one more lease of the default bench line on the final build (the round's median is taken over all of them) [+ the seg line]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
T=${1:-a}; O=$R/gpurun_out/r05lease_$T; mkdir -p $O
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
[ "$2" = seg ] && ( timeout 500 python bench.py --task seg 2> $O/bench_seg.err | tail -1 > $O/bench_seg.json; tail -2 $O/bench_seg.err | cut -c1-300 )
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["repeat"]["scenes_per_s"], d.get("train_step", {}).get("ms_per_step"), d.get("train_step", {}).get("step_ms"))
try:
    s = json.load(open("$O/bench_seg.json")); print("seg", s["value"], s["ms_per_step"], s["config"]["launch"], json.dumps(s.get("train_step"))[:400])
except Exception as e:
    print("seg -", e)
PY
