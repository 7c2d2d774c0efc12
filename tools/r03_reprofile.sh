#!/bin/bash
# profile set + the default / seg bench lines only (tools/r03_final.sh without the suite and the agent sweep)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03final
mkdir -p $O
cd $R
bash tools/r03_profile.sh > $O/profile.log 2>&1
P=$R/gpurun_out/r03prof
cp $P/rocprof_conv_sp.json profiles/r03_rocprof_conv_sp.json
cp $P/pmc_traffic_sp.json profiles/r03_pmc_traffic_sp.json
[ -s $P/pmc_traffic_seg.json ] && cp $P/pmc_traffic_seg.json profiles/r03_pmc_traffic_seg.json
cd $R
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
timeout 500 python bench.py --task seg --train-steps 4 2> $O/bench_seg.err | tail -1 > $O/bench_seg.json
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("rocprof", {}).get("conv_ms_per_step"), d.get("train_step", {}).get("ms_per_step"))
s = json.load(open("$O/bench_seg.json"))
print("seg", s["value"], s["ms_per_step"], s["roofline"]["frac"], s.get("train_step", {}).get("ms_per_step"))
PY
tail -3 $P/bench_layers.txt
