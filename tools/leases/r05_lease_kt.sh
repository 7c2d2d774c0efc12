#!/bin/bash
# kernel trace (rocprofv3 --kernel-trace --stats) of the default bench command AND the default line in the SAME lease, so that the
# line's live HIP-event conv time and the committed rocprofv3 summary are of one box (leases differ by up to 8 %).
R=${GRAFT_REPO_ROOT:-/root/repo}; RD=r05; O=$R/gpurun_out/${RD}kt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
V=$(python3 -c "import sys; sys.path.insert(0, '$R'); from disconet_amd import _lib; print(_lib.load().dn_version())")
B="python $R/bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-kernel-events --no-agent-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${RD}_kt -o kt -- $B > $O/kt.log 2>&1
for f in kernel_stats kernel_trace; do p=$(find /tmp/${RD}_kt -name "*${f}.csv" | head -1); [ -n "$p" ] && cp "$p" $O/$f.csv; done
python3 $R/tools/trace_last_step.py $O/kernel_trace.csv > $O/step_timeline.txt 2>&1
python3 $R/tools/layers_from_trace.py $O/kernel_trace.csv > $O/bench_layers.txt 2>&1
python3 $R/tools/rocprof_conv.py $O/kernel_trace.csv conv_sp_kernel,conv_spq_kernel,conv_pre_pair_kernel 19 5 $V > $O/rocprof_conv_sp.json 2> $O/rocprof_conv.err
cp $O/rocprof_conv_sp.json $R/profiles/${RD}_rocprof_conv_sp.json
cd $R
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json")); r = d["roofline"]
print("default", d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_per_step"], r.get("rocprof"), d.get("train_step", {}).get("ms_per_step"))
PY
tail -8 $O/bench_layers.txt
