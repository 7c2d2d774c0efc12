#!/bin/bash
# kernel trace of one rank's share of the 8-rank agent-sharded step (emulated on one GPU): which launches make up its 0.6 ms
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05share; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sh_kt -o kt -- python $R/bench.py --mode agent --no-pg --emulate-world 8 --steps 10 --warmup 2 > $O/kt.log 2>&1
p=$(find /tmp/sh_kt -name "*kernel_trace.csv" | head -1); [ -n "$p" ] && cp "$p" $O/kernel_trace.csv
python3 - <<PY
import csv, re
rows = sorted(csv.DictReader(open("$O/kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
# the emulated share replays graph A, a copy, graph B: take the last 40 product kernels and print them
prod = [r for r in rows if not any(k in r["Kernel_Name"] for k in ("at::native", "elementwise", "sp_range_collect"))]
last = prod[-60:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f +%7.1f us grid %7s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X") or r.get("Grid_Size"), n))
PY
