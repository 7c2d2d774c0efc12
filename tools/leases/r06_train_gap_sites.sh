#!/bin/bash
# where is the GPU idle inside a training step?  rocprofv3 kernel trace of tools/train_step_probe.py; for the last 3 steps the gaps
# above 8 us with the kernels on either side, and the gap between two steps -> gpurun_out/r06/train_gap_sites.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_gaps
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_gaps -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp --steps 8 > $O/train_gaps_probe.log 2>&1
f=$(find /tmp/rp_gaps -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/train_gap_sites.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
for a, b in zip(idx[-4:-1], idx[-3:]):
    seg = rows[a:b + 1]          # from the previous step's Adam launch to this step's
    t0, t1 = int(seg[0]["End_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg[1:])
    print("step (Adam end to Adam end): %.3f ms, busy %.3f ms, idle %.3f ms" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    for i in range(len(seg) - 1):
        g = int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])
        if g > 8000:
            print("   %7.1f us at kernel %3d: %s -> %s" % (g / 1e3, i, short(seg[i]["Kernel_Name"]), short(seg[i + 1]["Kernel_Name"])))
PY
cat $O/train_gap_sites.txt
