#!/bin/bash
# how much of the training step's wall time is the GPU idle between kernels?  rocprofv3 kernel trace of tools/train_step_probe.py,
# busy = sum of kernel durations / (end of the last kernel - start of the first) over the last 5 steps -> gpurun_out/r06/train_gaps.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_gaps
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_gaps -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp --steps 8 > $O/train_gaps_probe.log 2>&1
f=$(find /tmp/rp_gaps -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/train_gaps.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps end with adam_kernel
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
print("kernels", len(rows), "steps", len(idx))
for a, b in zip(idx[-6:-1], idx[-5:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = sorted((int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) for i in range(len(seg) - 1))
    big = sum(g for g in gaps if g > 20000)
    print("step: %d kernels, span %.3f ms, busy %.3f ms (%.1f %%), idle %.3f ms; gaps > 20 us sum %.3f ms; median gap %.2f us"
          % (len(seg), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6, big / 1e6, gaps[len(gaps) // 2] / 1e3))
PY
cat $O/train_gaps.txt
