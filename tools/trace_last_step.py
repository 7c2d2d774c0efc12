"""Print the kernels of the last bench step from a rocprofv3 kernel_trace.csv (in launch order)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
# a step starts at its scatter_dense memset
starts = [i for i, r in enumerate(rows) if "zero_fill_kernel" in r["Kernel_Name"]]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # the bench ends with an eager check step + extras: go back a few
first = starts[-skip] if len(starts) >= skip else max(0, len(rows) - n)
sel = rows[first:starts[-skip + 1]] if len(starts) >= skip and skip > 1 else rows[first:]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
for r in sel:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gx = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    wx = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    print("%8.1f +%7.1f us gap %5.1f grid %8s wg %4s lds %6s vgpr %4s q %s  %s" % (
        (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, gx, wx, r.get("LDS_Block_Size", "?"),
        r.get("VGPR_Count", "?"), r.get("Queue_Id", "?"), name[:100]))
    prev_end = max(prev_end, e)
print("step span %.1f us, %d launches" % ((prev_end - t0) / 1e3, len(sel)))
