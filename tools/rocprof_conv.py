"""Conv time per step from a rocprofv3 kernel trace of the DEFAULT bench command (graph replay): the sum of
the conv kernel's durations over the last `steps` steps / steps.  bench.py reports it beside its own HIP-event
figure (roofline.rocprof).     python tools/rocprof_conv.py kernel_trace.csv conv_sp_kernel,conv_spq_kernel 20 [round] [dn_version]
"""
import csv
import json
import sys


def main(path, kernel, steps, rnd=3, version=None):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at the voxel grid's zero fill
    starts = [i for i, r in enumerate(rows) if "zero_fill_kernel" in r["Kernel_Name"]]
    assert len(starts) >= steps, (len(starts), steps)
    sel = rows[starts[-steps]:]
    conv = [r for r in sel if any(k in r["Kernel_Name"] for k in kernel.split(","))]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in conv]
    # a K-sliced layer is two kernel launches (main + fix-up pass, the same instantiation back to back): one LAYER launch
    def ksl(r):
        kn = r["Kernel_Name"]
        a = kn.split("<")[1].split(">")[0].replace(" ", "").split(",") if "<" in kn else []
        return ("conv_spq_kernel" in kn and len(a) >= 4 and a[3] == "1") or ("conv_sp_kernel" in kn and len(a) >= 17 and a[16] == "1")
    fixups = sum(1 for p, q in zip(conv, conv[1:]) if ksl(q) and p["Kernel_Name"] == q["Kernel_Name"])
    names = sorted({r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in conv})
    print(json.dumps({"round": rnd, "kernel": kernel, "steps": steps, "launches_per_step": (len(conv) - fixups) / steps, "kernel_launches_per_step": len(conv) / steps,
                      "dn_version": version, "kernel_instantiations": names,
                      "conv_ms_per_step": sum(dur) / steps / 1e3, "avg_launch_us": sum(dur) / len(dur),
                      "all_kernels_ms_per_step": sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in sel) / steps / 1e6,
                      "note": "rocprofv3 --kernel-trace over `python bench.py` (hipGraph replay, one stream): kernel "
                              "durations of the last %d steps" % steps}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 3,
         int(sys.argv[5]) if len(sys.argv) > 5 else None)
