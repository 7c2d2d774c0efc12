"""Per-layer table of one bench step from a rocprofv3 kernel trace of the default command (graph replay):
launch order -> layer name, duration, algorithmic GFLOP (true channel counts), TFLOP/s.
    python tools/layers_from_trace.py gpurun_out/r03prof/kernel_trace.csv > profiles/r03_bench_layers.txt"""
import csv
import sys

N = 20                                        # 4 scenes x 5 agents
LAYERS = [  # name, out pixels per image, cin, cout, k   (launch order of DiscoNet.forward, conv_math sp)
    ("conv_pre_1", 256 * 256, 13, 32, 3), ("conv_pre_2", 256 * 256, 32, 32, 3), ("conv1_1 (s2)", 128 * 128, 32, 64, 3),
    ("conv1_2 + Conv3D 1x1", 128 * 128, 64, 64, 3), ("conv2_1 (s2)", 64 * 64, 64, 128, 3), ("conv2_2", 64 * 64, 128, 128, 3),
    ("conv3d_2 (1x1)", 64 * 64, 128, 128, 1), ("conv3_1 (s2)", 32 * 32, 128, 256, 3), ("conv3_2", 32 * 32, 256, 256, 3),
    ("conv4_1 (s2)", 16 * 16, 256, 512, 3), ("conv4_2", 16 * 16, 512, 512, 3),
    ("conv5_1 (up+cat)", 32 * 32, 768, 256, 3), ("conv5_2", 32 * 32, 256, 256, 3), ("conv6_1 (up+cat)", 64 * 64, 384, 128, 3),
    ("conv6_2", 64 * 64, 128, 128, 3), ("conv7_1 (up+cat)", 128 * 128, 192, 64, 3), ("conv7_2", 128 * 128, 64, 64, 3),
    ("conv8_1 (up+cat)", 256 * 256, 96, 32, 3), ("conv8_2", 256 * 256, 32, 32, 3), ("heads (3x3 + block-diag 1x1)", 256 * 256, 32, 64, 3),
]
EXTRA = {3: 2.0 * N * 128 * 128 * 64 * 64, 19: 2.0 * N * 256 * 256 * (32 * 12 + 32 * 36)}   # the fused 1x1 stages

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "zero_fill_kernel" in r["Kernel_Name"]]
skip = 4                                       # the bench ends with an eager check step and extras: go back a few
sel = rows[starts[-skip]:starts[-skip + 1]]
def is_conv(r):
    return any(k in r["Kernel_Name"] for k in ("conv_sp_kernel", "conv_spq_kernel", "conv_pre_pair_kernel"))


if any("conv_pre_pair_kernel" in r["Kernel_Name"] for r in sel):      # the stem's two layers in one launch (round 4)
    LAYERS = [("conv_pre_1 + conv_pre_2", 256 * 256, 0, 0, 3)] + LAYERS[2:]
    EXTRA = {2: EXTRA[3], 18: EXTRA[19], 0: 2.0 * N * 256 * 256 * 9 * (13 * 32 + 32 * 32)}


def is_ksl(r):      # K-sliced launch (template flag KSL = 1): may be followed by its fix-up pass, the same kernel again
    kn = r["Kernel_Name"]
    args = kn.split("<")[1].split(">")[0].replace(" ", "").split(",") if "<" in kn else []
    return ("conv_spq_kernel" in kn and len(args) >= 4 and args[3] == "1") or ("conv_sp_kernel" in kn and len(args) >= 17 and args[16] == "1")


conv = []
for r in sel:
    if not is_conv(r):
        continue
    if conv and is_ksl(r) and conv[-1]["Kernel_Name"] == r["Kernel_Name"] and not conv[-1].get("_fixup"):
        m = dict(conv[-1])          # main launch + fix-up pass of one layer: one row, durations added
        m["End_Timestamp"] = str(int(m["End_Timestamp"]) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        m["_fixup"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        conv[-1] = m
    else:
        conv.append(dict(r))
assert len(conv) == len(LAYERS), (len(conv), len(LAYERS))
print("%-30s %9s %10s %9s   %s" % ("layer", "us", "GFLOP", "TFLOP/s", "kernel configuration <KS,S,TH,TW,BN,TG,CA,WM,WN,WTM,WTN,POST,ABL,BSTAT,UPM,AHI>"))
tot_us = tot_gf = 0.0
for i, (r, (name, px, cin, cout, k)) in enumerate(zip(conv, LAYERS)):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gf = (2.0 * N * px * cout * cin * k * k + EXTRA.get(i, 0.0)) / 1e9
    kn = r["Kernel_Name"]
    cfg = ("quad-merged BN=" + kn.split("conv_spq_kernel<")[1].split(">")[0]) if "conv_spq_kernel" in kn else \
        "one launch: 16 x 32 tile, 8 waves, intermediate map in LDS" if "conv_pre_pair_kernel" in kn else \
        "<" + kn.split("conv_sp_kernel<")[1].split(">")[0].replace(" ", "") + ">"
    if r.get("_fixup"):
        cfg += "  K-sliced: %.1f us of it the fix-up pass" % r["_fixup"]
    print("%-30s %9.1f %10.2f %9.1f   %s" % (name, us, gf, gf / (us * 1e-6) / 1e3, cfg))
    tot_us += us
    tot_gf += gf
print("%-30s %9.1f %10.2f %9.1f" % ("all conv launches", tot_us, tot_gf, tot_gf / (tot_us * 1e-6) / 1e3))
for r in sel:
    if not is_conv(r):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        nm = nm.split("<")[0].split("(")[0] if not nm.startswith("at::") else "torch elementwise (num_agent cast)"
        print("%-30s %9.1f" % (nm[-30:], us))
print("step span %.1f us" % ((int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3))
