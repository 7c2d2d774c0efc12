"""Generates DESIGN.md and README.md from tools/templates/{DESIGN,README}.tmpl.md: fills their @@PLACEHOLDERS@@ and the per-layer table from the round's final GPU run (tools/final_set.sh r06 ->
gpurun_out/r06final, gpurun_out/r06prof) and copies that run's summaries into profiles/r06_*.  Run once, at the end:
    python tools/fill_design.py [--dry] [--ncpu N]      (N = tests of the CPU suite that passed)"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = "r06"
F, P = os.path.join(ROOT, "gpurun_out", RD + "final"), os.path.join(ROOT, "gpurun_out", RD + "prof")
# (label of tools/layers_from_trace.py, short name, shape, executed MFMA work / algorithmic: 3 products per multiply; 2 on conv_pre_1's
#  hi-only operand; 4 of 9 taps on the upsampled 2/3 of the tap-merged layers' channels)
LAYERS = [
    ("conv_pre_1 + conv_pre_2", "stem pair", "13→32→32 @256², one launch", (2 * 9.80 + 3 * 24.16) / 33.96),
    ("conv1_1 (s2)", "conv1_1", "32→64 s2 → 128²", 3.0),
    ("conv1_2 + Conv3D 1x1", "conv1_2+Conv3D", "64→64 (+1×1) @128²", 3.0),
    ("conv2_1 (s2)", "conv2_1", "64→128 s2 → 64²", 3.0),
    ("conv2_2", "conv2_2", "128→128 @64²", 3.0),
    ("conv3d_2 (1x1)", "conv3d_2", "128→128 1×1 @64²", 3.0),
    ("conv3_1 (s2)", "conv3_1", "128→256 s2 → 32²", 3.0),
    ("conv3_2", "conv3_2", "256→256 @32² (+ fp32 copy)", 3.0),
    ("conv4_1 (s2)", "conv4_1", "256→512 s2 → 16²", 3.0),
    ("conv4_2", "conv4_2", "512→512 @16²", 3.0),
    ("conv5_1 (up+cat)", "conv5_1", "768→256 @32², tap-merged, 4 K slices", 3.0 * 17 / 27),
    ("conv5_2", "conv5_2", "256→256 @32²", 3.0),
    ("conv6_1 (up+cat)", "conv6_1", "384→128 @64², tap-merged", 3.0 * 17 / 27),
    ("conv6_2", "conv6_2", "128→128 @64²", 3.0),
    ("conv7_1 (up+cat)", "conv7_1", "192→64 @128², tap-merged", 3.0 * 17 / 27),
    ("conv7_2", "conv7_2", "64→64 @128²", 3.0),
    ("conv8_1 (up+cat)", "conv8_1", "96→32 @256², tap-merged", 3.0 * 17 / 27),
    ("conv8_2", "conv8_2", "32→32 @256²", 3.0),
    ("heads (3x3 + block-diag 1x1)", "heads", "32→64→(12, 36) @256², fp32 out", 3.0),
]


def j(path):
    return json.load(open(path))


FROM_PROFILES = False


def srcpath(kind, name):
    """kind "F" (final_set outputs) / "P" (profile_set outputs) -> path; --from-profiles: the committed copy under profiles/"""
    if FROM_PROFILES:
        m = {"bench_default.json": "_bench_default.json", "bench_seg.json": "_bench_seg.json", "pytest_gpu.log": "_pytest_gpu.txt",
             "train_groups.json": "_train_groups.json", "bench_layers.txt": "_bench_layers.txt"}
        return os.path.join(ROOT, "profiles", RD + m[name])
    return os.path.join(F if kind == "F" else P, name)


def pmc_rows():
    if FROM_PROFILES:
        out = open(os.path.join(ROOT, "profiles", RD + "_pmc_conv_sp.txt")).read()
    else:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py"), P, "24"], capture_output=True, text=True, check=True).stdout
    return [l.split() for l in out.splitlines()[1:]]


def layer_table():
    names = {l[0] for l in LAYERS}
    layers = {}
    for line in open(srcpath("P", "bench_layers.txt")):
        m = re.match(r"^(\S.*?)\s{2,}([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(.*)$", line.rstrip())
        if m and m.group(1) in names:
            layers[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))      # us, GFLOP, TFLOP/s
    rows = pmc_rows()
    conv = [r for r in rows if r[0].startswith("sp<") or r[0].startswith("spq<") or r[0].startswith("conv_pre_pair")]
    merged = []
    for r in conv:       # the K-sliced layer's fix-up launch belongs to the row before it
        name = " ".join(r[:-9])
        vals = [float(x) for x in r[-9:]]      # blocks us GHz mfma wInst wAny act fetch write
        if merged and merged[-1][0] == name and vals[0] < 512 and "1>" in name.replace(" ", ""):
            merged[-1][1][7] += vals[7]; merged[-1][1][8] += vals[8]
            continue
        merged.append([name, vals])
    assert len(merged) == len(LAYERS) == 19 and len(layers) == 19, (len(merged), len(layers))
    lines = ["| layer | shape (20 images) | µs | floor µs | % of floor | MFMA busy | busy × GHz ÷ 2.4 | wAny | HBM r + w MB |", "|---|---|---|---|---|---|---|---|---|"]
    tot_us = tot_floor = 0.0
    for (label, short, shape, factor), (_, v) in zip(LAYERS, merged):
        us, gflop, _ = layers[label]
        # floor in us: executed GFLOP at 2.5 PFLOP/s (= 2.5e3 GFLOP per ms = 2.5 GFLOP per us) or PMC megabytes at 6.3 TB/s (= 6.3e3 MB per ms = 6.3 MB per us)
        floor = max(gflop * factor / 2.5, (v[7] + v[8]) / 6.3)
        tot_us += us; tot_floor += floor
        lines.append("| %s | %s | %.0f | %.0f | %.0f | %.0f %% | %.2f | %.0f %% | %.0f + %.0f |" % (
            short, shape, us, floor, 100 * floor / us, v[3], v[3] / 100 * v[2] / 2.4, v[5], v[7], v[8]))
    lines.append("| **all 19 launches** | | **%.0f** | **%.0f** | **%.0f** | | | | |" % (tot_us, tot_floor, 100 * tot_floor / tot_us))
    return "\n".join(lines)


def main(dry, ncpu):
    d = j(srcpath("F", "bench_default.json"))
    seg = j(srcpath("F", "bench_seg.json"))
    tg = j(srcpath("P", "train_groups.json"))
    r, c, a, t = d["roofline"], d["cpu_baseline"], d["alt_math"], d["train_step"]
    es = d["agent_sharded"]["emulated_share"]
    es16 = d["agent_sharded_batch16"]["emulated_share"]
    other = r["other_kernels_ms_per_step"]
    gpu_log = open(srcpath("F", "pytest_gpu.log")).read()
    m = re.search(r"(\d+) passed", gpu_log)
    ab = [l.split() for l in open(os.path.join(ROOT, "profiles", "r06_bias_ab.txt")) if l.startswith("fused_bias")]
    med = lambda xs: sorted(xs)[len(xs) // 2]
    b0 = med([float(x[2]) for x in ab if x[0] == "fused_bias=0"])
    b1 = med([float(x[2]) for x in ab if x[0] == "fused_bias=1" and x[1] == "blocks=1024"])
    g = lambda k: tg.get(k, {"ms_per_step": 0.0})["ms_per_step"]
    tj = os.path.join(ROOT, "gpurun_out", "r06_trajectory.json")
    traj = j(tj if (not FROM_PROFILES and os.path.exists(tj)) else os.path.join(ROOT, "profiles", "r06_trajectory.json"))
    pc = lambda x: "%.1f %%" % (100.0 * x)
    v = {
        "VALUE": "%.0f" % d["value"], "MS": "%.3f" % d["ms_per_step"], "CONVMS": "%.3f" % r["kernel_ms_per_step"],
        "FRAC": "%.3f" % r["frac"], "FRACX": "%.2f" % r["frac_executed"],
        "ALTV": "%.0f" % a["value"], "ALTFRAC": "%.2f" % a["roofline"]["frac"],
        "CPUV": "%.2f" % c["value"], "CPUC": "%d" % c["cores"],
        "TRAINMS": "%.1f" % t["ms_per_step"], "TRAINF32": "%.1f" % t["all_gradients_f32"]["ms_per_step"],
        "TRAINKD": "%.1f" % t["with_kd"]["ms_per_step"],
        "AGENTMS": "%.2f" % d["agent_sharded"]["ms_per_step"], "SHAREMS": "%.3f" % es["ms_per_step"],
        "PROJ": "%.2f" % es["projected_speedup"], "PROJ16": "%.2f" % es16["projected_speedup"],
        "PCLS": "%.1e" % c["parity_max_abs_err"]["cls"], "PLOC": "%.1e" % c["parity_max_abs_err"]["loc"],
        "NCPU": str(ncpu), "NGPU": m.group(1) if m else "?",
        "VOXUS": "%.0f" % d["voxelize"]["us_per_cloud"],
        "WARPUS": "%.0f" % (1e3 * other.get("warp", 0.0)), "FUSEUS": "%.0f" % (1e3 * other.get("fuse_mlp", 0.0)),
        "SEGV": "%.0f" % seg["value"], "SEGFRAC": "%.3f" % seg["roofline"]["frac"],
        "T_CONVSP": "%.2f" % g("conv_engine_sp"), "T_CONVNHWC": "%.2f" % g("conv_engine_nhwc"), "T_WGRAD": "%.2f" % g("wgrad"),
        "T_BNF": "%.2f" % g("bn_forward"), "T_BNB": "%.2f" % g("bn_backward"), "T_FOLDS": "%.2f" % g("folds_and_bias_sums"),
        "T_PACK": "%.2f" % (g("packing") + g("copies")),
        "T_MISC": "%.2f" % g("other (loss, warp, combine, Adam, torch elementwise)"), "T_TOTAL": "%.2f" % tg["total_ms_per_step"],
        "BIASAB": "%.2f → %.2f ms per step, medians of three interleaved runs" % (b0, b1),
        "TRAJ_FIRST40": pc(traj["first40_sp_vs_f32"]), "TRAJ_MAX": pc(traj["max_rel_sp_vs_f32"]),
        "TRAJ_FLOOR": pc(traj["max_rel_f32_nhwc_vs_f32_sp (floor)"]),
        "LAYER_TABLE": layer_table(),
    }
    # the documents are generated from tools/templates/*.tmpl.md (the same text with @@PLACEHOLDERS@@): edit the templates, re-run
    tdir = os.path.join(ROOT, "tools", "templates")
    text = open(os.path.join(tdir, "DESIGN.tmpl.md")).read()
    missing = sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", text)) - set(v))
    assert not missing, missing
    if dry:
        print(v["LAYER_TABLE"])
        print(json.dumps({k: x for k, x in v.items() if k != "LAYER_TABLE"}, indent=1))
        return
    for k, val in v.items():
        text = text.replace("@@%s@@" % k, val)
    open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
    readme = open(os.path.join(tdir, "README.tmpl.md")).read()      # the same placeholders in the README's status paragraph
    for k, val in v.items():
        readme = readme.replace("@@%s@@" % k, val)
    assert "@@" not in readme, sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", readme)))
    open(os.path.join(ROOT, "README.md"), "w").write(readme)
    if FROM_PROFILES:
        return
    cp = [(os.path.join(F, "bench_default.json"), RD + "_bench_default.json"), (os.path.join(F, "bench_seg.json"), RD + "_bench_seg.json"),
          (os.path.join(F, "agent_share.json"), RD + "_agent_share.json"), (os.path.join(F, "guard_bands.txt"), RD + "_guard_bands.txt"),
          (os.path.join(F, "pytest_gpu.log"), RD + "_pytest_gpu.txt"),
          (os.path.join(P, "kernel_stats.csv"), RD + "_bench_kernel_stats.csv"),
          (os.path.join(P, "kernel_trace.csv"), RD + "_bench_kernel_trace.csv"), (os.path.join(P, "bench_layers.txt"), RD + "_bench_layers.txt"),
          (os.path.join(P, "step_timeline.txt"), RD + "_step_timeline.txt"), (os.path.join(P, "rocprof_conv_sp.json"), RD + "_rocprof_conv_sp.json"),
          (os.path.join(P, "pmc_traffic_sp.json"), RD + "_pmc_traffic_sp.json"), (os.path.join(P, "pmc_traffic_seg.json"), RD + "_pmc_traffic_seg.json"),
          (os.path.join(P, "train_step_kernel_stats.csv"), RD + "_train_step_kernel_stats.csv"),
          (os.path.join(P, "train_groups.json"), RD + "_train_groups.json")]
    cp += [(os.path.join(F, "agent_share_b%d.json" % b), RD + "_agent_share_b%d.json" % b) for b in (8, 16, 32)]
    cp.append((os.path.join(ROOT, "gpurun_out", "r06_trajectory.json"), RD + "_trajectory.json"))      # written by the GPU suite's trajectory test
    for src, dst in cp:
        if os.path.exists(src):
            shutil.copy(src, os.path.join(ROOT, "profiles", dst))
        else:
            print("missing", src, file=sys.stderr)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py"), P, "24"], capture_output=True, text=True, check=True).stdout
    open(os.path.join(ROOT, "profiles", RD + "_pmc_conv_sp.txt"), "w").write(out)


if __name__ == "__main__":
    FROM_PROFILES = "--from-profiles" in sys.argv
    n = sys.argv[sys.argv.index("--ncpu") + 1] if "--ncpu" in sys.argv else "?"
    main("--dry" in sys.argv, n)
