"""Fills DESIGN.md's @@PLACEHOLDERS@@ and the per-layer table from the round's final GPU run (tools/final_set.sh r05 ->
gpurun_out/r05final, gpurun_out/r05prof) and copies that run's summaries into profiles/r05_*.  Run once, at the end:
    python tools/fill_design.py [--dry]      (RD below names the round; PREV = the previous round's per-layer microseconds)"""
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = "r05"
F, P = os.path.join(ROOT, "gpurun_out", RD + "final"), os.path.join(ROOT, "gpurun_out", RD + "prof")
R3 = {"conv_pre_1 + conv_pre_2": 98, "conv1_1 (s2)": 52, "conv1_2 + Conv3D 1x1": 82, "conv2_1 (s2)": 47, "conv2_2": 55,      # round 4 (its fastest lease)
      "conv3d_2 (1x1)": 20, "conv3_1 (s2)": 42, "conv3_2": 61, "conv4_1 (s2)": 54, "conv4_2": 65, "conv5_1 (up+cat)": 132, "conv5_2": 60,
      "conv6_1 (up+cat)": 123, "conv6_2": 54, "conv7_1 (up+cat)": 122, "conv7_2": 56, "conv8_1 (up+cat)": 152, "conv8_2": 89,
      "heads (3x3 + block-diag 1x1)": 151}
SHAPE = {"conv_pre_1 + conv_pre_2": "13→32→32 @256², one launch (occupancy words in, intermediate map in LDS)", "conv1_1 (s2)": "32→64 s2 → 128²",
         "conv1_2 + Conv3D 1x1": "64→64 (+1×1) @128²", "conv2_1 (s2)": "64→128 s2 → 64²", "conv2_2": "128→128 @64²", "conv3d_2 (1x1)": "128→128 1×1 @64²",
         "conv3_1 (s2)": "128→256 s2 → 32²", "conv3_2": "256→256 @32² (+ fp32 NHWC copy)", "conv4_1 (s2)": "256→512 s2 → 16²", "conv4_2": "512→512 @16²",
         "conv5_1 (up+cat)": "768→256 @32² (tap-merged BN 32, 4 K slices)", "conv5_2": "256→256 @32²", "conv6_1 (up+cat)": "384→128 @64² (tap-merged BN 32)",
         "conv6_2": "128→128 @64²", "conv7_1 (up+cat)": "192→64 @128² (tap-merged BN 64)", "conv7_2": "64→64 @128²",
         "conv8_1 (up+cat)": "96→32 @256² (tap-merged BN 32)", "conv8_2": "32→32 @256²", "heads (3x3 + block-diag 1x1)": "32→64→(12, 36) @256², fp32 out"}


def j(path):
    return json.load(open(path))


def layer_table():
    layers = []
    for line in open(os.path.join(P, "bench_layers.txt")):
        m = re.match(r"^(\S.*?)\s{2,}([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(.*)$", line.rstrip())
        if m and m.group(1) in R3:
            layers.append((m.group(1), float(m.group(2)), float(m.group(4))))
    # counters: the last 24 product launches of the eager pass = zero fill, scatter, 10 encoder convs, warp, fuse, conv5_1 (+ fix-up), 8 more
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py"), P, "24"], capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in out.splitlines()[1:]]
    conv = [r for r in rows if r[0].startswith("sp<") or r[0].startswith("spq<") or r[0].startswith("conv_pre_pair")]
    merged = []
    for r in conv:       # the K-sliced layer's fix-up launch belongs to the row before it
        name = " ".join(r[:-9])
        vals = [float(x) for x in r[-9:]]      # blocks us GHz mfma wInst wAny act fetch write
        if merged and merged[-1][0] == name and vals[0] < 512 and "1>" in name.replace(" ", ""):
            merged[-1][1][7] += vals[7]; merged[-1][1][8] += vals[8]
            continue
        merged.append([name, vals])
    assert len(merged) == len(layers) == 19, (len(merged), len(layers))
    lines = ["| layer (shape at batch 4 × 5 agents) | µs (round 4, its fastest lease) | TFLOP/s (alg.) | MFMA busy | clock | busy × GHz ÷ 2.4 | HBM r + w (MB) |", "|---|---|---|---|---|---|---|"]
    for (name, us, tf), (kname, v) in zip(layers, merged):
        lines.append("| %s %s | %.0f (%d) | %.0f | %.0f %% | %.2f GHz | %.2f | %.0f + %.0f |" % (
            name.split(" (")[0].replace(" + Conv3D 1x1", "+Conv3D"), SHAPE[name], us, R3[name], tf, v[3], v[2], v[3] / 100 * v[2] / 2.4, v[7], v[8]))
    return "\n".join(lines)


def main(dry):
    d = j(os.path.join(F, "bench_default.json"))
    seg = j(os.path.join(F, "bench_seg.json"))
    sh = {b: j(os.path.join(F, "agent_share%s.json" % ("" if b == 4 else "_b%d" % b))) for b in (4, 8, 16, 32)}
    rp = j(os.path.join(P, "rocprof_conv_sp.json"))
    tr = j(os.path.join(P, "pmc_traffic_sp.json"))
    r, c, a = d["roofline"], d["cpu_baseline"], d["alt_math"]
    es = d["agent_sharded"]["emulated_share"]
    flop = r["flop_per_step"]
    v = {
        "VALUE": "%.0f" % d["value"], "MS": "%.3f" % d["ms_per_step"], "CONV_ROCPROF_MS": "%.3f" % rp["conv_ms_per_step"],
        "CONV_EVENTS_MS": "%.3f" % r["kernel_ms_per_step"], "CONV_TF": "%.0f" % (flop / rp["conv_ms_per_step"] / 1e9),
        "FRAC": "%.3f" % (flop / rp["conv_ms_per_step"] / 1e9 / 2500), "F32_VALUE": "%.0f" % a["value"], "F32_MS": "%.2f" % a["ms_per_step"],
        "F32_TF": "%.0f" % a["roofline"]["achieved"], "F32_FRAC": "%.2f" % a["roofline"]["frac"],
        "SEG_VALUE": "%.0f" % seg["value"], "SEG_MS": "%.2f" % seg["ms_per_step"], "SEG_TF": "%.0f" % seg["roofline"]["achieved"],
        "SEG_FRAC": "%.3f" % seg["roofline"]["frac"], "CPU4": "%.2f" % c["value"], "CPU1": "%.2f" % c["batch_1"]["value"],
        "TRAIN_MS": "%.1f" % d["train_step"]["ms_per_step"], "TRAIN_SPS": "%.0f" % d["train_step"]["scenes_per_s"],
        "TRAINKD_MS": "%.1f" % d["train_step"]["with_kd"]["ms_per_step"], "SEGTRAIN_MS": "%.1f" % seg["train_step"]["ms_per_step"],
        "AG4_SPS": "%.0f" % d["agent_sharded"]["value"], "AG4_MS": "%.2f" % d["agent_sharded"]["ms_per_step"],
        "AG16_SPS": "%.0f" % d["agent_sharded_batch16"]["value"], "AG16_MS": "%.2f" % d["agent_sharded_batch16"]["ms_per_step"],
        "SH4R": "%.3f" % es["ms_per_step"], "SP4R": "%.2f" % es["projected_speedup"],
        "TRAFFIC_MB": "%.0f" % (tr["hbm_bytes_per_launch"] / 1e6), "ALG_MB": "%.0f" % (r["algorithmic_bytes_per_launch"] / 1e6),
        "PARITY": "cls %.1e / loc %.1e against logits of max |%.1f| / |%.1f|" % (
            c["parity_max_abs_err"]["cls"], c["parity_max_abs_err"]["loc"], c["parity_ref_max_abs"]["cls"], c["parity_ref_max_abs"]["loc"]),
        "AG8_MS": "%.2f" % sh[8]["ms_per_step"], "AG16B_MS": "%.2f" % sh[16]["ms_per_step"], "AG32_MS": "%.2f" % sh[32]["ms_per_step"],
    }
    for b in (4, 8, 16, 32):
        e = sh[b]["emulated_share"]
        v["SH%d" % b] = "%.3f" % e.get("ms_per_step_without_collective", e["ms_per_step"])
        v["SP%d" % b] = "%.2f" % e.get("projected_speedup_without_collective", e["projected_speedup"])
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", text)) - set(v))
    assert not missing, missing
    for k, val in v.items():
        text = text.replace("@@%s@@" % k, val)
    text = re.sub(r"(<!-- LAYER_TABLE_BEGIN[^\n]*-->\n).*?(<!-- LAYER_TABLE_END -->)", lambda m: m.group(1) + layer_table() + "\n" + m.group(2), text, flags=re.S)
    if dry:
        print(layer_table())
        print(json.dumps(v, indent=1))
        return
    open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
    cp = [(os.path.join(F, "bench_default.json"), RD + "_bench_default.json"), (os.path.join(F, "bench_seg.json"), RD + "_bench_seg.json"),
          (os.path.join(F, "agent_share.json"), RD + "_agent_share.json"), (os.path.join(P, "kernel_stats.csv"), RD + "_bench_kernel_stats.csv"),
          (os.path.join(P, "kernel_trace.csv"), RD + "_bench_kernel_trace.csv"), (os.path.join(P, "bench_layers.txt"), RD + "_bench_layers.txt"),
          (os.path.join(P, "step_timeline.txt"), RD + "_step_timeline.txt"), (os.path.join(P, "rocprof_conv_sp.json"), RD + "_rocprof_conv_sp.json"),
          (os.path.join(P, "pmc_traffic_sp.json"), RD + "_pmc_traffic_sp.json"), (os.path.join(P, "pmc_traffic_seg.json"), RD + "_pmc_traffic_seg.json"),
          (os.path.join(P, "train_step_kernel_stats.csv"), RD + "_train_step_kernel_stats.csv")]
    cp += [(os.path.join(F, "agent_share_b%d.json" % b), RD + "_agent_share_b%d.json" % b) for b in (8, 16, 32)]
    for src, dst in cp:
        shutil.copy(src, os.path.join(ROOT, "profiles", dst))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py"), P, "24"], capture_output=True, text=True, check=True).stdout
    open(os.path.join(ROOT, "profiles", RD + "_pmc_conv_sp.txt"), "w").write(out)


if __name__ == "__main__":
    main("--dry" in sys.argv)
