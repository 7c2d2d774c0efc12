"""The profile post-processing tools run on the committed rocprofv3 trace (no GPU): the per-layer table and the
rocprof-derived conv time that bench.py reports in `roofline.rocprof` can be regenerated from profiles/."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACE = os.path.join(ROOT, "profiles", "r04_bench_kernel_trace.csv")


def _run(*args):
    return subprocess.run([sys.executable] + list(args), cwd=ROOT, capture_output=True, text=True, check=True).stdout


def test_layer_table_from_the_committed_trace():
    out = _run("tools/layers_from_trace.py", TRACE)
    rows = [l for l in out.splitlines() if l.startswith("conv") or l.startswith("heads")]
    assert len(rows) == 19, out            # conv_pre_1 + conv_pre_2 are one launch (round 4)
    total = [l for l in out.splitlines() if l.startswith("all conv launches")][0].split()
    us, gflop = float(total[3]), float(total[4])
    assert 600 < gflop < 650 and 1000 < us < 3000
    committed = open(os.path.join(ROOT, "profiles", "r04_bench_layers.txt")).read()
    assert committed.strip() == out.strip()


def test_rocprof_conv_time_matches_the_committed_summary():
    got = json.loads(_run("tools/rocprof_conv.py", TRACE, "conv_sp_kernel,conv_spq_kernel,conv_pre_pair_kernel", "19"))
    want = json.load(open(os.path.join(ROOT, "profiles", "r04_rocprof_conv_sp.json")))
    assert got["launches_per_step"] == 19.0
    assert abs(got["conv_ms_per_step"] - want["conv_ms_per_step"]) < 1e-9
