"""Per-piece kernel time of the training step from a rocprofv3 --kernel-trace --stats CSV of tools/train_step_probe.py:

    python tools/train_groups.py kernel_stats.csv [steps]

Groups the step's kernels into the rows of DESIGN.md section 8 (forward convs, BatchNorm forward, data gradients, weight
gradients, BatchNorm backward, the rest) and prints ms per step (total over the run / steps; the probe's first step is the
fp32 calibration pass, so the averages sit slightly above the steady step).  Which conv launches are forward and which are
data gradients is read from the kernel template: fp32-row epilogue launches of the inference engine serve both, so the
split is by call count of the probe's layer graph (20 forward layers, 19 data-gradient layers per step)."""
import csv
import json
import sys

GROUPS = [
    ("wgrad", ("conv_wgrad_sp_s2", "conv_wgrad_sp_kernel", "conv_wgrad64", "conv_wgrad_kernel", "wgrad_reduce", "conv_wgrad")),
    ("bn_forward", ("bn_apply", "bn_stats")),
    ("bn_backward", ("bn_bwd_apply", "bn_bwd_reduce", "bn_param_grad")),
    ("folds_and_bias_sums", ("fold_", "channel_sum")),
    ("conv_engine_sp", ("conv_sp_kernel", "conv_spq_kernel", "conv_pre_pair")),
    ("conv_engine_nhwc", ("conv_mfma_kernel",)),
    ("packing", ("sp_pack", "pack", "sp_from_nhwc", "dgrad_weights", "dgrad_class")),
    ("copies", ("copyBuffer", "fillBuffer", "direct_copy")),
]


def group_of(name):
    for g, keys in GROUPS:
        if any(k in name for k in keys):
            return g
    return "other (loss, warp, combine, Adam, torch elementwise)"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    out = {}
    for r in rows:
        g = group_of(r["Name"])
        e = out.setdefault(g, [0, 0.0])
        e[0] += int(r["Calls"])
        e[1] += float(r["TotalDurationNs"])
    total = sum(v[1] for v in out.values())
    res = {g: {"calls_per_step": round(c / steps, 1), "ms_per_step": round(t / 1e6 / steps, 3)} for g, (c, t) in out.items()}
    res["total_ms_per_step"] = round(total / 1e6 / steps, 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
