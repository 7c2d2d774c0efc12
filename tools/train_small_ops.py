"""Which torch ops (copies, fills, casts, elementwise) the steady-state training step still launches, by Python call site:

    python tools/train_small_ops.py [--steps 3]

torch.profiler with stacks over `--steps` steps after the calibration step; prints per (op, first disconet_amd frame) the calls per
step and the device time.  The library's own kernels go through ctypes and do not show up here -- this lists what is NOT ours."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    from disconet_amd import CoDetModule, Config, DiscoNet
    from disconet_amd.synthetic import make_scene_batch, make_train_targets
    torch.manual_seed(0)
    model = DiscoNet(Config(map_hw=256), kd_flag=0, num_agent=5)
    model.conv_math = "sp"
    model.cuda()
    bevs, trans, na = make_scene_batch(4, 5, 256)
    labels, targets, mask = make_train_targets(bevs.shape[0], 256)
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": labels.cuda(),
            "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    mod = CoDetModule(model, lr=1e-3, dgrad_math="sp", wgrad_math="sp")
    for _ in range(3):
        mod.step(data, 4)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(args.steps):
            mod.step(data, 4)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dev_us = getattr(ev, "self_device_time_total", 0) or 0
        if dev_us <= 0 or not ev.name.startswith("aten::"):
            continue
        site = "?"
        for fr in ev.stack or []:
            if "disconet_amd" in fr or "tools/" in fr:
                site = fr.strip()
                break
        k = (ev.name, site[-110:])
        agg[k][0] += 1
        agg[k][1] += dev_us
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = 0.0
    for (name, site), (n, us) in rows:
        tot += us / args.steps
        print("%6.1f calls/step %8.1f us/step  %-28s %s" % (n / args.steps, us / args.steps, name, site))
    print("total %.1f us/step in %d torch-op launches per step" % (tot, sum(v[0] for v in agg.values()) / args.steps))


if __name__ == "__main__":
    main()
