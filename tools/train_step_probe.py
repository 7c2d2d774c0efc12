"""One CoDetModule.step configuration, timed alone (so that a rocprofv3 --kernel-trace --stats of this process holds the
training step's kernels and nothing else):

    python tools/train_step_probe.py [--dgrad f32|sp] [--wgrad f32|sp] [--math sp|f16x3|f32] [--steps 6]

BASELINE configs[1]'s batch (5 agents x batch 4, 256 x 256 x 13, synthetic).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dgrad", default="f32", choices=["f32", "sp"])
    ap.add_argument("--wgrad", default="f32", choices=["f32", "sp"])
    ap.add_argument("--math", default="sp")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--map", type=int, default=256)
    ap.add_argument("--gc", default="on", choices=["on", "off", "freeze"], help="Python's cyclic collector during the timed steps")
    args = ap.parse_args()
    from disconet_amd import CoDetModule, Config, DiscoNet, ops
    from disconet_amd.synthetic import make_scene_batch, make_train_targets
    torch.manual_seed(0)
    model = DiscoNet(Config(map_hw=args.map), kd_flag=0, num_agent=args.agents)
    model.conv_math = args.math
    model.cuda()
    bevs, trans, na = make_scene_batch(args.batch, args.agents, args.map)
    labels, targets, mask = make_train_targets(bevs.shape[0], args.map)
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": labels.cuda(),
            "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    mod = CoDetModule(model, lr=1e-3, dgrad_math=args.dgrad, wgrad_math=args.wgrad)
    first = mod.step(data, args.batch)
    mod.step(data, args.batch)
    torch.cuda.synchronize()
    import gc
    if args.gc == "off":
        gc.disable()
    elif args.gc == "freeze":
        gc.collect()
        gc.freeze()
    per = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        last = mod.step(data, args.batch)
        per.append(round(1e3 * (time.perf_counter() - t1), 3))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    gc.enable()
    print(json.dumps({"dgrad": args.dgrad, "wgrad": args.wgrad, "math": args.math, "ms_per_step": round(1e3 * dt, 3), "loss_first": first["loss"],
                      "loss_last": last["loss"], "step_ms": per, "gc": args.gc, "lifts": {k: v[0] for k, v in mod.engine._dz_lift.items()},
                      "range_flags": ops.sp_range_flags(reset=False)}))


if __name__ == "__main__":
    main()
