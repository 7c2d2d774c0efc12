#!/usr/bin/env python
"""Thin re-hosting of the reference's segmentation evaluation entry point for `--com disco`
(upstream:tools/seg/test_seg.py; the task is only mentioned at /root/reference/README.md:15, the
flag spelling follows the det tools at README.md:68-75).  The reference's tool body -- V2X-Sim
loading, mIoU bookkeeping, visualisation -- is out of scope; this shim builds
the model the way the tool does, loads `--resume` if given, and runs the eval forward + the
per-pixel cross entropy on synthetic scenes through the MI355X path.

    python tools/seg/test_seg.py --com disco [--resume ckpt.pth] [--num_agent 5] [--batch 1] [--rsu 0]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from disconet_amd import SegDiscoNet, SegModule  # noqa: E402
from disconet_amd.synthetic import make_scene_batch, randomize_bn_stats  # noqa: E402


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data", default=None, help="(unused here: synthetic scenes)")
    ap.add_argument("--com", default="disco")
    ap.add_argument("--resume", default="")
    ap.add_argument("--log", action="store_true")
    ap.add_argument("--logpath", default="")
    ap.add_argument("--nworker", type=int, default=0)
    ap.add_argument("--kd_flag", type=int, default=0)
    ap.add_argument("--num_agent", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rsu", type=int, default=0)
    ap.add_argument("--only_v2i", type=int, default=0)
    ap.add_argument("--map_hw", type=int, default=256)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--visualization", type=int, default=0, help="accepted; visualisation is out of scope")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.com != "disco":
        raise SystemExit("only --com disco is built on the MI355X path (SURVEY.md §2.1 #8)")
    num_agent = args.num_agent + (1 if args.rsu else 0)
    model = SegDiscoNet(num_agent=num_agent, kd_flag=bool(args.kd_flag), only_v2i=bool(args.only_v2i))
    if args.resume:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(checkpoint["model_state_dict"])
        print("loaded", args.resume, "epoch", checkpoint.get("epoch"))
    else:
        torch.manual_seed(0)
        randomize_bn_stats(model)
    model.eval().cuda()
    mod = SegModule(model)
    for frame in range(args.frames):
        bevs, trans, na = make_scene_batch(args.batch, num_agent, args.map_hw, jitter_seed=frame)
        labels = torch.randint(0, 8, (bevs.shape[0], args.map_hw, args.map_hw))
        data = {"bev_seq": bevs[:, 0].permute(0, 3, 1, 2).cuda(), "trans_matrices": trans.cuda(),
                "num_agent": na.cuda(), "labels": labels.cuda()}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = mod.evaluate(data, args.batch)
        torch.cuda.synchronize()
        print("frame %d: %.2f ms  pred %s  cross entropy %.4f" % (
            frame, 1e3 * (time.perf_counter() - t0), tuple(out["pred"].shape), out["loss"]))


if __name__ == "__main__":
    main()
