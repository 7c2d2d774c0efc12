#!/usr/bin/env python
"""Thin re-hosting of the reference's evaluation entry point for `--com disco`
(flag names from /root/reference/README.md:68-75).  The reference's tool body --
V2X-Sim loading, box decode, NMS, mAP, tracking dumps -- is out of scope
(SURVEY.md §8(f)); this shim builds the model the way the reference's tool does,
loads `--resume` if given, and runs the eval-mode forward on synthetic scenes
(there is no V2X-Sim data in this environment) through the MI355X path.

    python tools/det/test_codet.py --com disco [--resume ckpt.pth] [--nworker 0] \
        [--num_agent 5] [--batch 1] [--kd_flag 0] [--rsu 0]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from disconet_amd import Config, DiscoNet  # noqa: E402
from disconet_amd.synthetic import make_scene_batch, randomize_bn_stats  # noqa: E402


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data", default=None, help="(unused here: synthetic scenes)")
    ap.add_argument("--com", default="disco")
    ap.add_argument("--resume", default="")
    ap.add_argument("--log", action="store_true")
    ap.add_argument("--logpath", default="")
    ap.add_argument("--nworker", type=int, default=0)
    ap.add_argument("--layer", type=int, default=3)
    ap.add_argument("--kd_flag", type=int, default=0)
    ap.add_argument("--num_agent", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rsu", type=int, default=0)
    ap.add_argument("--compress_level", type=int, default=0)
    ap.add_argument("--only_v2i", type=int, default=0)
    ap.add_argument("--frames", type=int, default=4)
    # accepted for command-line compatibility (/root/reference/README.md:72,74): the tracking dump and
    # the PNG visualisation are downstream CPU consumers of the detections, out of scope here
    ap.add_argument("--tracking", action="store_true", help="accepted; the tracking dump is out of scope")
    ap.add_argument("--visualization", type=int, default=0, help="accepted; visualisation is out of scope")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.tracking or args.visualization:
        print("note: --tracking / --visualization are accepted for compatibility; this shim only runs "
              "the detector")
    if args.com != "disco":
        raise SystemExit("only --com disco is built on the MI355X path (SURVEY.md §2.1 #8)")
    num_agent = args.num_agent + (1 if args.rsu else 0)

    config = Config("test", binary=True, only_det=True)
    model = DiscoNet(config, layer=args.layer, kd_flag=args.kd_flag, num_agent=num_agent,
                     compress_level=args.compress_level, only_v2i=bool(args.only_v2i))
    if args.resume:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(checkpoint["model_state_dict"])
        print("loaded", args.resume, "epoch", checkpoint.get("epoch"))
    else:
        torch.manual_seed(0)
        randomize_bn_stats(model)
    model.eval().cuda()

    for frame in range(args.frames):
        bevs, trans, na = make_scene_batch(args.batch, num_agent, config.map_dims[0], jitter_seed=frame)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model(bevs.cuda(), trans.cuda(), na.cuda(), args.batch)
        torch.cuda.synchronize()
        result = out[0] if isinstance(out, tuple) else out
        cls = torch.softmax(result["cls"], dim=-1)
        print("frame %d: %.2f ms  cls %s loc %s  max fg score %.4f" % (
            frame, 1e3 * (time.perf_counter() - t0), tuple(result["cls"].shape),
            tuple(result["loc"].shape), float(cls[..., 1].max())))


if __name__ == "__main__":
    main()
