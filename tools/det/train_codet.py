#!/usr/bin/env python
"""Thin re-hosting of the reference's training entry point for `--com disco`
(flag names from /root/reference/README.md:54-63).  The reference's tool body -- V2X-Sim
loading, anchor / target assembly, logging -- is out of scope; this shim
builds the student (and, with --kd_flag 1, the teacher) the way the reference's tool does,
resumes from `--resume` / `--resume_teacher` if given, and runs CoDetModule.step on synthetic
scenes and targets (there is no V2X-Sim data in this environment) through the MI355X path,
writing `epoch_N.pth` with the reference's checkpoint keys.

    python tools/det/train_codet.py --com disco [--batch 4] [--nepoch 2] [--steps_per_epoch 8] \
        [--kd_flag 1 --resume_teacher teacher.pth --kd_weight 100000] [--resume epoch_1.pth] \
        [--logpath logs/] [--num_agent 5] [--layer 3] [--compress_level 0] [--only_v2i 0]

Data-parallel: one process per GPU under torch.distributed.run; every rank trains its own
scenes and the flat gradient buffer is averaged by one RCCL all-reduce per step.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from disconet_amd import CoDetModule, Config, DiscoNet, TeacherNet  # noqa: E402
from disconet_amd.synthetic import make_bevs, make_scene_batch, make_train_targets  # noqa: E402


def _rsu_from_leftovers(args, rest):
    """The README writes the last flag as `-- rsu [0/1]` (a space after the dashes,
    /root/reference/README.md:63): argparse then sees "--" and the words `rsu` and `<value>`.
    Accept that spelling; anything else left over is an error, as argparse would report."""
    rest = [r for r in rest if r != "--"]
    if len(rest) == 2 and rest[0] == "rsu":
        digits = "".join(ch for ch in rest[1] if ch.isdigit())
        args.rsu = int(digits[-1]) if digits else 0     # "[0/1]" -> the placeholder's second choice
        rest = []
    if rest:
        raise SystemExit("unrecognized arguments: " + " ".join(rest))
    return args


def newest_checkpoint(path):
    """--auto_resume_path: the epoch_N.pth with the largest N under `path` (None if there is none)"""
    best, best_n = None, -1
    if path and os.path.isdir(path):
        for name in os.listdir(path):
            if name.startswith("epoch_") and name.endswith(".pth"):
                try:
                    n = int(name[len("epoch_"):-len(".pth")])
                except ValueError:
                    continue
                if n > best_n:
                    best, best_n = os.path.join(path, name), n
    return best


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data", default=None, help="(unused here: synthetic scenes)")
    ap.add_argument("--com", default="disco")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--nepoch", type=int, default=2)
    ap.add_argument("--steps_per_epoch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--nworker", type=int, default=0)
    ap.add_argument("--log", action="store_true")
    ap.add_argument("--logpath", default="")
    ap.add_argument("--resume", default="")
    ap.add_argument("--resume_teacher", default="")
    ap.add_argument("--layer", type=int, default=3)
    ap.add_argument("--kd_flag", type=int, default=0)
    ap.add_argument("--kd_weight", type=float, default=100000.0)
    ap.add_argument("--num_agent", type=int, default=5)
    ap.add_argument("--rsu", type=int, default=0)
    ap.add_argument("--compress_level", type=int, default=0)
    ap.add_argument("--only_v2i", type=int, default=0)
    ap.add_argument("--auto_resume_path", default="",
                    help="resume from the newest epoch_N.pth under this directory, if any")
    return ap


def parse_args(argv=None):
    args, rest = build_parser().parse_known_args(argv)
    return _rsu_from_leftovers(args, rest)


def main(argv=None):
    args = parse_args(argv)
    if args.auto_resume_path and not args.resume:
        found = newest_checkpoint(args.auto_resume_path)
        if found:
            args.resume = found
    if args.com != "disco":
        raise SystemExit("only --com disco is built on the MI355X path (SURVEY.md §2.1 #8)")
    num_agent = args.num_agent + (1 if args.rsu else 0)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")

    config = Config("train", binary=True, only_det=True)
    hw = config.map_dims[0]
    torch.manual_seed(0)
    model = DiscoNet(config, layer=args.layer, kd_flag=args.kd_flag, num_agent=num_agent,
                     compress_level=args.compress_level, only_v2i=bool(args.only_v2i)).cuda()
    teacher = None
    if args.kd_flag:
        teacher = TeacherNet(config).cuda()
        if args.resume_teacher:
            teacher.load_state_dict(torch.load(args.resume_teacher, map_location="cpu")["model_state_dict"])
        teacher.eval()
    start_epoch = 1
    optimizer_state = None
    if args.resume:
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["model_state_dict"])
        optimizer_state = ck.get("optimizer_state_dict")
        start_epoch = int(ck.get("epoch", 0)) + 1
        print("resumed", args.resume, "-> epoch", start_epoch)
    fafmodule = CoDetModule(model, teacher, config, None, kd_flag=args.kd_flag, lr=args.lr)
    if optimizer_state is not None:
        fafmodule.engine.load_state_dict(optimizer_state)

    n_img = num_agent * args.batch
    for epoch in range(start_epoch, start_epoch + args.nepoch):
        t0, running = time.perf_counter(), 0.0
        for it in range(args.steps_per_epoch):
            seed = (epoch * 1000 + it) * world + rank
            bevs, trans, na = make_scene_batch(args.batch, num_agent, hw, jitter_seed=seed)
            labels, targets, mask = make_train_targets(n_img, hw, seed=seed)
            data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                    "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
            if args.kd_flag:
                data["bev_seq_teacher"] = make_bevs(args.batch, num_agent, hw, p=0.05).cuda()
                data["kd_weight"] = args.kd_weight
            out = fafmodule.step(data, args.batch)
            running += out["loss"]
        torch.cuda.synchronize()
        lr = fafmodule.scheduler_step(epoch)
        dt = time.perf_counter() - t0
        if rank == 0:
            fb = fafmodule.engine.f32_fallback_steps      # backward passes repeated on the fp32 kernels (a gradient outgrew its lift)
            print("epoch %d: mean loss %.4f  (%s)  %.1f scenes/s  lr %.2e%s" % (
                epoch, running / args.steps_per_epoch,
                ", ".join("%s %.4f" % (k, v) for k, v in out.items() if k != "loss"),
                world * args.batch * args.steps_per_epoch / dt, lr,
                "  [%d fp32 fallback step(s) so far]" % fb if fb else ""))
            if args.logpath:
                os.makedirs(args.logpath, exist_ok=True)
                torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                            "optimizer_state_dict": fafmodule.engine.state_dict(),
                            "scheduler_state_dict": {"lr": lr}, "loss": running / args.steps_per_epoch},
                           os.path.join(args.logpath, "epoch_%d.pth" % epoch))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
