"""Host-side cost of the training step: cProfile over `--steps` steady-state steps of tools/train_step_probe.py's configuration,
top functions by own time and by cumulative time (the step is launch-bound on the host through its first ~100 launches:
profiles/r06_train_gap_sites.txt).   python tools/train_host_profile.py [--steps 20]"""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=35)
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
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        mod.step(data, 4)
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumtime"):
        print("==== by %s, per step = total / %d" % (key, args.steps))
        pstats.Stats(pr).strip_dirs().sort_stats(key).print_stats(args.top)


if __name__ == "__main__":
    main()
