"""Does the training step read memory nobody wrote?  torch.empty is replaced by a NaN / 0xFF fill for the whole
step; the losses must still match the committed float64 golden.  Repeats to expose order-dependent garbage."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cases  # noqa: E402
from disconet_amd import CoDetModule, Config, DiscoNet  # noqa: E402
from disconet_amd.synthetic import make_scene_batch, make_train_targets  # noqa: E402

gold = np.load(os.path.join(ROOT, "tests", "golden", "train_step.npz"))
real_empty, real_empty_like = torch.empty, torch.empty_like


def poison(t):
    if t.is_cuda:
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.uint8, torch.int32, torch.int64):
            t.fill_(-1 if t.dtype != torch.uint8 else 255)
    return t


for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for case in ("cfg1", "ragged_a4"):
        for math in ("f32", "f16x3"):
            c = cases.TRAIN_CASES[case]
            ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0)
            model = DiscoNet(Config(map_hw=c["map_hw"]), kd_flag=0, num_agent=c["agents"])
            model.load_state_dict(ref.state_dict())
            model = model.cuda()
            model.conv_math = math
            bevs, trans, na = make_scene_batch(c["batch"], c["agents"], c["map_hw"], live=c["live"], jitter_seed=c["jitter"])
            labels, targets, mask = make_train_targets(bevs.shape[0], c["map_hw"], p_fg=0.02)
            mod = CoDetModule(model, lr=1e-3)
            data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                    "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
            if trial % 2 == 1:
                torch.empty = lambda *a, **k: poison(real_empty(*a, **k))
                torch.empty_like = lambda *a, **k: poison(real_empty_like(*a, **k))
            out = mod.step(data, c["batch"])
            torch.empty, torch.empty_like = real_empty, real_empty_like
            want = gold["%s/det/losses" % case]
            rel = max(abs(out["cls_loss"] - want[0]) / want[0], abs(out["loc_loss"] - want[1]) / want[1])
            print("trial %d %-10s %-6s poisoned=%d  cls %.4f loc %.4f  rel err %.2e %s" % (
                trial, case, math, trial % 2, out["cls_loss"], out["loc_loss"], rel, "" if rel < 2e-5 else "  <-- MISMATCH"))
            junk = [real_empty(1 << (10 + 2 * k), device="cuda").normal_() for k in range(8)]   # allocator churn
            del junk
