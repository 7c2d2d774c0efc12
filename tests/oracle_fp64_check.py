"""How far is the fp32 oracle's own backward from exact arithmetic?  Runs the oracle's
training forward/backward in float32 and in float64 (same parameters, inputs and warp grids)
and prints the per-tensor gradient deviation -- the noise floor a HIP-vs-oracle gradient
comparison cannot go below.   python tests/oracle_fp64_check.py [case]"""
import copy
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from oracle.train_ref import det_loss  # noqa: E402
from tests import cases  # noqa: E402
from tests.test_gpu_train_step import STEP_CASES  # noqa: E402
from disconet_amd.synthetic import make_scene_batch, make_train_targets  # noqa: E402

_orig = F.grid_sample
F.grid_sample = lambda inp, grid, **kw: _orig(inp, grid.to(inp.dtype), **kw)


def main(case="cfg1", init="kaiming"):
    c = STEP_CASES[case]
    ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0, init=init)
    ref64 = copy.deepcopy(ref).double()
    ref64.u_encoder.conv_pre_1.register_forward_pre_hook(lambda m, inp: (inp[0].double(),))
    bevs, trans, na = make_scene_batch(c["batch"], c["agents"], c["map_hw"], live=c["live"],
                                       jitter_seed=c["jitter"])
    labels, targets, mask = make_train_targets(bevs.shape[0], c["map_hw"], p_fg=0.02)

    def run(m):
        m.train()
        out = m(bevs, trans, na, c["batch"])
        lc, ll = det_loss(out, labels, targets, mask, norm=bevs.shape[0])
        (lc + ll).backward()
        return float(lc.detach()), float(ll.detach())

    print("loss fp32", run(ref), "fp64", run(ref64))
    r64 = dict(ref64.named_parameters())
    worst = 0.0
    for n, p in ref.named_parameters():
        if p.grad is None:
            continue
        g64 = r64[n].grad
        e = float((p.grad.double() - g64).abs().max() / g64.abs().max())
        if float(g64.abs().max()) > 1e-2:
            worst = max(worst, e)
        print("%-48s max|g| %.3e  fp32-vs-fp64 err/max %.2e" % (n, float(g64.abs().max()), e))
    print("worst (tensors with max|g| > 1e-2): %.2e" % worst)


if __name__ == "__main__":
    main(*sys.argv[1:])
