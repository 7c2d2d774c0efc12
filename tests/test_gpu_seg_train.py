"""SegModule.step on the HIP path (disconet_amd/seg_train.py; BASELINE.json configs[3]) against the oracle: torch-CPU
autograd of oracle/seg_ref.py in train() mode through F.cross_entropy + torch.optim.Adam, on identical inputs and
parameters -- loss, every parameter's gradient (float64 oracle run as the truth, criteria of
tests/test_gpu_train_step.py), BatchNorm running statistics, the next step's loss -- the four UNet training kernels
against torch on their own, and size-independent properties at the configs[3] map size."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests import cases
from tests.test_gpu_train_step import _assert_grads

pytestmark = pytest.mark.gpu


def test_unet_training_kernels_vs_torch():
    """max-pool / bilinear-upsample forward and backward on fp32 NHWC maps vs ATen (ties in the max-pool windows --
    the zero windows of post-ReLU maps -- route the gradient to the first maximum, as torch does)"""
    from disconet_amd import train_ops as T
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 12, 20, 24, generator=g).clamp_(min=0)              # NHWC, many exact zeros
    xt = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = F.max_pool2d(xt, 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    got_y = T.maxpool2(x.cuda())
    assert torch.equal(got_y.cpu(), y.detach().permute(0, 2, 3, 1))
    wide = torch.zeros(3, 6, 10, 40)
    wide[..., 8:32] = dy.permute(0, 2, 3, 1)
    got_dx = T.maxpool2_backward(x.cuda(), wide.cuda()[..., 8:32])         # gradient as a channel slice
    assert torch.equal(got_dx.cpu(), xt.grad.permute(0, 2, 3, 1))

    z = torch.randn(2, 9, 7, 16, generator=g)
    zt = z.permute(0, 3, 1, 2).clone().requires_grad_(True)
    u = F.interpolate(zt, scale_factor=2, mode="bilinear", align_corners=True)
    du = torch.randn(u.shape, generator=g)
    u.backward(du)
    got_u = T.upsample2_bilinear(z.cuda())
    assert (got_u.cpu() - u.detach().permute(0, 2, 3, 1)).abs().max().item() <= 1e-6
    wide = torch.zeros(2, 18, 14, 24)
    wide[..., 4:20] = du.permute(0, 2, 3, 1)
    got_dz = T.upsample2_bilinear_backward(wide.cuda()[..., 4:20])
    assert (got_dz.cpu() - zt.grad.permute(0, 2, 3, 1)).abs().max().item() <= 2e-6 * float(zt.grad.abs().max())


def _setup(case):
    from disconet_amd import SegDiscoNet
    c = cases.SEG_CASES[case]
    ref = cases.seg_ref_model(c["agents"])
    model = SegDiscoNet(num_agent=c["agents"])
    model.load_state_dict(ref.state_dict())
    model = model.cuda()
    return c, ref, model, cases.seg_inputs(case)


def _fp64_grads(ref, x, trans, na, batch, labels, monkeypatch):
    from oracle.seg_ref import seg_train_loss
    orig = F.grid_sample
    monkeypatch.setattr(F, "grid_sample", lambda inp, grid, **kw: orig(inp, grid.to(inp.dtype), **kw))
    ref64 = copy.deepcopy(ref).double().train()
    loss = seg_train_loss(ref64(x.double(), trans, na, batch), labels, x)    # padded agents' images dropped
    loss.backward()
    monkeypatch.undo()
    return {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("case", list(cases.SEG_CASES))
def test_seg_train_step_matches_oracle(case, monkeypatch):
    from disconet_amd import SegModule
    from oracle.seg_train_ref import seg_train_step
    c, ref, model, (x, trans, na, labels) = _setup(case)
    g64 = _fp64_grads(ref, x, trans, na, c["batch"], labels, monkeypatch)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    l_ref = seg_train_step(ref, opt, x, trans, na, c["batch"], labels)

    mod = SegModule(model, lr=1e-3)
    data = {"bev_seq": x.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": labels.cuda()}
    out = mod.step(data, c["batch"])
    assert abs(out["loss"] - l_ref) < 2e-5 * abs(l_ref), (out, l_ref)

    eng = mod._trainer.engine
    # the committed float64 golden (tests/golden/seg_train_step.npz): loss and gradient slices
    import os
    import numpy as np
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "seg_train_step.npz"))
    assert abs(out["loss"] - float(gold["%s/loss" % case])) < 2e-5 * abs(float(gold["%s/loss" % case]))
    named = dict(model.named_parameters())
    for n in cases.SEG_GOLDEN_GRAD_TENSORS:
        got = cases.grad_slice(eng.g(named[n]).cpu())
        scale = float(gold["%s/%s/absmax" % (case, n)])
        assert abs(got - gold["%s/%s" % (case, n)]).max() < 0.05 * scale, n      # ~1 % fp32 noise floor
    ref_named = dict(ref.named_parameters())
    gmax = max(float(g.abs().max()) for g in g64.values())
    rows = {}
    for name, p in model.named_parameters():
        t = g64[name]
        den = max(float(t.abs().max()), 1e-4 * gmax)
        g = eng.g(p).cpu().double()
        e_hip = float((g - t).abs().max()) / den
        e_ora = float((ref_named[name].grad.double() - t).abs().max()) / den
        cos = float((g * t).sum() / (g.norm() * t.norm()).clamp_min(1e-300))
        rows[name] = (e_hip, e_ora, cos, float(t.abs().max()) > 1e-4 * gmax)
    _assert_grads(rows)

    ref_buf = dict(ref.named_buffers())
    for name, b in model.named_buffers():
        r = ref_buf[name]
        if name.endswith("num_batches_tracked"):
            assert int(b) == int(r), name
        else:
            assert float((b.cpu() - r).abs().max()) < 1e-4 * max(float(r.abs().max()), 1.0), name

    l_ref2 = seg_train_step(ref, opt, x, trans, na, c["batch"], labels)
    out2 = mod.step(data, c["batch"])
    assert abs(out2["loss"] - l_ref2) < 2e-3 * abs(l_ref2), (out2, l_ref2)
    assert out2["loss"] < out["loss"]

    # eval() afterwards: the oracle loaded with the HIP model's trained state agrees with the HIP eval forward
    model.eval()
    ref_eval = cases.seg_ref_model(c["agents"])
    ref_eval.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    ref_eval.eval()
    with torch.no_grad():
        want = ref_eval(x, trans, na, c["batch"])
        got = model(x.cuda(), trans.cuda(), na.cuda(), c["batch"])
    assert float((got.cpu() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()) / 5.0)


def test_seg_train_step_at_baseline_size_properties():
    """configs[3] shape (5 agents, 256 x 256, batch 2 here): the loss is finite and falls over three steps on a fixed
    batch, every gradient is finite, BatchNorm statistics move, ignored pixels (-100) change nothing but the divisor"""
    from disconet_amd import SegDiscoNet, SegModule
    from disconet_amd.synthetic import make_scene_batch
    A, B, hw = 5, 2, 256
    torch.manual_seed(0)
    model = SegDiscoNet(num_agent=A).cuda()
    bevs, trans, na = make_scene_batch(B, A, hw)
    x = bevs[:, 0].permute(0, 3, 1, 2).contiguous()
    g = torch.Generator().manual_seed(4)
    labels = torch.randint(0, 8, (A * B, hw, hw), generator=g)
    labels[torch.rand(labels.shape, generator=g) < 0.1] = -100
    mod = SegModule(model, lr=1e-3)
    data = {"bev_seq": x.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": labels.cuda()}
    rm0 = model.inc.double_conv[1].running_mean.clone()
    losses = [mod.step(data, B)["loss"] for _ in range(3)]
    assert all(l == l and l < 1e3 for l in losses), losses
    assert losses[2] < losses[0], losses
    eng = mod._trainer.engine
    assert torch.isfinite(eng.flat_g).all()
    assert float(eng.flat_g.abs().max()) > 0
    assert not torch.equal(rm0, model.inc.double_conv[1].running_mean)
    assert int(model.up4.conv.double_conv[4].num_batches_tracked) == 3
    # determinism: the same three steps from the same start give the same bits (no atomics on the gradient path)
    torch.manual_seed(0)
    model2 = SegDiscoNet(num_agent=A).cuda()
    mod2 = SegModule(model2, lr=1e-3)
    losses2 = [mod2.step(data, B)["loss"] for _ in range(3)]
    assert torch.equal(mod2._trainer.engine.flat_g, eng.flat_g) and torch.equal(mod2._trainer.engine.flat_p, eng.flat_p)
    assert abs(losses2[2] - losses[2]) <= 1e-12 * abs(losses[2])


def test_seg_step_drops_the_images_of_padded_agents():
    """upstream SegModule.step drops pred / labels of every image whose BEV is empty (padded agent slots) before the
    criterion: whatever the labels of those images are, loss and gradients are the same; an optimizer's lr schedule
    reaches the engine"""
    from disconet_amd import SegDiscoNet, SegModule
    from disconet_amd.synthetic import make_scene_batch
    A, B, hw = 3, 2, 64
    bevs, trans, na = make_scene_batch(B, A, hw, live=[3, 1])
    x = bevs[:, 0].permute(0, 3, 1, 2).contiguous()
    g = torch.Generator().manual_seed(9)
    labels = torch.randint(0, 8, (A * B, hw, hw), generator=g)
    empty = [i for i in range(A * B) if float(x[i].sum()) <= 1e-4]
    assert empty == [3, 5]                                   # agents 1, 2 of scene 1 (image = a * B + b)
    outs = []
    for variant in range(2):
        torch.manual_seed(0)
        model = SegDiscoNet(num_agent=A).cuda()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        mod = SegModule(model, optimizer=opt)
        lab = labels.clone()
        if variant:
            lab[empty] = 7 - lab[empty]                      # other labels on the dropped images
        data = {"bev_seq": x.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": lab.cuda()}
        loss = mod.step(data, B)["loss"]
        outs.append((loss, mod._trainer.engine.flat_g.clone()))
        opt.param_groups[0]["lr"] = 5e-4                     # what a MultiStepLR does
        mod.step(data, B)
        assert mod._trainer.engine.lr == 5e-4
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


def test_seg_step_takes_uint8_labels_and_skips_a_batch_without_live_images():
    """the reference casts with labels.long(), so uint8 label maps are legal -- the ignore index (-100) used
    to be written into the caller's dtype (full_like on uint8); and a batch whose images are ALL empty has a zero divisor:
    the step is skipped, parameters and optimizer state untouched"""
    from disconet_amd import SegDiscoNet, SegModule
    from disconet_amd.synthetic import make_scene_batch
    A, B, hw = 2, 1, 64
    bevs, trans, na = make_scene_batch(B, A, hw, live=[1])
    x = bevs[:, 0].permute(0, 3, 1, 2).contiguous()
    g = torch.Generator().manual_seed(4)
    labels = torch.randint(0, 8, (A * B, hw, hw), generator=g)
    losses = []
    for dtype in (torch.int64, torch.uint8):
        torch.manual_seed(0)
        mod = SegModule(SegDiscoNet(num_agent=A).cuda())
        data = {"bev_seq": x.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": labels.to(dtype).cuda()}
        losses.append((mod.step(data, B)["loss"], mod._trainer.engine.flat_g.clone()))
    assert losses[0][0] == losses[1][0] and torch.equal(losses[0][1], losses[1][1])
    assert losses[0][0] == losses[0][0]                                   # not a NaN
    eng = mod._trainer.engine
    before, steps = eng.flat_p.clone(), eng.step_count
    out = mod.step({"bev_seq": torch.zeros_like(x).cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                    "labels": labels.cuda()}, B)
    assert out.get("skipped") and out["loss"] != out["loss"]
    assert torch.equal(eng.flat_p, before) and eng.step_count == steps
