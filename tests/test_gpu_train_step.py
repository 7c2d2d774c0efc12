"""Whole training step on the HIP path against the oracle (torch-CPU autograd of the oracle
model in train() mode + oracle/train_ref.py losses + torch.optim.Adam) on identical inputs
and parameters: losses, every parameter's gradient, BatchNorm running statistics, the
parameters after Adam, and the loss of the following step.

Gradients: this backward is ill-conditioned in fp32 -- BatchNorm's backward subtracts the
per-channel mean of a gradient that is ~1e4 x larger than what is left (the cls head's
shift-invariant component), so the fp32 oracle itself sits 0.2-2 % (relative to each tensor's
largest gradient) away from the same oracle run in float64 (tests/oracle_fp64_check.py).
The test therefore takes the float64 run as the truth and requires, per tensor,
    err(HIP vs fp64) <= max(5 x err(fp32 oracle vs fp64), 2e-3)
(see _assert_grads for the ReLU-mask-flip allowance) and a cosine similarity > 0.9995 -- the HIP
step must be as close to exact arithmetic as ATen's fp32 autograd is.  Losses agree to 2e-5 relative."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests import cases

pytestmark = pytest.mark.gpu

STEP_CASES = cases.TRAIN_CASES


def _setup(case, math, only_v2i=False, compress_level=0, layer=3):
    from disconet_amd import Config, DiscoNet
    from disconet_amd.synthetic import make_scene_batch, make_train_targets
    c = STEP_CASES[case]
    ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0, only_v2i=only_v2i,
                          compress_level=compress_level, layer=layer)
    cfg = Config(map_hw=c["map_hw"])
    model = DiscoNet(cfg, kd_flag=0, num_agent=c["agents"], only_v2i=only_v2i,
                     compress_level=compress_level, layer=layer)
    model.load_state_dict(ref.state_dict())
    model = model.cuda()
    model.conv_math = math
    bevs, trans, na = make_scene_batch(c["batch"], c["agents"], c["map_hw"], live=c["live"],
                                       jitter_seed=c["jitter"])
    labels, targets, mask = make_train_targets(bevs.shape[0], c["map_hw"], p_fg=0.02)
    return c, ref, model, (bevs, trans, na), (labels, targets, mask)


def _fp64_grads(ref, inputs, targets, batch, monkeypatch):
    """the oracle's gradients in float64: same parameters, inputs and (fp32) warp grids"""
    from oracle.train_ref import det_loss
    orig = F.grid_sample
    monkeypatch.setattr(F, "grid_sample", lambda inp, grid, **kw: orig(inp, grid.to(inp.dtype), **kw))
    ref64 = copy.deepcopy(ref).double().train()
    ref64.u_encoder.conv_pre_1.register_forward_pre_hook(lambda m, inp: (inp[0].double(),))
    bevs, trans, na = inputs
    out = ref64(bevs, trans, na, batch)
    l_cls, l_loc = det_loss(out, *targets, norm=bevs.shape[0])
    (l_cls + l_loc).backward()
    monkeypatch.undo()
    return {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}


def _grad_report(g64, ref, engine, model):
    ref_named = dict(ref.named_parameters())
    gmax = max(float(g.abs().max()) for g in g64.values())
    rows = {}
    for name, p in model.named_parameters():
        t = g64[name]
        den = max(float(t.abs().max()), 1e-4 * gmax)
        g = engine.g(p).cpu().double()
        e_hip = float((g - t).abs().max()) / den
        e_ora = float((ref_named[name].grad.double() - t).abs().max()) / den
        cos = float((g * t).sum() / (g.norm() * t.norm()).clamp_min(1e-300))
        rows[name] = (e_hip, e_ora, cos, float(t.abs().max()) > 1e-4 * gmax)
    return rows


def _assert_grads(rows):
    """per tensor: max error within 5 x the fp32 oracle's own (floor 2e-3 of the tensor's largest
    gradient) -- or, where a ReLU mask flipped (train-mode activations differ by ~3e-5 between any two
    fp32 implementations, so ~1e-5 of the elements sit on the other side of 0 and move single
    elements of a gradient by a few % of its maximum; the fp32 oracle has the same flips against
    float64), a relative L2 error below 1.5 % with no element off by more than 15 %.  Always: cosine
    similarity > 0.9995 for tensors that carry a real gradient."""
    bad = {}
    for k, (e_hip, e_ora, cos, real) in rows.items():
        rel_l2 = (2.0 * max(0.0, 1.0 - cos)) ** 0.5
        ok = e_hip <= max(5 * e_ora, 2e-3) or (real and rel_l2 < 1.5e-2 and e_hip < 0.15)
        if not ok or (real and cos < 0.9995):
            bad[k] = (e_hip, e_ora, cos)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("case", ["cfg1", "ragged_a4"])
def test_train_step_matches_oracle(case, math, monkeypatch):
    from disconet_amd import CoDetModule
    from oracle.train_ref import train_step
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup(case, math)
    g64 = _fp64_grads(ref, (bevs, trans, na), (labels, targets, mask), c["batch"], monkeypatch)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    l_ref = train_step(ref, opt, bevs, trans, na, c["batch"], labels, targets, mask)

    mod = CoDetModule(model, lr=1e-3)
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    out = mod.step(data, c["batch"])
    assert abs(out["cls_loss"] - l_ref[0]) < 2e-5 * abs(l_ref[0]), (out, l_ref)
    assert abs(out["loc_loss"] - l_ref[1]) < 2e-5 * abs(l_ref[1]), (out, l_ref)
    # and the committed float64 golden (tests/golden/train_step.npz): losses and gradient slices
    import os
    import numpy as np
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_step.npz"))
    assert np.allclose([out["cls_loss"], out["loc_loss"]], gold["%s/det/losses" % case], rtol=2e-5)
    for n in cases.GOLDEN_GRAD_TENSORS:
        got = cases.grad_slice(mod.engine.g(dict(model.named_parameters())[n]).cpu())
        scale = float(gold["%s/det/%s/absmax" % (case, n)])
        assert abs(got - gold["%s/det/%s" % (case, n)]).max() < 0.05 * scale, n    # ~1 % fp32 noise floor

    rows = _grad_report(g64, ref, mod.engine, model)
    _assert_grads(rows)

    # BatchNorm running statistics (momentum update, unbiased variance, call order of the MLP's BNs)
    ref_buf = dict(ref.named_buffers())
    for name, b in model.named_buffers():
        r = ref_buf[name]
        if name.endswith("num_batches_tracked"):
            assert int(b) == int(r), name
        else:
            assert float((b.cpu() - r).abs().max()) < 1e-4 * max(float(r.abs().max()), 1.0), name

    # parameters after Adam: the first update is lr * sign(g) wherever g is above the ~1 %
    # gradient noise discussed above (below it the sign itself is noise, here and in the oracle)
    ref_named = dict(ref.named_parameters())
    gmax = max(float(g.abs().max()) for g in g64.values())
    for name, p in model.named_parameters():
        gr = ref_named[name].grad
        if float(gr.abs().max()) < 1e-4 * gmax:
            continue                     # a conv bias in front of a BatchNorm: the gradient is exactly 0
        sel = gr.abs() > 0.1 * gr.abs().max()
        diff = (p.detach().cpu() - ref_named[name].detach())[sel].abs()
        assert float(diff.max()) < 2e-5, name

    # the next step starts from matching state: its loss agrees
    l_ref2 = train_step(ref, opt, bevs, trans, na, c["batch"], labels, targets, mask)
    out2 = mod.step(data, c["batch"])
    assert abs(out2["loss"] - (l_ref2[0] + l_ref2[1])) < 2e-3 * abs(l_ref2[0] + l_ref2[1]), (out2, l_ref2)
    assert out2["loss"] < out["loss"]

    # eval() afterwards runs on the trained parameters and running statistics: the oracle loaded
    # with the HIP model's state_dict must agree with the HIP eval forward.  (Comparing against the
    # oracle's OWN trained weights is not meaningful at 1e-4: Adam moves every conv bias in front
    # of a BatchNorm by +-lr with the sign of a rounding-noise gradient, in both implementations.)
    model.eval()
    ref_eval = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0)
    ref_eval.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=False)
    ref_eval.eval()
    with torch.no_grad():
        r = ref_eval(bevs, trans, na, c["batch"])
        g = model(data["bev_seq"], data["trans_matrices"], data["num_agent"], c["batch"])
    for k in ("cls", "loc"):     # 1e-4 for O(5) logits, scaled where the half-trained net's are larger
        tol = 1e-4 * max(1.0, float(r[k].abs().max()) / 5.0)
        assert float((g[k].cpu() - r[k]).abs().max()) < tol, (k, float(r[k].abs().max()))


@pytest.mark.parametrize("case,dgrad,wgrad", [("cfg1", "f32", "sp"), ("cfg1", "sp", "sp"), ("ragged_a4", "sp", "f32"), ("ragged_a4", "sp", "sp")])
def test_split_f16_data_gradients_meet_the_fp32_criteria(case, dgrad, wgrad, monkeypatch):
    """dgrad_math = "sp": the 3x3 stride-1 data gradients on the inference engine's split-f16 kernels, dz pre-split and LIFTED by
    the BatchNorm backward.  wgrad_math = "sp": the 3x3 stride-1 weight gradients (>= 32 channels a side) on the f16 MFMA, dz
    (same lift) and x split while they are staged (dn_conv_wgrad_sp).  Same parameters, three backward passes: the calibration
    pass (lifts unknown: every gradient in fp32 -- bit for bit the "f32" engine's gradients), then the split-f16 pass, which
    must meet the criteria the fp32 step is held to (float64 oracle run as the truth, _assert_grads) and stay within 2.5 x of
    the fp32 pass's own error per tensor; the per-tensor errors of both go to gpurun_out/sp_grad_errors_<case>_<dgrad>_<wgrad>.json
    (DESIGN.md 8)."""
    import json
    import os
    from disconet_amd import CoDetModule
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup(case, "f32")
    g64 = _fp64_grads(ref, (bevs, trans, na), (labels, targets, mask), c["batch"], monkeypatch)
    from oracle.train_ref import det_loss
    out = ref.train()(bevs, trans, na, c["batch"])
    l_cls, l_loc = det_loss(out, labels, targets, mask, norm=bevs.shape[0])
    (l_cls + l_loc).backward()                      # the fp32 oracle's own gradients (the yardstick of _grad_report)
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    plain = CoDetModule(model, lr=1e-3, dgrad_math="f32", wgrad_math="f32")
    plain.step(data, c["batch"], update=False)
    g_plain = plain.engine.flat_g.clone()
    buffers = {k: v.clone() for k, v in model.named_buffers()}
    mod = CoDetModule(model, lr=1e-3, dgrad_math=dgrad, wgrad_math=wgrad)       # (a new engine over the same parameters: flat_p is re-pointed)
    for k, v in model.named_buffers():
        v.copy_(buffers[k])
    o1 = mod.step(data, c["batch"], update=False)
    assert torch.equal(mod.engine.flat_g, g_plain)           # calibration pass == the fp32 engine
    assert len(mod.engine._dz_lift) >= 14, sorted(mod.engine._dz_lift)
    rows_f32 = _grad_report(g64, ref, mod.engine, model)
    o2 = mod.step(data, c["batch"], update=False)
    assert abs(o1["loss"] - o2["loss"]) <= 1e-12 * abs(o1["loss"])      # (the forward is the same; f64 atomics in the loss scalars)
    assert not torch.equal(mod.engine.flat_g, g_plain)       # the split-f16 kernels did run
    rows_sp = _grad_report(g64, ref, mod.engine, model)
    worst = {k: (e_sp, rows_f32[k][0], e_ora, cos) for k, (e_sp, e_ora, cos, real) in rows_sp.items() if real}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/sp_grad_errors_%s_%s_%s.json" % (case, dgrad, wgrad), "w") as f:
        json.dump({"case": case, "dgrad_math": dgrad, "wgrad_math": wgrad, "columns": ["err split-f16", "err fp32", "err fp32 oracle (ATen)", "cosine (split-f16)"],
                   "note": "max |g - g64| / max |g64| per parameter tensor; g64 = the oracle run in float64",
                   "lifts": {k: v[0] for k, v in mod.engine._dz_lift.items()},
                   "tensors": {k: [float("%.3g" % x) for x in v] for k, v in sorted(worst.items())}}, f, indent=1)
    _assert_grads(rows_sp)
    for k, (e_sp, e_f32, e_ora, cos) in worst.items():
        assert e_sp <= max(2.5 * e_f32, 2.5 * e_ora, 2e-3), (k, e_sp, e_f32, e_ora)
    # and the optimizer step goes through with it
    o3 = mod.step(data, c["batch"])
    o4 = mod.step(data, c["batch"])
    assert o4["loss"] < o3["loss"]


def test_reference_style_step_through_autograd():
    """model.train(); out = model(...); loss.backward() -- the reference's own step code --
    fills p.grad with the same gradients as the native step"""
    from disconet_amd import CoDetModule
    from oracle.train_ref import det_loss
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("cfg1", "f16x3")
    model.train()
    out = model(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
    l_cls, l_loc = det_loss(out, labels.cuda(), targets.cuda(), mask.cuda(), norm=bevs.shape[0])
    (l_cls + l_loc).backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    assert all(g is not None for g in grads.values())

    c2, ref2, model2, _, _ = _setup("cfg1", "f16x3")
    mod = CoDetModule(model2, lr=1e-3)
    with torch.no_grad():
        res = mod.engine.forward(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
    from disconet_amd import train_ops
    _, dcls, dloc = train_ops.det_loss(res["cls"].reshape(-1, 2), labels.cuda().reshape(-1, 2).float(),
                                       res["loc"].reshape(-1, 6), targets.cuda().reshape(-1, 6).float(),
                                       mask.cuda().reshape(-1).float(), norm=bevs.shape[0])
    mod.engine.backward(dcls, dloc)
    gmax = max(float(g.abs().max()) for g in grads.values())
    for name, p in model2.named_parameters():
        a, b = grads[name], mod.engine.g(p)
        if float(b.abs().max()) < 1e-4 * gmax:
            # a conv bias in front of a BatchNorm: the gradient is exactly zero, what is stored is
            # the rounding noise of a sum of +-1e3-sized terms (and the float atomics of the warp
            # scatter make it differ from run to run)
            assert float(a.abs().max()) < 1e-4 * gmax, name
            continue
        assert float((a - b).abs().max()) < 1e-3 * float(b.abs().max()), name


def _teacher_pair(c):
    """oracle teacher + the HIP TeacherNet with the same weights, and holistic-view voxels"""
    from disconet_amd import Config, TeacherNet
    from disconet_amd.synthetic import make_bevs
    from oracle.disconet_ref import RefConfig
    from oracle.teacher_ref import build_teacher
    t_ref = build_teacher(RefConfig(c["map_hw"]))
    t_hip = TeacherNet(Config(map_hw=c["map_hw"]))
    t_hip.load_state_dict(t_ref.state_dict())
    t_hip = t_hip.cuda().eval()
    bevs_t = make_bevs(c["batch"], c["agents"], c["map_hw"], p=0.05)      # denser: everyone's points
    return t_ref, t_hip, bevs_t


def test_teacher_forward_matches_oracle():
    c = STEP_CASES["cfg1"]
    t_ref, t_hip, bevs_t = _teacher_pair(c)
    with torch.no_grad():
        want = t_ref(bevs_t)
    got = t_hip(bevs_t.cuda())
    assert len(got) == 6
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert float((g.cpu() - w).abs().max()) < 1e-4


def test_kd_kernel_matches_torch_kl_div():
    from disconet_amd import train_ops
    g = torch.Generator().manual_seed(4)
    for c_ in (64, 128, 256):
        s = (torch.randn(3, 16, 16, c_, generator=g) * 3).double().requires_grad_(True)
        t = torch.randn(3, 16, 16, c_, generator=g) * 3
        ref = 1e5 * F.kl_div(F.log_softmax(s.reshape(-1, c_), 1), F.softmax(t.double().reshape(-1, c_), 1),
                             reduction="mean")
        ref.backward()
        loss = torch.full((1,), 5.0, dtype=torch.float64, device="cuda")
        d = train_ops.kd_kl_loss(s.detach().float().cuda(), t.cuda(), 1e5, loss)
        assert abs(float(loss) - 5.0 - float(ref)) < 2e-5 * abs(float(ref))
        assert float((d.cpu().double() - s.grad).abs().max()) < 2e-5 * float(s.grad.abs().max())


@pytest.mark.parametrize("case", ["cfg1", "ragged_a4"])
def test_kd_train_step_matches_oracle(case, monkeypatch):
    """kd_flag = 1: detection losses + kd_weight * KL(student || teacher) on x5, x6, x7, fused"""
    from disconet_amd import CoDetModule
    from oracle.train_ref import det_loss
    from oracle.teacher_ref import kd_loss
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup(case, "f16x3")
    ref.kd_flag = model.kd_flag = 1
    t_ref, t_hip, bevs_t = _teacher_pair(c)
    kd_weight = 1e5

    def oracle_step(m, tm, dt):
        m.train()
        res, x8, x7, x6, x5, fused = m(bevs, trans, na, c["batch"])
        with torch.no_grad():
            t8, t7, t6, t5, t3, t2 = tm(bevs_t.to(dt))
        l_cls, l_loc = det_loss(res, labels, targets, mask, norm=bevs.shape[0])
        l_kd = kd_loss((x5, x6, x7, fused), (t5, t6, t7, t3), kd_weight)
        (l_cls + l_loc + l_kd).backward()
        return float(l_cls.detach()), float(l_loc.detach()), float(l_kd.detach())

    orig = F.grid_sample
    monkeypatch.setattr(F, "grid_sample", lambda inp, grid, **kw: orig(inp, grid.to(inp.dtype), **kw))
    ref64, t64 = copy.deepcopy(ref).double(), copy.deepcopy(t_ref).double()
    for m_ in (ref64.u_encoder, t64.stpn):
        m_.conv_pre_1.register_forward_pre_hook(lambda m, inp: (inp[0].double(),))
    oracle_step(ref64, t64, torch.float64)
    monkeypatch.undo()
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    l_ref = oracle_step(ref, t_ref, torch.float32)

    mod = CoDetModule(model, t_hip, None, None, kd_flag=1, lr=1e-3)
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda(),
            "bev_seq_teacher": bevs_t.cuda(), "kd_weight": kd_weight}
    out = mod.step(data, c["batch"])
    assert abs(out["cls_loss"] - l_ref[0]) < 2e-5 * abs(l_ref[0])
    assert abs(out["loc_loss"] - l_ref[1]) < 2e-5 * abs(l_ref[1])
    assert abs(out["kd_loss"] - l_ref[2]) < 1e-4 * abs(l_ref[2]), (out, l_ref)
    rows = _grad_report(g64, ref, mod.engine, model)
    _assert_grads(rows)


def test_kd_through_autograd_node():
    """the reference's own KD step: loss on (result, x8, x7, x6, x5, fused) from model(...)"""
    from oracle.train_ref import det_loss
    from oracle.teacher_ref import kd_loss
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("cfg1", "f16x3")
    model.kd_flag = 1
    t_ref, t_hip, bevs_t = _teacher_pair(c)
    model.train()
    res, x8, x7, x6, x5, fused = model(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
    t8, t7, t6, t5, t3, t2 = t_hip(bevs_t.cuda())
    l_cls, l_loc = det_loss(res, labels.cuda(), targets.cuda(), mask.cuda(), norm=bevs.shape[0])
    l_kd = kd_loss((x5, x6, x7, fused), (t5, t6, t7, t3), 1e5)
    (l_cls + l_loc + l_kd).backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}

    from disconet_amd import CoDetModule, train_ops
    c2, ref2, model2, _, _ = _setup("cfg1", "f16x3")
    model2.kd_flag = 1
    mod = CoDetModule(model2, t_hip, kd_flag=1)
    eng = mod.engine
    with torch.no_grad():
        r = eng.forward(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
        _, dcls, dloc = train_ops.det_loss(r["cls"].reshape(-1, 2), labels.cuda().reshape(-1, 2).float(),
                                           r["loc"].reshape(-1, 6), targets.cuda().reshape(-1, 6).float(),
                                           mask.cuda().reshape(-1).float(), norm=bevs.shape[0])
        kd = torch.zeros(1, dtype=torch.float64, device="cuda")
        tn = t_hip.forward_nhwc(bevs_t.cuda())
        dkd = {k: train_ops.kd_kl_loss(eng.outs[k], t, 1e5, kd)
               for k, t in (("x5", tn[3]), ("x6", tn[2]), ("x7", tn[1]), ("fused", tn[4]))}
        eng.backward(dcls, dloc, dkd=dkd)
    assert abs(float(kd) - float(l_kd)) < 1e-4 * abs(float(l_kd))
    gmax = max(float(g.abs().max()) for g in grads.values())
    for name, p in model2.named_parameters():
        a, b = grads[name], eng.g(p)
        if float(b.abs().max()) < 1e-4 * gmax:
            assert float(a.abs().max()) < 1e-4 * gmax, name
            continue
        assert float((a - b).abs().max()) < 1e-3 * float(b.abs().max()), name


def test_checkpoint_resume_reproduces_the_next_step(tmp_path):
    """epoch_N.pth round trip: model_state_dict + the engine's optimizer state -> a fresh model
    resumed from the file takes the same next step"""
    from disconet_amd import CoDetModule
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("cfg1", "f16x3")
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    mod = CoDetModule(model, lr=1e-3)
    mod.step(data, c["batch"])
    path = str(tmp_path / "epoch_1.pth")
    torch.save({"epoch": 1, "model_state_dict": model.state_dict(),
                "optimizer_state_dict": mod.engine.state_dict()}, path)
    want = mod.step(data, c["batch"])

    c2, _, model2, _, _ = _setup("cfg1", "f16x3")
    ck = torch.load(path, weights_only=False)
    model2.load_state_dict({"module." + k: v for k, v in ck["model_state_dict"].items()})   # DataParallel keys
    mod2 = CoDetModule(model2, lr=123.0)
    mod2.engine.load_state_dict(ck["optimizer_state_dict"])
    got = mod2.step(data, c["batch"])
    assert abs(got["loss"] - want["loss"]) < 1e-4 * abs(want["loss"])
    for (n, a), (_, b) in zip(model.named_parameters(), model2.named_parameters()):
        assert float((a - b).abs().max()) < 2.1e-3, n      # at most one lr-sized sign flip of a noise gradient
    assert mod.scheduler_step(50) == 0.5e-3 and mod.scheduler_step(51) == 0.5e-3


def test_train_step_only_v2i(monkeypatch):
    """--only_v2i: vehicles exchange with the RSU (agent 0) only -- fewer warps / pairs per scene"""
    from disconet_amd import CoDetModule
    from oracle.train_ref import train_step
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("ragged_a4", "f16x3", only_v2i=True)
    g64 = _fp64_grads(ref, (bevs, trans, na), (labels, targets, mask), c["batch"], monkeypatch)
    l_ref = train_step(ref, torch.optim.Adam(ref.parameters(), lr=1e-3), bevs, trans, na, c["batch"],
                       labels, targets, mask)
    mod = CoDetModule(model, lr=1e-3)
    assert mod.engine is model.__dict__["_train_engine"]
    out = mod.step({"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                    "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()},
                   c["batch"])
    assert mod.engine.F["n_warps"] == 2 * 2 + 2 * 1          # scene 0: 3 live -> 4 warps, scene 1: 2 live -> 2
    assert abs(out["cls_loss"] - l_ref[0]) < 2e-5 * abs(l_ref[0])
    assert abs(out["loc_loss"] - l_ref[1]) < 2e-5 * abs(l_ref[1])
    rows = _grad_report(g64, ref, mod.engine, model)
    _assert_grads(rows)


def test_train_step_with_compression(monkeypatch):
    """--compress_level 2: the exchanged map goes through 1x1 compress (256 -> 64) / decompress"""
    from disconet_amd import CoDetModule
    from oracle.train_ref import train_step
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("cfg1", "f16x3", compress_level=2)
    g64 = _fp64_grads(ref, (bevs, trans, na), (labels, targets, mask), c["batch"], monkeypatch)
    l_ref = train_step(ref, torch.optim.Adam(ref.parameters(), lr=1e-3), bevs, trans, na, c["batch"],
                       labels, targets, mask)
    mod = CoDetModule(model, lr=1e-3)
    out = mod.step({"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                    "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()},
                   c["batch"])
    assert abs(out["cls_loss"] - l_ref[0]) < 2e-5 * abs(l_ref[0])
    assert abs(out["loc_loss"] - l_ref[1]) < 2e-5 * abs(l_ref[1])
    rows = _grad_report(g64, ref, mod.engine, model)
    assert "u_encoder.com_compresser.weight" in rows and "u_encoder.bn_decompress.bias" in rows
    _assert_grads(rows)


@pytest.mark.parametrize("layer", [4, 2, 1])
def test_train_step_fusion_on_another_layer(layer, monkeypatch):
    """--layer 4 / 2 / 1: the exchange and the DiscoGraph fusion sit on x4 (512 ch, behind the
    decoder's upsample), x2 (128 ch), x1 (64 ch)"""
    from disconet_amd import CoDetModule
    from oracle.train_ref import train_step
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("cfg1", "f16x3", layer=layer)
    g64 = _fp64_grads(ref, (bevs, trans, na), (labels, targets, mask), c["batch"], monkeypatch)
    l_ref = train_step(ref, torch.optim.Adam(ref.parameters(), lr=1e-3), bevs, trans, na, c["batch"],
                       labels, targets, mask)
    mod = CoDetModule(model, lr=1e-3)
    out = mod.step({"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                    "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()},
                   c["batch"])
    assert abs(out["cls_loss"] - l_ref[0]) < 2e-5 * abs(l_ref[0])
    assert abs(out["loc_loss"] - l_ref[1]) < 2e-5 * abs(l_ref[1])
    rows = _grad_report(g64, ref, mod.engine, model)
    _assert_grads(rows)


def test_train_step_scene_with_a_single_live_agent(monkeypatch):
    """live = [1, 2] of 3 slots: scene 0 has no neighbour to warp (its ego fuses with itself only),
    two slots per scene are padding that passes through the fusion untouched"""
    from disconet_amd import CoDetModule
    from oracle.train_ref import train_step
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("lonely_a3", "f16x3")
    g64 = _fp64_grads(ref, (bevs, trans, na), (labels, targets, mask), c["batch"], monkeypatch)
    l_ref = train_step(ref, torch.optim.Adam(ref.parameters(), lr=1e-3), bevs, trans, na, c["batch"],
                       labels, targets, mask)
    mod = CoDetModule(model, lr=1e-3)
    out = mod.step({"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                    "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()},
                   c["batch"])
    assert mod.engine.F["n_warps"] == 2 and mod.engine.F["n_calls"] == 1 + 2 * 2
    assert abs(out["cls_loss"] - l_ref[0]) < 2e-5 * abs(l_ref[0])
    assert abs(out["loc_loss"] - l_ref[1]) < 2e-5 * abs(l_ref[1])
    _assert_grads(_grad_report(g64, ref, mod.engine, model))
    ref_buf = dict(ref.named_buffers())
    for name, b in model.named_buffers():          # MLP BNs saw 5 calls, in the reference's order
        if "pixel_weighted_fusion" in name and name.endswith("num_batches_tracked"):
            assert int(b) == int(ref_buf[name]) == 5, name


def test_backward_after_a_second_forward_fails_loudly():
    """the HIP engine keeps ONE set of saved activations: a backward whose forward has been
    overwritten by a later forward (gradient accumulation over micro-batches, a no_grad monitoring
    pass in train() mode) must raise instead of returning gradients of the wrong batch"""
    from oracle.train_ref import det_loss
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup("cfg1", "f16x3")
    model.train()
    out1 = model(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
    with torch.no_grad():
        model(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])          # overwrites the saved state
    l_cls, l_loc = det_loss(out1, labels.cuda(), targets.cuda(), mask.cuda(), norm=bevs.shape[0])
    with pytest.raises(RuntimeError, match="saved activations"):
        (l_cls + l_loc).backward()
    # the straight sequence still works afterwards
    out2 = model(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
    l_cls, l_loc = det_loss(out2, labels.cuda(), targets.cuda(), mask.cuda(), norm=bevs.shape[0])
    (l_cls + l_loc).backward()
    assert all(p.grad is not None for p in model.parameters())


def test_moved_parameters_fail_loudly():
    """Parameters are views into the engine's flat buffer; re-pointing them (model.float(),
    load_state_dict(assign=True), ...) must be reported, not silently train a stale copy"""
    from disconet_amd import CoDetModule
    c, ref, model, (bevs, trans, na), _ = _setup("cfg1", "f16x3")
    mod = CoDetModule(model, optimizer=torch.optim.Adam(model.parameters(), lr=2e-3, betas=(0.8, 0.95), eps=1e-6))
    assert mod.engine.lr == 2e-3 and mod.engine.betas == (0.8, 0.95) and mod.engine.eps == 1e-6
    p = next(model.parameters())
    p.data = p.data.clone()                                   # what model.to()/.float() does
    with pytest.raises(RuntimeError, match="no longer aliases"):
        mod.engine.forward(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])


def test_kd_train_step_at_baseline_size_properties():
    """BASELINE configs[2] size (5 agents, batch 4, 256x256x13, kd_flag = 1), through properties that
    do not need the CPU oracle at this size: finite losses; the train()-mode forward of a scene
    does not depend on its batch slot's NEIGHBOURS' voxels beyond BatchNorm's batch statistics --
    checked in eval(); a zero kd_weight leaves exactly the gradients of the plain step; the KD term
    moves the gradient when switched on; a second step lowers the loss"""
    from disconet_amd import CoDetModule, Config, DiscoNet, TeacherNet
    from disconet_amd.synthetic import make_bevs, make_scene_batch, make_train_targets
    A, B, hw = 5, 4, 256
    torch.manual_seed(0)
    model = DiscoNet(Config(map_hw=hw), kd_flag=1, num_agent=A).cuda()
    teacher = TeacherNet(Config(map_hw=hw)).cuda().eval()
    bevs, trans, na = make_scene_batch(B, A, hw, jitter_seed=4)
    labels, targets, mask = make_train_targets(A * B, hw, p_fg=0.02)
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda(),
            "bev_seq_teacher": make_bevs(B, A, hw, p=0.05).cuda(), "kd_weight": 0.0}
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def grads_of(kd_flag, kd_weight):
        model.load_state_dict(state)
        mod = CoDetModule(model, teacher if kd_flag else None, kd_flag=kd_flag, lr=0.0)   # lr 0: Adam moves nothing
        out = mod.step(dict(data, kd_weight=kd_weight), B)
        return out, mod.engine.flat_g.clone()

    plain, g_plain = grads_of(0, 0.0)
    zero_kd, g_zero = grads_of(1, 0.0)
    with_kd, g_kd = grads_of(1, 1e5)
    for out in (plain, zero_kd, with_kd):
        assert all(v == v and abs(v) != float("inf") for v in out.values()), out
    assert zero_kd["kd_loss"] == 0.0 and with_kd["kd_loss"] > 0.0
    # the backward is deterministic (fixed-order per-channel sums and wave reductions since round 3; rigid poses take
    # the gather form of the warp backward): the same step twice gives the same bits ...
    plain2, g_plain2 = grads_of(0, 0.0)
    assert torch.equal(g_plain, g_plain2)
    # ... a zeroed teacher term adds exact zeros to the decoder's gradients, and the live KD term moves them
    gmax = float(g_plain.abs().max())
    noise = float((g_plain - g_zero).abs().max())
    assert noise <= 1e-6 * gmax
    assert float((g_kd - g_plain).abs().max()) > 1e-4 * gmax
    # training moves the loss down
    model.load_state_dict(state)
    mod = CoDetModule(model, teacher, kd_flag=1, lr=1e-3)
    first = mod.step(dict(data, kd_weight=1e5), B)
    second = mod.step(dict(data, kd_weight=1e5), B)
    assert second["loss"] < first["loss"]
    # eval(): batch-slot independence of the forward at this size, kd outputs included
    model.eval()
    with torch.no_grad():
        full = model(bevs.cuda(), trans.cuda(), na.cuda(), B)
        sel = torch.tensor([a * B + 1 for a in range(A)])
        one = model(bevs[sel].cuda(), trans[1:2].cuda(), na[1:2].cuda(), 1)
    assert torch.equal(full[0]["cls"][sel.cuda()], one[0]["cls"])
    for i in range(1, 6):
        assert torch.equal(full[i][sel.cuda()], one[i]), i


@pytest.mark.parametrize("case", ["cfg1", "ragged_a4"])
def test_steps_on_the_one_launch_weight_packs_are_the_per_layer_packs_steps_bit_for_bit(case, monkeypatch):
    """TrainEngine._pack_multi (default: every known weight form packed by one launch at the start of the forward) against the
    per-layer launches (DN_TRAIN_PACK_MULTI=0): seven steps from one state -- every parameter the same bits at the end, the losses
    equal (their last bits are the reduction's, not repeatable run to run); an in-place torch update the engine is not told about
    is picked up, update = False steps and a forced fallback pass read the same images."""
    from disconet_amd import CoDetModule
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup(case, "f16x3")
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(multi):
        monkeypatch.setenv("DN_TRAIN_PACK_MULTI", multi)
        model.load_state_dict(state)
        mod = CoDetModule(model, lr=1e-3)
        losses = [mod.step(data, c["batch"])["loss"] for _ in range(3)]
        with torch.no_grad():                            # an update the engine is not told about
            next(iter(model.parameters())).mul_(1.0009765625)
        losses.append(mod.step(data, c["batch"], update=False)["loss"])
        losses.append(mod.step(data, c["batch"], update=False)["loss"])
        mod.engine._force_range_flags = [1]              # one forced fallback: the second pass reads the same images
        losses.append(mod.step(data, c["batch"])["loss"])
        assert mod.engine.f32_fallback_steps == 1
        losses.append(mod.step(data, c["batch"])["loss"])
        sets = mod.engine.__dict__.get("_packset", {})
        return losses, mod.engine.flat_p.clone(), {e: ps["set"].n for e, ps in sets.items() if ps["set"] is not None}


    l0, p0, n0 = run("0")
    l1, p1, n1 = run("1")
    assert n0 == {} and n1["sp"] >= 20 and n1["nhwc"] >= 8, n1
    assert all(abs(a - b) <= 1e-12 * abs(a) for a, b in zip(l0, l1)) and abs(l0[3] - l0[4]) <= 1e-12 * abs(l0[3])
    assert torch.equal(p0.view(torch.int32), p1.view(torch.int32))


@pytest.mark.parametrize("case,kw", [("cfg1", {}), ("ragged_a4", {}), ("cfg1", {"dgrad_math": "f32", "wgrad_math": "f32"})])
def test_bias_gradient_folds_launched_together_give_the_same_gradients_bit_for_bit(case, kw, monkeypatch):
    """The folds of a pass's bias gradients in one launch behind it (train_ops.DeferredFolds, default) against a fold behind
    every sum (DN_TRAIN_DEFER_FOLDS=0): every gradient of three steps the same bits -- the first is the all-fp32 calibration pass,
    the third runs after a forced fallback (two passes, two fold launches)."""
    from disconet_amd import CoDetModule
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup(case, "f16x3")
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(defer):
        monkeypatch.setenv("DN_TRAIN_DEFER_FOLDS", defer)
        model.load_state_dict(state)
        mod = CoDetModule(model, lr=1e-3, **kw)
        grads = []
        for s in range(3):
            if s == 2:
                mod.engine._force_range_flags = [1]
            mod.step(data, c["batch"])
            grads.append(mod.engine.flat_g.clone())
        return grads, mod.engine.__dict__.get("_folds")

    g0, f0 = run("0")
    g1, f1 = run("1")
    assert f0 is None and f1 is not None and len(f1._ws) >= 20 and not f1._jobs
    for a, b in zip(g0, g1):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("case", ["cfg1", "ragged_a4"])
def test_steps_without_the_fp32_copy_of_dz_give_the_same_gradients_bit_for_bit(case, monkeypatch):
    """Default: where a layer's data gradient, weight gradient (dn_conv_wgrad_sp_z) and bias gradient all read dz through the
    BatchNorm backward's other outputs, its fp32 dz is not written.  Against DN_TRAIN_DZ_SP_ONLY=0 + DN_TRAIN_WGRAD_ZSP=0 (fp32
    dz written, the weight gradient splits it while staging): every gradient of four steps the same bits -- calibration step,
    two plain steps, a step with a forced fallback -- and the layers that skipped the copy are counted."""
    from disconet_amd import CoDetModule, train_ops as T
    c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup(case, "f16x3")
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
    state = {k: v.clone() for k, v in model.state_dict().items()}
    real = T.bn_backward
    skipped = {"n": 0}

    def counting(*a, **kw):
        out = real(*a, **kw)
        skipped["n"] += out is None
        return out

    def run(on):
        monkeypatch.setenv("DN_TRAIN_DZ_SP_ONLY", on)
        monkeypatch.setenv("DN_TRAIN_WGRAD_ZSP", on)
        model.load_state_dict(state)
        mod = CoDetModule(model, lr=1e-3)
        grads = []
        skipped["n"] = 0
        for s in range(4):
            if s == 3:
                mod.engine._force_range_flags = [1]
            mod.step(data, c["batch"])
            grads.append(mod.engine.flat_g.clone())
        return grads, skipped["n"]

    monkeypatch.setattr(T, "bn_backward", counting)
    g0, n0 = run("0")
    g1, n1 = run("1")
    assert n0 == 0 and n1 >= 2 * 15, (n0, n1)      # steps 1 and 2, fifteen or more layers each (step 0 calibrates, step 3 falls back)
    for a, b in zip(g0, g1):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_long_trajectory_split_f16_gradients_track_fp32_through_refreshes_and_an_overflow():
    """VERDICT round 5 (weak 6a/6b): CoDetModule.step must never throw on a finite loss, and no test crossed a lift refresh or
    an overflow.  160 steps from ONE seed on eight batches in rotation, three times:
      A  every gradient on the fp32 kernels, forward on the fp32-NHWC engine
      B  every gradient on the fp32 kernels, forward on the SP engine          (A vs B: two fp32-gradient runs that differ in
                                                                                a summation order only -- the noise floor)
      C  the default: split-f16 data / weight gradients, forward on the SP engine
    C crosses two periodic re-measurements of the gradient lifts (steps 64 and 154) and ONE REAL overflow: at step 90 the loss
    gradient handed to the backward is scaled by 1e4 (all runs), 40 x past the lifts' 256 x head room, so the pre-split dz
    copies clamp, the range guard trips, and backward() must drop the lifts, repeat the pass on the fp32 kernels from the
    saved activations and APPLY it.  Asserted: no step raised or was dropped (160 optimizer steps each); exactly the overflow
    step took the fp32 pass; the lifts were re-measured by it and again 64 steps later; the split-f16 loss curve stays within
    max(0.5 %, 1.5 x the A-vs-B floor over the same steps) of the fp32 curve for the first 40 steps and -- a training trajectory
    under Adam amplifies ANY rounding difference -- within max(2 %, 3 x the floor) over the whole run, never beyond 10 %; the
    run trains.  (Two builds of the kernels whose results differ in last bits give different curves: 0.4 % / 1.0 % (sp / floor,
    first 40 steps) and 3.8 % / 6.6 % overall before the lane-parallel warp gathers, 0.9 % / 0.9 % and 1.4 % / 2.4 % after.)
    The three curves go to gpurun_out/r06_trajectory.json."""
    from disconet_amd import CoDetModule, Config, DiscoNet, train_ops as T
    from disconet_amd.synthetic import make_scene_batch, make_train_targets
    A, B, hw, steps, k_over = 2, 2, 128, 160, 90
    batches = []
    for i in range(8):
        bevs, trans, na = make_scene_batch(B, A, hw, jitter_seed=i)
        g = torch.Generator().manual_seed(100 + i)
        bevs = (torch.rand(bevs.shape, generator=g) < 0.03).float()
        labels, targets, mask = make_train_targets(A * B, hw, p_fg=0.03)
        batches.append({"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": labels.cuda(),
                        "reg_targets": (targets + 0.05 * i).cuda(), "reg_loss_mask": mask.cuda()})
    real_det_loss = T.det_loss
    scale = {"v": 1.0}

    def scaled_det_loss(*a, **kw):
        losses, dcls, dloc = real_det_loss(*a, **kw)
        if scale["v"] != 1.0:
            dcls, dloc = dcls * scale["v"], dloc * scale["v"]
        return losses, dcls, dloc

    curves, engines = {}, {}
    T.det_loss = scaled_det_loss
    try:
        for name, grad, fwd in (("A", "f32", "nhwc"), ("B", "f32", "sp"), ("C", "sp", "sp")):
            torch.manual_seed(7)
            model = DiscoNet(Config(map_hw=hw), kd_flag=0, num_agent=A)
            for m in model.modules():      # O(1) activations and gradients (the parity tests' init)
                if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d)):
                    torch.nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
            model.conv_math = "f16x3"
            model.cuda()
            mod = CoDetModule(model, lr=2e-4, dgrad_math=grad, wgrad_math=grad)
            mod.engine.fwd_math = fwd
            out = []
            for s in range(steps):
                scale["v"] = 1e4 if s == k_over else 1.0
                out.append(mod.step(batches[s % len(batches)], B)["loss"])      # must not raise
            curves[name], engines[name] = out, mod.engine
    finally:
        T.det_loss = real_det_loss
    for name in "ABC":
        assert engines[name].step_count == steps                            # no step dropped
    assert engines["A"].f32_fallback_steps == 0 and engines["B"].f32_fallback_steps == 0
    esp = engines["C"]
    assert esp.f32_fallback_steps == 1 and esp.last_fallback_step == k_over, (esp.f32_fallback_steps, esp.last_fallback_step)
    assert len(esp._dz_lift) >= 15                                        # the lifts were re-measured by the fp32 pass
    measured_at = {v[1] for v in esp._dz_lift.values()}
    assert max(measured_at) >= k_over + 64 - 1, measured_at               # ... and again 64 steps later (a periodic refresh)
    t = {k: torch.tensor(v, dtype=torch.float64) for k, v in curves.items()}
    for v in t.values():
        assert torch.isfinite(v).all()
    rel = lambda x, y: ((x - y).abs() / y.abs().clamp_min(1e-12))
    floor, dev = rel(t["A"], t["B"]), rel(t["C"], t["B"])
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r06_trajectory.json"), "w") as f:
        json.dump({"steps": steps, "overflow_step": k_over, "curves": curves, "max_rel_sp_vs_f32": float(dev.max()),
                   "max_rel_f32_nhwc_vs_f32_sp (floor)": float(floor.max()), "first40_sp_vs_f32": float(dev[:40].max()),
                   "first40_floor": float(floor[:40].max())}, f)
    assert float(dev[:40].max()) < max(5e-3, 1.5 * float(floor[:40].max())), (float(dev[:40].max()), float(floor[:40].max()))
    bound = max(0.02, 3.0 * float(floor.max()))
    assert float(dev.max()) < min(bound, 0.10), (int(dev.argmax()), float(dev.max()), float(floor.max()))
    assert t["C"][-8:].mean() < 0.7 * t["C"][:8].mean()                    # and the run trains


@pytest.mark.parametrize("case", ["cfg1", "ragged_a4"])
def test_training_forward_on_the_sp_engine_agrees_with_the_nhwc_engine(case, monkeypatch):
    """Round 6: the training forward's convs on the inference engine's split-f16 LDS-DMA kernels (fwd_math = "sp", the default:
    every BatchNorm apply writes y a second time as an SP tensor, dn_spconv2d_nhwc reads it and writes z as fp32 rows) against
    rounds 2-5's fp32-NHWC engine with the split on the VALU (fwd_math = "nhwc"): the same arithmetic in another summation
    order -- losses to 1e-6, the outputs to 2e-5 of their magnitude, and the gradients of BOTH under the float64-oracle
    criteria; the SP path must really have run (an SP twin on every conv layer's input)."""
    from disconet_amd import CoDetModule
    outs = {}
    for mode in ("nhwc", "sp"):
        c, ref, model, (bevs, trans, na), (labels, targets, mask) = _setup(case, "f16x3")
        mod = CoDetModule(model, lr=1e-3)
        mod.engine.fwd_math = mode
        data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
                "labels": labels.cuda(), "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda()}
        out = mod.step(data, c["batch"], update=False)
        eng = mod.engine
        twins = sum(1 for lay in eng.L.values() if getattr(lay.ctx["src0"], "_dn_sp", None) is not None)
        outs[mode] = (out, eng.last_result["cls"].clone(), eng.last_result["loc"].clone(), twins, eng, model, ref)
    assert outs["nhwc"][3] == 0 and outs["sp"][3] >= 18, (outs["nhwc"][3], outs["sp"][3])
    for k in ("cls_loss", "loc_loss"):
        assert abs(outs["sp"][0][k] - outs["nhwc"][0][k]) <= 1e-6 * abs(outs["nhwc"][0][k]), k
    for i in (1, 2):
        a, b = outs["sp"][i], outs["nhwc"][i]
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    c, ref, _, inputs, tg = _setup(case, "f16x3")
    g64 = _fp64_grads(ref, inputs, tg, c["batch"], monkeypatch)
    from oracle.train_ref import train_step
    train_step(ref, torch.optim.Adam(ref.parameters(), lr=1e-3), *inputs, c["batch"], *tg)      # fills ref's fp32 gradients
    for mode in ("nhwc", "sp"):
        _assert_grads(_grad_report(g64, ref, outs[mode][4], outs[mode][5]))


def test_fallback_pass_is_the_fp32_engine_bit_for_bit_with_kd():
    """The second, all-fp32 backward that replaces a pass with a clamped gradient map (TrainEngine.backward) must be exactly the
    fp32 engine's pass -- it reads the same saved activations and its own arguments, nothing the first pass wrote (the KD
    gradient of the fused map is ADDED to the decoder's: a copy, since round 6).  kd_flag = 1, two engines from one state:
    A (split-f16 gradients) takes the calibration step, then a step whose range poll is forced to report a clamp;
    B (dgrad = wgrad = "f32") takes the same two steps.  Parameters, Adam moments and BatchNorm buffers equal bit for bit."""
    from disconet_amd import CoDetModule
    c, ref, model_a, (bevs, trans, na), (labels, targets, mask) = _setup("cfg1", "f16x3")
    _, _, model_b, _, _ = _setup("cfg1", "f16x3")
    _, t_hip, bevs_t = _teacher_pair(c)
    data = {"bev_seq": bevs.cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(), "labels": labels.cuda(),
            "reg_targets": targets.cuda(), "reg_loss_mask": mask.cuda(), "bev_seq_teacher": bevs_t.cuda(), "kd_weight": 1e5}
    model_a.kd_flag = model_b.kd_flag = 1
    mod_a = CoDetModule(model_a, t_hip, None, None, kd_flag=1, lr=1e-3, dgrad_math="sp", wgrad_math="sp")
    mod_b = CoDetModule(model_b, t_hip, None, None, kd_flag=1, lr=1e-3, dgrad_math="f32", wgrad_math="f32")
    outs = []
    for mod in (mod_a, mod_b):
        o1 = mod.step(data, c["batch"])
        if mod is mod_a:
            assert len(mod.engine._dz_lift) > 10
            mod.engine._force_range_flags = [1]
        o2 = mod.step(data, c["batch"])
        outs.append((o1, o2))
    assert mod_a.engine.f32_fallback_steps == 1 and mod_b.engine.f32_fallback_steps == 0
    assert torch.equal(mod_a.engine.flat_p, mod_b.engine.flat_p)
    assert torch.equal(mod_a.engine.flat_m, mod_b.engine.flat_m) and torch.equal(mod_a.engine.flat_v, mod_b.engine.flat_v)
    for (n1, b1), (n2, b2) in zip(model_a.named_buffers(), model_b.named_buffers()):
        assert n1 == n2 and torch.equal(b1, b2), n1
    for k in ("cls_loss", "loc_loss", "kd_loss"):
        for i in (0, 1):
            assert abs(outs[0][i][k] - outs[1][i][k]) <= 1e-12 * abs(outs[1][i][k]), (k, i)
