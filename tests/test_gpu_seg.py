"""Segmentation variant (BASELINE.json configs[3]) through the C ABI against
the CPU oracle (oracle/seg_ref.py) and the committed goldens.  Tolerance 1e-4 absolute on
kaiming-initialised weights (activations O(1)), as for the det model."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_maxpool_and_bilinear_upsample_vs_torch():
    from disconet_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 40, 12, 20, generator=g)                       # NCHW, 40 channels (2.5 chunks)
    sp = ops.SpTensor.from_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda())
    pooled = ops.sp_maxpool2(sp)
    assert pooled.shape == (3, 6, 10, 40)
    # exact: the (hi, lo) pair of the maximum is copied
    assert torch.equal(pooled.nhwc().cpu().permute(0, 3, 1, 2), F.max_pool2d(sp.nhwc().cpu().permute(0, 3, 1, 2), 2))
    up = ops.sp_upsample2_bilinear(sp)
    want = F.interpolate(sp.nhwc().cpu().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    got = up.nhwc().cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape == (3, 40, 24, 40)
    assert (got - want).abs().max().item() <= 1e-5      # fp32 lerp + the 22-bit hi/lo split of O(4) values


def test_cross_entropy_kernel_vs_torch():
    from disconet_amd import ops
    g = torch.Generator().manual_seed(3)
    z = (torch.randn(2, 50, 36, 8, generator=g) * 3).requires_grad_(True)
    y = torch.randint(0, 8, (2, 50, 36), generator=g)
    want = F.cross_entropy(z.permute(0, 3, 1, 2), y)
    want.backward()
    loss, grad = ops.seg_ce_loss(z.detach().cuda().contiguous(), y.cuda())
    assert abs(float(loss) - float(want)) <= 1e-6 * abs(float(want))
    assert (grad.cpu() - z.grad).abs().max().item() <= 1e-9 + 1e-6 * z.grad.abs().max().item()


@pytest.mark.parametrize("classes", [8, 5])
def test_cross_entropy_ignore_index_and_class_counts(classes):
    """nn.CrossEntropyLoss semantics: the mean is over the NON-ignored pixels (ignore_index = -100), ignored
    pixels get zero gradient rows, labels outside [0, classes) that are not -100 are refused (torch raises)."""
    from disconet_amd import ops
    from disconet_amd._lib import DnError
    g = torch.Generator().manual_seed(11)
    z = (torch.randn(2, 30, 20, classes, generator=g) * 2).requires_grad_(True)
    y = torch.randint(0, classes, (2, 30, 20), generator=g)
    y[torch.rand(2, 30, 20, generator=g) < 0.3] = -100
    want = F.cross_entropy(z.permute(0, 3, 1, 2), y)
    want.backward()
    loss, grad = ops.seg_ce_loss(z.detach().cuda().contiguous(), y.cuda())
    assert abs(float(loss) - float(want)) <= 1e-6 * abs(float(want))
    assert (grad.cpu() - z.grad).abs().max().item() <= 1e-9 + 1e-6 * z.grad.abs().max().item()
    assert torch.equal(grad.cpu()[y == -100], torch.zeros_like(grad.cpu()[y == -100]))
    y_bad = y.clone()
    y_bad[0, 0, 0] = classes
    with pytest.raises(DnError):
        ops.seg_ce_loss(z.detach().cuda().contiguous(), y_bad.cuda())


@pytest.mark.parametrize("case", list(cases.SEG_CASES))
def test_seg_model_vs_oracle_and_golden(case, golden_dir):
    from disconet_amd import SegDiscoNet, SegModule
    c = cases.SEG_CASES[case]
    ref = cases.seg_ref_model(c["agents"], kd_flag=True)
    want, want_loss = cases.run_seg_ref(case, ref)
    x, trans, na, labels = cases.seg_inputs(case)
    m = SegDiscoNet(num_agent=c["agents"], kd_flag=True).eval()
    m.load_state_dict({"module." + k: v for k, v in ref.state_dict().items()})
    m.cuda()
    with torch.no_grad():
        logits, x9, x8, x7, x6, x5, fused = m(x.cuda(), trans.cuda(), na.cuda(), c["batch"])
    got = {"logits": logits.cpu(), "x9": x9.cpu(), "x6": x6.cpu(), "fused": fused.cpu()}
    g = np.load(os.path.join(golden_dir, "seg_cases.npz"))
    for name in want:
        assert got[name].shape == want[name].shape, name
        err = (got[name] - want[name]).abs().max().item()
        assert err <= TOL, "%s/%s max abs err %.3e" % (case, name, err)
        assert np.abs(cases.seg_subsample(name, got[name]) - g["%s/%s" % (case, name)]).max() <= TOL
    loss, dlogits = SegModule(m).loss(logits, labels.cuda())
    assert abs(loss - want_loss) <= 1e-4 * abs(want_loss)
    assert abs(loss - float(g["%s/loss" % case])) <= 1e-4 * abs(loss)
    z = want["logits"].clone().requires_grad_(True)
    F.cross_entropy(z, labels).backward()
    assert (dlogits.cpu() - z.grad).abs().max().item() <= 1e-4 * z.grad.abs().max().item()


def test_seg_at_baseline_size_properties():
    """configs[3] size (5 agents, 256 x 256 x 13): finite logits, a scene's result does not depend
    on its batch slot (bitwise), identical agents under identity poses fuse to themselves, and the
    split-planar voxel batch is accepted in place of the dense tensor"""
    from disconet_amd import SegDiscoNet, ops
    from disconet_amd.synthetic import make_scene_batch, make_sparse_scene_batch, make_trans_matrices
    A, B, hw = 5, 2, 256
    torch.manual_seed(0)
    m = SegDiscoNet(num_agent=A).eval().cuda()
    bevs, trans, na = make_scene_batch(B, A, hw, jitter_seed=3)
    x = bevs[:, 0].permute(0, 3, 1, 2)
    with torch.no_grad():
        full = m(x.cuda(), trans.cuda(), na.cuda(), B)
        sel = torch.tensor([a * B + 1 for a in range(A)])
        one = m(x[sel].cuda(), trans[1:2].cuda(), na[1:2].cuda(), 1)
    assert full.shape == (A * B, 8, hw, hw) and torch.isfinite(full).all()
    assert torch.equal(full[sel.cuda()], one)
    # identity poses + identical agents: the fused bottleneck equals the un-fused one
    xx = x[:1].repeat(A, 1, 1, 1)
    eye = torch.eye(4).repeat(1, A, A, 1, 1)
    mk = SegDiscoNet(num_agent=A, kd_flag=True).eval().cuda()
    mk.load_state_dict(m.state_dict())
    with torch.no_grad():
        outs = mk(xx.cuda(), eye.cuda(), torch.full((1, A), A).cuda(), 1)
        alone = mk(xx.cuda(), eye.cuda(), torch.full((1, A), 1).cuda(), 1)     # one live agent: no fusion
    assert (outs[-1][0] - alone[-1][0]).abs().max().item() <= 1e-5
    # sparse voxel lists scattered straight into the engine's layout
    indices, offsets, _ = make_sparse_scene_batch(1, A, hw)
    tr1 = make_trans_matrices(1, A, jitter_seed=5).cuda()
    na1 = torch.full((1, A), A).cuda()
    with torch.no_grad():
        dense = ops.scatter_dense(indices.cuda(), offsets.cuda(), A, (hw, hw, 13))
        a = m(dense[:, 0].permute(0, 3, 1, 2), tr1, na1, 1)
        b = m(ops.scatter_dense_sp(indices.cuda(), offsets.cuda(), A, (hw, hw, 13)), tr1, na1, 1)
    assert torch.equal(a, b)


def test_seg_forward_in_train_mode_fails_loudly():
    """training runs through SegModule.step (the explicit HIP training graph, tests/test_gpu_seg_train.py); the module's
    forward() is the eval plan and refuses train() mode instead of silently using running statistics"""
    from disconet_amd import SegDiscoNet
    m = SegDiscoNet(num_agent=2).cuda().train()
    x = torch.zeros(2, 13, 32, 32).cuda()
    with pytest.raises(NotImplementedError, match="SegModule.step"):
        m(x, torch.eye(4).repeat(1, 2, 2, 1, 1).cuda(), torch.full((1, 2), 2).cuda(), 1)
