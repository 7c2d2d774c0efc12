"""Training-step kernels (include/disconet_train.h) against torch-CPU autograd of the same
op in float64 (these are floating-point kernels; the reference is the op's own definition,
the whole-step parity against the oracle model is tests/test_gpu_train_step.py).

Tolerances are relative to the largest reference magnitude of each tensor: 2e-5 for the
exact-fp32 kernels (fp32 accumulation over up to 1e5 terms), 1e-4 where the split-f16
conv engine takes part."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def rel_err(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


WGRAD_CASES = [
    # n, h, w, c0, c1, up0, c_out, k, stride
    (2, 32, 32, 32, 0, 0, 32, 3, 1),
    (2, 20, 24, 13, 0, 0, 32, 3, 1),        # voxel input: 13 channels, ragged tiles
    (3, 32, 32, 64, 0, 0, 128, 3, 2),
    (2, 18, 22, 32, 0, 0, 64, 3, 2),
    (2, 32, 32, 64, 32, 1, 32, 3, 1),       # decoder: up(64) || skip(32)
    (2, 16, 16, 512, 256, 1, 256, 3, 1),    # conv5_1's channel structure
    (2, 32, 32, 64, 0, 0, 64, 1, 1),        # Conv3D(1,1,1)
    (2, 32, 32, 32, 0, 0, 12, 1, 1),        # cls head conv2
    (4, 32, 32, 8, 0, 0, 1, 1, 1),          # attention MLP's last layer
    (1, 256, 256, 32, 0, 0, 32, 3, 1),      # full-resolution layer, many slices
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_matches_autograd(case):
    from disconet_amd import ops, train_ops
    n, h, w, c0, c1, up0, c_out, k, stride = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    hs, ws = (h // 2, w // 2) if up0 else (h, w)
    x0 = torch.randn(n, c0, hs, ws, generator=g)
    x1 = torch.randn(n, c1, h, w, generator=g) if c1 else None
    wgt = (torch.randn(c_out, c0 + c1, k, k, generator=g) * 0.1).double().requires_grad_(True)
    xin = F.interpolate(x0, scale_factor=2) if up0 else x0
    if c1:
        xin = torch.cat([xin, x1], 1)
    z = F.conv2d(xin.double(), wgt, None, stride, k // 2)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz.double())

    d = ops.conv_desc(n, h, w, c0, c_out, ksize=k, stride=stride, c1=c1, up0=up0, relu=False)
    dw = torch.full((c_out, c0 + c1, k, k), 7.0, device=_dev())
    train_ops.conv_wgrad(d, nhwc(x0).to(_dev()), nhwc(x1).to(_dev()) if c1 else None,
                         nhwc(dz).to(_dev()), dw)
    assert rel_err(dw, wgt.grad) < 2e-5
    # accumulate adds to what is there
    train_ops.conv_wgrad(d, nhwc(x0).to(_dev()), nhwc(x1).to(_dev()) if c1 else None,
                         nhwc(dz).to(_dev()), dw, accumulate=True)
    assert rel_err(dw, 2 * wgt.grad) < 2e-5


WGRAD_SP_CASES = [
    # n, h, w, c0, c1, up0, c_out -- 3x3 stride 1; the split-f16 kernel's block (64 / 32) follows the channel counts
    (2, 32, 32, 64, 0, 0, 64),
    (2, 20, 24, 128, 0, 0, 64),         # ragged tiles: 24 = 1.5 x 16 columns, 20 = 5 x 4 rows
    (2, 16, 16, 512, 256, 1, 256),      # conv5_1's channel structure: up(512) || skip(256)
    (2, 32, 32, 64, 64, 1, 192),        # c_out = 3 x 64
    (1, 64, 64, 64, 32, 1, 32),         # conv8_1's structure -> 32-channel blocks, up + concat
    (2, 20, 40, 32, 0, 0, 32),          # 32 -> 32, ragged (20 = 2.5 x 8 rows, 40 = 2.5 x 16 columns)
    (1, 32, 32, 32, 0, 0, 64),          # the heads' first convs
    (1, 256, 256, 32, 0, 0, 32),        # full-resolution layer, many slices
    (2, 40, 48, 13, 0, 0, 32),          # conv_pre_1: the 13-channel voxel grid (rows not 16-byte loadable: dword loads)
    (1, 32, 32, 72, 0, 0, 40),          # one source ending in a partial block; c_out = 32 + 8
    (2, 32, 32, 64, 0, 0, 128, 2),      # stride 2 (conv2_1's structure): 64-channel blocks, 1 x 16 output tiles
    (2, 18, 22, 32, 0, 0, 64, 2),       # stride 2, 32-channel blocks (conv1_1), ragged: 9 x 11 outputs
    (1, 17, 33, 128, 0, 0, 64, 2),      # stride 2, odd input sizes
    (3, 16, 64, 32, 0, 0, 32, 2),       # stride 2, 32 -> 32
]


@pytest.mark.parametrize("case", WGRAD_SP_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_split_f16_matches_autograd(case):
    """dn_conv_wgrad_sp (f16 hi + lo operands split while staging, three MFMAs per product) against float64 autograd: the
    fp32 kernels' 2e-5, with a gradient map of training-like magnitude (1e-4) under the engine's lift rule."""
    import math
    from disconet_amd import ops, train_ops
    n, h, w, c0, c1, up0, c_out = case[:7]
    stride = case[7] if len(case) > 7 else 1
    g = torch.Generator().manual_seed(hash(case) % 1000)
    hs, ws = (h // 2, w // 2) if up0 else (h, w)
    x0 = torch.randn(n, c0, hs, ws, generator=g)
    x1 = torch.randn(n, c1, h, w, generator=g) if c1 else None
    wgt = (torch.randn(c_out, c0 + c1, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
    xin = F.interpolate(x0, scale_factor=2) if up0 else x0
    if c1:
        xin = torch.cat([xin, x1], 1)
    z = F.conv2d(xin.double(), wgt, None, stride, 1)
    dz = torch.randn(z.shape, generator=g) * 1e-4
    z.backward(dz.double())
    d = ops.conv_desc(n, h, w, c0, c_out, ksize=3, stride=stride, c1=c1, up0=up0, relu=False)
    assert train_ops.conv_wgrad_sp_supported(d)
    lift = float(2.0 ** (8 - math.floor(math.log2(float(dz.abs().max())))))
    a0, a1, adz = nhwc(x0).to(_dev()), nhwc(x1).to(_dev()) if c1 else None, nhwc(dz).to(_dev())
    dw = torch.full((c_out, c0 + c1, 3, 3), 7.0, device=_dev())
    train_ops.conv_wgrad(d, a0, a1, adz, dw, sp_lift=lift)
    ref = wgt.grad
    per_tap = [rel_err(dw[:, :, t // 3, t % 3], ref[:, :, t // 3, t % 3]) for t in range(9)]
    per_src = [rel_err(dw[:, :c0], ref[:, :c0])] + ([rel_err(dw[:, c0:], ref[:, c0:])] if c1 else [])
    assert rel_err(dw, ref) < 2e-5, (per_tap, per_src)
    first = dw.clone()
    train_ops.conv_wgrad(d, a0, a1, adz, dw, sp_lift=lift)
    assert torch.equal(dw, first)                       # fixed summation order: bitwise repeatable
    train_ops.conv_wgrad(d, a0, a1, adz, dw, accumulate=True, sp_lift=lift)
    assert rel_err(dw, 2 * ref) < 2e-5
    fp32 = torch.empty_like(dw)
    train_ops.conv_wgrad(d, a0, a1, adz, fp32)
    assert rel_err(dw, 2 * fp32.double().cpu()) < 2e-5      # and against the exact-fp32 kernels
    assert ops.sp_range_flags(reset=True) & 5 == 0
    if c_out % 16 == 0:
        # dz from its SP copy (dn_conv_wgrad_sp_z; the tensor the BatchNorm backward writes for the data gradient): the same halves
        # the fp32 form derives while staging -- the same bits, with and without the fp32 tensor at hand
        zsp = ops.SpTensor.from_nhwc(adz * lift)
        for dz_arg in (adz, None):
            via_sp = torch.full_like(dw, -3.0)
            train_ops.conv_wgrad(d, a0, a1, dz_arg, via_sp, sp_lift=lift, dz_sp=zsp)
            assert torch.equal(via_sp.view(torch.int32), first.view(torch.int32))


BASELINE_LAYERS = [
    # name, map, c0, c1, up0, c_out, stride: every 3x3 layer of the detector at BASELINE configs[1]'s size (20 images of 256 x 256)
    ("conv_pre_1", 256, 13, 0, 0, 32, 1), ("conv_pre_2", 256, 32, 0, 0, 32, 1), ("conv1_1", 256, 32, 0, 0, 64, 2),
    ("conv1_2", 128, 64, 0, 0, 64, 1), ("conv2_1", 128, 64, 0, 0, 128, 2), ("conv2_2", 64, 128, 0, 0, 128, 1),
    ("conv3_1", 64, 128, 0, 0, 256, 2), ("conv3_2", 32, 256, 0, 0, 256, 1), ("conv4_1", 32, 256, 0, 0, 512, 2),
    ("conv4_2", 16, 512, 0, 0, 512, 1), ("conv5_1", 32, 512, 256, 1, 256, 1), ("conv6_1", 64, 256, 128, 1, 128, 1),
    ("conv7_1", 128, 128, 64, 1, 64, 1), ("conv8_1", 256, 64, 32, 1, 32, 1), ("heads1", 256, 32, 0, 0, 64, 1),
]


@pytest.mark.parametrize("layer", BASELINE_LAYERS, ids=lambda l: l[0])
def test_conv_wgrad_split_f16_equals_the_fp32_kernels_at_baseline_size(layer):
    """the benchmarked batch (5 agents x batch 4, 256 x 256): every 3x3 layer's weight gradient by dn_conv_wgrad_sp against the
    exact-fp32 kernels on the same operands (2e-5 of the largest entry: both sit ~1e-6 from the float64 sum), bitwise repeatable,
    no range flag -- post-ReLU-like activations, a 1e-4-sized gradient map under the engine's lift rule"""
    import math
    from disconet_amd import ops, train_ops
    name, hw, c0, c1, up0, c_out, stride = layer
    n = 20
    g = torch.Generator(device=_dev()).manual_seed(hash(name) % 1000)
    hs = hw // 2 if up0 else hw
    x0 = torch.randn(n, hs, hs, c0, generator=g, device=_dev()).clamp_(min=0) if c0 != 13 else \
        (torch.rand(n, hs, hs, c0, generator=g, device=_dev()) < 0.05).float()
    x1 = torch.randn(n, hw, hw, c1, generator=g, device=_dev()).clamp_(min=0) if c1 else None
    ho = hw // stride
    dz = torch.randn(n, ho, ho, c_out, generator=g, device=_dev()) * 1e-4
    d = ops.conv_desc(n, hw, hw, c0, c_out, ksize=3, stride=stride, c1=c1, up0=up0, relu=False)
    assert train_ops.conv_wgrad_sp_supported(d)
    lift = float(2.0 ** (8 - math.floor(math.log2(float(dz.abs().max())))))
    ops.sp_range_flags(reset=True)
    want = torch.empty(c_out, c0 + c1, 3, 3, device=_dev())
    train_ops.conv_wgrad(d, x0, x1, dz, want)
    got = torch.empty_like(want)
    train_ops.conv_wgrad(d, x0, x1, dz, got, sp_lift=lift)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    again = torch.empty_like(want)
    train_ops.conv_wgrad(d, x0, x1, dz, again, sp_lift=lift)
    assert torch.equal(got, again)
    via_sp = torch.empty_like(want)                        # dz from its SP copy: the same bits
    train_ops.conv_wgrad(d, x0, x1, None, via_sp, sp_lift=lift, dz_sp=ops.SpTensor.from_nhwc(dz * lift))
    assert torch.equal(got.view(torch.int32), via_sp.view(torch.int32))
    assert ops.sp_range_flags(reset=True) & 5 == 0


def test_conv_wgrad_split_f16_refuses_other_layers_and_flags_an_outgrown_lift():
    from disconet_amd import _lib, ops, train_ops
    for kw in (dict(c0=48, c_out=64, stride=2), dict(c0=64, c_out=64, ksize=1), dict(c0=64, c_out=16)):
        d = ops.conv_desc(1, 32, 32, kw["c0"], kw["c_out"], ksize=kw.get("ksize", 3), stride=kw.get("stride", 1), relu=False)
        assert not train_ops.conv_wgrad_sp_supported(d)
    d = ops.conv_desc(1, 32, 32, 64, 64, ksize=1, relu=False)
    with pytest.raises(_lib.DnError):
        train_ops.conv_wgrad(d, torch.zeros(1, 32, 32, 64, device=_dev()), None, torch.zeros(1, 32, 32, 64, device=_dev()),
                             torch.zeros(64, 64, 1, 1, device=_dev()), sp_lift=1.0)
    d = ops.conv_desc(1, 32, 32, 64, 64, ksize=3, relu=False)
    x, dz, dw = (torch.ones(1, 32, 32, 64, device=_dev()), torch.ones(1, 32, 32, 64, device=_dev()),
                 torch.zeros(64, 64, 3, 3, device=_dev()))
    with pytest.raises(_lib.DnError):
        train_ops.conv_wgrad(d, x, None, dz, dw, sp_lift=3.0)            # not a power of two
    ops.sp_range_flags(reset=True)
    train_ops.conv_wgrad(d, x, None, dz, dw, sp_lift=2.0 ** 17)          # 1 * 2^17 > 65504: clamped and flagged
    assert ops.sp_range_flags(reset=True) & 1


def test_conv_wgrad_column_block_and_strided_dz():
    """the attention MLP's W1 = [W_ego | W_nbr]: dW written into a column block; dz as a
    channel slice of a wider tensor"""
    from disconet_amd import ops, train_ops
    g = torch.Generator().manual_seed(3)
    n, h, w, c_in, c_out = 3, 32, 32, 256, 128
    x = torch.randn(n, h, w, c_in, generator=g)
    dz_wide = torch.randn(n, h, w, c_out + 32, generator=g)
    dz = dz_wide[..., 16:16 + c_out]
    ref = torch.einsum("nhwo,nhwi->oi", dz.double(), x.double())
    dw = torch.zeros(c_out, 2 * c_in, 1, 1, device=_dev())
    d = ops.conv_desc(n, h, w, c_in, c_out, ksize=1, relu=False, ldo=c_out + 32)
    dzg = dz_wide.to(_dev())
    train_ops.conv_wgrad(d, x.to(_dev()), None, dzg[..., 16:16 + c_out], dw[:, c_in:], dw_cin_total=2 * c_in)
    assert rel_err(dw[:, c_in:, 0, 0], ref) < 2e-5
    assert float(dw[:, :c_in].abs().max()) == 0.0


DGRAD_CASES = [
    (2, 32, 32, 64, 64, 3, 1), (2, 32, 32, 32, 64, 3, 2), (2, 20, 24, 128, 256, 3, 2),
    (2, 32, 32, 96, 32, 3, 1), (2, 32, 32, 64, 64, 1, 1), (2, 32, 32, 32, 36, 1, 1),
]


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("case", DGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad_through_the_forward_engine(case, math):
    """dx = conv(dz, flipped / transposed weights), stride-2 layers read dz zero-stuffed"""
    from disconet_amd import ops, train_ops
    n, h, w, c_in, c_out, k, stride = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, c_in, h, w, generator=g).double().requires_grad_(True)
    wgt = torch.randn(c_out, c_in, k, k, generator=g) * 0.1
    z = F.conv2d(x, wgt.double(), None, stride, k // 2)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz.double())

    wt = train_ops.dgrad_weights(wgt.to(_dev()))
    d = ops.conv_desc(n, h, w, c_out, c_in, ksize=k, stride=1, up0=2 if stride == 2 else 0,
                      relu=False, math=math)
    packed = ops.pack_conv_weights(d, wt)
    one = torch.ones(c_in, device=_dev())
    zero = torch.zeros(c_in, device=_dev())
    dx = ops.conv2d(d, nhwc(dz).to(_dev()), packed, one, zero)
    assert rel_err(dx, nhwc(x.grad)) < (2e-5 if math == "f32" else 1e-4)


@pytest.mark.parametrize("n,h,w,c_in,c_out,ci_first,take", [(2, 32, 32, 32, 64, 0, None), (3, 20, 24, 128, 256, 0, None),
                                                            (2, 64, 64, 64, 128, 0, None), (2, 16, 24, 96, 48, 32, 64)])
def test_parity_phase_stride2_dgrad(n, h, w, c_in, c_out, ci_first, take):
    """data gradient of a stride-2 3x3 layer as four tap-masked stride-1 convs over dz, one per input-pixel parity
    class (dn_conv_dgrad_class_weights + dn_conv2d_taps): against float64 autograd, and against the zero-stuffed form
    (same exact-fp32 products, another summation order); a column block of a wider weight; a strided dx destination"""
    from disconet_amd import ops, train_ops
    g = torch.Generator().manual_seed(5 + h)
    x = torch.randn(n, c_in, h, w, generator=g).double().requires_grad_(True)
    wgt = torch.randn(c_out, c_in, 3, 3, generator=g) * 0.1
    z = F.conv2d(x, wgt.double(), None, 2, 1)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz.double())
    cnt = c_in - ci_first if take is None else take
    want = nhwc(x.grad)[..., ci_first:ci_first + cnt]

    dev = _dev()
    dzd, wd = nhwc(dz).to(dev), wgt.to(dev)
    wide = torch.full((n, h, w, cnt + 8), 7.0, device=dev)              # dx lands in a channel slice of a wider tensor
    dx = wide[..., 4:4 + cnt]
    dd = ops.conv_desc(n, h // 2, w // 2, c_out, cnt, 3, 1, False, math="f32")
    one, zero = torch.ones(cnt, device=dev), torch.zeros(cnt, device=dev)
    masks = []
    for py in (0, 1):
        for px in (0, 1):
            v, mask = train_ops.dgrad_class_weights(wd, py, px, ci_first, cnt)
            masks.append(bin(mask).count("1"))
            ops.conv2d_taps(dd, dzd, ops.pack_conv_weights(dd, v), one, zero, dx[:, py::2, px::2, :], mask)
    assert masks == [1, 2, 2, 4]
    assert rel_err(dx, want) < 2e-5
    assert float(wide[..., :4].min()) == 7.0 and float(wide[..., 4 + cnt:].max()) == 7.0      # nothing outside the slice
    wt = train_ops.dgrad_weights(wd, ci_first, cnt)
    ds = ops.conv_desc(n, h, w, c_out, cnt, 3, 1, False, up0=2, math="f32")
    stuffed = ops.conv2d(ds, dzd, ops.pack_conv_weights(ds, wt), one, zero)
    assert rel_err(dx, stuffed.cpu()) < 2e-6


@pytest.mark.parametrize("shape,groups", [((4, 32, 32, 64), 1), ((6, 16, 16, 128), 6),
                                          ((2, 64, 64, 13), 1), ((5, 32, 32, 8), 5),
                                          ((3, 32, 32, 1), 3), ((2, 16, 16, 512), 1)])
def test_bn_train_forward_and_backward(shape, groups):
    from disconet_amd import train_ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(5)
    z = torch.randn(shape, generator=g) * 2 + 0.5
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.2
    eps = 1e-5
    zd = z.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    # per group: F.batch_norm over that group's images (training mode)
    ys = []
    for k in range(groups):
        zg = zd[k * (n // groups):(k + 1) * (n // groups)].permute(0, 3, 1, 2)
        ys.append(F.relu(F.batch_norm(zg, None, None, gd, bd, True, 0.1, eps)).permute(0, 2, 3, 1))
    y_ref = torch.cat(ys, 0)
    dy = torch.randn(shape, generator=g)
    dy2 = torch.randn(shape, generator=g)
    y_ref.backward((dy + dy2).double())

    zg_, gm, bt = z.to(_dev()), gamma.to(_dev()), beta.to(_dev())
    mean, var = train_ops.bn_stats(zg_, groups)
    y = train_ops.bn_apply(zg_, mean, var, gm, bt, eps, relu=True)
    assert rel_err(y, y_ref.detach()) < 2e-5
    dgamma, dbeta = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
    # second gradient as a channel slice of a wider tensor
    wide = torch.zeros(n, h, w, c + 8, device=_dev())
    wide[..., 8:] = dy2.to(_dev())
    dz = train_ops.bn_backward(dy.to(_dev()), y, zg_, mean, var, gm, eps, dgamma, dbeta,
                               dy_b=wide[..., 8:])
    assert rel_err(dz, zd.grad) < 5e-5
    assert rel_err(dgamma, gd.grad) < 2e-5
    assert rel_err(dbeta, bd.grad) < 2e-5


def test_bn_backward_of_upsampled_gradient_and_running_stats():
    from disconet_amd import train_ops
    n, h, w, c = 2, 16, 16, 64
    g = torch.Generator().manual_seed(8)
    z = torch.randn(n, h, w, c, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    zd = z.double().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(c).double()
    bn.weight.data, bn.bias.data = gamma.double(), beta.double()
    bn.running_mean.data = torch.randn(c, generator=g).double()
    rm0, rv0 = bn.running_mean.clone().float(), bn.running_var.clone().float()
    y_ref = F.relu(bn(zd.permute(0, 3, 1, 2)))
    up = F.interpolate(y_ref, scale_factor=2)                       # decoder's nearest upsample
    cat = torch.cat([up, torch.zeros(n, 32, 2 * h, 2 * w, dtype=torch.float64)], 1)
    dcat = torch.randn(n, 2 * h, 2 * w, c + 32, generator=g)        # NHWC gradient of the concat
    cat.backward(dcat.permute(0, 3, 1, 2).double())

    zg_, gm = z.to(_dev()), gamma.to(_dev())
    mean, var = train_ops.bn_stats(zg_)
    y = train_ops.bn_apply(zg_, mean, var, gm, beta.to(_dev()), bn.eps)
    dgamma, dbeta = torch.zeros(c, device=_dev()), torch.zeros(c, device=_dev())
    dz = train_ops.bn_backward(dcat.to(_dev())[..., :c], y, zg_, mean, var, gm, bn.eps, dgamma, dbeta,
                               up_a=True)
    assert rel_err(dz, zd.grad) < 5e-5
    assert rel_err(dgamma, bn.weight.grad) < 2e-5
    rm, rv = rm0.to(_dev()), rv0.to(_dev())
    train_ops.bn_update_running(mean, var, n * h * w, rm, rv, 0.1)
    assert rel_err(rm, bn.running_mean) < 1e-6 and rel_err(rv, bn.running_var) < 1e-6


@pytest.mark.parametrize("shape,up_a,lift", [((3, 16, 32, 64), False, 2.0 ** 9), ((2, 8, 8, 512), False, 2.0 ** 12),
                                             ((2, 32, 32, 32), True, 2.0 ** 14)])
def test_bn_backward_sp_copy_is_the_split_of_the_lifted_dz(shape, up_a, lift):
    """dn_bn_train_backward_finish_sp: dz itself is bit for bit the plain call's, and the SP copy holds exactly the f16 hi / lo
    split of dz * lift (what dn_sp_from_nhwc makes of it) -- pieces of two threads' values, ragged wavefronts, the x2 gradient."""
    from disconet_amd import ops, train_ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(21)
    z = (torch.randn(shape, generator=g) * 2 + 0.5).to(_dev())
    gm = (torch.rand(c, generator=g) + 0.5).to(_dev())
    bt = (torch.randn(c, generator=g) * 0.2).to(_dev())
    mean, var = train_ops.bn_stats(z)
    y = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True)
    dy = (torch.randn((n, 2 * h, 2 * w, c + 16) if up_a else (n, h, w, c + 16), generator=g) * 1e-3).to(_dev())[..., 16:]
    dg0, db0 = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
    dg1, db1 = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
    dz0 = train_ops.bn_backward(dy, y, z, mean, var, gm, 1e-5, dg0, db0, up_a=up_a)
    sp = ops.SpTensor(n, h, w, c, device=_dev())
    sp.data.fill_(7.0)
    dz1 = train_ops.bn_backward(dy, y, z, mean, var, gm, 1e-5, dg1, db1, up_a=up_a, sp_out=sp, sp_lift=lift)
    assert torch.equal(dz0, dz1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    want = ops.SpTensor.from_nhwc(dz0 * lift)
    assert torch.equal(sp.data.view(torch.int16), want.data.view(torch.int16))
    assert float((sp.nhwc() / lift - dz0).abs().max()) <= 2.0 ** -21 * float(dz0.abs().max())


@pytest.mark.parametrize("shape,up_a,sp", [((3, 16, 32, 64), False, False), ((2, 9, 7, 36), False, False), ((2, 32, 32, 32), True, True)])
def test_bn_relu_byte_mask_replaces_y_in_the_backward_bit_for_bit(shape, up_a, sp):
    """dn_bn_train_apply_mask: y is the plain call's, the mask is (y > 0) packed four channels to a byte; the backward reading
    the mask (relu = 2) equals the backward reading y bit for bit -- dz, the SP copy, dgamma, dbeta; ragged sizes included."""
    from disconet_amd import ops, train_ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(31)
    z = (torch.randn(shape, generator=g) * 2 + 0.5).to(_dev())
    gm = (torch.rand(c, generator=g) + 0.5).to(_dev())
    bt = (torch.randn(c, generator=g) * 0.2).to(_dev())
    mean, var = train_ops.bn_stats(z)
    y0 = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True)
    mask = torch.full((z.numel() // 4,), 0xAA, dtype=torch.uint8, device=_dev())
    y1 = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True, relu_mask=mask)
    assert torch.equal(y0, y1)
    bits = (y0.reshape(-1, 4) > 0).to(torch.uint8)
    assert torch.equal(mask, bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3))
    dy = (torch.randn((n, 2 * h, 2 * w, c + 16) if up_a else (n, h, w, c + 16), generator=g) * 1e-3).to(_dev())[..., 16:]
    dyb = None if up_a else torch.randn(shape, generator=g).to(_dev()) * 1e-3
    out = []
    for m in (None, mask):
        dg, db = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
        spt = ops.SpTensor(n, h, w, c, device=_dev()) if sp else None
        dz = train_ops.bn_backward(dy, y0, z, mean, var, gm, 1e-5, dg, db, up_a=up_a, dy_b=dyb, sp_out=spt, sp_lift=2.0 ** 12,
                                   relu_mask=m)
        out.append((dz, dg, db, spt.data.clone() if sp else dz))
    for a, b in zip(*out):
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16),
                           b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16))


@pytest.mark.parametrize("shape,sp", [((2, 16, 24, 32), True), ((3, 8, 8, 6), False), ((1, 32, 32, 64), False)])
def test_bn_backward_reads_a_space_to_depth_gradient_in_place(shape, sp):
    """up_a = 2: dy_a as [n, h/2, w/2, 4c] with pixel (y, x) in channel group (y & 1) * 2 + (x & 1) -- the layout of the one-launch
    stride-2 data gradient -- gives bit for bit what the dense gradient gives (vector and scalar kernels, SP copy, second consumer)."""
    from disconet_amd import ops, train_ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(41)
    z = (torch.randn(shape, generator=g) * 2 + 0.5).to(_dev())
    gm = (torch.rand(c, generator=g) + 0.5).to(_dev())
    bt = (torch.randn(c, generator=g) * 0.2).to(_dev())
    mean, var = train_ops.bn_stats(z)
    y = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True)
    dy = (torch.randn(shape, generator=g) * 1e-3).to(_dev())
    dyb = (torch.randn(shape, generator=g) * 1e-3).to(_dev())
    s2d = dy.view(n, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4 * c).contiguous()
    out = []
    for src, up in ((dy, 0), (s2d, 2)):
        dg, db = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
        spt = ops.SpTensor(n, h, w, c, device=_dev()) if sp else None
        dz = train_ops.bn_backward(src, y, z, mean, var, gm, 1e-5, dg, db, up_a=up, dy_b=dyb, sp_out=spt, sp_lift=2.0 ** 12)
        out.append((dz, dg, db, spt.data.clone().view(torch.int16) if sp else dz))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    with pytest.raises(ops._lib.DnError):       # odd maps have no space-to-depth image
        zz = z[:, :h - 1].contiguous()
        train_ops.bn_backward(s2d, y[:, :h - 1].contiguous(), zz, mean, var, gm, 1e-5, dg, db, up_a=2)


_BN_AB_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r)
from disconet_amd import ops, train_ops
dev = torch.device("cuda:0")
out = []
for shape, up_a, sp, two in (((20, 64, 64, 128), 0, True, True), ((3, 16, 32, 64), 1, True, False), ((2, 32, 32, 32), 2, True, True),
                             ((4, 32, 32, 512), 0, False, False), ((2, 16, 24, 16), 0, True, True), ((1, 256, 256, 32), 0, True, False)):
    n, h, w, c = shape
    g = torch.Generator().manual_seed(51)
    z = (torch.randn(shape, generator=g) * 2 + 0.5).to(dev)
    gm = (torch.rand(c, generator=g) + 0.5).to(dev)
    bt = (torch.randn(c, generator=g) * 0.2).to(dev)
    mean, var = train_ops.bn_stats(z)
    mask = torch.zeros(z.numel() // 4, dtype=torch.uint8, device=dev)
    y = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True, relu_mask=mask)
    y2 = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=False)
    dshape = {0: (n, h, w, c), 1: (n, 2 * h, 2 * w, c), 2: (n, h // 2, w // 2, 4 * c)}[up_a]
    dy = (torch.randn(dshape, generator=g) * 1e-3).to(dev)
    dyb = (torch.randn(shape, generator=g) * 1e-3).to(dev) if two else None
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    spt = ops.SpTensor(n, h, w, c, device=dev) if sp else None
    dz = train_ops.bn_backward(dy, y, z, mean, var, gm, 1e-5, dg, db, up_a=up_a, dy_b=dyb, sp_out=spt, sp_lift=2.0 ** 12, relu_mask=mask)
    dz2 = train_ops.bn_backward(dy, y, z, mean, var, gm, 1e-5, dg, db, up_a=up_a, dy_b=dyb)
    bias = train_ops.channel_sum(dz, torch.empty(c, device=dev))
    out += [t.cpu() for t in (mean, var, y, y2, mask, dz, dz2, dg, db, bias)] + ([spt.data.view(torch.int16).cpu()] if sp else [])
torch.save(out, sys.argv[1])
"""


def test_bn_fast_kernels_equal_the_general_kernels_bit_for_bit(tmp_path):
    """The one-group fast forms of the BatchNorm apply / backward-apply kernels (per-channel constants once per workgroup in LDS,
    shift-and-mask indexing) and the reductions that fetch several rows before they add them (in row order) against the general
    kernels / one row per iteration (DN_BN_LEGACY=1, a child process: the switch is read once): mean, var, y, the ReLU byte mask,
    dz, the SP copy, dgamma, dbeta, the bias sum -- bit for bit, over the gradient forms (dense, x2 block sum, space-to-depth, second
    consumer) and layer shapes incl. a 20-image batch and a full-resolution map."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    files = []
    for legacy in ("0", "1"):
        f = str(tmp_path / ("bn_%s.pt" % legacy))
        env = dict(os.environ, DN_BN_LEGACY=legacy)
        subprocess.run([sys.executable, "-c", _BN_AB_SCRIPT % ROOT, f], check=True, env=env, timeout=600)
        files.append(torch.load(f))
    assert len(files[0]) == len(files[1]) > 60
    for a, b in zip(*files):
        assert a.dtype == b.dtype and a.shape == b.shape
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))


def test_bn_backward_sp_copy_flags_a_gradient_that_outgrows_its_lift():
    from disconet_amd import ops, train_ops
    n, h, w, c = 1, 8, 8, 32
    g = torch.Generator().manual_seed(22)
    z = torch.randn(n, h, w, c, generator=g).to(_dev())
    gm, bt = torch.ones(c, device=_dev()), torch.zeros(c, device=_dev())
    mean, var = train_ops.bn_stats(z)
    y = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True)
    dy = torch.randn(n, h, w, c, generator=g).to(_dev())
    dg, db = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
    ops.sp_range_flags(reset=True)
    sp = ops.SpTensor(n, h, w, c, device=_dev())
    train_ops.bn_backward(dy, y, z, mean, var, gm, 1e-5, dg, db, sp_out=sp, sp_lift=2.0 ** 8)
    assert ops.sp_range_flags(reset=True) & 1 == 0
    train_ops.bn_backward(dy, y, z, mean, var, gm, 1e-5, dg, db, sp_out=sp, sp_lift=2.0 ** 20)
    assert ops.sp_range_flags(reset=True) & 1 == 1          # clamped to +-65504: the caller must discard the gradients


@pytest.mark.parametrize("n,h,w,c_out,c_in", [(2, 16, 16, 256, 768), (3, 32, 64, 32, 96), (1, 64, 64, 64, 64), (2, 8, 8, 512, 512)])
def test_split_f16_data_gradient_through_the_inference_engine(n, h, w, c_out, c_in):
    """dx = conv(dz; flipped / transposed weights) on the split-planar engine with fp32 rows as its only output
    (dn_spconv2d_nhwc), dz lifted and pre-split: against float64 autograd at the engine's 2^-22-per-operand accuracy, written
    into a channel slice, and bit for bit the fp32 copy of dn_spconv2d_dual."""
    from disconet_amd import ops, train_ops
    g = torch.Generator().manual_seed(31)
    wgt = torch.randn(c_out, c_in, 3, 3, generator=g) * (2.0 / (9 * c_in)) ** 0.5
    dz = torch.randn(n, h, w, c_out, generator=g) * 3e-4                  # a gradient map's magnitudes: f16 subnormal lo halves without a lift
    x = torch.zeros(n, c_in, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, wgt.double(), padding=1).backward(dz.permute(0, 3, 1, 2).double())
    want = x.grad.permute(0, 2, 3, 1)
    import math
    lift = 2.0 ** (8 - math.floor(math.log2(float(dz.abs().max()))))
    dzd = dz.to(_dev())
    sp = ops.SpTensor.from_nhwc(dzd * lift)
    wt = train_ops.dgrad_weights(wgt.to(_dev()), 0, None)
    dd = ops.conv_desc(n, h, w, c_out, c_in, 3, 1, False)
    packed, wmul = ops.sp_pack_conv_weights(dd, wt)
    scale = torch.full((c_in,), 1.0 / (lift * wmul), device=_dev())
    shift = torch.zeros(c_in, device=_dev())
    wide = torch.full((n, h, w, c_in + 8), 5.0, device=_dev())
    dx = ops.sp_conv2d_nhwc(dd, sp, packed, scale, shift, wide[..., 8:])
    assert rel_err(dx, want) < 3e-6
    assert float(wide[..., :8].min()) == 5.0 and float(wide[..., :8].max()) == 5.0
    _, flat = ops.sp_conv2d(dd, sp, packed, scale, shift, nhwc_copy=True)
    assert torch.equal(flat, dx)
    # without the lift the same engine loses the lo halves (f16 subnormals): the measurement behind rounds 2-4's "fp32 only"
    sp0 = ops.SpTensor.from_nhwc(dzd)
    dx0 = ops.sp_conv2d_nhwc(dd, sp0, packed, torch.full((c_in,), 1.0 / wmul, device=_dev()), shift,
                             torch.empty(n, h, w, c_in, device=_dev()))
    assert rel_err(dx0, want) > 20 * rel_err(dx, want)


def test_channel_sum_add_rows_and_pair_kernels():
    from disconet_amd import train_ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 32, 32, 48, generator=g)
    xg = x.to(_dev())
    out = torch.ones(36, device=_dev())
    train_ops.channel_sum(xg[..., 4:40], out, accumulate=True)
    assert rel_err(out, 1 + x[..., 4:40].double().sum((0, 1, 2))) < 1e-6
    a = torch.randn(3, 32, 32, 36, generator=g)
    ag = a.to(_dev())
    train_ops.add_rows(ag, xg[..., 4:40])
    assert rel_err(ag, a + x[..., 4:40]) < 1e-6
    # pairs: z1[p] += e[ego[p]];  de[img] = sum of dz1 over the image's pairs
    e = torch.randn(4, 8, 8, 16, generator=g)
    z1 = torch.randn(7, 8, 8, 16, generator=g)
    ego = torch.tensor([0, 1, 1, 3, 3, 3, 2], dtype=torch.int32)
    got = train_ops.pair_add_ego(z1.to(_dev()), e.to(_dev()), ego.to(_dev()))
    assert rel_err(got, z1 + e[ego.long()]) < 1e-6
    first = torch.tensor([0, 1, 3, 4, 7], dtype=torch.int32)
    pairs = torch.tensor([0, 1, 2, 6, 3, 4, 5], dtype=torch.int32)
    de = train_ops.pair_sum_ego(z1.to(_dev()), first.to(_dev()), pairs.to(_dev()), 4)
    ref = torch.stack([z1[pairs[first[i]:first[i + 1]].long()].sum(0) for i in range(4)])
    assert rel_err(de, ref) < 1e-6


def _poses(n, seed=4):
    rng = np.random.RandomState(seed)
    out = np.zeros((n, 4, 4), dtype=np.float32)
    for k in range(n):
        yaw, tx, ty = rng.uniform(-0.6, 0.6), rng.uniform(-12, 12), rng.uniform(-12, 12)
        out[k] = np.eye(4)
        out[k, 0, 0], out[k, 0, 1], out[k, 1, 0], out[k, 1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
        out[k, 0, 3], out[k, 1, 3] = tx, ty
    out[0] = np.eye(4)
    out[1, 0, 3], out[1, 1, 3] = 500.0, 500.0          # out of frame
    return torch.from_numpy(out)


def _warp_ref(nb, pose):
    """the two-pass warp of the oracle (upstream feature_transformation), differentiable"""
    theta_rot = torch.tensor([[pose[0, 0], pose[0, 1], 0.0], [pose[1, 0], pose[1, 1], 0.0]],
                             dtype=nb.dtype).unsqueeze(0)
    theta_trans = torch.tensor([[1.0, 0.0, 4 * pose[0, 3] / 128], [0.0, 1.0, -4 * pose[1, 3] / 128]],
                               dtype=nb.dtype).unsqueeze(0)
    size = (1,) + tuple(nb.shape)
    r = F.grid_sample(nb.unsqueeze(0), F.affine_grid(theta_rot, size, align_corners=False),
                      align_corners=False)
    return F.grid_sample(r, F.affine_grid(theta_trans, size, align_corners=False),
                         align_corners=False)[0]


@pytest.mark.parametrize("rigid,c", [(False, 64), (True, 64), (True, 512)])
def test_warp_list_and_backward_match_autograd(rigid, c):
    """rigid=False: scatter with float atomics (any pose); True: the deterministic gather form"""
    from disconet_amd import train_ops
    g = torch.Generator().manual_seed(9)
    m, hw, n = 4, 32, 7
    maps = torch.randn(m, c, hw, hw, generator=g).double().requires_grad_(True)
    poses = _poses(n)
    src = torch.tensor([0, 1, 2, 3, 1, 1, 0], dtype=torch.int32)
    ref = torch.stack([_warp_ref(maps[int(src[k])], poses[k].double()) for k in range(n)])
    dwarp = torch.randn(ref.shape, generator=g)
    ref.backward(dwarp.double())

    mg = nhwc(maps.detach().float()).to(_dev())
    warped = train_ops.warp_list(mg, poses.to(_dev()), src.to(_dev()))
    assert rel_err(warped, nhwc(ref.detach())) < 1e-5
    d_src = torch.ones(m, hw, hw, c, device=_dev())      # accumulates on top of what is there
    train_ops.warp_backward(nhwc(dwarp).to(_dev()), poses.to(_dev()), src.to(_dev()), d_src, rigid=rigid)
    assert rel_err(d_src - 1, nhwc(maps.grad)) < 2e-5
    if rigid:      # deterministic: a second run gives the same bits
        again = torch.ones(m, hw, hw, c, device=_dev())
        train_ops.warp_backward(nhwc(dwarp).to(_dev()), poses.to(_dev()), src.to(_dev()), again, rigid=True)
        assert torch.equal(again, d_src)


_WARP_AB_SCRIPT = """
import sys
sys.path.insert(0, %r)
import math, torch
from disconet_amd import train_ops
out = []
for n_src, n, hw, c, seed in ((4, 7, 32, 256, 1), (20, 80, 32, 256, 2), (3, 70, 16, 1024, 3), (2, 5, 24, 64, 4)):
    g = torch.Generator().manual_seed(seed)
    poses = torch.zeros(n, 4, 4)
    for k in range(n):
        a = float(torch.rand((), generator=g)) * 2 * math.pi
        poses[k] = torch.eye(4)
        poses[k, 0, 0], poses[k, 0, 1], poses[k, 1, 0], poses[k, 1, 1] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
        poses[k, 0, 3], poses[k, 1, 3] = (torch.rand(2, generator=g) - 0.5) * 40
    poses[0] = torch.eye(4)                                  # taps on the grid: weights of exactly 0 and 1
    src = torch.randint(0, n_src, (n,), generator=g, dtype=torch.int32)
    d_warped = torch.randn(n, hw, hw, c, generator=g).cuda()
    d_src = torch.randn(n_src, hw, hw, c, generator=g).cuda()
    train_ops.warp_backward(d_warped, poses.cuda(), src.cuda(), d_src, rigid=True)
    out.append(d_src.cpu())
torch.save(out, sys.argv[1])
"""


def test_lane_parallel_warp_gathers_agree_with_the_scalar_kernels(tmp_path):
    """dn_warp_backward's rigid form: the kernels whose lanes evaluate one candidate pixel each (round 6) against the ones that walk
    the candidates one after the other (DN_WARP_GATHER_LEGACY=1, a child process: the switch is read once) -- the same hits added in
    the same order; the tap weights may differ in their last bit (the compiler contracts the pose arithmetic per kernel), so the
    gradients agree to a few ulp, not bit for bit; 80 warps of 20 source images (the detector's batch), more warps than lanes (two
    ballots), 1024 channels (four rows per lane), a map that is no multiple of the pixels per workgroup."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    files = []
    for legacy in ("0", "1"):
        f = str(tmp_path / ("warp_%s.pt" % legacy))
        subprocess.run([sys.executable, "-c", _WARP_AB_SCRIPT % ROOT, f], check=True, env=dict(os.environ, DN_WARP_GATHER_LEGACY=legacy),
                       timeout=600)
        files.append(torch.load(f))
    assert len(files[0]) == len(files[1]) == 4
    for a, b in zip(*files):
        assert a.shape == b.shape and torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


@pytest.mark.parametrize("seed", [0, 1])
def test_one_launch_pack_of_many_weight_forms_writes_the_single_launches_bytes(seed):
    """dn_spconv_pack_weights_multi (ops.SpPackSet) against the calls it stands for -- dn_spconv_pack_weights of the weight (mode 0),
    of dn_conv_dgrad_weights' flipped / transposed cut (mode 1), of the four dn_conv_dgrad_class_weights classes stacked (mode 2):
    every packed image byte for byte, over 3x3 and 1x1 layers, channel counts off the 16 / 64 padding, column cuts of a concat
    layer's weight, and a second run after the weights and the lifts moved; a tap-merged layer is refused."""
    from disconet_amd import ops, train_ops
    g = torch.Generator().manual_seed(seed)
    dev = _dev()
    jobs, singles = [], []

    def weight(c_out, c_in, k):
        return (torch.randn(c_out, c_in, k, k, generator=g) * 0.05).to(dev)

    def single(j, wmul):
        d, w, mode, cin_total, ci_first, n_in = j
        if mode == 0:
            return ops.sp_pack_conv_weights(d, w, wmul)[0]
        if mode == 1:
            return ops.sp_pack_conv_weights(d, train_ops.dgrad_weights(w, ci_first, d.c_out), wmul)[0]
        wt = torch.empty((4 * n_in, w.shape[0], 3, 3), device=dev)
        for py in (0, 1):
            for px in (0, 1):
                k = py * 2 + px
                train_ops.dgrad_class_weights(w, py, px, ci_first, n_in, out=wt[k * n_in:(k + 1) * n_in])
        return ops.sp_pack_conv_weights(d, wt, wmul)[0]

    for c_out, c_in, k in ((32, 13, 3), (64, 32, 3), (72, 40, 3), (512, 256, 3), (32, 64, 1), (12, 32, 1)):
        w = weight(c_out, c_in, k)
        jobs.append((ops.conv_desc(3, 16, 16, c_in, c_out, k, 1, False), w, 0, c_in, 0, 0))
    for c_out, cin_total, ci_first, n_in in ((64, 96, 0, 64), (64, 96, 64, 32), (128, 64, 0, 64), (32, 13, 0, 13)):
        w = weight(c_out, cin_total, 3)
        jobs.append((ops.conv_desc(2, 16, 16, c_out, n_in, 3, 1, False), w, 1, cin_total, ci_first, n_in))
    for c_out, cin_total, ci_first, n_in in ((64, 32, 0, 32), (128, 64, 0, 64), (64, 48, 16, 32)):
        w = weight(c_out, cin_total, 3)
        jobs.append((ops.conv_desc(2, 8, 8, c_out, 4 * n_in, 3, 1, False), w, 2, cin_total, ci_first, n_in))
    ps = ops.SpPackSet(jobs, dev)
    for trial in range(2):
        wmuls = [float(2.0 ** (3 + (i + trial) % 5)) for i in range(len(jobs))]
        if trial:
            for j in jobs:
                j[1].mul_(1.37)
        for b in ps.buffers:
            b.fill_(0xAB)
        ps.run(wmuls)
        for i, (j, b) in enumerate(zip(jobs, ps.buffers)):
            want = single(j, wmuls[i])
            assert want.numel() == b.numel() and torch.equal(want, b), (trial, i)
    up = ops.conv_desc(2, 16, 16, 64, 32, 3, 1, False, c1=32, up0=1)
    assert not ops.SpPackSet.supported(up) and ops.SpPackSet.supported(jobs[0][0])


@pytest.mark.parametrize("math", [0, 1])
def test_one_launch_pack_of_the_nhwc_engines_weight_forms_writes_the_single_launches_bytes(math):
    """dn_conv_pack_weights_multi (ops.PackSet, engine "nhwc") against dn_conv_pack_weights of the weight times its lift, of a
    column cut of it (the attention MLP's W1 halves) and of dn_conv_dgrad_weights' flipped / transposed cut -- fp32 rows
    (math 0) and split-f16 rows (math 1), 1x1 and 3x3, byte for byte, before and after the weights and lifts moved."""
    from disconet_amd import ops, train_ops
    g = torch.Generator().manual_seed(5 + math)
    dev = _dev()
    jobs = []
    for c_out, cin_total, ci_first, c_in, k, mode in ((128, 512, 0, 256, 1, 0), (128, 512, 256, 256, 1, 0), (32, 128, 0, 128, 1, 0),
                                                      (8, 32, 0, 32, 1, 0), (1, 8, 0, 8, 1, 0), (12, 32, 0, 32, 1, 0), (64, 40, 0, 40, 3, 0),
                                                      (128, 512, 256, 256, 1, 1), (32, 128, 0, 128, 1, 1), (36, 32, 0, 32, 1, 1),
                                                      (64, 96, 32, 64, 3, 1)):
        w = (torch.randn(c_out, cin_total, k, k, generator=g) * 0.05).to(dev)
        d = (ops.conv_desc(2, 8, 8, c_in, c_out, k, 1, False, math=math) if mode == 0 else
             ops.conv_desc(2, 8, 8, c_out, c_in, k, 1, False, math=math))
        jobs.append((d, w, mode, cin_total, ci_first, c_in if mode else 0))
    ps = ops.PackSet(jobs, dev, "nhwc")
    for trial in range(2):
        wmuls = [float(2.0 ** ((i + trial) % 4)) for i in range(len(jobs))]
        if trial:
            for j in jobs:
                j[1].mul_(0.77)
        for b in ps.buffers:
            b.fill_(123.0)
        ps.run(wmuls)
        for i, ((d, w, mode, cin_total, ci_first, n_in), b) in enumerate(zip(jobs, ps.buffers)):
            if mode == 0:
                wc = w[:, ci_first:ci_first + d.c0].contiguous()
            else:
                wc = train_ops.dgrad_weights(w, ci_first, d.c_out)
            want = ops.pack_conv_weights(d, wc * wmuls[i])
            assert want.numel() == b.numel() and torch.equal(want.view(torch.int32), b.view(torch.int32)), (trial, i)


def test_fuse_combine_forward_backward():
    from disconet_amd import train_ops
    g = torch.Generator().manual_seed(12)
    hw, c = 16, 256
    n_maps, n_pairs = 7, 6
    maps = torch.randn(n_maps, hw, hw, c, generator=g).double().requires_grad_(True)
    z4 = (torch.randn(n_pairs, hw, hw, 1, generator=g) * 2).double().requires_grad_(True)
    # ego 0: pairs 0,1,2 over maps 0,4,5; ego 1: pairs 3,4,5 over maps 1,6,2; ego 2 not live: itself
    lists = [[(0, 0), (1, 4), (2, 5)], [(3, 1), (4, 6), (5, 2)], [(-1, 3)]]
    ego_out = [0, 1, 3]
    fused_ref = torch.zeros(4, hw, hw, c, dtype=torch.float64)
    outs = []
    for lst in lists:
        s = [F.relu(z4[p, ..., 0]) if p >= 0 else torch.zeros(hw, hw, dtype=torch.float64) for p, _ in lst]
        e = [torch.exp(v) for v in s]
        tot = sum(e)
        outs.append(sum((ek / tot).unsqueeze(-1) * maps[mi] for ek, (_, mi) in zip(e, lst)))
    dfused = torch.randn(4, hw, hw, c + 64, generator=g)
    loss = sum((o * dfused[eo, ..., 64:].double()).sum() for o, eo in zip(outs, ego_out))
    loss.backward()

    first = torch.tensor([0, 3, 6, 7], dtype=torch.int32).to(_dev())
    pair_index = torch.tensor([p for l in lists for p, _ in l], dtype=torch.int32).to(_dev())
    map_image = torch.tensor([m for l in lists for _, m in l], dtype=torch.int32).to(_dev())
    eo = torch.tensor(ego_out, dtype=torch.int32).to(_dev())
    mg, zg = maps.detach().float().to(_dev()), z4.detach().float().to(_dev())
    fused = torch.zeros(4, hw, hw, c, device=_dev())
    wts = train_ops.fuse_combine(zg, mg, first, pair_index, map_image, eo, fused)
    for o, e_ in zip(outs, ego_out):
        assert rel_err(fused[e_], o.detach()) < 1e-5
    dmaps = torch.zeros_like(mg)
    dz4 = train_ops.fuse_combine_backward(dfused.to(_dev())[..., 64:], zg, wts, mg, first, pair_index,
                                          map_image, eo, dmaps)
    assert rel_err(dmaps, maps.grad) < 1e-5
    assert rel_err(dz4, z4.grad) < 2e-5


def test_det_loss_and_gradients_match_the_oracle():
    from disconet_amd import train_ops
    from oracle.train_ref import det_loss
    g = torch.Generator().manual_seed(6)
    n_img, hw, a, code = 2, 16, 6, 6
    n = n_img * hw * hw * a
    cls = (torch.randn(n_img, hw * hw * a, 2, generator=g) * 3).double().requires_grad_(True)
    loc = torch.randn(n_img, hw, hw, a, 1, code, generator=g).double().requires_grad_(True)
    fg = torch.rand(n, generator=g) < 0.05
    ignore = torch.rand(n, generator=g) < 0.02
    labels = torch.stack([(~fg).float(), fg.float()], -1)
    labels[ignore] = 0
    targets = torch.randn(n, code, generator=g) * 0.5
    mask = fg.float()
    l_cls, l_loc = det_loss({"cls": cls, "loc": loc}, labels, targets, mask, norm=n_img)
    (l_cls + l_loc).backward()
    losses, dcls, dloc = train_ops.det_loss(
        cls.detach().float().reshape(-1, 2).to(_dev()), labels.to(_dev()),
        loc.detach().float().reshape(-1, code).to(_dev()), targets.to(_dev()), mask.to(_dev()), norm=n_img)
    assert abs(float(losses[0]) - float(l_cls)) < 1e-5 * abs(float(l_cls))
    assert abs(float(losses[1]) - float(l_loc)) < 1e-5 * abs(float(l_loc))
    assert rel_err(dcls, cls.grad.reshape(-1, 2)) < 2e-5
    assert rel_err(dloc, loc.grad.reshape(-1, code)) < 2e-5
    # round 6: the float4-stream kernel (even n, 16-byte aligned tensors) against the per-anchor kernel (what an unaligned tensor
    # still gets): the same formulas per element -> dcls / dloc bit for bit, the loss values to the rounding of another summation order
    cd = cls.detach().float().reshape(-1, 2).to(_dev())
    pad = torch.zeros(cd.numel() + 2, device=_dev())
    pad[2:] = cd.reshape(-1)
    cls_unaligned = pad[2:].view(-1, 2)
    assert cls_unaligned.data_ptr() % 16 == 8 and cd.data_ptr() % 16 == 0
    args = (labels.to(_dev()), loc.detach().float().reshape(-1, code).to(_dev()), targets.to(_dev()), mask.to(_dev()))
    l1, dc1, dl1 = train_ops.det_loss(cd, *args, norm=n_img)
    l0, dc0, dl0 = train_ops.det_loss(cls_unaligned, *args, norm=n_img)
    assert torch.equal(dc0, dc1) and torch.equal(dl0, dl1)
    assert float((l0 - l1).abs().max()) <= 1e-12 * float(l0.abs().max())


def test_adam_matches_torch_optim():
    from disconet_amd import train_ops
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(10007, generator=g)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p, m, v = p0.to(_dev()), torch.zeros(10007, device=_dev()), torch.zeros(10007, device=_dev())
    for step in range(1, 4):
        grad = torch.randn(10007, generator=g) * (10.0 ** (step - 2))
        p_ref.grad = grad.clone()
        opt.step()
        train_ops.adam_step(p, grad.to(_dev()), m, v, step, lr=1e-3)
    assert float((p.cpu() - p_ref.detach()).abs().max()) < 2e-6


@pytest.mark.parametrize("shape", [(3, 16, 24, 64), (2, 32, 32, 256), (1, 64, 64, 32), (2, 8, 8, 512)])
def test_bn_apply_writes_y_a_second_time_as_an_sp_tensor(shape):
    """Round 6 (the training forward on the inference engine): dn_bn_train_apply_mask_sp = dn_bn_train_apply_mask bit for bit
    (y, the byte mask) + y as the split-planar f16 hi / lo tensor -- the bits dn_sp_from_nhwc makes of that y."""
    from disconet_amd import ops, train_ops
    g = torch.Generator().manual_seed(17)
    n, h, w, c = shape
    z = (torch.randn(shape, generator=g) * 2 + 0.3).to(_dev())
    gm, bt = (torch.rand(c, generator=g) + 0.5).to(_dev()), (torch.randn(c, generator=g) * 0.2).to(_dev())
    mean, var = train_ops.bn_stats(z)
    m0 = torch.empty(z.numel() // 4, dtype=torch.uint8, device=_dev())
    y0 = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True, relu_mask=m0)
    m1 = torch.zeros_like(m0)
    sp = ops.SpTensor(n, h, w, c, device=_dev())
    sp.data.fill_(7.0)
    y1 = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True, relu_mask=m1, sp_out=sp)
    assert torch.equal(y0, y1) and torch.equal(m0, m1)
    assert torch.equal(sp.data, ops.SpTensor.from_nhwc(y0).data)
    assert ops.sp_range_flags(reset=True) & 5 == 0
    assert not train_ops.bn_apply_sp_supported(torch.empty(1, 4, 4, 24))          # c % 16
    assert not train_ops.bn_apply_sp_supported(torch.empty(1, 4, 4, 48))          # c / 4 not a power of two
    with pytest.raises(Exception):
        train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True, sp_out=sp)      # needs the mask


@pytest.mark.parametrize("n,h,w,c0,c1,c_out", [(2, 32, 32, 512, 256, 256), (3, 16, 64, 64, 32, 32), (2, 24, 40, 128, 64, 64), (1, 64, 64, 256, 0, 96)])
def test_tap_merged_up_conv_writes_fp32_rows(n, h, w, c0, c1, c_out):
    """Round 6: the decoder's upsample + concat + 3x3 layers (conv_spq_kernel) also take dn_spconv2d_nhwc / _dual -- the
    training forward's z.  The fp32 rows are the epilogue's values BEFORE the split: they agree with the SP output to the split's
    2^-22, the SP output of the dual launch is bit for bit the plain launch's, and a channel slice of a wider tensor is honoured."""
    from disconet_amd import ops
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(n, h // 2, w // 2, c0, generator=g).relu().to(_dev())
    x1 = torch.randn(n, h, w, c1, generator=g).relu().to(_dev()) if c1 else None
    wgt = (torch.randn(c_out, c0 + c1, 3, 3, generator=g) * (2.0 / (9 * (c0 + c1))) ** 0.5).to(_dev())
    bias = (torch.randn(c_out, generator=g) * 0.1).to(_dev())
    d = ops.conv_desc(n, h, w, c0, c_out, 3, 1, False, c1=c1, up0=True)
    packed, wmul = ops.sp_pack_conv_weights(d, wgt)
    scale = torch.full((c_out,), 1.0 / wmul, device=_dev())
    s0, s1 = ops.SpTensor.from_nhwc(x0), (ops.SpTensor.from_nhwc(x1) if c1 else None)
    want_sp = ops.sp_conv2d(d, s0, packed, scale, bias, src1=s1)
    wide = torch.full((n, h, w, c_out + 4), -3.0, device=_dev())
    rows = ops.sp_conv2d_nhwc(d, s0, packed, scale, bias, wide[..., 4:], src1=s1)
    assert float(wide[..., :4].min()) == -3.0 and float(wide[..., :4].max()) == -3.0
    ref = F.conv2d(torch.cat([F.interpolate(x0.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").double().cpu()]
                             + ([x1.permute(0, 3, 1, 2).double().cpu()] if c1 else []), 1),
                   wgt.double().cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
    assert rel_err(rows, ref) < 3e-6
    assert float((want_sp.nhwc() - rows).abs().max()) <= 2.0 ** -21 * float(rows.abs().max())
    got_sp, flat = ops.sp_conv2d(d, s0, packed, scale, bias, src1=s1, nhwc_copy=True)
    assert torch.equal(got_sp.data, want_sp.data) and torch.equal(flat, rows)


@pytest.mark.parametrize("shape,up_a,sp", [((2, 16, 24, 32), False, True), ((3, 32, 32, 64), True, True), ((20, 64, 64, 128), False, False),
                                           ((2, 8, 8, 512), False, True), ((1, 40, 72, 16), False, False)])
def test_bn_backward_fuses_the_conv_bias_gradient(shape, up_a, sp):
    """Round 6: bn_backward(dbias=...) -- the sum of dz per channel (the gradient of the conv bias in front of the BatchNorm)
    leaves the launch that writes dz.  dz, its SP copy, dgamma, dbeta are bit for bit the plain call's; dbias equals the
    float64 sum of that dz to fp32 rounding of the terms, equals dn_channel_sum's to the same, and is bitwise repeatable."""
    from disconet_amd import ops, train_ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(41)
    z = (torch.randn(shape, generator=g) * 2 + 0.5).to(_dev())
    gm, bt = (torch.rand(c, generator=g) + 0.5).to(_dev()), (torch.randn(c, generator=g) * 0.2).to(_dev())
    mean, var = train_ops.bn_stats(z)
    mask = torch.empty(z.numel() // 4, dtype=torch.uint8, device=_dev())
    y = train_ops.bn_apply(z, mean, var, gm, bt, 1e-5, relu=True, relu_mask=mask)
    dy = (torch.randn((n, 2 * h, 2 * w, c) if up_a else shape, generator=g) * 1e-3).to(_dev())
    assert train_ops.bn_backward_bias_supported(z)
    runs = []
    for fused in (False, True, True):
        dg, db = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
        spt = ops.SpTensor(n, h, w, c, device=_dev()) if sp else None
        dbias = torch.full((c,), 7.0, device=_dev())
        dz = train_ops.bn_backward(dy, y, z, mean, var, gm, 1e-5, dg, db, up_a=up_a, sp_out=spt, sp_lift=2.0 ** 10, relu_mask=mask,
                                   dbias=dbias if fused else None)
        if not fused:
            train_ops.channel_sum(dz, dbias)
        runs.append((dz, dg, db, spt.data.clone() if sp else dz, dbias))
    for k in range(4):
        a, b = runs[0][k], runs[1][k]
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16),
                           b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16)), k
    assert torch.equal(runs[1][4], runs[2][4])                                   # repeatable bit for bit
    want = runs[0][0].double().sum((0, 1, 2)).cpu()
    scale = float(runs[0][0].double().abs().sum((0, 1, 2)).max())
    assert float((runs[1][4].double().cpu() - want).abs().max()) <= 2e-6 * scale
    assert float((runs[0][4].double().cpu() - want).abs().max()) <= 2e-6 * scale
    assert not train_ops.bn_backward_bias_supported(torch.empty(1, 4, 4, 24))    # c / 4 = 6: the general kernels, channel_sum stays
