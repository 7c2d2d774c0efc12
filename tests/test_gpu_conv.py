"""K2/K3/K7: the MFMA implicit-GEMM conv against torch-CPU fp32 conv2d (the
plain fp32 reference of the same op) -- tolerance 1e-4 absolute on O(1) data."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4


MATHS = ["f32", "f16x3", "sp"]


def _run(n, h, w, c0, c_out, k, stride=1, relu=True, c1=0, up0=False, bn=True, seed=0, math="f32"):
    from disconet_amd import ops
    g = torch.Generator().manual_seed(seed)
    cin = c0 + c1
    wgt = torch.randn(c_out, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(c_out, generator=g) * 0.1
    h0, w0 = (h // 2, w // 2) if up0 else (h, w)
    x0 = torch.randn(n, c0, h0, w0, generator=g)
    x1 = torch.randn(n, c1, h, w, generator=g) if c1 else None
    xin = F.interpolate(x0, scale_factor=(2, 2)) if up0 else x0
    if c1:
        xin = torch.cat((xin, x1), 1)
    y = F.conv2d(xin, wgt, bias, stride=stride, padding=k // 2)
    bn_mod = None
    if bn:
        bn_mod = torch.nn.BatchNorm2d(c_out).eval()
        bn_mod.running_mean.copy_(torch.randn(c_out, generator=g) * 0.1)
        bn_mod.running_var.copy_(torch.rand(c_out, generator=g) + 0.5)
        with torch.no_grad():
            bn_mod.weight.copy_(torch.rand(c_out, generator=g) + 0.5)
            bn_mod.bias.copy_(torch.randn(c_out, generator=g) * 0.1)
            y = bn_mod(y)
    if relu:
        y = F.relu(y)
    d = ops.conv_desc(n, h, w, c0, c_out, k, stride, relu, c1=c1, up0=up0, math=math)
    scale, shift = ops.fold_bn(bias.cuda(), bn_mod.cuda() if bn_mod else None, c_out)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    if math == "sp":      # split-planar engine: NHWC -> SP, conv, SP -> NHWC
        packed, wmul = ops.sp_pack_conv_weights(d, wgt.cuda())
        out = ops.sp_conv2d(d, ops.SpTensor.from_nhwc(nhwc(x0)), packed, scale / wmul, shift,
                            src1=ops.SpTensor.from_nhwc(nhwc(x1)) if c1 else None).nhwc()
    else:
        packed = ops.pack_conv_weights(d, wgt.cuda())
        out = ops.conv2d(d, nhwc(x0), packed, scale, shift, src1=nhwc(x1) if c1 else None)
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2)
    err = (got - y).abs().max().item()
    assert got.shape == y.shape
    assert err <= TOL, "max abs err %.3e (ref absmax %.3f)" % (err, y.abs().max().item())
    return err


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("c0,c_out", [(13, 32), (32, 32), (32, 64), (64, 64), (64, 128),
                                      (128, 256), (256, 512)])
def test_conv3x3_stride1(c0, c_out, math):
    _run(2, 32, 32, c0, c_out, 3, math=math)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("c0,c_out", [(32, 64), (64, 128), (128, 256), (256, 512)])
def test_conv3x3_stride2(c0, c_out, math):
    _run(2, 32, 32, c0, c_out, 3, stride=2, math=math)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("c0,c1,c_out", [(512, 256, 256), (256, 128, 128), (128, 64, 64),
                                         (64, 32, 32)])
def test_conv3x3_upsample_concat(c0, c1, c_out, math):
    _run(2, 32, 32, c0, c_out, 3, c1=c1, up0=True, math=math)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("c0,c_out,relu,bn", [(32, 12, False, False), (32, 36, False, False),
                                              (64, 64, True, True), (128, 128, True, True),
                                              (256, 256, False, False), (256, 128, False, False)])
def test_conv1x1(c0, c_out, relu, bn, math):
    _run(2, 32, 32, c0, c_out, 1, relu=relu, bn=bn, math=math)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("h,w", [(8, 8), (20, 44), (17, 33), (256, 256)])
def test_conv3x3_ragged_and_full_size_tiles(h, w, math):
    # sizes that do not divide the 8x32 / 8x16 tiles exercise the edge guards;
    # 256x256 is the BASELINE plane size
    _run(1, h, w, 32, 32, 3, math=math)
    if h % 2 == 0 and h < 256:
        _run(1, h, w, 32, 64, 3, stride=2, math=math)


@pytest.mark.parametrize("math", MATHS)
def test_conv_many_tiles_per_workgroup(math):
    """enough tiles that the persistent workgroups each walk several items, with a
    ragged right/bottom edge"""
    _run(24, 250, 252, 32, 32, 3, math=math)
    _run(24, 124, 120, 64, 64, 3, math=math)


@pytest.mark.parametrize("engine", ["f16x3", "sp"])
def test_split_f16_extremes(engine):
    """split-f16 keeps small magnitudes (fp16-subnormal lo parts) and large ones"""
    from disconet_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 32, 16, 16, generator=g)
    x[:, :8] *= 1e-4
    x[:, 8:16] *= 300.0
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.05
    y = F.conv2d(x.double(), w.double(), None, padding=1).float()
    d = ops.conv_desc(1, 16, 16, 32, 32, 3, 1, False, math=engine)
    scale, shift = ops.fold_bn(torch.zeros(32).cuda(), None, 32)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    if engine == "sp":
        packed, wmul = ops.sp_pack_conv_weights(d, w.cuda())
        assert wmul >= 1024.0            # |w| ~ 0.05: lifted out of the f16 subnormal range
        out = ops.sp_conv2d(d, ops.SpTensor.from_nhwc(xin), packed, scale / wmul, shift).nhwc()
    else:
        packed = ops.pack_conv_weights(d, w.cuda())
        out = ops.conv2d(d, xin, packed, scale, shift)
    got = out.cpu().permute(0, 3, 1, 2)
    rel = ((got - y).abs().max() / y.abs().max()).item()
    assert rel <= 2e-6, rel


def test_conv_rejects_bad_arguments():
    from disconet_amd import ops, _lib
    d = ops.conv_desc(1, 8, 8, 32, 32, 5)
    with pytest.raises(_lib.DnError):
        ops.pack_conv_weights(d, torch.zeros(32, 32, 5, 5, device="cuda"))
    with pytest.raises(_lib.DnError):
        ops.conv2d(ops.conv_desc(1, 8, 8, 32, 32, 3), torch.zeros(1, 8, 8, 32), None, None, None)


@pytest.mark.parametrize("engine", ["f16x3", "sp"])
@pytest.mark.parametrize("c_in,c_out2,split,relu2", [(32, 48, 12, False), (64, 64, 64, True),
                                                     (96, 8, 4, False)])
def test_conv3x3_fused_1x1_stage(c_in, c_out2, split, relu2, engine):
    """dn_conv2d_post1x1: 3x3 conv (64 ch) + affine + ReLU, then 1x1 + affine (+ReLU),
    one or two outputs, vs the two torch ops"""
    from disconet_amd import ops
    g = torch.Generator().manual_seed(4)
    n, h, w = 3, 40, 24                      # not a multiple of the 8x16 tile
    x = torch.randn(n, c_in, h, w, generator=g)
    w1 = torch.randn(64, c_in, 3, 3, generator=g) * (2.0 / (c_in * 9)) ** 0.5
    s1, t1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(c_out2, 64, generator=g) * (2.0 / 64) ** 0.5
    s2, t2 = torch.rand(c_out2, generator=g) + 0.5, torch.randn(c_out2, generator=g) * 0.1
    hmid = F.relu(F.conv2d(x, w1, None, padding=1) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1))
    y = F.conv2d(hmid, w2.view(c_out2, 64, 1, 1)) * s2.view(1, -1, 1, 1) + t2.view(1, -1, 1, 1)
    if relu2:
        y = F.relu(y)
    d = ops.conv_desc(n, h, w, c_in, 64, 3, 1, True, math=engine)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    out_a = torch.empty(n, h, w, split, device="cuda")
    out_b = torch.empty(n, h, w, c_out2 - split, device="cuda") if split < c_out2 else None
    if engine == "sp":
        packed, m1 = ops.sp_pack_conv_weights(d, w1.cuda())
        packed2, m2 = ops.sp_pack_post1x1_weights(w2.cuda())
        if out_b is None:                 # single output: the SP form
            out_a = ops.SpTensor(n, h, w, c_out2, device="cuda")
        ops.sp_conv2d_post1x1(d, ops.SpTensor.from_nhwc(xin), packed, s1.cuda() / m1, t1.cuda(), packed2,
                              s2.cuda() / m2, t2.cuda(), c_out2, split, relu2, out_a, out_b)
        if out_b is None:
            out_a = out_a.nhwc()
    else:
        packed = ops.pack_conv_weights(d, w1.cuda())
        packed2 = ops.pack_post1x1_weights(w2.cuda())
        ops.conv2d_post1x1(d, xin, packed, s1.cuda(), t1.cuda(),
                           packed2, s2.cuda(), t2.cuda(), c_out2, split, relu2, out_a, out_b)
    torch.cuda.synchronize()
    got = out_a.cpu() if out_b is None else torch.cat([out_a.cpu(), out_b.cpu()], -1)
    err = (got.permute(0, 3, 1, 2) - y).abs().max().item()
    assert err <= TOL, "max abs err %.3e" % err


def test_conv_random_shapes_fuzz():
    """seeded sweep over odd sizes / channel counts / modes: ragged tiles, channel tails
    (c_in not a multiple of the chunk, c_out not a multiple of 4 or 32), 1x1 and 3x3,
    stride 2, upsample+concat, both math modes"""
    import random
    rnd = random.Random(1234)
    fails = []
    for case in range(48):
        k = rnd.choice([1, 3, 3])
        stride = rnd.choice([1, 1, 2]) if k == 3 else 1
        math = rnd.choice(MATHS)
        n = rnd.randint(1, 3)
        up0 = k == 3 and stride == 1 and rnd.random() < 0.3
        h, w = rnd.randint(3, 40), rnd.randint(3, 70)
        if up0:
            h, w = 2 * ((h + 1) // 2), 2 * ((w + 1) // 2)
        kcp = 16 if k == 3 else 32
        if up0 or rnd.random() < 0.25:
            c0 = rnd.choice([kcp, 2 * kcp, 4 * kcp])           # concat: c0 multiple of the packed chunk
            c1 = rnd.choice([4, 8, 20, 32, 36])
        else:
            c0, c1 = rnd.choice([3, 13, 16, 24, 32, 40, 64, 100]), 0
        c_out = rnd.choice([1, 6, 12, 32, 36, 64, 72, 100, 128])
        relu, bn = rnd.random() < 0.7, rnd.random() < 0.7
        if math == "sp" and c1 and (k == 1 or c0 % 16):
            c1 = 0                        # the SP engine concatenates only in front of 3x3 convs
        try:
            _run(n, h, w, c0, c_out, k, stride=stride, relu=relu, c1=c1, up0=up0, bn=bn,
                 seed=100 + case, math=math)
        except AssertionError as e:
            fails.append((case, n, h, w, c0, c1, c_out, k, stride, up0, math, str(e)[:80]))
    assert not fails, fails


@pytest.mark.parametrize("h,w,c0,c_out,k,stride", [(32, 32, 256, 256, 3, 1), (20, 28, 48, 36, 3, 1), (32, 32, 64, 128, 3, 2),
                                                   (24, 40, 128, 128, 1, 1)])
def test_sp_conv_dual_output(h, w, c0, c_out, k, stride):
    """dn_spconv2d_dual: the fp32 NHWC copy written by the same epilogue is the unsplit value -- the SP output
    is its hi + lo split, bit for bit -- and the SP output equals the plain launch's"""
    from disconet_amd import ops
    g = torch.Generator().manual_seed(h + c0)
    wgt = (torch.randn(c_out, c0, k, k, generator=g) * (2.0 / (c0 * k * k)) ** 0.5).cuda()
    x = ops.SpTensor.from_nhwc(torch.randn(3, h, w, c0, generator=g).cuda())
    scale = (torch.rand(c_out, generator=g) + 0.5).cuda()
    shift = (torch.randn(c_out, generator=g) * 0.1).cuda()
    d = ops.conv_desc(3, h, w, c0, c_out, k, stride, True, math="sp")
    packed, wmul = ops.sp_pack_conv_weights(d, wgt)
    plain = ops.sp_conv2d(d, x, packed, scale / wmul, shift)
    sp, flat = ops.sp_conv2d(d, x, packed, scale / wmul, shift, nhwc_copy=True)
    torch.cuda.synchronize()
    assert torch.equal(sp.data, plain.data)
    assert torch.equal(ops.SpTensor.from_nhwc(flat).data, sp.data)
    assert (flat - sp.nhwc()).abs().max().item() <= 2e-6 * max(1.0, flat.abs().max().item())


@pytest.mark.parametrize("n,h,w,c0,c1,c_out,stride,up0,ks", [
    (20, 32, 32, 512, 256, 256, 1, True, 4),     # conv5_1 (the tap-merged kernel): 640 tiles, the last 128 split
    (12, 32, 32, 256, 0, 256, 1, False, 4),      # conv3_2 / conv5_2 shape
    (9, 16, 16, 512, 0, 512, 1, False, 4),       # conv4_2
    (10, 32, 32, 256, 0, 512, 2, False, 2),      # conv4_1 (stride 2), two slices
    (5, 20, 44, 64, 32, 48, 1, True, 2),         # ragged up + concat, c_out not a tile multiple
    (6, 24, 40, 128, 0, 80, 1, False, 4),
])
def test_k_sliced_conv_is_batch_invariant(n, h, w, c0, c1, c_out, stride, up0, ks):
    """dn_spconv2d_ks: a layer's K slices define its result -- the sum, in slice order, of the slices' accumulation
    chains -- however a launch distributes them: all tiles whole (no workspace), the under-filled round split
    through the workspace, or the first images as a launch of their own give the SAME BITS; the values follow the
    torch-CPU conv within the engine's tolerance; a second fp32 NHWC output rides along"""
    from disconet_amd import ops, _lib
    import ctypes
    g = torch.Generator().manual_seed(n * 100 + c_out)
    cin = c0 + c1
    wgt = torch.randn(c_out, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    h0, w0 = (h // 2, w // 2) if up0 else (h, w)
    x0 = torch.randn(n, h0, w0, c0, generator=g).clamp_(min=0)
    x1 = torch.randn(n, h, w, c1, generator=g).clamp_(min=0) if c1 else None
    scale = torch.rand(c_out, generator=g) + 0.5
    shift = torch.randn(c_out, generator=g) * 0.1
    xin = x0.permute(0, 3, 1, 2)
    if up0:
        xin = F.interpolate(xin, scale_factor=(2, 2))
    if c1:
        xin = torch.cat((xin, x1.permute(0, 3, 1, 2)), 1)
    want = F.relu(F.conv2d(xin, wgt, None, stride=stride, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    want = want.permute(0, 2, 3, 1)

    s0 = ops.SpTensor.from_nhwc(x0.cuda())
    s1 = ops.SpTensor.from_nhwc(x1.cuda()) if c1 else None
    lib = _lib.load()

    def run(n_img, workspace):
        d = ops.conv_desc(n_img, h, w, c0, c_out, 3, stride, True, c1=c1, up0=up0, math="sp")
        packed, wmul = ops.sp_pack_conv_weights(d, wgt.cuda())
        ho, wo = ops.conv_out_hw(d)
        out = ops.SpTensor(n_img, ho, wo, c_out, device="cuda")
        out.data.fill_(255)
        nb = int(lib.dn_spconv_workspace_bytes(ctypes.byref(d), ks)) if workspace else 0
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
        sc, sh = (scale / wmul).cuda(), shift.cuda()         # (named: a temporary would be freed before the launch reads it)
        ops.check(lib.dn_spconv2d_ks(ctypes.byref(d), ks, ops._ptr(s0.data), ops._ptr(s1.data) if s1 is not None else None,
                                     ops._ptr(packed), ops._ptr(sc), ops._ptr(sh),
                                     ops._ptr(out.data), None, 0, ops._ptr(ws) if workspace else None, nb,
                                     ops._stream()), "dn_spconv2d_ks")
        torch.cuda.synchronize()
        return out

    split = run(n, True)
    whole = run(n, False)
    few = run(min(n, 3), True)
    assert torch.equal(split.data, whole.data)
    per_img = split.data.numel() // n
    assert torch.equal(few.data.reshape(-1), split.data.reshape(-1)[:per_img * min(n, 3)])
    got = split.nhwc().cpu()
    assert (got - want).abs().max().item() <= TOL * max(1.0, want.abs().max().item() / 4)
    # through ops.sp_conv2d (what the plan calls), with the second output of the exchanged level
    if not up0:
        d = ops.conv_desc(n, h, w, c0, c_out, 3, stride, True, c1=c1, up0=up0, math="sp")
        packed, wmul = ops.sp_pack_conv_weights(d, wgt.cuda())
        sp, flat = ops.sp_conv2d(d, s0, packed, (scale / wmul).cuda(), shift.cuda(), src1=s1, nhwc_copy=True, kslices=ks)
        torch.cuda.synchronize()
        assert torch.equal(sp.data, split.data)
        assert torch.equal(ops.SpTensor.from_nhwc(flat).data, sp.data)


@pytest.mark.parametrize("n,h,w,c,c_out", [
    (4, 64, 64, 13, 32),        # conv_pre_1's shape at a small map
    (3, 20, 44, 13, 32),        # ragged: partial tiles right and bottom
    (2, 40, 72, 29, 24),        # two 16-channel chunks of bits, c_out below the tile
    (5, 8, 32, 5, 32),          # exactly one tile per image
    (1, 3, 5, 16, 8),           # smaller than a tile
])
def test_bit_grid_source_equals_hi_only_source(n, h, w, c, c_out):
    """dn_spconv2d with math = 4 (source 0 an occupancy bit grid, expanded into the patch stage by VALU + ds_write)
    gives the bits of math = 3 on the hi-only SP tensor of the same grid and of the full SP form, and follows the
    torch-CPU conv; many tiles per workgroup (persistent loop: first / steady / last tile) are covered by n x tiles"""
    from disconet_amd import ops
    g = torch.Generator().manual_seed(c * 7 + h)
    occ = (torch.rand(n, h, w, c, generator=g) < 0.15).float()
    occ[0, 0, 0, :] = 1.0                     # corners: halo rows / columns of the patch are zero padding
    occ[-1, -1, -1, :] = 1.0
    wgt = torch.randn(c_out, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5
    scale = torch.rand(c_out, generator=g) + 0.5
    shift = torch.randn(c_out, generator=g) * 0.1
    want = F.relu(F.conv2d(occ.permute(0, 3, 1, 2), wgt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    want = want.permute(0, 2, 3, 1)

    full = ops.SpTensor.from_nhwc(occ.cuda())
    hi = ops.SpTensor(n, h, w, c, device="cuda", hi_only=True, data=full.data[:, :, :2].contiguous())
    words = (occ.to(torch.int64) << torch.arange(c, dtype=torch.int64)).sum(-1)
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)      # bit 31 set: the int32 pattern
    bits = ops.SpTensor(n, h, w, c, device="cuda", bits=True, data=words.cuda())
    assert torch.equal(bits.nhwc(), occ.cuda())

    outs = []
    for src in (full, hi, bits):
        d = ops.conv_desc(n, h, w, c, c_out, 3, 1, True, math="sp")
        packed, wmul = ops.sp_pack_conv_weights(d, wgt.cuda())
        sc, sh = (scale / wmul).cuda(), shift.cuda()
        out = ops.SpTensor(n, h, w, c_out, device="cuda")
        out.data.fill_(255)
        outs.append(ops.sp_conv2d(d, src, packed, sc, sh, out=out))
        torch.cuda.synchronize()
    assert torch.equal(outs[1].data, outs[0].data)
    assert torch.equal(outs[2].data, outs[0].data)
    assert (outs[2].nhwc().cpu() - want).abs().max().item() <= TOL * max(1.0, want.abs().max().item() / 4)


def test_bit_grid_source_refusals():
    """a bit grid is source 0 of a 3x3 stride-1 single-source layer whose weights stay in LDS; anything else is refused"""
    from disconet_amd import ops, _lib
    bits = ops.SpTensor(1, 16, 16, 13, device="cuda", bits=True, data=torch.zeros(1, 16, 16, dtype=torch.int32, device="cuda"))
    g = torch.Generator().manual_seed(0)

    def attempt(c_out, k, stride):
        d = ops.conv_desc(1, 16, 16, 13, c_out, k, stride, True, math="sp")
        packed, wmul = ops.sp_pack_conv_weights(d, torch.randn(c_out, 13, k, k, generator=g).cuda())
        sc, sh = torch.ones(c_out, device="cuda"), torch.zeros(c_out, device="cuda")
        return ops.sp_conv2d(d, bits, packed, sc, sh)

    attempt(32, 3, 1)
    for c_out, k, stride in ((64, 3, 1), (32, 3, 2), (32, 1, 1)):
        with pytest.raises(_lib.DnError):
            attempt(c_out, k, stride)
    with pytest.raises(_lib.DnError):
        ops.SpTensor(1, 16, 16, 33, device="cuda", bits=True)


@pytest.mark.parametrize("n,h,w,c,c_out", [
    (3, 64, 64, 13, 32),        # the stem's shape at a small map: 2 x 2 tiles per image
    (2, 37, 70, 13, 32),        # ragged: partial tiles right and bottom, odd sizes
    (5, 16, 32, 16, 32),        # exactly one tile per image, a full 16-channel chunk of bits
    (1, 5, 3, 7, 20),           # smaller than a tile, c_out below 32 (second chunk partly empty)
    (2, 48, 96, 13, 16),        # c_out = 16: one output chunk
    (40, 32, 64, 13, 32),       # more tiles than workgroups' first round: the persistent loop
])
def test_stem_pair_launch_equals_two_launches(n, h, w, c, c_out):
    """dn_spconv2d_pre_pair (conv_pre_1 -> conv_pre_2 in one launch, the intermediate map in LDS) writes the bytes of
    dn_spconv2d(math 4) followed by dn_spconv2d, and follows the torch-CPU reference of the two layers"""
    from disconet_amd import ops
    g = torch.Generator().manual_seed(h * 131 + c_out)
    occ = (torch.rand(n, h, w, c, generator=g) < 0.1).float()
    occ[0, 0, 0, :] = 1.0
    occ[-1, -1, -1, :] = 1.0
    w1 = torch.randn(32, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5
    w2 = torch.randn(c_out, 32, 3, 3, generator=g) * (2.0 / (32 * 9)) ** 0.5
    s1, t1 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2     # shifts of both signs: ReLU bites
    s2, t2 = torch.rand(c_out, generator=g) + 0.5, torch.randn(c_out, generator=g) * 0.2
    x = occ.permute(0, 3, 1, 2)
    mid = F.relu(F.conv2d(x, w1, None, padding=1) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1))
    want = F.relu(F.conv2d(mid, w2, None, padding=1) * s2.view(1, -1, 1, 1) + t2.view(1, -1, 1, 1)).permute(0, 2, 3, 1)

    words = (occ.to(torch.int64) << torch.arange(c, dtype=torch.int64)).sum(-1).to(torch.int32)
    bits = ops.SpTensor(n, h, w, c, device="cuda", bits=True, data=words.cuda())
    d1 = ops.conv_desc(n, h, w, c, 32, 3, 1, True, math="sp")
    d2 = ops.conv_desc(n, h, w, 32, c_out, 3, 1, True, math="sp")
    p1, m1 = ops.sp_pack_conv_weights(d1, w1.cuda())
    p2, m2 = ops.sp_pack_conv_weights(d2, w2.cuda())
    sc1, sh1, sc2, sh2 = (s1 / m1).cuda(), t1.cuda(), (s2 / m2).cuda(), t2.cuda()
    assert ops.sp_conv2d_pre_pair_supported(d1, d2)
    two = ops.sp_conv2d(d2, ops.sp_conv2d(d1, bits, p1, sc1, sh1), p2, sc2, sh2)
    out = ops.SpTensor(n, h, w, c_out, device="cuda")
    out.data.fill_(255)
    one = ops.sp_conv2d_pre_pair(d1, d2, bits, p1, sc1, sh1, p2, sc2, sh2, out=out)
    torch.cuda.synchronize()
    assert torch.equal(one.data, two.data)
    assert (one.nhwc().cpu() - want).abs().max().item() <= TOL * max(1.0, want.abs().max().item() / 4)


def test_stem_pair_refusals():
    from disconet_amd import ops, _lib
    mk = lambda c0, c_out, k=3, s=1, hw=16: ops.conv_desc(2, hw, hw, c0, c_out, k, s, True, math="sp")
    assert ops.sp_conv2d_pre_pair_supported(mk(13, 32), mk(32, 32))
    for d1, d2 in ((mk(17, 32), mk(32, 32)), (mk(13, 64), mk(64, 32)), (mk(13, 32), mk(32, 64)), (mk(13, 32), mk(32, 32, s=2)),
                   (mk(13, 32, k=1), mk(32, 32)), (mk(13, 32), mk(32, 32, hw=32))):
        assert not ops.sp_conv2d_pre_pair_supported(d1, d2)
    bits = ops.SpTensor(2, 16, 16, 13, device="cuda", bits=True, data=torch.zeros(2, 16, 16, dtype=torch.int32, device="cuda"))
    z = torch.zeros(64, device="cuda")
    pk = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.DnError):
        ops.sp_conv2d_pre_pair(mk(13, 32), mk(32, 64), bits, pk, z, z, pk, z, z)


def test_three_weight_stages_are_bit_identical_to_two():
    """round 5: the 8 x 8-pixel tiles request a step's weights two steps ahead through a third LDS stage (SpTile NB = 3, counted
    vmcnt waits from an issue history).  Same operands, same MFMA order: the outputs of the layers that take those tiles -- stride-2
    layers, a 16 x 16 long-K layer, ragged maps, one- and many-chunk K loops, more tiles than resident workgroups -- must equal the
    two-stage form's bit for bit.  DN_SP_B3 is read once per process, so each form runs in a process of its own."""
    import os
    import subprocess
    import sys
    code = r'''
import torch
from disconet_amd import ops
out = []
for (n, h, w, c0, c_out, stride) in [(20, 64, 64, 128, 256, 2), (3, 40, 72, 32, 96, 2), (20, 16, 16, 512, 512, 1), (2, 24, 24, 16, 64, 2),
                                      (5, 16, 24, 64, 128, 1), (64, 32, 32, 32, 64, 2)]:
    g = torch.Generator().manual_seed(n * 100 + c0)
    d = ops.conv_desc(n, h, w, c0, c_out, 3, stride, True, math="sp")
    wt = (torch.randn(c_out, c0, 3, 3, generator=g) * (2.0 / (9 * c0)) ** 0.5).cuda()
    packed, wmul = ops.sp_pack_conv_weights(d, wt)
    x = ops.SpTensor.from_nhwc(torch.randn(n, h, w, c0, generator=g).clamp_(min=0).cuda())
    y = ops.sp_conv2d(d, x, packed, (torch.ones(c_out) / wmul).cuda(), (torch.randn(c_out, generator=g) * 0.1).cuda())
    torch.cuda.synchronize()
    out.append(int(y.data.view(torch.int32).long().sum()) ^ (int(y.data.view(torch.int32)[::7].long().sum()) << 1))
print("CHECKSUMS", out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = {}
    for b3 in ("0", "3"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, DN_SP_B3=b3), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-600:]
        sums[b3] = [l for l in r.stdout.splitlines() if l.startswith("CHECKSUMS")][0]
    assert sums["0"] == sums["3"], sums
