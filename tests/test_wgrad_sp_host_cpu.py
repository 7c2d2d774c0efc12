"""Host-side logic of the split-f16 weight gradient (include/disconet_train.h :: dn_conv_wgrad_sp, csrc/wgrad_sp.inl) that needs no
GPU: which layers of the detector get which kernel, the workspace formula, argument refusal -- and a SPECIFICATION test of the
kernels' LDS image and fragment maps: the index arithmetic documented in wgrad_sp.inl (pixel pairs as dwords, tap column 1 / 2 as
v_alignbit of neighbouring dwords, the stride-2 kernel's column-parity planes, waves as quadrants or as rows) restated in numpy and
checked against the definition of dW.  The device code itself is checked on the GPU (tests/test_gpu_train_ops.py)."""
import ctypes

import numpy as np
import pytest

from disconet_amd import _lib, ops

# (name, h_in, c0, c1, up0, c_out, stride) of the detector's 3x3 layers at 256 x 256 (disconet_amd/train.py :: _graph)
LAYERS = [
    ("conv_pre_1", 256, 13, 0, 0, 32, 1, 32), ("conv_pre_2", 256, 32, 0, 0, 32, 1, 32), ("conv1_1", 256, 32, 0, 0, 64, 2, 32),
    ("conv1_2", 128, 64, 0, 0, 64, 1, 64), ("conv2_1", 128, 64, 0, 0, 128, 2, 64), ("conv2_2", 64, 128, 0, 0, 128, 1, 64),
    ("conv3_1", 64, 128, 0, 0, 256, 2, 64), ("conv3_2", 32, 256, 0, 0, 256, 1, 64), ("conv4_1", 32, 256, 0, 0, 512, 2, 64),
    ("conv4_2", 16, 512, 0, 0, 512, 1, 64), ("conv5_1", 32, 512, 256, 1, 256, 1, 64), ("conv5_2", 32, 256, 0, 0, 256, 1, 64),
    ("conv6_1", 64, 256, 128, 1, 128, 1, 64), ("conv6_2", 64, 128, 0, 0, 128, 1, 64), ("conv7_1", 128, 128, 64, 1, 64, 1, 64),
    ("conv7_2", 128, 64, 0, 0, 64, 1, 64), ("conv8_1", 256, 64, 32, 1, 32, 1, 32), ("conv8_2", 256, 32, 0, 0, 32, 1, 32),
    ("heads1", 256, 32, 0, 0, 64, 1, 32),
]


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: l[0])
def test_every_3x3_layer_of_the_detector_has_a_split_f16_weight_gradient(layer):
    name, hw, c0, c1, up0, c_out, stride, want = layer
    lib = _lib.load()
    d = ops.conv_desc(20, hw, hw, c0, c_out, ksize=3, stride=stride, c1=c1, up0=up0, relu=False)
    cb = lib.dn_conv_wgrad_sp_supported(ctypes.byref(d))
    assert cb == want
    # one resident generation of workgroups, each with its own partial block of 9 taps x cb x cb floats
    ho = hw // stride
    th = {(64, 1): 4, (32, 1): 8, (64, 2): 1, (32, 2): 4}[(cb, stride)]
    n_tiles = 20 * ((ho + th - 1) // th) * ((ho + 15) // 16)
    blocks = -(-c_out // cb) * (-(-c0 // cb) + -(-c1 // cb))
    s = min(512 // blocks, n_tiles // 4)
    s = s & ~7 if s >= 8 else max(s, 1)
    assert lib.dn_conv_wgrad_sp_workspace(ctypes.byref(d)) == s * blocks * 9 * cb * cb * 4
    assert s * blocks <= 512


def test_split_f16_weight_gradient_refuses_what_it_cannot_take():
    lib = _lib.load()
    for kw in (dict(c0=64, c_out=64, ksize=1), dict(c0=64, c_out=16), dict(c0=48, c_out=64, stride=2), dict(c0=64, c1=13, c_out=64),
               dict(c0=64, c_out=30)):
        d = ops.conv_desc(2, 32, 32, kw["c0"], kw["c_out"], ksize=kw.get("ksize", 3), stride=kw.get("stride", 1), c1=kw.get("c1", 0),
                          relu=False)
        assert lib.dn_conv_wgrad_sp_supported(ctypes.byref(d)) == 0 and lib.dn_conv_wgrad_sp_workspace(ctypes.byref(d)) == 0
        assert lib.dn_conv_wgrad_sp(ctypes.byref(d), 16, None, 16, 16, 16, 0, 0, 256.0, 16.0, None) != 0
    d = ops.conv_desc(2, 32, 32, 64, 64, ksize=3, relu=False)
    assert lib.dn_conv_wgrad_sp(ctypes.byref(d), 16, None, 16, 16, 16, 0, 0, 3.0, 16.0, None) != 0      # a lift that is no power of two
    assert b"powers of two" in lib.dn_last_error()
    assert lib.dn_conv_wgrad_sp(ctypes.byref(d), 8, None, 16, 16, 16, 0, 0, 256.0, 16.0, None) != 0     # a source off 16 bytes
    assert b"aligned" in lib.dn_last_error()


def test_bn_backward_refuses_bad_gate_and_gradient_modes():
    """relu = 2 (byte mask) needs c % 4 == 0; up_a = 2 (space-to-depth gradient) needs even maps and ld_a >= 4 c"""
    lib = _lib.load()
    p = ctypes.c_void_p(16)

    def partial(relu, up_a, h, w, c, ld_a):
        return lib.dn_bn_train_backward_partial(p, ld_a, up_a, None, 0, p, p, p, p, 1e-5, relu, 1, h, w, 2, c, p, 1 << 30, p, p, 0, None)
    assert partial(3, 0, 8, 8, 32, 32) != 0 and b"relu" in lib.dn_last_error()
    assert partial(2, 0, 8, 8, 6, 6) != 0 and b"c % 4" in lib.dn_last_error()
    assert partial(1, 3, 8, 8, 32, 32) != 0 and b"up_a" in lib.dn_last_error()
    assert partial(1, 2, 7, 8, 32, 128) != 0 and b"up_a" in lib.dn_last_error()
    assert partial(1, 2, 8, 8, 32, 64) != 0 and b"up_a" in lib.dn_last_error()


# ---- the index maps of csrc/wgrad_sp.inl, restated ---------------------------------------------------------------------------
def _frag(dwords):
    """4 dwords (each a (low half, high half) pair of values) -> the 8 K entries of a lane's fragment"""
    return [v for d in dwords for v in d]


def _alignbit(hi, lo):
    """v_alignbit_b32(hi, lo, 16): low half = lo's high half, high half = hi's low half"""
    return (lo[1], hi[0])


def _emulate(cb, stride, h_in, w_in, rng):
    s1 = stride == 1
    th = (4 if cb == 64 else 8) if s1 else (1 if cb == 64 else 4)
    rows_per_wave = (th if cb == 64 else th // 4)
    ph = th + 2 if s1 else 2 * th + 1
    rp = 9 if s1 else 17                                   # pixel pairs per patch row
    qn, pitch = cb // 4, cb + 8
    h_out, w_out = ((h_in - 1) // stride + 1, (w_in - 1) // stride + 1)
    x, dz = rng.standard_normal((h_in, w_in, cb)), rng.standard_normal((h_out, w_out, cb))
    xp = np.pad(x, ((1, 2), (1, 2), (0, 0)))
    ref = np.stack([np.einsum("hwo,hwi->oi", dz, xp[ty:ty + stride * h_out:stride, tx:tx + stride * w_out:stride])
                    for ty in range(3) for tx in range(3)])
    acc = np.zeros((4, 9, 32, 32))
    for oy0 in range(0, h_out, th):
        for ox0 in range(0, w_out, 16):
            X = np.zeros((ph * rp * pitch, 2))
            D = np.zeros((th * 8 * pitch, 2))
            for idx in range(ph * rp * qn):                # the staging pass: item = (pixel pair, channel quad)
                pr, q = divmod(idx, qn)
                prow, pp = divmod(pr, rp)
                iy = (oy0 - 1 + prow) if s1 else (2 * oy0 - 1 + prow)
                for e in range(2):
                    ix = (ox0 - 1 + 2 * pp + e) if s1 else (2 * (ox0 + 2 * pp + e) - 1 if pp < 9 else 2 * (ox0 + 2 * (pp - 9) + e))
                    if 0 <= iy < h_in and 0 <= ix < w_in:
                        X[pr * pitch + 4 * q:pr * pitch + 4 * q + 4, e] = x[iy, ix, 4 * q:4 * q + 4]
            for idx in range(th * 8 * qn):
                pr, q = divmod(idx, qn)
                oy = oy0 + (pr >> 3)
                for e in range(2):
                    ox = ox0 + 2 * (pr & 7) + e
                    if oy < h_out and ox < w_out:
                        D[pr * pitch + 4 * q:pr * pitch + 4 * q + 4, e] = dz[oy, ox, 4 * q:4 * q + 4]
            for wave in range(4):
                wm, wn = (wave >> 1, wave & 1) if cb == 64 else (0, 0)
                row0 = 0 if cb == 64 else wave * rows_per_wave
                for r in range(rows_per_wave):
                    for ty in range(3):
                        prow = (row0 + r + ty) if s1 else 2 * (row0 + r) + ty
                        A, B = np.zeros((32, 16)), np.zeros((3, 16, 32))
                        for lh in range(2):
                            for li in range(32):
                                col = lambda pair: tuple(X[(prow * rp + pair) * pitch + wn * 32 + li])
                                b = [col(lh * 4 + i) for i in range(5)]
                                if s1:
                                    f = [b[:4], [_alignbit(b[i + 1], b[i]) for i in range(4)], b[1:5]]
                                else:
                                    ev = [col(9 + lh * 4 + i) for i in range(4)]
                                    f = [b[:4], ev, [_alignbit(b[i + 1], b[i]) for i in range(4)]]
                                for tx in range(3):
                                    B[tx, lh * 8:lh * 8 + 8, li] = _frag(f[tx])
                                a = [tuple(D[((row0 + r) * 8 + lh * 4 + i) * pitch + wm * 32 + li]) for i in range(4)]
                                A[li, lh * 8:lh * 8 + 8] = _frag(a)
                        for tx in range(3):
                            acc[wave, ty * 3 + tx] += A @ B[tx]
    if cb == 64:
        got = np.zeros((9, 64, 64))
        for wave in range(4):
            got[:, (wave >> 1) * 32:(wave >> 1) * 32 + 32, (wave & 1) * 32:(wave & 1) * 32 + 32] = acc[wave]
    else:
        got = acc.sum(0)
    return got, ref


@pytest.mark.parametrize("cb,stride,h,w", [(64, 1, 8, 32), (32, 1, 10, 24), (64, 2, 6, 40), (32, 2, 10, 18), (64, 2, 7, 21)])
def test_lds_image_and_fragment_maps_of_the_split_f16_weight_gradient(cb, stride, h, w):
    got, ref = _emulate(cb, stride, h, w, np.random.default_rng(cb + stride + h))
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
