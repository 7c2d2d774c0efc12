"""Tensor-level wrappers over the training C ABI (include/disconet_train.h).

Same rules as ops.py: float32 HIP tensors, torch's current stream, no CPU path.
Maps are NHWC; a tensor argument that is a channel slice of a wider map is passed
as (tensor_view, ld) with the view's data_ptr at the first channel.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check
from .ops import _need_gpu, _ptr, _stream, _stream_handle, conv_desc


def _ws(device, nbytes, cache={}):
    """grow-only scratch buffer per (device, stream): two streams may run backward kernels side by
    side (train.py), each needs its own partial-sum space"""
    key = (device.index, _stream_handle())       # (the launches that use it go to the current device's current stream)
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        cache[key] = buf
    return buf


def _ld(t):
    """row stride (floats) of an NHWC map or a channel slice of one"""
    assert t.stride(-1) == 1, "channel axis must be contiguous"
    return t.stride(-2)


# ---- conv backward ------------------------------------------------------------------------
def conv_wgrad(desc, src0, src1, dz, dw, dw_cin_total=0, accumulate=False, sp_lift=None, x_lift=16.0, dz_sp=None):
    """dw [c_out, c_in(, k, k)] (a column block of a [c_out, dw_cin_total, k, k] tensor when
    dw_cin_total is given) = weight gradient of the layer `desc` (desc.ldo = row stride of dz).
    sp_lift: the power-of-two lift of dz (max |dz| * sp_lift ~ 2^8) -> the split-f16 kernel (dn_conv_wgrad_sp; the layer must
    pass conv_wgrad_sp_supported); None: the exact-fp32 MFMA kernels.
    dz_sp (with sp_lift): the ops.SpTensor that holds dz * sp_lift (bn_backward's sp_out) -- read in place of dz, which may then
    be None (dn_conv_wgrad_sp_z: the same bits)."""
    _need_gpu(src0, src1, dz, dw)
    lib = _lib.load()
    if sp_lift is not None:
        nbytes = lib.dn_conv_wgrad_sp_workspace(ctypes.byref(desc))
        if nbytes == 0:
            raise _lib.DnError("conv_wgrad: this layer has no split-f16 weight-gradient kernel (conv_wgrad_sp_supported)")
        ws = _ws(dw.device, nbytes)
        if dz_sp is not None:
            if dz_sp.hi_only or dz_sp.bits or desc.c_out % 16:
                raise _lib.DnError("conv_wgrad: dz_sp must be a full SP tensor of a layer with c_out % 16 == 0")
            check(lib.dn_conv_wgrad_sp_z(ctypes.byref(desc), _ptr(src0), _ptr(src1), _ptr(dz_sp.data), _ptr(ws), _ptr(dw),
                                         int(dw_cin_total), int(bool(accumulate)), float(sp_lift), float(x_lift), _stream()),
                  "dn_conv_wgrad_sp_z")
            return dw
        check(lib.dn_conv_wgrad_sp(ctypes.byref(desc), _ptr(src0), _ptr(src1), _ptr(dz), _ptr(ws), _ptr(dw), int(dw_cin_total),
                                   int(bool(accumulate)), float(sp_lift), float(x_lift), _stream()), "dn_conv_wgrad_sp")
        return dw
    nbytes = lib.dn_conv_wgrad_workspace(ctypes.byref(desc))
    if nbytes == 0:
        check(-1, "dn_conv_wgrad_workspace")
    ws = _ws(dz.device, nbytes)
    check(lib.dn_conv_wgrad(ctypes.byref(desc), _ptr(src0), _ptr(src1), _ptr(dz), _ptr(ws), _ptr(dw),
                            int(dw_cin_total), int(bool(accumulate)), _stream()), "dn_conv_wgrad")
    return dw


def conv_wgrad_sp_supported(desc):
    return bool(_lib.load().dn_conv_wgrad_sp_supported(ctypes.byref(desc)))


def dgrad_weights(w, ci_first=0, c_in=None):
    """w [c_out, cin_total, k, k] -> wt [c_in, c_out, k, k], taps flipped (weights of the
    data-gradient conv for input columns ci_first .. ci_first + c_in)."""
    _need_gpu(w)
    c_out, cin_total, k = w.shape[0], w.shape[1], w.shape[-1] if w.dim() == 4 else 1
    c_in = cin_total - ci_first if c_in is None else c_in
    wt = torch.empty((c_in, c_out, k, k), dtype=torch.float32, device=w.device)
    check(_lib.load().dn_conv_dgrad_weights(_ptr(w), c_out, cin_total, ci_first, c_in, k, _ptr(wt),
                                            _stream()), "dn_conv_dgrad_weights")
    return wt


def dgrad_class_weights(w, py, px, ci_first=0, c_in=None, out=None):
    """parity class (py, px) of a stride-2 3x3 layer's data gradient: -> (v [c_in, c_out, 3, 3], tap mask).
    out: write v there (a contiguous [c_in, c_out, 3, 3] view, e.g. one class's rows of the four-class weight tensor)"""
    _need_gpu(w, out)
    c_out, cin_total = w.shape[0], w.shape[1]
    c_in = cin_total - ci_first if c_in is None else c_in
    if out is not None and (tuple(out.shape) != (c_in, c_out, 3, 3) or not out.is_contiguous() or out.dtype != torch.float32):
        raise _lib.DnError("dgrad_class_weights: out must be a contiguous float32 [c_in, c_out, 3, 3] tensor")
    v = torch.empty((c_in, c_out, 3, 3), dtype=torch.float32, device=w.device) if out is None else out
    mask = ctypes.c_int(0)
    check(_lib.load().dn_conv_dgrad_class_weights(_ptr(w), c_out, cin_total, ci_first, c_in, py, px, _ptr(v),
                                                  ctypes.byref(mask), _stream()), "dn_conv_dgrad_class_weights")
    return v, mask.value


# ---- batch norm, training mode ------------------------------------------------------------
def _folded(sums, n_groups, c):
    """the folded sums [n_groups, 2, c] float64 at the start of a reduction workspace (what a SyncBN exchange all-reduces)"""
    return sums[:n_groups * 2 * c * 8].view(torch.float64)


def bn_stats(z, n_groups=1, sync=None, norm_rows=None, running=None):
    """z [..., c] dense NHWC rows -> (mean, biased var) each [n_groups, c].
    sync (agent-parallel training): callable that all-reduces a float64 tensor in place -- this rank's sums are reduced,
    `sync` adds the other ranks', and the statistics are normalised by `norm_rows` rows per group (the global count).
    running = (running_mean, running_var, momentum): one group, no sync -- the momentum update of the running statistics leaves
    the launch that finishes the statistics (dn_bn_train_stats_running), bit for bit bn_update_running's."""
    _need_gpu(z)
    c = z.shape[-1]
    rows = z.numel() // c
    assert rows % n_groups == 0
    mean = torch.empty((n_groups, c), dtype=torch.float32, device=z.device)
    var = torch.empty_like(mean)
    lib = _lib.load()
    sums = _ws(z.device, lib.dn_reduce_workspace_bytes(n_groups, rows // n_groups, c))
    if running is not None:
        if sync is not None or n_groups != 1:
            raise _lib.DnError("bn_stats: the fused running-statistics update takes one group and no sync")
        rm, rv, momentum = running
        _need_gpu(rm, rv)
        check(lib.dn_bn_train_stats_running(_ptr(z), rows, c, c, _ptr(sums), sums.numel(), _ptr(mean), _ptr(var), _ptr(rm), _ptr(rv),
                                            float(momentum), _stream()), "dn_bn_train_stats_running")
        return mean, var
    if sync is None:
        check(lib.dn_bn_train_stats(_ptr(z), n_groups, rows // n_groups, c, c, _ptr(sums), sums.numel(),
                                    _ptr(mean), _ptr(var), _stream()), "dn_bn_train_stats")
        return mean, var
    check(lib.dn_bn_train_stats_partial(_ptr(z), n_groups, rows // n_groups, c, c, _ptr(sums), sums.numel(), _stream()),
          "dn_bn_train_stats_partial")
    sync(_folded(sums, n_groups, c))
    check(lib.dn_bn_train_stats_finish(_ptr(sums), n_groups, int(norm_rows if norm_rows is not None else rows // n_groups), c,
                                       _ptr(mean), _ptr(var), _stream()), "dn_bn_train_stats_finish")
    return mean, var


def bn_apply_sp_supported(z, n_groups=1):
    """can bn_apply(..., sp_out=...) also write y as an SP tensor?  (one group, [n, h, w, c] with c % 16 == 0, c / 4 a power of two)"""
    c = z.shape[-1]
    return z.dim() == 4 and n_groups == 1 and c % 16 == 0 and ((c // 4) & (c // 4 - 1)) == 0 and c <= 1024


def bn_apply(z, mean, var, gamma, beta, eps, relu=True, out=None, relu_mask=None, sp_out=None):
    """relu_mask: a uint8 tensor of z.numel() / 4 bytes that receives the backward's ReLU gate, one byte per four channels
    (dn_bn_train_apply_mask; relu must be on, c % 4 == 0) -- bn_backward(relu_mask=...) reads it instead of y.
    sp_out: an ops.SpTensor of z's shape that ALSO receives y as f16 hi / lo planes (dn_bn_train_apply_mask_sp: the operand of
    the next layer's forward conv on the split-f16 engine; needs relu_mask, bn_apply_sp_supported)."""
    _need_gpu(z, mean, var, gamma, beta, relu_mask)
    c = z.shape[-1]
    n_groups = mean.shape[0]
    rows = z.numel() // c
    y = torch.empty_like(z) if out is None else out
    if sp_out is not None:
        if relu_mask is None or not relu or not bn_apply_sp_supported(z, n_groups) or tuple(sp_out.shape) != tuple(z.shape) \
                or sp_out.hi_only or sp_out.bits or relu_mask.dtype != torch.uint8 or relu_mask.numel() * 4 != z.numel():
            raise _lib.DnError("bn_apply: sp_out needs relu + relu_mask, one group, c % 16 == 0 with c / 4 a power of two, and a "
                               "full SP tensor of z's shape")
        check(_lib.load().dn_bn_train_apply_mask_sp(_ptr(z), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), float(eps), rows,
                                                    z.shape[1] * z.shape[2], c, c, _ptr(y), _ptr(relu_mask), _ptr(sp_out.data),
                                                    _stream()), "dn_bn_train_apply_mask_sp")
        return y
    if relu_mask is not None:
        if not relu or relu_mask.dtype != torch.uint8 or relu_mask.numel() * 4 != z.numel() or not relu_mask.is_contiguous():
            raise _lib.DnError("bn_apply: relu_mask must be a contiguous uint8 tensor of z.numel() / 4 bytes, with relu on")
        check(_lib.load().dn_bn_train_apply_mask(_ptr(z), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), float(eps), n_groups,
                                                 rows // n_groups, c, c, _ptr(y), _ptr(relu_mask), _stream()),
              "dn_bn_train_apply_mask")
        return y
    check(_lib.load().dn_bn_train_apply(_ptr(z), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta),
                                        float(eps), int(relu), n_groups, rows // n_groups, c, c,
                                        _ptr(y), _stream()), "dn_bn_train_apply")
    return y


def bn_update_running(mean, var, rows_per_group, running_mean, running_var, momentum=0.1, order=None):
    n_groups, c = mean.shape
    if order is not None:            # only the listed groups, in that order
        n_groups = order.numel()
    check(_lib.load().dn_bn_update_running(_ptr(mean), _ptr(var), n_groups, int(rows_per_group), c,
                                           _ptr(order), float(momentum), _ptr(running_mean),
                                           _ptr(running_var), _stream()), "dn_bn_update_running")


_BIAS_WS = {}      # the fused bias gradient's partials: never the buffer the reduction's sums live in (read by the same launch)


def bn_backward_bias_supported(z, n_groups=1):
    """can bn_backward(..., dbias=...) fuse the conv bias gradient (sum of dz per channel) into its apply launch?"""
    c = z.shape[-1]
    return (z.dim() == 4 and n_groups == 1 and c % 4 == 0 and ((c // 4) & (c // 4 - 1)) == 0 and c <= 1024
            and z.numel() // 4 < (1 << 31) and os.environ.get("DN_BN_LEGACY", "0") != "1"
            and os.environ.get("DN_BN_FUSED_BIAS", "1") != "0")


def bn_backward(dy_a, y, z, mean, var, gamma, eps, dgamma, dbeta, relu=True, dy_b=None, up_a=False,
                accumulate=False, out=None, sync=None, norm_rows=None, sp_out=None, sp_lift=1.0, relu_mask=None, dbias=None,
                folds=None, want_dz=True):
    """z, y [n, h, w, c] dense.  dy_a: [n, h, w, c'] view (or [n, 2h, 2w, c'] when up_a = 1 / True; up_a = 2: the
    space-to-depth image [n, h/2, w/2, 4c] of the gradient, train.py :: _dgrad's one-launch stride-2 data gradient), dy_b
    optional second gradient (same resolution as y).  Returns dz; fills dgamma / dbeta.
    sync / norm_rows (agent-parallel training, see bn_stats): dgamma / dbeta are then THIS rank's share (sums over its rows).
    sp_out / sp_lift: an ops.SpTensor [n, h, w, c] that also receives dz * sp_lift as f16 hi / lo planes
    (dn_bn_train_backward_finish_sp: the operand of the split-f16 data gradient; one group, c % 16 == 0).
    relu_mask: bn_apply's byte mask of (y > 0) -- read in place of y by both passes (relu = 2 of the C entry points).
    dbias [c]: also receives sum over this call's rows of dz -- the gradient of the conv bias in front of this BatchNorm --
    from the launch that writes dz (dn_bn_train_backward_finish_bias; bn_backward_bias_supported), instead of a channel_sum
    pass over dz.  folds (a DeferredFolds): dbias' fold joins it instead of being launched here -- dbias is valid after
    folds.run().  want_dz = False (with sp_out and dbias): the fp32 dz is not written, None is returned -- for a layer whose
    weight gradient reads the SP copy (conv_wgrad(..., dz_sp=))."""
    _need_gpu(dy_a, dy_b, y, z, mean, var, gamma, relu_mask, dbias)
    n, h, w, c = z.shape
    if relu_mask is not None:
        if not relu or relu_mask.dtype != torch.uint8 or relu_mask.numel() * 4 != z.numel():
            raise _lib.DnError("bn_backward: relu_mask must be bn_apply's uint8 mask of z.numel() / 4 bytes")
        y, relu = relu_mask, 2
    n_groups = mean.shape[0]
    assert n % n_groups == 0
    if not want_dz and (sp_out is None or dbias is None or out is not None):
        raise _lib.DnError("bn_backward: want_dz = False needs sp_out and dbias (the fused forms) and no out")
    dz = (torch.empty_like(z) if out is None else out) if want_dz else None
    lib = _lib.load()
    sums = _ws(z.device, lib.dn_reduce_workspace_bytes(n_groups, (n // n_groups) * h * w, c))
    if dbias is not None:
        if not bn_backward_bias_supported(z, n_groups) or dbias.numel() != c or not dbias.is_contiguous():
            raise _lib.DnError("bn_backward: dbias needs one group, c / 4 a power of two and a contiguous [c] tensor")
        if sp_out is not None and (tuple(sp_out.shape) != (n, h, w, c) or sp_out.hi_only or sp_out.bits):
            raise _lib.DnError("bn_backward: sp_out must be a full SP tensor of z's shape")
        src = (_ptr(dy_a), _ld(dy_a), int(up_a), _ptr(dy_b), _ld(dy_b) if dy_b is not None else 0, _ptr(y), _ptr(z),
               _ptr(mean), _ptr(var))
        check(lib.dn_bn_train_backward_partial(*src, float(eps), int(relu), n_groups, h, w, n, c, _ptr(sums),
                                               sums.numel(), _ptr(dgamma), _ptr(dbeta), int(bool(accumulate)), _stream()),
              "dn_bn_train_backward_partial")
        if sync is not None:
            sync(_folded(sums, n_groups, c))
        rows = int(norm_rows if norm_rows is not None else n * h * w)
        nb = int(lib.dn_bn_bias_workspace_bytes(n * h * w, c))
        if folds is not None:
            bws = folds.workspace(dbias, nb)
            blocks = ctypes.c_int(0)
            check(lib.dn_bn_train_backward_finish_bias_deferred(
                *src, _ptr(gamma), float(eps), int(relu), h, w, n, c, _ptr(sums), rows, _ptr(dz),
                _ptr(sp_out.data) if sp_out is not None else None, float(sp_lift) if sp_out is not None else 1.0,
                _ptr(bws), bws.numel(), ctypes.byref(blocks), _stream()), "dn_bn_train_backward_finish_bias_deferred")
            folds.add(bws, blocks.value, c, dbias, False)
            return dz
        bws = _ws(z.device, nb, _BIAS_WS)
        check(lib.dn_bn_train_backward_finish_bias(*src, _ptr(gamma), float(eps), int(relu), h, w, n, c, _ptr(sums), rows, _ptr(dz),
                                                   _ptr(sp_out.data) if sp_out is not None else None,
                                                   float(sp_lift) if sp_out is not None else 1.0, _ptr(dbias),
                                                   _ptr(bws), bws.numel(), _stream()), "dn_bn_train_backward_finish_bias")
        return dz
    if sp_out is not None:
        if tuple(sp_out.shape) != (n, h, w, c) or sp_out.hi_only or sp_out.bits:
            raise _lib.DnError("bn_backward: sp_out must be a full SP tensor of z's shape")
        src = (_ptr(dy_a), _ld(dy_a), int(up_a), _ptr(dy_b), _ld(dy_b) if dy_b is not None else 0, _ptr(y), _ptr(z),
               _ptr(mean), _ptr(var))
        check(lib.dn_bn_train_backward_partial(*src, float(eps), int(relu), n_groups, h, w, n // n_groups, c, _ptr(sums),
                                               sums.numel(), _ptr(dgamma), _ptr(dbeta), int(bool(accumulate)), _stream()),
              "dn_bn_train_backward_partial")
        if sync is not None:
            sync(_folded(sums, n_groups, c))
        rows = int(norm_rows if norm_rows is not None else (n // n_groups) * h * w)
        check(lib.dn_bn_train_backward_finish_sp(*src, _ptr(gamma), float(eps), int(relu), n_groups, h, w, n // n_groups, c,
                                                 _ptr(sums), rows, _ptr(dz), _ptr(sp_out.data), float(sp_lift), _stream()),
              "dn_bn_train_backward_finish_sp")
        return dz
    if sync is None:
        check(lib.dn_bn_train_backward(
            _ptr(dy_a), _ld(dy_a), int(up_a), _ptr(dy_b), _ld(dy_b) if dy_b is not None else 0,
            _ptr(y), _ptr(z), _ptr(mean), _ptr(var), _ptr(gamma), float(eps), int(relu), n_groups, h, w,
            n // n_groups, c, _ptr(sums), sums.numel(), _ptr(dz), _ptr(dgamma), _ptr(dbeta), int(bool(accumulate)),
            _stream()), "dn_bn_train_backward")
        return dz
    src = (_ptr(dy_a), _ld(dy_a), int(up_a), _ptr(dy_b), _ld(dy_b) if dy_b is not None else 0, _ptr(y), _ptr(z),
           _ptr(mean), _ptr(var))
    check(lib.dn_bn_train_backward_partial(*src, float(eps), int(relu), n_groups, h, w, n // n_groups, c, _ptr(sums),
                                           sums.numel(), _ptr(dgamma), _ptr(dbeta), int(bool(accumulate)), _stream()),
          "dn_bn_train_backward_partial")
    sync(_folded(sums, n_groups, c))
    rows = int(norm_rows if norm_rows is not None else (n // n_groups) * h * w)
    check(lib.dn_bn_train_backward_finish(*src, _ptr(gamma), float(eps), int(relu), n_groups, h, w, n // n_groups, c,
                                          _ptr(sums), rows, _ptr(dz), _stream()), "dn_bn_train_backward_finish")
    return dz


class DeferredFolds:
    """The channel-sum folds of one backward pass, launched together (dn_channel_sum_fold_multi).  The sums are leaves of the
    backward (bias gradients): bn_backward(..., dbias=, folds=) and channel_sum(..., folds=) leave their per-workgroup partials
    in a workspace of their own -- kept per output tensor across steps -- and run() folds them all in one launch; the outputs
    are valid after it."""

    def __init__(self):
        self._ws, self._jobs, self._keep = {}, [], []

    def workspace(self, out, nbytes):
        key = (out.data_ptr(), out.numel(), sum(1 for _, o in self._keep if o.data_ptr() == out.data_ptr()))   # (a second sum into one tensor: its own partials)
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self._ws[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=out.device)
        return buf

    def add(self, ws, n_blocks, c, out, accumulate):
        self._jobs.append((ws.data_ptr() + 8 * c, ws.data_ptr(), out.data_ptr(), int(n_blocks), int(c), int(bool(accumulate))))
        self._keep.append((ws, out))

    def run(self):
        if not self._jobs:
            return
        jobs = (_lib.FoldJob * len(self._jobs))()
        for q, (part, sums, out, nb, c, acc) in zip(jobs, self._jobs):
            q.partials, q.sums, q.out, q.n_blocks, q.c, q.accumulate = part, sums, out, nb, c, acc
        check(_lib.load().dn_channel_sum_fold_multi(jobs, len(self._jobs), _stream()), "dn_channel_sum_fold_multi")
        self._jobs, self._keep = [], []


def channel_sum(x, out, accumulate=False, folds=None):
    """out[c] (+)= sum over all rows of x [..., c] (x may be a channel slice).  folds (a DeferredFolds): the fold joins it
    instead of being launched here -- out is valid after folds.run()."""
    _need_gpu(x, out)
    c = x.shape[-1]
    rows = x.numel() // c
    if folds is not None:
        lib = _lib.load()
        ws = folds.workspace(out, lib.dn_reduce_workspace_bytes(1, rows, c))
        blocks = ctypes.c_int(0)
        check(lib.dn_channel_sum_partial(_ptr(x), rows, c, _ld(x), _ptr(ws), ws.numel(), ctypes.byref(blocks), _stream()),
              "dn_channel_sum_partial")
        folds.add(ws, blocks.value, c, out, accumulate)
        return out
    sums = _ws(x.device, _lib.load().dn_reduce_workspace_bytes(1, rows, c))
    check(_lib.load().dn_channel_sum(_ptr(x), rows, c, _ld(x), _ptr(sums), sums.numel(), _ptr(out),
                                     int(bool(accumulate)), _stream()), "dn_channel_sum")
    return out


def upsample2_sum(g):
    """g [n, 2h, 2w, c] (may be a channel slice) -> dense [n, h, w, c] of 2 x 2 block sums"""
    _need_gpu(g)
    n, h2, w2, c = g.shape
    out = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.float32, device=g.device)
    check(_lib.load().dn_upsample2_sum(_ptr(g), _ld(g), n, h2 // 2, w2 // 2, c, _ptr(out), _stream()),
          "dn_upsample2_sum")
    return out


def add_rows(a, b):
    """a += b for NHWC maps / channel slices of equal shape"""
    _need_gpu(a, b)
    c = a.shape[-1]
    rows = a.numel() // c
    check(_lib.load().dn_add_rows(_ptr(a), _ld(a), _ptr(b), _ld(b), rows, c, _stream()), "dn_add_rows")
    return a


# ---- UNet pieces of the segmentation variant (include/disconet_seg.h) -----------------------
def maxpool2(x):
    """nn.MaxPool2d(2) on a dense NHWC map"""
    _need_gpu(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    check(_lib.load().dn_maxpool2_nhwc(_ptr(x), n, h, w, c, _ptr(y), _stream()), "dn_maxpool2_nhwc")
    return y


def maxpool2_backward(x, dy):
    """x: the pooled map's INPUT [n, h, w, c] dense; dy [n, h/2, w/2, c] (may be a channel slice) -> dx dense"""
    _need_gpu(x, dy)
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().dn_maxpool2_nhwc_backward(_ptr(x), _ptr(dy), _ld(dy), n, h, w, c, _ptr(dx), _stream()),
          "dn_maxpool2_nhwc_backward")
    return dx


def upsample2_bilinear(x):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) on a dense NHWC map"""
    _need_gpu(x)
    n, h, w, c = x.shape
    y = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
    check(_lib.load().dn_upsample2_bilinear_nhwc(_ptr(x), n, h, w, c, _ptr(y), _stream()), "dn_upsample2_bilinear_nhwc")
    return y


def upsample2_bilinear_backward(dy):
    """dy [n, 2h, 2w, c] (may be a channel slice) -> dx [n, h, w, c] dense"""
    _need_gpu(dy)
    n, h2, w2, c = dy.shape
    dx = torch.empty((n, h2 // 2, w2 // 2, c), dtype=torch.float32, device=dy.device)
    check(_lib.load().dn_upsample2_bilinear_nhwc_backward(_ptr(dy), _ld(dy), n, h2 // 2, w2 // 2, c, _ptr(dx),
                                                          _stream()), "dn_upsample2_bilinear_nhwc_backward")
    return dx


# ---- fusion, training form ----------------------------------------------------------------
def pair_add_ego(z1, e, ego_image):
    n_pairs, rows, c = z1.shape[0], z1[0].numel() // z1.shape[-1], z1.shape[-1]
    check(_lib.load().dn_pair_add_ego(_ptr(z1), _ptr(e), _ptr(ego_image), n_pairs, rows, c, _stream()),
          "dn_pair_add_ego")
    return z1


def pair_sum_ego(dz1, first, pairs, n_images):
    rows, c = dz1[0].numel() // dz1.shape[-1], dz1.shape[-1]
    de = torch.empty((n_images,) + tuple(dz1.shape[1:]), dtype=torch.float32, device=dz1.device)
    check(_lib.load().dn_pair_sum_ego(_ptr(dz1), _ptr(first), _ptr(pairs), n_images, rows, c, _ptr(de),
                                      _stream()), "dn_pair_sum_ego")
    return de


def fuse_combine(z4, maps, first, pair_index, map_image, ego_out, fused):
    """z4 [P, h, w, 1]; maps [M, h, w, c]; lists per ego; writes fused[ego_out[e]]; returns weights"""
    _need_gpu(z4, maps, fused)
    hw, c = maps.shape[1] * maps.shape[2], maps.shape[3]
    weights = torch.empty_like(z4)
    check(_lib.load().dn_fuse_combine(_ptr(z4), _ptr(maps), _ptr(first), _ptr(pair_index),
                                      _ptr(map_image), _ptr(ego_out), ego_out.numel(), hw, c,
                                      _ptr(weights), _ptr(fused), _stream()), "dn_fuse_combine")
    return weights


def fuse_combine_backward(dfused, z4, weights, maps, first, pair_index, map_image, ego_out, dmaps):
    hw, c = maps.shape[1] * maps.shape[2], maps.shape[3]
    dz4 = torch.zeros_like(z4)
    check(_lib.load().dn_fuse_combine_backward(
        _ptr(dfused), _ld(dfused), _ptr(z4), _ptr(weights), _ptr(maps), _ptr(first), _ptr(pair_index),
        _ptr(map_image), _ptr(ego_out), ego_out.numel(), hw, c, _ptr(dmaps), _ptr(dz4), _stream()),
        "dn_fuse_combine_backward")
    return dz4


def warp_list(src, poses, src_image, out=None):
    """src [M, h, w, c]; poses [n, 4, 4]; src_image [n] int32 -> warped [n, h, w, c]"""
    _need_gpu(src, poses, src_image)
    n = src_image.numel()
    _, h, w, c = src.shape
    warped = torch.empty((n, h, w, c), dtype=torch.float32, device=src.device) if out is None else out
    check(_lib.load().dn_warp_list(_ptr(src), _ptr(poses), _ptr(src_image), n, h, w, c, _ptr(warped),
                                   _stream()), "dn_warp_list")
    return warped


def warp_backward(d_warped, poses, src_image, d_src, rigid=False):
    """adds into d_src [M, h, w, c]; rigid=True (all poses rotation + translation): gather form"""
    n, h, w, c = d_warped.shape
    scratch = _ws(d_warped.device, 4 * d_warped.numel())
    check(_lib.load().dn_warp_backward(_ptr(d_warped), _ptr(poses), _ptr(src_image), n, d_src.shape[0], h,
                                       w, c, int(bool(rigid)), _ptr(scratch), _ptr(d_src), _stream()),
          "dn_warp_backward")
    return d_src


# ---- loss / optimiser ---------------------------------------------------------------------
def det_loss(cls, labels, loc, targets, mask, norm, alpha=0.25, gamma=2.0, sigma=3.0):
    """-> (losses [2] float64 = (cls, loc), dcls, dloc)"""
    _need_gpu(cls, labels, loc, targets, mask)
    code = loc.shape[-1]
    n = loc.numel() // code
    assert cls.numel() == 2 * n and mask.numel() == n
    losses = torch.empty(2, dtype=torch.float64, device=cls.device)
    dcls, dloc = torch.empty_like(cls), torch.empty_like(loc)
    check(_lib.load().dn_det_loss(_ptr(cls), _ptr(labels), _ptr(loc), _ptr(targets), _ptr(mask), n,
                                  code, float(alpha), float(gamma), float(sigma), float(norm),
                                  _ptr(losses), _ptr(dcls), _ptr(dloc), _stream()), "dn_det_loss")
    return losses, dcls, dloc


def kd_kl_loss(student, teacher, kd_weight, loss, zero=False, norm_rows=None):
    """student, teacher [..., c] dense NHWC maps of equal shape; adds kd_weight * KLDiv (mean over
    all elements) to loss[0] (float64, zeroed first if zero) and returns d(term)/d(student).
    norm_rows (agent-parallel training): the rows the mean runs over when this call sees only THIS rank's share of them
    (the reference's KLDivLoss averages over all A * B images' pixels); the ranks' terms then sum to the un-sharded one."""
    _need_gpu(student, teacher, loss)
    assert student.shape == teacher.shape and student.is_contiguous() and teacher.is_contiguous()
    c = student.shape[-1]
    rows = student.numel() // c
    d = torch.empty_like(student)
    check(_lib.load().dn_kd_kl_loss(_ptr(student), _ptr(teacher), rows, c,
                                    float(kd_weight) / float((rows if norm_rows is None else int(norm_rows)) * c), _ptr(loss), _ptr(d), int(zero),
                                    _stream()), "dn_kd_kl_loss")
    return d


def adam_step(p, g, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    _need_gpu(p, g, m, v)
    check(_lib.load().dn_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr),
                                   float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                   int(step), _stream()), "dn_adam_step")


__all__ = [n for n in dir() if not n.startswith("_")] + ["conv_desc"]
