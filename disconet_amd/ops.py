"""Tensor-level wrappers over the C ABI (include/disconet_hip.h).

torch is plumbing here: device memory, the current HIP stream, nothing else.
Every function requires float32 CUDA(HIP) tensors and launches on torch's
current stream.  No CPU fallback: a CPU tensor raises.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvDesc, FuseMlpParams, MlpTailParams, Post1x1Desc, check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream_handle():
    """the current device's current stream as an integer handle -- torch's raw getter: torch.cuda.current_stream() builds a Stream
    object behind three device-index lookups (one of them an environment read), ~3 us of the ~15 us a launch costs on the host,
    and the training step is launch-bound on the host through its first hundred launches (profiles/r06_train_gap_sites.txt)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return ctypes.c_void_p(_stream_handle())


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.DnError("disconet_amd ops need tensors on the MI355X (got a %s tensor); "
                               "there is no CPU path" % t.device)


def unsafe_overlap_allowed():
    """Running two of this library's kernels CONCURRENTLY (a second HIP stream, two graphs in flight)
    has produced silently wrong values on MI355X / ROCm 7.2: lanes 48..63 of a VGPR of one kernel are
    replaced while a split-f16 conv kernel shares its SIMD (DESIGN.md 3.6 (B),
    profiles/r02_hazard_repro.txt; not root-caused).  The library's contract is ONE stream per GPU;
    the overlap toggles exist for the hazard study only and need DISCONET_UNSAFE_OVERLAP=1."""
    import os
    return os.environ.get("DISCONET_UNSAFE_OVERLAP", "0") == "1"


def check_overlap_request(value, what):
    if value and not unsafe_overlap_allowed():
        raise _lib.DnError(
            "%s = True would co-schedule kernels on two HIP streams; that has produced silently wrong "
            "results beside the split-f16 conv kernels (DESIGN.md 3.6 (B)).  The library runs in stream "
            "order; set DISCONET_UNSAFE_OVERLAP=1 to override for measurements that checksum every "
            "result (bench.py --in-flight, tools/hazard/)." % what)
    return bool(value)


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.DnError("%s must be contiguous float32 (got %s, contiguous=%s)"
                           % (name, t.dtype, t.is_contiguous()))


# ---------------------------------------------------------------------------
# K1
# ---------------------------------------------------------------------------
def _geom(voxel_size, extents):
    vs = (ctypes.c_double * 3)(*[float(v) for v in voxel_size])
    ext = (ctypes.c_double * 6)(*[float(e) for row in extents for e in row])
    return vs, ext


def voxelize_occupy(pts, voxel_size, extents, dims):
    """pts [N, >=3] float32 cuda -> dense [X, Y, Z] float32 occupancy."""
    _need_gpu(pts)
    _f32c(pts, "pts")
    vs, ext = _geom(voxel_size, extents)
    d = (ctypes.c_int * 3)(*[int(v) for v in dims])
    dense = torch.empty(tuple(int(v) for v in dims), dtype=torch.float32, device=pts.device)
    check(_lib.load().dn_voxelize_occupy(_ptr(pts), pts.shape[0], pts.shape[1], vs, ext, d,
                                         _ptr(dense), _stream()), "dn_voxelize_occupy")
    return dense


def voxel_compact(dense, capacity=None):
    """dense [X, Y, Z] -> (indices [M, 3] int32 in lexsort(x, y, z) order)."""
    _need_gpu(dense)
    _f32c(dense, "dense")
    lib = _lib.load()
    d = (ctypes.c_int * 3)(*dense.shape)
    capacity = int(dense.numel() if capacity is None else capacity)
    ws = torch.empty(max(1, lib.dn_voxel_compact_workspace(d)), dtype=torch.uint8, device=dense.device)
    idx = torch.empty((capacity, 3), dtype=torch.int32, device=dense.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=dense.device)
    check(lib.dn_voxel_compact(_ptr(dense), d, _ptr(idx), capacity, _ptr(cnt), _ptr(ws), _stream()),
          "dn_voxel_compact")
    m = int(cnt.item())
    return idx[:min(m, capacity)], m


def scatter_dense(indices, offsets, n_images, dims):
    """Batched dense rebuild: indices [Mtot, 3] int32, offsets [n_images + 1] int32
    -> dense [n_images, 1, X, Y, Z] float32 (the bevs layout of the reference)."""
    _need_gpu(indices, offsets)
    if indices.dtype != torch.int32 or offsets.dtype != torch.int32:
        raise _lib.DnError("scatter_dense needs int32 indices/offsets")
    d = (ctypes.c_int * 3)(*[int(v) for v in dims])
    dense = torch.empty((n_images, 1) + tuple(int(v) for v in dims), dtype=torch.float32,
                        device=indices.device)
    check(_lib.load().dn_scatter_dense(_ptr(indices), _ptr(offsets), n_images, indices.shape[0], d,
                                       _ptr(dense), _stream()), "dn_scatter_dense")
    return dense


# ---------------------------------------------------------------------------
# K2/K3/K7
# ---------------------------------------------------------------------------
# "f32": exact-fp32 MFMA; "f16x3": split-f16 x3 MFMA on fp32 NHWC activations (conv_mfma.hip);
# "sp": the same split-f16 x3 arithmetic on activations kept pre-split in HBM (split-planar,
# conv_sp.hip) -- the inference default.  Training kernels take NHWC: they read "sp" as "f16x3".
MATH_MODES = {"f32": 0, "f16x3": 1, "sp": 2}


def nhwc_math(mode):
    """math value for the NHWC engine (dn_conv2d): the SP engine's arithmetic is mode 1"""
    m = MATH_MODES.get(mode, mode)
    return 1 if m == 2 else m


def conv_desc(n_images, h_in, w_in, c0, c_out, ksize, stride=1, relu=True, c1=0, up0=False,
              ld0=None, ld1=None, ldo=None, math=0):
    d = ConvDesc()
    d.n_images, d.h_in, d.w_in = n_images, h_in, w_in
    d.c0, d.c1, d.up0 = c0, c1, int(up0)   # 1: nearest x2 upsample, 2: zero-stuffed (dgrad)
    d.c_out, d.ksize, d.stride, d.relu = c_out, ksize, stride, int(bool(relu))
    d.ld0 = c0 if ld0 is None else ld0
    d.ld1 = c1 if ld1 is None else ld1
    d.ldo = c_out if ldo is None else ldo
    d.math = MATH_MODES.get(math, math)
    return d


def conv_out_hw(d):
    pad = d.ksize // 2
    return ((d.h_in + 2 * pad - d.ksize) // d.stride + 1,
            (d.w_in + 2 * pad - d.ksize) // d.stride + 1)


def pack_conv_weights(d, weight):
    """weight [c_out, c_in, k, k] (or a Conv3d (1,1,1) weight) -> packed float32 buffer."""
    _need_gpu(weight)
    w = weight.detach().reshape(d.c_out, d.c0 + d.c1, d.ksize, d.ksize).contiguous().float()
    lib = _lib.load()
    n = lib.dn_conv_packed_weight_floats(ctypes.byref(d))
    if n == 0:
        check(-1, "dn_conv_packed_weight_floats")
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    check(lib.dn_conv_pack_weights(ctypes.byref(d), _ptr(w), _ptr(packed), _stream()),
          "dn_conv_pack_weights")
    return packed


def fold_bn(bias, bn=None, channels=None):
    """-> (scale, shift) float32 device vectors for y = acc * scale + shift."""
    ref = bias if bias is not None else bn.weight
    _need_gpu(ref)
    channels = channels or ref.numel()
    scale = torch.empty(channels, dtype=torch.float32, device=ref.device)
    shift = torch.empty(channels, dtype=torch.float32, device=ref.device)
    b = bias.detach().float().contiguous() if bias is not None else None
    if bn is not None:
        args = [bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous(),
                bn.running_mean.float().contiguous(), bn.running_var.float().contiguous()]
        eps = float(bn.eps)
    else:
        args, eps = [None] * 4, 0.0
    check(_lib.load().dn_fold_bn(_ptr(b), *[_ptr(t) for t in args], eps, channels, _ptr(scale),
                                 _ptr(shift), _stream()), "dn_fold_bn")
    return scale, shift


def conv2d(d, src0, packed, scale, shift, src1=None, out=None):
    """NHWC float32 conv; returns [n_images, h_out, w_out, c_out] (ldo == c_out)."""
    _need_gpu(src0, packed, scale, shift, src1)
    ho, wo = conv_out_hw(d)
    if out is None:
        out = torch.empty((d.n_images, ho, wo, d.ldo), dtype=torch.float32, device=src0.device)
    check(_lib.load().dn_conv2d(ctypes.byref(d), _ptr(src0), _ptr(src1), _ptr(packed), _ptr(scale),
                                _ptr(shift), _ptr(out), _stream()), "dn_conv2d")
    return out


def conv2d_taps(d, src0, packed, scale, shift, out_view, tap_mask):
    """3x3 NHWC conv restricted to the taps of `tap_mask`, written into `out_view` -- a strided [n, h_out, w_out, c_out]
    view (channels contiguous), e.g. dx[:, py::2, px::2, :] -- through dn_conv2d_taps."""
    _need_gpu(src0, packed, scale, shift, out_view)
    ho, wo = conv_out_hw(d)
    if tuple(out_view.shape) != (d.n_images, ho, wo, d.c_out) or out_view.stride(3) != 1:
        raise _lib.DnError("conv2d_taps: output view %s / strides %s does not match [%d, %d, %d, %d] with contiguous channels"
                           % (tuple(out_view.shape), out_view.stride(), d.n_images, ho, wo, d.c_out))
    check(_lib.load().dn_conv2d_taps(ctypes.byref(d), _ptr(src0), None, _ptr(packed), _ptr(scale), _ptr(shift),
                                     _ptr(out_view), int(tap_mask), out_view.stride(0), out_view.stride(1),
                                     out_view.stride(2), _stream()), "dn_conv2d_taps")
    return out_view


# ---------------------------------------------------------------------------
# split-planar (SP) activations and the SP conv engine
# ---------------------------------------------------------------------------
class SpTensor:
    """An activation map in the SP layout of include/disconet_hip.h:
    data [n, ceil(c/16), 4, h, w, 8] float16 (quarter = 2*part + oct; part 0 = half(x),
    part 1 = half(x - half(x))).  `shape` is the logical NHWC shape."""

    __slots__ = ("data", "n", "h", "w", "c", "hi_only", "bits")

    def __init__(self, n, h, w, c, device=None, data=None, hi_only=False, bits=False):
        """hi_only: the HI-ONLY form [n, ceil(c/16), 2, h, w, 8] of values that are exact in binary16 (the 0/1
        occupancy grid of scatter_dense_sp): accepted as source 0 of a 3x3 stride-1 SP conv.
        bits: an occupancy BIT grid, data int32 [n, h, w] with bit k = channel k (c <= 32; scatter_dense_bits):
        accepted as source 0 of a 3x3 stride-1 SP conv of <= 32 output channels."""
        self.n, self.h, self.w, self.c = int(n), int(h), int(w), int(c)
        self.hi_only, self.bits = bool(hi_only), bool(bits)
        if self.bits and (self.hi_only or self.c > 32):
            raise _lib.DnError("SpTensor: a bit grid holds <= 32 channels and has no hi-only form")
        if data is None:
            if self.bits:
                data = torch.empty((self.n, self.h, self.w), dtype=torch.int32, device=device)
            else:
                data = torch.empty((self.n, (self.c + 15) // 16, 2 if hi_only else 4, self.h, self.w, 8),
                                   dtype=torch.float16, device=device)
        self.data = data

    shape = property(lambda self: (self.n, self.h, self.w, self.c))
    device = property(lambda self: self.data.device)
    is_cuda = property(lambda self: self.data.is_cuda)

    def numel(self):
        return self.n * self.h * self.w * self.c

    def data_ptr(self):
        return self.data.data_ptr()

    def record_stream(self, stream):
        self.data.record_stream(stream)

    def nhwc(self, out=None):
        """-> float32 [n, h, w, c] (x = hi + lo)"""
        if self.bits:         # not a hot path either
            x = ((self.data.unsqueeze(-1) >> torch.arange(self.c, device=self.data.device, dtype=torch.int32)) & 1).float()
            if out is None:
                return x.contiguous()
            out.copy_(x)
            return out
        if self.hi_only:      # not a hot path (training entry, tests): torch reshapes the two octet planes
            x = self.data.permute(0, 3, 4, 1, 2, 5).reshape(self.n, self.h, self.w, -1)[..., :self.c].float()
            if out is None:
                return x.contiguous()
            out.copy_(x)
            return out
        if out is None:
            out = torch.empty((self.n, self.h, self.w, self.c), dtype=torch.float32, device=self.data.device)
        check(_lib.load().dn_sp_to_nhwc(_ptr(self.data), self.n, self.h, self.w, self.c, out.stride(2),
                                        _ptr(out), _stream()), "dn_sp_to_nhwc")
        return out

    @staticmethod
    def from_nhwc(x):
        """float32 [n, h, w, c] (pixel stride x.stride(2)) -> SpTensor"""
        _need_gpu(x)
        if x.dtype != torch.float32 or x.stride(3) != 1:
            raise _lib.DnError("SpTensor.from_nhwc needs float32 channels-last data")
        n, h, w, c = x.shape
        if x.stride(1) != w * x.stride(2) or x.stride(0) != h * x.stride(1):
            x = x.contiguous()
        t = SpTensor(n, h, w, c, device=x.device)
        check(_lib.load().dn_sp_from_nhwc(_ptr(x), n, h, w, c, x.stride(2), _ptr(t.data), _stream()),
              "dn_sp_from_nhwc")
        return t


def sp_range_flags(reset=True):
    """Sticky range flags of the split-f16 engines on the current device (include/disconet_hip.h ::
    dn_sp_range_flags): bit 1 = a value above 2^14 was stored as an f16 hi/lo pair, bit 0 = a value was clamped
    to +-65504 (results no longer follow the fp32 reference), bit 2 = a NaN reached an epilogue.  Blocking:
    validation time, not per step (check_sp_range is the asynchronous guard the model uses)."""
    if reset:
        _range_guard.pending.pop(torch.cuda.current_device(), None)
    flags = int(_lib.load().dn_sp_range_flags(1 if reset else 0)) & 0xffffffff
    if flags & 0x80000000:       # the read itself failed (device sync / allocation / copy): never "no clamp, no NaN"
        msg = _lib.load().dn_last_error()
        raise _lib.DnError("dn_sp_range_flags: the flags could not be read (%s)" % (msg.decode() if msg else "HIP error"))
    return flags


def sp_range_flags_into(word, zero_first=False, reset=False):
    """Stream-ordered collect of the sticky range flags into the int32 device tensor `word` (dn_sp_range_flags_async): kernel
    launches only, legal inside a stream capture.  zero_first: the word is zeroed by a kernel of the same call (a captured step
    cannot rely on a memset node); reset: the sticky flags are cleared once collected."""
    _need_gpu(word)
    check(_lib.load().dn_sp_range_flags_async(_ptr(word), (1 if reset else 0) | (2 if zero_first else 0), _stream()),
          "dn_sp_range_flags_async")
    return word


def _raise_on_range_flags(flags, what):
    if flags & 0x80000000:
        raise _lib.DnError("%s: the split-f16 range flags could not be read" % what)
    if flags & 1:
        raise _lib.DnError("%s: a value was clamped to +-65504 by the split-f16 (hi + lo binary16) activation format; "
                           "the outputs do not follow the fp32 reference%s.  Rescale the layer (fold a power of two into "
                           "its BatchNorm) or run conv_math = 'f32'."
                           % (what, " (and a NaN reached a later epilogue)" if flags & 4 else ""))
    if flags & 4:
        raise _lib.DnError("%s: a NaN reached a conv / fusion epilogue of the split-f16 engines (ReLU and the clamp of the "
                           "split turn it into a finite number, so the outputs hold plausible garbage).  Check the inputs "
                           "and the weights." % what)
    if flags & 2:
        import warnings
        warnings.warn("%s: activations above 2^14 were stored as f16 hi/lo pairs (limit 65504)" % what)


class _RangeGuard:
    """The range guard without a synchronisation: behind a forward the sticky device flags are collected into a
    device word on the forward's stream (dn_sp_range_flags_async) and copied to pinned host memory; the NEXT
    call looks at the copy if its event has completed and raises.  A clamp is therefore reported one call late (or by
    drain(), which waits) -- but it is reported by default, and the cost is four tiny launches per polled call."""

    def __init__(self):
        self.pending = {}       # device index -> (event, pinned host word, device word, what)

    def poll(self, what):
        dev = torch.cuda.current_device()
        self._look(dev, wait=False)
        if dev in self.pending or torch.cuda.is_current_stream_capturing():
            return
        word = torch.zeros(1, dtype=torch.int32, device="cuda")
        check(_lib.load().dn_sp_range_flags_async(_ptr(word), 0, _stream()), "dn_sp_range_flags_async")
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending[dev] = (ev, host, word, what)

    def _look(self, dev, wait):
        ent = self.pending.get(dev)
        if ent is None:
            return
        ev, host, _, what = ent
        if wait:
            ev.synchronize()
        elif not ev.query():
            return
        del self.pending[dev]
        flags = int(host[0]) & 0xffffffff
        if flags & 7:
            _lib.load().dn_sp_range_flags(1)       # the flags are sticky: cleared here, where they are reported
        _raise_on_range_flags(flags, what)

    def drain(self):
        """wait for the outstanding read of the current device and raise if it carried a flag"""
        self._look(torch.cuda.current_device(), wait=True)


_range_guard = _RangeGuard()


def check_sp_range(what="forward"):
    """Range guard of the split-f16 engines, called behind a forward.  Default: asynchronous (see _RangeGuard) --
    a clamped activation / a NaN raises at the next call, with no synchronisation.  DN_SP_CHECK=1: blocking read,
    this very call raises; DN_SP_CHECK=0: off.  Never inside a stream capture."""
    import os
    mode = os.environ.get("DN_SP_CHECK", "")
    if mode == "0" or torch.cuda.is_current_stream_capturing():
        return
    if mode == "1":
        _range_guard.pending.pop(torch.cuda.current_device(), None)
        _raise_on_range_flags(sp_range_flags(reset=True), what)
        return
    _range_guard.poll(what)


def drain_sp_range():
    """block until the outstanding asynchronous range-guard read (if any) has landed; raises like check_sp_range"""
    _range_guard.drain()


def sp_upmode():
    """The process's up-conv form as the library reads it (csrc/conv_sp.hip :: sp_upmode): DN_SP_UPMERGE, default 2 = the
    row- and column-merged kernel (4 of 9 taps on the upsampled source), 1 = row-merged (6 of 9), 0 = plain taps.
    (dn_spconv_set_upmode() overrides are a tools / test matter and not mirrored here: this only feeds the executed-work
    figure of the profiling regions.)"""
    import os
    try:
        return int(os.environ.get("DN_SP_UPMERGE", "2"))
    except ValueError:
        return 2


def as_sp(x):
    return x if isinstance(x, SpTensor) else SpTensor.from_nhwc(x)


def as_nhwc(x):
    return x.nhwc() if isinstance(x, SpTensor) else x


def _pow2_lift(weight):
    """power of two that lifts max|w| to [2^12, 2^13): the lo halves of the split then sit in
    the f16 normal range; 1/wmul is folded into the layer's scale (exact in fp32)"""
    m = float(weight.detach().abs().max())
    if not (m > 0.0) or m != m or m == float("inf"):
        return 1.0
    import math
    return float(2.0 ** max(-20, min(30, 12 - math.floor(math.log2(m)))))


def sp_pack_conv_weights(d, weight, wmul=None):
    """-> (packed uint8 buffer, wmul).  wmul: the power-of-two lift, when the caller already knows it (the training engine
    caches it per parameter: _pow2_lift reads max |w| back from the device)"""
    _need_gpu(weight)
    w = weight.detach().reshape(d.c_out, d.c0 + d.c1, d.ksize, d.ksize).contiguous().float()
    lib = _lib.load()
    nbytes = lib.dn_spconv_packed_weight_bytes(ctypes.byref(d))
    if nbytes == 0:
        check(-1, "dn_spconv_packed_weight_bytes")
    if wmul is None:
        wmul = _pow2_lift(w)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    check(lib.dn_spconv_pack_weights(ctypes.byref(d), _ptr(w), wmul, _ptr(packed), _stream()),
          "dn_spconv_pack_weights")
    return packed, wmul


class PackSet:
    """Many weight packs of one conv engine as ONE launch (dn_spconv_pack_weights_multi: engine "sp", plain layout only;
    dn_conv_pack_weights_multi: engine "nhwc").  A job = (conv descriptor, forward weight tensor [c_out, cin_total, taps], mode,
    cin_total, ci_first, n_in) -- include/disconet_hip.h :: dn_pack_job; its packed image lives in a buffer this object owns.
    run(wmuls) packs every job from the weights as they are now."""
    _FN = {"sp": ("dn_spconv_pack_multi_table_bytes", "dn_spconv_pack_multi_prepare", "dn_spconv_pack_weights_multi"),
           "nhwc": ("dn_conv_pack_multi_table_bytes", "dn_conv_pack_multi_prepare", "dn_conv_pack_weights_multi")}

    def __init__(self, jobs, device, engine="sp"):
        lib = _lib.load()
        self.engine, self.n = engine, len(jobs)
        self._bytes, self._prepare, self._launch = (getattr(lib, f) for f in self._FN[engine])
        self._jobs = (_lib.PackJob * self.n)()
        self.buffers = []
        self._keep = []
        for q, (d, weight, mode, cin_total, ci_first, n_in) in zip(self._jobs, jobs):
            _need_gpu(weight)
            if weight.dtype != torch.float32 or not weight.is_contiguous():
                raise _lib.DnError("PackSet: weights must be contiguous float32 tensors (the views are read in place)")
            if engine == "sp":
                nbytes = lib.dn_spconv_packed_weight_bytes(ctypes.byref(d))
                buf = torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None
            else:
                nfloats = lib.dn_conv_packed_weight_floats(ctypes.byref(d))
                buf = torch.empty(nfloats, dtype=torch.float32, device=device) if nfloats else None
            if buf is None:
                check(-1, "packed weight size")
            self.buffers.append(buf)
            self._keep.append(weight)
            q.desc, q.weight, q.packed = d, weight.data_ptr(), buf.data_ptr()
            q.mode, q.cin_total, q.ci_first, q.n_in, q.wmul = int(mode), int(cin_total), int(ci_first), int(n_in), 1.0
        self._host = torch.empty(self._bytes(self.n), dtype=torch.uint8).pin_memory()
        self._table = torch.empty(self._host.numel(), dtype=torch.uint8, device=device)
        self._wmuls, self._blocks = None, 0

    @classmethod
    def supported(cls, d, engine="sp"):
        """can the layer `d`'s pack be a job?  (the split-planar engine: not when it is packed tap-merged)"""
        lib = _lib.load()
        probe = (_lib.PackJob * 1)()
        probe[0].desc, probe[0].weight, probe[0].packed, probe[0].wmul = d, 16, 16, 1.0
        probe[0].cin_total = d.c0 + d.c1
        host = (ctypes.c_ubyte * int(getattr(lib, cls._FN[engine][0])(1)))()
        blocks = ctypes.c_int(0)
        return getattr(lib, cls._FN[engine][1])(probe, 1, host, ctypes.byref(blocks)) == 0

    def run(self, wmuls):
        wmuls = [float(v) for v in wmuls]
        if wmuls != self._wmuls:
            if self._wmuls is not None:
                torch.cuda.current_stream().synchronize()      # the copy of the previous image out of the pinned buffer
            for q, v in zip(self._jobs, wmuls):
                q.wmul = v
            blocks = ctypes.c_int(0)
            check(self._prepare(self._jobs, self.n, ctypes.c_void_p(self._host.data_ptr()), ctypes.byref(blocks)),
                  self._FN[self.engine][1])
            self._table.copy_(self._host, non_blocking=True)
            self._wmuls, self._blocks = wmuls, blocks.value
        check(self._launch(_ptr(self._table), self.n, self._blocks, _stream()), self._FN[self.engine][2])


SpPackSet = PackSet


_KS_WORKSPACE_CAP = 32 << 20      # bytes of partial sums per K-sliced launch (the launcher splits fewer tiles beyond)


def sp_conv2d(d, src0, packed, scale, shift, src1=None, out=None, nhwc_copy=False, kslices=1):
    """SP conv: src0 / src1 SpTensors -> SpTensor [n_images, h_out, w_out, c_out].
    nhwc_copy: also return the output as a float32 NHWC tensor written by the same launch (dn_spconv2d_dual)
    -> (SpTensor, tensor).
    kslices > 1: the layer's K-sliced form (dn_spconv2d_ks): results independent of the batch, the launch's last
    round / small launches handed out slice by slice through a scratch buffer."""
    _need_gpu(src0, packed, scale, shift, src1)
    ho, wo = conv_out_hw(d)
    if out is None:
        out = SpTensor(d.n_images, ho, wo, d.c_out, device=src0.device)
    if src1 is not None and (src1.hi_only or src1.bits):
        raise _lib.DnError("sp_conv2d: only source 0 may be a hi-only SP tensor or a bit grid")
    if src0.hi_only:
        d.math = 3            # include/disconet_hip.h: source 0 is a hi-only SP tensor
    elif src0.bits:
        d.math = 4            # ... an occupancy bit grid
    lib = _lib.load()
    flat = torch.empty((d.n_images, ho, wo, d.c_out), dtype=torch.float32, device=src0.device) if nhwc_copy else None
    p1 = _ptr(src1.data) if src1 is not None else None
    if kslices > 1 and not lib.dn_spconv_ks_supported(ctypes.byref(d), kslices):
        kslices = 1      # the process's up-conv form (DN_SP_UPMERGE=1) or a short layer has no K-sliced form: run whole
    if kslices > 1:
        nbytes = min(int(lib.dn_spconv_workspace_bytes(ctypes.byref(d), kslices)), _KS_WORKSPACE_CAP)
        if os.environ.get("DN_SP_KS_NOSPLIT", "0") == "1":      # A/B runs: every tile whole (the slices folded in registers)
            nbytes = 0
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=src0.device)
        check(lib.dn_spconv2d_ks(ctypes.byref(d), kslices, _ptr(src0.data), p1, _ptr(packed), _ptr(scale), _ptr(shift),
                                 _ptr(out.data), _ptr(flat), d.c_out if nhwc_copy else 0, _ptr(ws) if nbytes else None, nbytes,
                                 _stream()),
              "dn_spconv2d_ks")
        return (out, flat) if nhwc_copy else out
    if nhwc_copy:
        check(lib.dn_spconv2d_dual(ctypes.byref(d), _ptr(src0.data), p1, _ptr(packed), _ptr(scale), _ptr(shift),
                                   _ptr(out.data), _ptr(flat), d.c_out, _stream()), "dn_spconv2d_dual")
        return out, flat
    check(_lib.load().dn_spconv2d(ctypes.byref(d), _ptr(src0.data), _ptr(src1.data) if src1 is not None else None,
                                  _ptr(packed), _ptr(scale), _ptr(shift), _ptr(out.data), _stream()),
          "dn_spconv2d")
    return out


def sp_conv2d_nhwc(d, src0, packed, scale, shift, out, src1=None):
    """SP conv whose only output is float32 NHWC rows (dn_spconv2d_nhwc): `out` [n_images, h_out, w_out, c_out], possibly a
    channel slice of a wider tensor (pixel stride out.stride(2)).  The training step's split-f16 data gradient."""
    _need_gpu(src0, packed, scale, shift, src1, out)
    ho, wo = conv_out_hw(d)
    if (tuple(out.shape) != (d.n_images, ho, wo, d.c_out) or out.dtype != torch.float32 or out.stride(3) != 1
            or out.stride(1) != wo * out.stride(2) or out.stride(0) != ho * out.stride(1)):
        raise _lib.DnError("sp_conv2d_nhwc: out must be float32 [n, h_out, w_out, c_out] rows (a channel slice is fine)")
    if src0.hi_only or src0.bits or (src1 is not None and (src1.hi_only or src1.bits)):
        raise _lib.DnError("sp_conv2d_nhwc: full SP sources only")
    check(_lib.load().dn_spconv2d_nhwc(ctypes.byref(d), _ptr(src0.data), _ptr(src1.data) if src1 is not None else None,
                                       _ptr(packed), _ptr(scale), _ptr(shift), _ptr(out), out.stride(2), _stream()),
          "dn_spconv2d_nhwc")
    return out


def sp_conv2d_pre_pair_supported(d1, d2):
    return bool(_lib.load().dn_spconv2d_pre_pair_supported(ctypes.byref(d1), ctypes.byref(d2)))


def sp_conv2d_pre_pair(d1, d2, bits, packed1, scale1, shift1, packed2, scale2, shift2, out=None):
    """The encoder stem's first two 3x3 layers in one launch (dn_spconv2d_pre_pair): `bits` an SpTensor(bits=True), the
    layers' packed weights and folded affines -> SpTensor [n, h, w, d2.c_out]; bit-identical to the two launches."""
    _need_gpu(bits, packed1, packed2, scale1, shift1, scale2, shift2)
    if not getattr(bits, "bits", False):
        raise _lib.DnError("sp_conv2d_pre_pair: the source must be an occupancy bit grid (ops.scatter_dense_bits)")
    if out is None:
        out = SpTensor(d2.n_images, d2.h_in, d2.w_in, d2.c_out, device=bits.device)
    check(_lib.load().dn_spconv2d_pre_pair(ctypes.byref(d1), ctypes.byref(d2), _ptr(bits.data), _ptr(packed1), _ptr(scale1),
                                           _ptr(shift1), _ptr(packed2), _ptr(scale2), _ptr(shift2), _ptr(out.data), _stream()),
          "dn_spconv2d_pre_pair")
    return out


def sp_pack_post1x1_weights(weight):
    """weight [c_out2, c_in2(, 1, 1)] -> (packed A-operand fragments, wmul)"""
    _need_gpu(weight)
    w = weight.detach().reshape(weight.shape[0], -1).contiguous().float()
    lib = _lib.load()
    wmul = _pow2_lift(w)
    packed = torch.empty(lib.dn_sp_post1x1_packed_bytes(), dtype=torch.uint8, device=weight.device)
    check(lib.dn_sp_post1x1_pack_weights(_ptr(w), w.shape[0], w.shape[1], wmul, _ptr(packed), _stream()),
          "dn_sp_post1x1_pack_weights")
    return packed, wmul


def sp_pack_heads_weights(weight, split):
    """block-diagonal 1x1 stage [c_out2, 64]: rows < split read stage-1 channels 0..31, the rest
    channels 32..63 -> (packed fragments for sp_conv2d_post1x1(block_diag=True), wmul)"""
    _need_gpu(weight)
    w = weight.detach().reshape(weight.shape[0], 64).contiguous().float()
    lib = _lib.load()
    wmul = _pow2_lift(w)
    packed = torch.empty(lib.dn_sp_post1x1_packed_bytes(), dtype=torch.uint8, device=weight.device)
    check(lib.dn_sp_post1x1_pack_heads(_ptr(w), w.shape[0], split, wmul, _ptr(packed), _stream()),
          "dn_sp_post1x1_pack_heads")
    return packed, wmul


def sp_conv2d_post1x1(d, src0, packed, scale, shift, packed2, scale2, shift2, c_out2, split, relu2,
                      out_a, out_b=None, block_diag=False):
    """SP 3x3 conv (64 ch) + affine + ReLU fused with a 1x1 stage.  out_a an SpTensor (one SP
    output of c_out2 channels) or a float32 NHWC tensor (+ out_b: two-headed fp32 output)."""
    _need_gpu(src0, packed, packed2)
    p = Post1x1Desc()
    p.c_out2, p.relu2, p.split = c_out2, int(bool(relu2)), split
    f32 = not isinstance(out_a, SpTensor)
    p.block_diag = int(bool(block_diag))
    p.ldo_a = out_a.shape[-1] if f32 else 0
    p.ldo_b = out_b.shape[-1] if out_b is not None else 0
    check(_lib.load().dn_spconv2d_post1x1(ctypes.byref(d), ctypes.byref(p), _ptr(src0.data), None,
                                          _ptr(packed), _ptr(scale), _ptr(shift), _ptr(packed2),
                                          _ptr(scale2), _ptr(shift2), int(f32),
                                          _ptr(out_a if f32 else out_a.data), _ptr(out_b), _stream()),
          "dn_spconv2d_post1x1")
    return out_a, out_b


def scatter_dense_sp(indices, offsets, n_images, dims, hi_only=False):
    """Batched dense rebuild straight into the conv engine's layout: -> SpTensor [n_images, X, Y, Z]
    (what DiscoNet.forward accepts in place of the float32 bevs tensor).  hi_only: the half-size hi-only
    form (0/1 is exact in binary16; conv_pre_1 then reads half the bytes and runs 2 MFMAs per product)."""
    _need_gpu(indices, offsets)
    if indices.dtype != torch.int32 or offsets.dtype != torch.int32:
        raise _lib.DnError("scatter_dense_sp needs int32 indices/offsets")
    d = (ctypes.c_int * 3)(*[int(v) for v in dims])
    out = SpTensor(n_images, int(dims[0]), int(dims[1]), int(dims[2]), device=indices.device, hi_only=hi_only)
    fn = _lib.load().dn_scatter_dense_sp_hi if hi_only else _lib.load().dn_scatter_dense_sp
    check(fn(_ptr(indices), _ptr(offsets), n_images, indices.shape[0], d, _ptr(out.data), _stream()),
          "dn_scatter_dense_sp")
    return out


def scatter_dense_bits(indices, offsets, n_images, dims):
    """Batched dense rebuild as one occupancy word per pixel: -> SpTensor(bits=True) [n_images, X, Y, Z <= 32] --
    1/32 of the float32 bevs tensor, 1/8 of the hi-only SP form.  DiscoNet.forward accepts it in place of bevs
    (conv_pre_1 expands the words on their way into LDS; results bit-identical to the dense input's)."""
    _need_gpu(indices, offsets)
    if indices.dtype != torch.int32 or offsets.dtype != torch.int32:
        raise _lib.DnError("scatter_dense_bits needs int32 indices/offsets")
    d = (ctypes.c_int * 3)(*[int(v) for v in dims])
    out = SpTensor(n_images, int(dims[0]), int(dims[1]), int(dims[2]), device=indices.device, bits=True)
    check(_lib.load().dn_scatter_dense_bits(_ptr(indices), _ptr(offsets), n_images, indices.shape[0], d, _ptr(out.data),
                                            _stream()), "dn_scatter_dense_bits")
    return out


def sp_maxpool2(x):
    """nn.MaxPool2d(2) on an SpTensor"""
    out = SpTensor(x.n, x.h // 2, x.w // 2, x.c, device=x.device)
    check(_lib.load().dn_sp_maxpool2(_ptr(x.data), x.n, x.h, x.w, x.c, _ptr(out.data), _stream()),
          "dn_sp_maxpool2")
    return out


def sp_upsample2_bilinear(x):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) on an SpTensor"""
    out = SpTensor(x.n, 2 * x.h, 2 * x.w, x.c, device=x.device)
    check(_lib.load().dn_sp_upsample2_bilinear(_ptr(x.data), x.n, x.h, x.w, x.c, _ptr(out.data),
                                               _stream()), "dn_sp_upsample2_bilinear")
    return out


def seg_ce_loss(logits_nhwc, labels, want_grad=True, check_labels=True):
    """nn.CrossEntropyLoss (mean over the non-ignored pixels, ignore_index = -100) of float32 NHWC
    logits [n, h, w, classes] vs int labels [n, h, w] -> (loss double scalar tensor, dlogits or None).
    check_labels: refuse labels that are neither in [0, classes) nor -100, as torch does (one host
    sync); the training step passes False and stays asynchronous."""
    _need_gpu(logits_nhwc, labels)
    _f32c(logits_nhwc, "logits")
    classes = logits_nhwc.shape[-1]
    pixels = logits_nhwc.numel() // classes
    lab = labels.to(device=logits_nhwc.device, dtype=torch.int32).contiguous()
    if lab.numel() != pixels:
        raise _lib.DnError("seg_ce_loss: %d labels for %d pixels" % (lab.numel(), pixels))
    lib = _lib.load()
    counts = torch.empty(2, dtype=torch.int32, device=logits_nhwc.device)
    check(lib.dn_seg_label_count(_ptr(lab), pixels, classes, _ptr(counts), _stream()), "dn_seg_label_count")
    loss = torch.empty(1, dtype=torch.float64, device=logits_nhwc.device)
    grad = torch.empty_like(logits_nhwc) if want_grad else None
    check(lib.dn_seg_ce_loss(_ptr(logits_nhwc), _ptr(lab), pixels, classes, classes, 1.0, _ptr(counts),
                             _ptr(loss), _ptr(grad), _stream()), "dn_seg_ce_loss")
    if check_labels:
        live, bad = counts.tolist()
        if bad:
            raise _lib.DnError("seg_ce_loss: %d labels outside [0, %d) (and not the ignore index -100)"
                               % (bad, classes))
    return loss / counts[0].clamp(min=1).to(torch.float64), grad


def pack_post1x1_weights(weight):
    """weight [c_out2, c_in2(, 1, 1)] -> packed split-f16 rows for dn_conv2d_post1x1."""
    _need_gpu(weight)
    w = weight.detach().reshape(weight.shape[0], -1).contiguous().float()
    lib = _lib.load()
    packed = torch.empty(lib.dn_post1x1_packed_floats(), dtype=torch.float32, device=weight.device)
    check(lib.dn_post1x1_pack_weights(_ptr(w), w.shape[0], w.shape[1], _ptr(packed), _stream()),
          "dn_post1x1_pack_weights")
    return packed


def conv2d_post1x1(d, src0, packed, scale, shift, packed2, scale2, shift2, c_out2, split, relu2,
                   out_a, out_b=None, src1=None):
    """3x3 conv (64 ch) + affine + ReLU fused with a 1x1 stage; writes out_a (and out_b)."""
    _need_gpu(src0, packed, packed2, out_a)
    p = Post1x1Desc()
    p.c_out2, p.relu2, p.split = c_out2, int(bool(relu2)), split
    p.ldo_a = out_a.shape[-1]
    p.ldo_b = out_b.shape[-1] if out_b is not None else 0
    check(_lib.load().dn_conv2d_post1x1(ctypes.byref(d), ctypes.byref(p), _ptr(src0), _ptr(src1),
                                        _ptr(packed), _ptr(scale), _ptr(shift), _ptr(packed2),
                                        _ptr(scale2), _ptr(shift2), _ptr(out_a), _ptr(out_b),
                                        _stream()), "dn_conv2d_post1x1")
    return out_a, out_b


# ---------------------------------------------------------------------------
# K4-K6
# ---------------------------------------------------------------------------
def warp_fm_supported(h, w, c):
    """can the warped maps take the fragment-major form (dn_warp_neighbors_fm / dn_disco_fuse_mlp_fm)?"""
    return bool(_lib.load().dn_warp_fm_supported(h, w, c))


def warp_neighbors(feat, trans, num_agent, batch, agents, only_v2i=False, ego_first=0,
                   ego_count=None, out=None, fm=False):
    """feat [A*B, h, w, C] agent-major NHWC (all agents) -> warped
    [B, ego_count, A-1, h, w, C] for the egos [ego_first, ego_first + ego_count).
    fm: every [h, w, C] block in the fragment-major order dn_disco_fuse_mlp_fm reads (an opaque intermediate)."""
    _need_gpu(feat, trans, num_agent)
    _f32c(feat, "feat")
    _f32c(trans, "trans_matrices")
    ego_count = agents if ego_count is None else ego_count
    n, h, w, c = feat.shape
    warped = out if out is not None else torch.empty(
        (batch, ego_count, max(agents - 1, 0), h, w, c), dtype=torch.float32, device=feat.device)
    lib = _lib.load()
    fn = lib.dn_warp_neighbors_fm if fm else lib.dn_warp_neighbors
    check(fn(_ptr(feat), _ptr(trans), _ptr(num_agent), batch, agents, h, w, c, int(only_v2i), ego_first, ego_count,
             _ptr(warped), _stream()), "dn_warp_neighbors_fm" if fm else "dn_warp_neighbors")
    return warped


def disco_fuse_tail(feat, warped, g, fw, num_agent, tail_params, batch, agents, only_v2i=False,
                    want_weights=False, ego_first=0, ego_count=None, out=None):
    """feat: maps of ALL agents; g / warped / fw and the result: the served egos only."""
    _need_gpu(feat, g, num_agent)
    ego_count = agents if ego_count is None else ego_count
    n, h, w, c = feat.shape
    fused = out if out is not None else torch.empty((ego_count * batch, h, w, c),
                                                     dtype=torch.float32, device=feat.device)
    weights = (torch.zeros((batch, ego_count, agents, h * w), dtype=torch.float32,
                           device=feat.device) if want_weights else None)
    check(_lib.load().dn_disco_fuse_tail(_ptr(feat), _ptr(warped), _ptr(g), _ptr(fw),
                                         _ptr(num_agent), ctypes.byref(tail_params), batch, agents,
                                         h * w, c, int(only_v2i), ego_first, ego_count, _ptr(fused),
                                         _ptr(weights), _stream()), "dn_disco_fuse_tail")
    return (fused, weights) if want_weights else fused


def fuse_mlp_supported(c):
    return bool(_lib.load().dn_fuse_mlp_supported(int(c)))


def set_fuse_mlp_waves(waves):
    """tools / tests: the launch form of dn_disco_fuse_mlp -- 4 or 1 wave(s) per 32-pixel tile, 2 = one wave per tile with
    the layer-1 weights staged in LDS per workgroup of tiles; 0 = chosen per launch.  Bit-identical results."""
    check(_lib.load().dn_fuse_mlp_set_waves(int(waves)), "dn_fuse_mlp_set_waves")


def make_fuse_mlp_params(w1, b1, bn1, w2, b2, bn2, w3, b3, bn3, w4, b4, c):
    """Packs the attention MLP for dn_disco_fuse_mlp.  w1 [128, 2c], w2 [32, 128], w3 [8, 32],
    w4 [8]; bn* = (scale, shift) of the eval BatchNorm after each of the first three layers.
    Returns (FuseMlpParams, tensors-to-keep-alive)."""
    lib = _lib.load()
    dev = w1.device
    w1, w2, w3 = (t.detach().float().contiguous() for t in (w1, w2, w3))
    m1, m2, m3 = _pow2_lift(w1), _pow2_lift(w2), _pow2_lift(w3)
    packed = torch.empty(lib.dn_fuse_mlp_packed_bytes(c), dtype=torch.uint8, device=dev)
    check(lib.dn_fuse_mlp_pack(_ptr(w1), _ptr(w2), _ptr(w3), c, m1, m2, m3, _ptr(packed), _stream()),
          "dn_fuse_mlp_pack")
    keep = {"packed": packed}
    for name, (scale, shift), bias, m in (("1", bn1, b1, m1), ("2", bn2, b2, m2), ("3", bn3, b3, m3)):
        # relu(bn(acc / m + bias)) = relu(acc * (scale / m) + (scale * bias + shift))
        keep["s" + name] = (scale.float() / m).contiguous()
        keep["t" + name] = (scale.float() * bias.detach().float() + shift.float()).contiguous()
    keep["w4"] = w4.detach().float().reshape(8).contiguous()
    keep["b4"] = b4.detach().float().reshape(1).contiguous()
    p = FuseMlpParams()
    for name, _ in FuseMlpParams._fields_:
        setattr(p, name, keep[name].data_ptr())
    return p, keep


def disco_fuse_mlp(feat, warped, num_agent, params, batch, agents, only_v2i=False, want_weights=False,
                   ego_first=0, ego_count=None, sp_out=False, fm=False):
    """Attention MLP + agent softmax + weighted sum in one launch.  feat [A*B, h, w, C] NHWC (all
    agents), warped [B, E, A-1, h, w, C] (fm: in the fragment-major form of warp_neighbors(fm=True));
    -> SpTensor (sp_out) or float32 NHWC [E*B, h, w, C] (+ weights [B, E, A, h*w] when want_weights)."""
    _need_gpu(feat, num_agent, warped)
    _f32c(feat, "feat")
    ego_count = agents if ego_count is None else ego_count
    n, h, w, c = feat.shape
    out_sp = SpTensor(ego_count * batch, h, w, c, device=feat.device) if sp_out else None
    out = None if sp_out else torch.empty((ego_count * batch, h, w, c), dtype=torch.float32, device=feat.device)
    weights = (torch.zeros((batch, ego_count, agents, h * w), dtype=torch.float32, device=feat.device)
               if want_weights else None)
    lib = _lib.load()
    fn = lib.dn_disco_fuse_mlp_fm if fm else lib.dn_disco_fuse_mlp
    check(fn(_ptr(feat), _ptr(warped), _ptr(num_agent), ctypes.byref(params), batch, agents, h * w, c, int(only_v2i),
             ego_first, ego_count, _ptr(out_sp.data) if sp_out else None, _ptr(out), _ptr(weights), _stream()),
          "dn_disco_fuse_mlp_fm" if fm else "dn_disco_fuse_mlp")
    res = out_sp if sp_out else out
    return (res, weights) if want_weights else res


def live_agent_counts(num_agent_tensor, device):
    """The kernels' [B] int32 live-agent counts from the reference's num_agent_tensor [B, A] (column 0 is the
    count).  A 1-D int32 tensor already on the device is taken as is -- no cast / gather kernel inside the step
    (pass `num_agent_tensor[:, 0].int().contiguous()` once, outside a captured step)."""
    t = num_agent_tensor
    if t.dim() == 1 and t.dtype == torch.int32 and t.device == torch.device(device) and t.is_contiguous():
        return t
    if t.dim() == 1:
        return t.to(device=device, dtype=torch.int32).contiguous()
    return t[:, 0].to(device=device, dtype=torch.int32).contiguous()


def make_tail_params(tensors):
    """tensors: dict name -> float32 device tensor for every dn_mlp_tail_params field."""
    p = MlpTailParams()
    for name, _ in MlpTailParams._fields_:
        setattr(p, name, tensors[name].data_ptr())
    return p
