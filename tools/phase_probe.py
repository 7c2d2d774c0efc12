"""Phase timing of the weight-stationary conv kernel (a -DDN_PHASE_TIMING=1 build of libdisconet_hip.so: tools/ab/build.sh
DN_PHASE_TIMING 1): share of wave 0's cycles in: wait for the patch + barrier / issue of the next patch / MFMA loop / epilogue / decode."""
import ctypes
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from disconet_amd import ops, _lib

lib = _lib.load()
lib.dn_sp_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.dn_sp_phase_cycles.restype = ctypes.c_int


def run(name, n, hw, c0, c_out, bits=False):
    g = torch.Generator().manual_seed(1)
    d = ops.conv_desc(n, hw, hw, c0, c_out, 3, 1, True, math="sp")
    w = (torch.randn(c_out, c0, 3, 3, generator=g) * 0.1).cuda()
    packed, wmul = ops.sp_pack_conv_weights(d, w)
    sc, sh = (torch.ones(c_out) / wmul).cuda(), torch.zeros(c_out).cuda()
    if bits:
        x = ops.SpTensor(n, hw, hw, c0, device="cuda", bits=True,
                         data=(torch.rand(n, hw, hw, generator=g) < 0.2).to(torch.int32).cuda() * 5)
    else:
        x = ops.SpTensor.from_nhwc(torch.randn(n, hw, hw, c0, generator=g).clamp_(min=0).cuda())
    out = ops.SpTensor(n, hw, hw, c_out, device="cuda")
    for _ in range(5):
        ops.sp_conv2d(d, x, packed, sc, sh, out=out)
    buf = (ctypes.c_ulonglong * 8)()
    lib.dn_sp_phase_cycles(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.sp_conv2d(d, x, packed, sc, sh, out=out)
    e1.record()
    lib.dn_sp_phase_cycles(buf, 1)
    v = list(buf)
    tot = v[6] or 1
    print("%-12s %7.1f us | workgroups' wave 0: tiles %d, cycles/tile %.0f | wait+barrier %.1f %%  issue next patch %.1f %%  MFMA loop %.1f %%  "
          "epilogue %.1f %%  decode/zero %.1f %%  (unaccounted %.1f %%)" % (
              name, e0.elapsed_time(e1) * 1e3, v[5], tot / max(v[5], 1), 100 * v[0] / tot, 100 * v[1] / tot, 100 * v[2] / tot, 100 * v[3] / tot,
              100 * v[4] / tot, 100 * (tot - sum(v[:5])) / tot))


def run_pair(n, hw):
    g = torch.Generator().manual_seed(2)
    d1 = ops.conv_desc(n, hw, hw, 13, 32, 3, 1, True, math="sp")
    d2 = ops.conv_desc(n, hw, hw, 32, 32, 3, 1, True, math="sp")
    p1, m1 = ops.sp_pack_conv_weights(d1, (torch.randn(32, 13, 3, 3, generator=g) * 0.1).cuda())
    p2, m2 = ops.sp_pack_conv_weights(d2, (torch.randn(32, 32, 3, 3, generator=g) * 0.1).cuda())
    one, zero = torch.ones(32).cuda(), torch.zeros(32).cuda()
    bits = ops.SpTensor(n, hw, hw, 13, device="cuda", bits=True,
                        data=((torch.rand(n, hw, hw, 13, generator=g) < 0.02).to(torch.int64) << torch.arange(13)).sum(-1).to(torch.int32).cuda())
    out = ops.SpTensor(n, hw, hw, 32, device="cuda")
    for _ in range(5):
        ops.sp_conv2d_pre_pair(d1, d2, bits, p1, one / m1, zero, p2, one / m2, zero, out=out)
    buf = (ctypes.c_ulonglong * 8)()
    lib.dn_sp_phase_cycles(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.sp_conv2d_pre_pair(d1, d2, bits, p1, one / m1, zero, p2, one / m2, zero, out=out)
    e1.record()
    lib.dn_sp_phase_cycles(buf, 1)
    v = list(buf)
    tot = v[6] or 1
    print("stem pair    %7.1f us | waves 0 and 7: tiles %d, cycles/tile %.0f | stage 1 %.1f %%  barrier after it %.1f %%  stage-2 MFMA loop %.1f %%  "
          "epilogue %.1f %%  words + barrier %.1f %%" % (e0.elapsed_time(e1) * 1e3, v[5], tot / max(v[5], 1), 100 * v[0] / tot,
                                                        100 * v[1] / tot, 100 * v[2] / tot, 100 * v[3] / tot, 100 * v[4] / tot))


run_pair(20, 256)
run("conv_pre_2", 20, 256, 32, 32)
run("conv_pre_1", 20, 256, 13, 32, bits=True)
run("pre_2 @4img", 4, 256, 32, 32)
