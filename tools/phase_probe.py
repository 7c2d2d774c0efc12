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


run("conv_pre_2", 20, 256, 32, 32)
run("conv_pre_1", 20, 256, 13, 32, bits=True)
run("pre_2 @4img", 4, 256, 32, 32)
