"""Is the conv engine power-limited?  The same launch on random operands and on ALL-ZERO operands (no toggling in the multipliers:
MI355X_MICROARCH.md, DVFS give-back: zero-filled inputs ran +19 % on a GEMM at equal wave cycles).  If a layer is held by the power
budget its zero-data run is faster by about that; if it is held by latencies / issue it does not care what the bits are.
    python tools/power_probe.py            (on the GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disconet_amd import ops

LAYERS = [  # name, n, h, w, c0, c1, up0, c_out, ksize, stride
    ("conv2_2 128->128 @64^2", 20, 64, 64, 128, 0, 0, 128, 3, 1),
    ("conv5_2 256->256 @32^2", 20, 32, 32, 256, 0, 0, 256, 3, 1),
    ("conv7_2 64->64 @128^2", 20, 128, 128, 64, 0, 0, 64, 3, 1),
    ("conv4_2 512->512 @16^2", 20, 16, 16, 512, 0, 0, 512, 3, 1),
    ("conv8_2 32->32 @256^2", 20, 256, 256, 32, 0, 0, 32, 3, 1),
    ("conv3_1 128->256 s2", 20, 64, 64, 128, 0, 0, 256, 3, 2),
    ("conv6_1 384->128 @64^2 up+cat", 20, 64, 64, 256, 128, 1, 128, 3, 1),
    ("conv8_1 96->32 @256^2 up+cat", 20, 256, 256, 64, 32, 1, 32, 3, 1),
]


def run(name, n, h, w, c0, c1, up0, c_out, k, stride, zero):
    g = torch.Generator().manual_seed(1)
    d = ops.conv_desc(n, h, w, c0, c_out, k, stride, True, c1=c1, up0=bool(up0), math="sp")
    wt = torch.randn(c_out, c0 + c1, k, k, generator=g) * (2.0 / ((c0 + c1) * k * k)) ** 0.5
    hs, ws = (h // 2, w // 2) if up0 else (h, w)
    x0 = torch.randn(n, hs, ws, c0, generator=g).clamp_(min=0)
    x1 = torch.randn(n, h, w, c1, generator=g).clamp_(min=0) if c1 else None
    if zero:
        wt.zero_(); x0.zero_()
        if x1 is not None:
            x1.zero_()
    packed, wmul = ops.sp_pack_conv_weights(d, wt.cuda())
    sc, sh = (torch.ones(c_out) / wmul).cuda(), torch.zeros(c_out).cuda()
    s0 = ops.SpTensor.from_nhwc(x0.cuda())
    s1 = ops.SpTensor.from_nhwc(x1.cuda()) if x1 is not None else None
    ho, wo = ops.conv_out_hw(d)
    out = ops.SpTensor(n, ho, wo, c_out, device="cuda")
    for _ in range(10):
        ops.sp_conv2d(d, s0, packed, sc, sh, src1=s1, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.sp_conv2d(d, s0, packed, sc, sh, src1=s1, out=out)
    e1.record()
    torch.cuda.synchronize()
    return 20.0 * e0.elapsed_time(e1)      # us per launch


for L in LAYERS:
    r1, z1, r2, z2 = run(*L, zero=False), run(*L, zero=True), run(*L, zero=False), run(*L, zero=True)
    print("%-34s random %6.1f / %6.1f us   all-zero %6.1f / %6.1f us   zero / random %.2f" % (L[0], r1, r2, z1, z2, (z1 + z2) / (r1 + r2)))
