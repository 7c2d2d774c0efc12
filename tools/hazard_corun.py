"""DESIGN.md 3.6 reproducer, full-size: the ROUND-1 pose-warp kernel (tools/hazard_warp_r1.hip, the
kernel as it was before its channel loop was made wave-uniform) beside the conv engine's split-f16
launches on a second HIP stream.  For every variant of the loop, counts launches whose output differs
from that variant's own serial result.

    for v in 0 1 2 3 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DHZ_VARIANT=$v \
        tools/hazard_warp_r1.hip -o tools/hazard_warp_v$v.so; done
    python tools/hazard_corun.py
"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from disconet_amd import ops  # noqa: E402
from disconet_amd.synthetic import make_trans_matrices  # noqa: E402

VARIANTS = {0: "round-1 loop: for (c4 = lane; c4 < c4n; c4 += 64)", 1: "wave-uniform trip count (shipped fix)",
            2: "round-1 loop + 2 x s_nop 15 at the loop top", 3: "round-1 loop + s_waitcnt vmcnt(0) lgkmcnt(0) at the bottom",
            4: "round-1 loop + counter of iterations with c4 >= c4n",
            5: "round-1 loop + probes (EXEC full? lanes agree on uniform values?)",
            6: "round-1 loop, divisions -> reciprocal multiplies (no v_div_scale/v_div_fmas)",
            7: "shipped loop form + division-free coordinates",
            8: "round-1 loop, IEEE division written out with builtins (div_scale x2, rcp, fma x5, div_fmas, div_fixup)",
            9: "as 8, v_div_fmas -> plain v_fma",
            10: "as 8, v_div_fmas with a constant-false flag (VCC = 0 by s_mov)",
            11: "as 8, + 2 x s_nop 15 before v_div_fmas",
            12: "as 8, without v_div_fixup",
            13: "as 8, numerator-side v_div_scale removed (no VALU write of VCC)",
            14: "round-1 loop; every lane then stores its copy of the wave-uniform tap masks / indices / weights"}
if len(sys.argv) > 1:
    VARIANTS = {int(v): VARIANTS[int(v)] for v in sys.argv[1].split(",")}
torch.manual_seed(0)
B, A, h, w, c = 4, 5, 32, 32, 256
feat = torch.randn(A * B, h, w, c, device="cuda")
trans = make_trans_matrices(B, A, jitter_seed=0).cuda()
na = torch.full((B,), A, dtype=torch.int32).cuda()
trace = torch.zeros(8, dtype=torch.int32, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())


def conv_runner(n, hh, ww, cin, cout, math):
    x = torch.randn(n, hh, ww, cin, device="cuda")
    d = ops.conv_desc(n, hh, ww, cin, cout, 3, math=math)
    pk = ops.pack_conv_weights(d, torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)
    one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    y = torch.empty(n, hh, ww, cout, device="cuda")
    return lambda: ops.conv2d(d, x, pk, one, zero, out=y)


def sp_conv_runner(n, hh, ww, cin, cout):
    x = ops.SpTensor.from_nhwc(torch.randn(n, hh, ww, cin, device="cuda"))
    d = ops.conv_desc(n, hh, ww, cin, cout, 3, math="sp")
    pk, m = ops.sp_pack_conv_weights(d, torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)
    one, zero = torch.ones(cout, device="cuda") / m, torch.zeros(cout, device="cuda")
    y = ops.SpTensor(n, hh, ww, cout, device="cuda")
    return lambda: ops.sp_conv2d(d, x, pk, one, zero, out=y)


runners = {"nothing": lambda: None,
           "conv_mfma f16x3 256ch@32 (256x32 tile: round 1's trigger)": conv_runner(20, 32, 32, 256, 256, 1),
           "conv_mfma f32   256ch@32": conv_runner(20, 32, 32, 256, 256, 0),
           "conv_sp (LDS-DMA engine) 256ch@32": sp_conv_runner(20, 32, 32, 256, 256)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for v, what in VARIANTS.items():
    lib = ctypes.CDLL(os.path.join(HERE, "hazard_warp_v%d.so" % v))

    def warp(out):
        rc = lib.hz_warp_neighbors(P(feat), P(trans), P(na), B, A, h, w, c, P(out), P(trace),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                   P(dbg[dbg_slot[0]]) if v == 14 else None)
        assert rc == 0
        return out

    dbg_slot = [0]
    dbg = torch.zeros((9, B * A * (A - 1) * h * w, 64, 4), dtype=torch.int32, device="cuda") if v == 14 else None
    ref = warp(torch.empty((B, A, A - 1, h, w, c), device="cuda")).clone()
    torch.cuda.synchronize()
    outs = [torch.empty_like(ref) for _ in range(8)]
    print("variant %d: %s" % (v, what))
    for name, co in runners.items():
        co()
        torch.cuda.synchronize()
        trace.zero_()
        bad, where = 0, None
        for trial in range(2):
            for i in range(8):
                with torch.cuda.stream(s2):
                    co()
                    co()
                with torch.cuda.stream(s1):
                    dbg_slot[0] = i + 1
                    warp(outs[i])
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    if where is None:
                        idx = (o != ref).nonzero()
                        ch = idx[:, -1]
                        where = "%d values, channels %d..%d, first at %s" % (len(idx), int(ch.min()), int(ch.max()),
                                                                              idx[0].tolist())
                        # whose values are they?  look the wrong 64-channel run up among all pixels of the pair
                        bb, ii, jj, yy, xx, _ = idx[0].tolist()
                        run = o[bb, ii, jj, yy, xx, 192:256]
                        same = (ref[bb, ii, jj, :, :, 192:256] == run).all(-1).nonzero()
                        where += "; equals the serial values of pixel(s) %s of the same map" % same.tolist()[:3]
        if v == 14:
            names = ["q-pixel in-frame masks (SGPR lane masks)", "tap indices", "pass-2 weights", "pass-1 weights"]
            for i in range(8):
                d = (dbg[i + 1] != dbg[0])
                if d.any():
                    idx = d.nonzero()
                    lanes = idx[:, 1]
                    print("      replay %d: %d lane records differ from the serial run's; lanes %d..%d; fields: %s" % (
                        i, int(d.any(-1).sum()), int(lanes.min()), int(lanes.max()),
                        ", ".join("%s x%d" % (names[k], int(d[..., k].sum())) for k in range(4) if d[..., k].any())))
                    pix, ln = idx[0, 0].item(), idx[0, 1].item()
                    print("         e.g. pixel record %d lane %d: serial %s, beside %s; lane 0 beside: %s" % (
                        pix, ln, [hex(x & 0xffffffff) for x in dbg[0, pix, ln].tolist()],
                        [hex(x & 0xffffffff) for x in dbg[i + 1, pix, ln].tolist()],
                        [hex(x & 0xffffffff) for x in dbg[i + 1, pix, 0].tolist()]))
        extra = "  iterations with c4 >= c4n: %d" % int(trace[1]) if v == 4 else ""
        if v == 5:
            extra = "  probes: EXEC-not-full %d, pixel-index mismatch %d, tap mismatch %d, weight mismatch %d" % tuple(
                int(t) for t in trace[1:5])
        print("   beside %-58s differing from the serial result: %2d of 16%s%s" % (
            name, bad, extra, ("   [" + where + "]") if where else ""))
