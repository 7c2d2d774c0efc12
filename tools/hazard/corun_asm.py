"""Runs the assembled variants of the round-1 warp kernel (asm_edit.py: e*.hsaco) beside the conv
engine's launches on a second stream and counts replays whose output differs from that variant's own
serial result.      python tools/hazard/asm_edit.py && python tools/hazard/corun_asm.py [e0,e1,...]
"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from disconet_amd import ops  # noqa: E402
from disconet_amd.synthetic import make_trans_matrices  # noqa: E402
from asm_edit import EDITS  # noqa: E402

KERNEL = b"_ZN12_GLOBAL__N_121warp_neighbors_kernelEPKfS1_PKiiiiiiiiiPfPjS5_"
host = ctypes.CDLL(os.path.join(HERE, "hsaco_host.so"))
host.hz_load.restype = ctypes.c_void_p
host.hz_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
host.hz_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3

torch.manual_seed(0)
B, A, h, w, c = 4, 5, 32, 32, 256
feat = torch.randn(A * B, h, w, c, device="cuda")
trans = make_trans_matrices(B, A, jitter_seed=0).cuda()
na = torch.full((B,), A, dtype=torch.int32).cuda()
trace = torch.zeros(2048, dtype=torch.int32, device="cuda")


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


_ga, _gb = torch.randn(4096, 4096, device="cuda", dtype=torch.half), torch.randn(4096, 4096, device="cuda", dtype=torch.half)
_fa, _fb = _ga.float(), _gb.float()
_bf_a, _bf_b = _ga.bfloat16(), _gb.bfloat16()
if os.environ.get("HZ_GEMM", "0") == "2":
    agg = ctypes.CDLL(os.path.join(HERE, "aggressors.so"))
    agg.hz_aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    KINDS = ["v_cvt_pk_f16_f32", "v_cvt_f32_f16_sdwa (WORD_1)", "v_cvt_f16_f32 + v_cvt_f32_f16", "v_permlane32_swap",
             "v_mfma_f32_32x32x16_f16", "ds_read_b128", "v_pk_add_f32", "v_and_or_b32"]

    def agg_runner(kind):
        def go():
            rc = agg.hz_aggressor(kind, 400, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        return go
    runners = {"nothing": lambda: None}
    runners.update({KINDS[k]: agg_runner(k) for k in range(len(KINDS))})
elif os.environ.get("HZ_GEMM", "0") == "1":
    runners = {"nothing": lambda: None, "torch f16 GEMM 4096^3 (hipBLASLt)": lambda: _ga @ _gb,
               "torch bf16 GEMM 4096^3": lambda: _bf_a @ _bf_b, "torch f32 GEMM 4096^3": lambda: _fa @ _fb}
else:
  runners = {"nothing": lambda: None,
             "conv_mfma f16x3 256ch@32": conv_runner(20, 32, 32, 256, 256, 1),
             "conv_sp 256ch@32": sp_conv_runner(20, 32, 32, 256, 256)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(EDITS)
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name in names:
    fn = host.hz_load(os.path.join(HERE, name + ".hsaco").encode(), KERNEL)
    assert fn, name

    def warp(out):
        rc = host.hz_launch(fn, feat.data_ptr(), trans.data_ptr(), na.data_ptr(), B, A, h, w, c, out.data_ptr(),
                            trace.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return out

    ref = warp(torch.empty((B, A, A - 1, h, w, c), device="cuda")).clone()
    torch.cuda.synchronize()
    outs = [torch.empty_like(ref) for _ in range(8)]
    line = "%-4s %-100s" % (name, EDITS[name][0])
    shown = False
    for rname, co in runners.items():
        co()
        torch.cuda.synchronize()
        bad = 0
        trace.zero_()
        for trial in range(ROUNDS):
            for i in range(8):
                with torch.cuda.stream(s2):
                    co()
                    co()
                with torch.cuda.stream(s1):
                    warp(outs[i])
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
            if name == "e0" and bad and not shown and rname != "nothing":
                shown = True
                o = next(o for o in outs if not torch.equal(o, ref))
                idx = (o != ref).nonzero()
                recs = sorted(set(tuple(r[:5]) for r in idx.tolist()))
                print("   analysis of one differing replay beside %s: %d wrong values in %d pixel rows" % (rname, len(idx), len(recs)))
                for (bb, ii, jj, yy, xx) in recs[:6]:
                    wrong, right = o[bb, ii, jj, yy, xx], ref[bb, ii, jj, yy, xx]
                    ch = (wrong != right).nonzero().flatten()
                    msg = "      pair (b%d i%d j%d) pixel (%d,%d): channels %d..%d (%d values)" % (bb, ii, jj, yy, xx, int(ch.min()), int(ch.max()), len(ch))
                    w64 = wrong[192:256]
                    plane = ref[bb, ii, jj]
                    for sft in (0, 64, 128, 192):
                        hit = (plane[:, :, sft:sft + 64] == w64).all(-1).nonzero()
                        if len(hit):
                            msg += "; == serial values of pixel %s channels %d..%d" % (hit[0].tolist(), sft, sft + 63)
                    if (w64 == 0).all():
                        msg += "; all zero"
                    print(msg)
                    print("         right[192:200] = %s" % ["%.5f" % v for v in right[192:200].tolist()])
                    print("         wrong[192:200] = %s" % ["%.5f" % v for v in wrong[192:200].tolist()])
                    # is the wrong run the right run of a neighbouring pixel row of the same wave (p +- 4k) or adjacent pixels?
                    p0 = yy * w + xx
                    for dp in (-8, -4, -1, 1, 4, 8):
                        q = p0 + dp
                        if 0 <= q < h * w and torch.equal(plane.view(h * w, c)[q, 192:256], w64):
                            print("         == serial values of pixel index p%+d" % dp)
        line += " | beside %s: %2d/%d" % (rname, bad, 8 * ROUNDS)
        if name == "e20" and int(trace[768:].abs().sum()):
            ex = torch.cat([trace[1024:1024 + 256].view(64, 4), trace[768:768 + 64].view(64, 1)], 1)
            fl = ex.view(torch.float32)
            print("   beside %s: v_floor_f32 examples left by differing lanes (lane: probe, input, own result, lane 0's result)" % rname)
            if int(ex[48, 4]) == 94:
                # whose value is it?  ix of the first rotated-map tap (k = 0) for every (b, i, j, pixel), as the kernel computes it
                tm = trans.view(B, A, A, 16)
                pix = torch.arange(h * w, device="cuda")
                px, py = (pix % w).float(), (pix // w).float()
                coords = torch.zeros(8, B, A, A, h * w, device="cuda")      # [2 * k + (0: ix, 1: iy)]
                for bb in range(B):
                    for ii in range(A):
                        for jj in range(A):
                            m = tm[bb, ii, jj]
                            gx = (2 * px + 1) / w - 1 + 4 * m[3] / 128
                            gy = (2 * py + 1) / h - 1 - 4 * m[7] / 128
                            x0, y0 = torch.floor(((gx + 1) * w - 1) * 0.5), torch.floor(((gy + 1) * h - 1) * 0.5)
                            for k in range(4):
                                qbx, qby = (2 * (x0 + (k & 1)) + 1) / w - 1, (2 * (y0 + (k >> 1)) + 1) / h - 1
                                coords[2 * k, bb, ii, jj] = ((m[0] * qbx + m[1] * qby + 1) * w - 1) * 0.5
                                coords[2 * k + 1, bb, ii, jj] = ((m[4] * qbx + m[5] * qby + 1) * h - 1) * 0.5
                stale, f0, pvict = float(fl[48, 0]), float(fl[48, 2]), int(ex[48, 1]) + int(ex[48, 3])
                src = ((coords - stale).abs() < 3e-5).nonzero().tolist()
                vict = ((torch.floor(coords[:, :, :, :, pvict]) == f0)).nonzero().tolist()
                print("      the wave worked on pixel %d (workgroup pixels %d..%d); (coordinate, b, i, j) whose floor is lane 0's result: %s" % (
                    pvict, int(ex[48, 1]), int(ex[48, 1]) + 31, vict[:10]))
                print("      lanes 48..63 hold %r = coordinate of (which, b, i, j, pixel): %s" % (stale, src[:10]))
            for ln in range(64):
                if int(ex[ln, 4]):
                    print("      lane %2d: probe %d  input %r (0x%08x)  block pixel base %d  lane 0's result %r  pixel-loop counter pp (wave + 4 * iteration) = %d" % (
                        ln, int(ex[ln, 4]), float(fl[ln, 0]), int(ex[ln, 0]) & 0xffffffff, int(ex[ln, 1]), float(fl[ln, 2]), int(ex[ln, 3])))
        if name in ("e19", "e20") and int(trace[:768].sum()):
            idx = {int(t.split()[0]): t for t in open(os.path.join(HERE, "e19_index.txt")).read().splitlines()}
            nz = trace[:768].nonzero().flatten().tolist()
            print("   beside %s: per-instruction count of lanes that differ from lane 0 (first 24 non-zero)" % rname)
            for k in nz[:int(os.environ.get('HZ_TOP', 24))]:
                print("      %6d  %s" % (int(trace[k]), idx.get(k, "?")))
        if name in ("e17", "e18"):
            line += " [lanes off: weights %d / offsets %d before the loop, %d / %d after]" % tuple(int(t) for t in trace[1:5])
    print(line, flush=True)
