"""Does a kernel's output change when another kernel runs beside it on a second stream?
Victim: the pose warp (dn_warp_neighbors); co-runners: the conv engine's tiles, torch kernels.
   python tools/corun_check.py"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from disconet_amd import ops  # noqa: E402
from disconet_amd.synthetic import make_trans_matrices  # noqa: E402

torch.manual_seed(0)
B, A, h, w, c = 4, 5, 32, 32, 256
feat = torch.randn(A * B, h, w, c, device="cuda")
trans = make_trans_matrices(B, A, jitter_seed=0).cuda()
na = torch.full((B,), A, dtype=torch.int32).cuda()


def warp(out):
    ops.warp_neighbors(feat, trans, na, B, A, False, 0, A, out=out)
    return out


def conv_runner(n, hh, ww, cin, cout, k, math, stride=1):
    x = torch.randn(n, hh, ww, cin, device="cuda")
    d = ops.conv_desc(n, hh, ww, cin, cout, k, stride=stride, math=math)
    pk = ops.pack_conv_weights(d, torch.randn(cout, cin, k, k, device="cuda") * 0.02)
    one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    ho, wo = ops.conv_out_hw(d)
    y = torch.empty(n, ho, wo, cout, device="cuda")
    return lambda: ops.conv2d(d, x, pk, one, zero, out=y)


ref = warp(torch.empty((B, A, A - 1, h, w, c), device="cuda")).clone()
torch.cuda.synchronize()
big = torch.randn(20, 256, 256, 32, device="cuda")
z = torch.empty_like(big)
runners = {
    "nothing": lambda: None,
    "torch elementwise": lambda: torch.mul(big, 1.5, out=z),
    "conv 3x3 f32   256ch@32 (256x32 tile)": conv_runner(20, 32, 32, 256, 256, 3, 0),
    "conv 3x3 f16x3 256ch@32 (256x32 tile)": conv_runner(20, 32, 32, 256, 256, 3, 1),
    "conv 3x3 f16x3 64ch@128 (256x64 tile)": conv_runner(20, 128, 128, 64, 64, 3, 1),
    "conv 3x3 f16x3 32ch@256 (256x32 tile)": conv_runner(20, 256, 256, 32, 32, 3, 1),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
outs = [torch.empty_like(ref) for _ in range(8)]
for name, co in runners.items():
    co()
    torch.cuda.synchronize()
    bad = 0
    for trial in range(2):
        for i in range(8):
            with torch.cuda.stream(s2):
                co()
                co()
            with torch.cuda.stream(s1):
                warp(outs[i])
        torch.cuda.synchronize()
        bad += sum(1 for o in outs if not torch.equal(o, ref))
    print("beside %-40s warp outputs differing from the serial result: %2d of 16" % (name, bad))
