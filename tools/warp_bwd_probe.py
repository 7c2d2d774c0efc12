"""us per dn_warp_backward (rigid poses) at the detector's training shape: 80 warps of 20 source maps, 32 x 32 x 256
(HIP events around 50 calls).  DN_WARP_GATHER_LEGACY=1: round 5's scalar candidate loops."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disconet_amd import train_ops

n_src, n, hw, c = 20, 80, 32, 256
g = torch.Generator().manual_seed(1)
poses = torch.zeros(n, 4, 4)
for k in range(n):
    a = 0.15 * (k % 5) - 0.15 * (k // 16)
    poses[k] = torch.eye(4)
    poses[k, 0, 0], poses[k, 0, 1], poses[k, 1, 0], poses[k, 1, 1] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
    poses[k, 0, 3], poses[k, 1, 3] = 6.0 * (k % 5) - 9.0, 4.0 * (k % 3) - 3.0
src = torch.tensor([(k // 4) % n_src for k in range(n)], dtype=torch.int32).cuda()
d_warped = torch.randn(n, hw, hw, c, generator=g).cuda()
d_src = torch.zeros(n_src, hw, hw, c).cuda()
pc = poses.cuda()
for _ in range(5):
    train_ops.warp_backward(d_warped, pc, src, d_src, rigid=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    train_ops.warp_backward(d_warped, pc, src, d_src, rigid=True)
e1.record()
torch.cuda.synchronize()
print("legacy=%s  %.1f us per call (both passes)" % (os.environ.get("DN_WARP_GATHER_LEGACY", "0"), e0.elapsed_time(e1) * 20.0))
