"""Timing of the fusion block's two launches at the bench shape (and checksums for A/B runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disconet_amd import Config, DiscoNet, ops
from disconet_amd.synthetic import make_trans_matrices, randomize_bn_stats
torch.manual_seed(0)
B, A, h, w, c = 4, 5, 32, 32, 256
model = DiscoNet(Config(map_hw=256), kd_flag=0, num_agent=A)
randomize_bn_stats(model)
model.eval().cuda()
P = model._get_plan()
feat = torch.randn(A * B, h, w, c, device="cuda")
trans = make_trans_matrices(B, A, jitter_seed=1).cuda()
na = torch.tensor([A] * (B - 1) + [A - 1], dtype=torch.int32).cuda()
warped = torch.empty((B, A, A - 1, h, w, c), device="cuda")
FM = os.environ.get("DN_FUSE_FM", "1") != "0"
def run_warp():
    ops.warp_neighbors(feat, trans, na, B, A, False, 0, A, out=warped, fm=FM)
run_warp()
def run():
    return ops.disco_fuse_mlp(feat, warped, na, P["_fuse_mlp"], B, A, False, False, 0, A, sp_out=False, fm=FM)
out = run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
e0.record()
for _ in range(20): run_warp()
e1.record(); torch.cuda.synchronize()
t_warp = 50 * e0.elapsed_time(e1)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("fm=%d warp %.1f us" % (FM, t_warp))
print("fuse_mlp B%d A%d %dx%dx%d: %.1f us  checksum %d" % (B, A, h, w, c, 50 * e0.elapsed_time(e1), int(out.view(torch.int32).long().sum())))
