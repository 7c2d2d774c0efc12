import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from disconet_amd import ops
from disconet_amd.synthetic import make_trans_matrices
torch.manual_seed(0)
for (B, A, h, w, c) in ((4, 5, 32, 32, 256), (2, 3, 20, 28, 64), (1, 5, 32, 32, 512)):
    feat = torch.randn(A * B, h, w, c, device="cuda")
    trans = make_trans_matrices(B, A, jitter_seed=1).cuda()
    na = torch.tensor([A] * (B - 1) + [max(1, A - 1)], dtype=torch.int32).cuda()
    out = torch.empty((B, A, A - 1, h, w, c), device="cuda")
    ops.warp_neighbors(feat, trans, na, B, A, False, 0, A, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.warp_neighbors(feat, trans, na, B, A, False, 0, A, out=out)
    e1.record(); torch.cuda.synchronize()
    print("tiled=%s B%d A%d %dx%dx%d: %.1f us  checksum %d" % (os.environ.get("DN_WARP_TILED", "1"), B, A, h, w, c, 50 * e0.elapsed_time(e1),
          int(out.view(torch.int32).long().sum())))
