"""Soak: the captured default step (bench.py's workload, conv_math sp) replayed N times on one stream; an on-device
checksum of cls + loc after every replay must equal the first one's.    python tools/soak_replay.py [replays]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disconet_amd import Config, DiscoNet, _lib, ops  # noqa: E402
from disconet_amd.graph import GraphedStep  # noqa: E402
from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
AGENTS, BATCH, HW = 5, 4, 256
torch.manual_seed(0)
model = DiscoNet(Config(map_hw=HW), kd_flag=0, num_agent=AGENTS)
randomize_bn_stats(model)
model.eval().cuda()
indices, offsets, _ = make_sparse_scene_batch(BATCH, AGENTS, HW)
indices, offsets = indices.cuda(), offsets.cuda()
trans = make_trans_matrices(BATCH, AGENTS, jitter_seed=0).cuda()
na = ops.live_agent_counts(torch.full((BATCH, AGENTS), AGENTS, dtype=torch.int64).cuda(), "cuda:0")


def step():
    with torch.no_grad():
        return model(ops.scatter_dense_bits(indices, offsets, AGENTS * BATCH, (HW, HW, 13)), trans, na, BATCH)


g = GraphedStep(step)
out = g()
torch.cuda.synchronize()
sums = torch.zeros(N, dtype=torch.int64, device="cuda")
for i in range(N):
    out = g()
    sums[i] = out["cls"].view(torch.int32).sum(dtype=torch.int64) + 3 * out["loc"].view(torch.int32).sum(dtype=torch.int64)
torch.cuda.synchronize()
bad = int((sums != sums[0]).sum().item())
print("default step, hipGraph replay on one stream: %d of %d replays differ from the first (dn_version %d, range flags %d)"
      % (bad, N, _lib.load().dn_version(), ops.sp_range_flags()))
sys.exit(1 if bad else 0)
