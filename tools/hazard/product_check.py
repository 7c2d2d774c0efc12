"""Product-level companion of the hazard reproducer: the SHIPPED step (bench.py's workload, conv_math sp)
replayed from two captured graphs on two alternating streams; every replay is checksummed on its stream
and compared with the checksum of a replay run alone.    python tools/hazard/product_check.py [replays]
"""
import os
os.environ.setdefault("DISCONET_UNSAFE_OVERLAP", "1")   # hazard study: co-scheduling on purpose
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from disconet_amd import Config, DiscoNet, ops  # noqa: E402
from disconet_amd.graph import GraphedStep  # noqa: E402
from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
AGENTS, BATCH, HW = 5, 4, 256
torch.manual_seed(0)
model = DiscoNet(Config(map_hw=HW), kd_flag=0, num_agent=AGENTS)
randomize_bn_stats(model)
model.eval().cuda()
indices, offsets, _ = make_sparse_scene_batch(BATCH, AGENTS, HW)
indices, offsets = indices.cuda(), offsets.cuda()
trans = make_trans_matrices(BATCH, AGENTS, jitter_seed=0).cuda()
na = torch.full((BATCH, AGENTS), AGENTS, dtype=torch.int64).cuda()


def step():
    with torch.no_grad():
        return model(ops.scatter_dense_bits(indices, offsets, AGENTS * BATCH, (HW, HW, 13)), trans, na, BATCH)


def checksum(out):
    return out["cls"].view(torch.int64).sum() + 3 * out["loc"].view(torch.int64).sum()


for overlap in (False, True):
    model.overlap_streams = overlap
    slots = [(GraphedStep(step), torch.cuda.Stream()) for _ in range(2)]
    want = checksum(slots[0][0]())
    torch.cuda.synchronize()
    checks = torch.zeros(N, dtype=torch.int64, device="cuda")
    for i in range(N):
        g, st = slots[i % 2]
        with torch.cuda.stream(st):
            checks[i].copy_(checksum(g()))
    torch.cuda.synchronize()
    bad = int((checks != want).sum())
    print("shipped step, 2 graphs on 2 streams, intra-step side stream %s: %d of %d replays differ from the serial replay"
          % ("on " if overlap else "off", bad, N))

# where and how much: overlap off, a few replays kept whole
model.overlap_streams = False
slots = [(GraphedStep(step), torch.cuda.Stream()) for _ in range(2)]
ref = {k: v.clone() for k, v in slots[0][0]().items() if torch.is_tensor(v)}
torch.cuda.synchronize()
for mode in ("alternating streams, device idle between replays", "alternating streams, back to back"):
    kept = []
    for i in range(8):
        g, st = slots[i % 2]
        with torch.cuda.stream(st):
            out = g()
            kept.append({k: v.clone() for k, v in out.items() if torch.is_tensor(v)})
        if mode.endswith("between replays"):
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for i, o in enumerate(kept):
        msg = []
        for k in ref:
            d = (o[k] != ref[k])
            if d.any():
                idx = d.nonzero()
                msg.append("%s: %d of %d values differ, max |diff| %.3g, first at %s" % (
                    k, int(d.sum()), d.numel(), float((o[k] - ref[k]).abs().max()), idx[0].tolist()))
        print("   [%s] replay %d: %s" % (mode, i, "; ".join(msg) if msg else "identical"))
