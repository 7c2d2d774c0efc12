"""Bit-exactness of the forward when two steps run concurrently (two streams, eager and as
captured graphs) against a serial run.   python tools/det_check.py"""
import os
os.environ.setdefault("DISCONET_UNSAFE_OVERLAP", "1")   # hazard study: co-scheduling on purpose
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from disconet_amd import Config, DiscoNet, ops  # noqa: E402
from disconet_amd.graph import GraphedStep  # noqa: E402
from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats  # noqa: E402

torch.manual_seed(0)
model = DiscoNet(Config(map_hw=256), kd_flag=1, num_agent=5)
randomize_bn_stats(model)
model.eval().cuda()
indices, offsets, _ = make_sparse_scene_batch(4, 5, 256)
indices, offsets = indices.cuda(), offsets.cuda()
trans = make_trans_matrices(4, 5, jitter_seed=0).cuda()
na = torch.full((4, 5), 5, dtype=torch.int64).cuda()


def step():
    bevs = ops.scatter_dense(indices, offsets, 20, (256, 256, 13))
    with torch.no_grad():
        res, x8, x7, x6, x5, fused = model(bevs, trans, na, 4)
    return {"fused": fused, "x5": x5, "x8": x8, "cls": res["cls"], "loc": res["loc"]}


def nz(a, b):
    d = {k: float((a[k].float() - b[k].float()).abs().max()) for k in a}
    return {k: v for k, v in d.items() if v > 0}


for ov in (False, True):
    model.overlap_streams = ov
    ref = {k: v.clone() for k, v in step().items()}
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for trial in range(4):
        outs = [None, None]
        for i in range(6):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                outs[i % 2] = step()
        torch.cuda.synchronize()
        bad += sum(1 for o in outs if nz(o, ref))
    print("overlap", ov, "eager 2-stream runs that differ from the serial result:", bad, "of 8")
    g1, g2 = GraphedStep(step), GraphedStep(step)
    badg = 0
    for trial in range(4):
        for i in range(8):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                (g1 if i % 2 == 0 else g2)()
        torch.cuda.synchronize()
        badg += sum(1 for g in (g1, g2) if nz(g.outputs, ref))
    print("overlap", ov, "graph 2-in-flight runs that differ:", badg, "of 8", nz(g1.outputs, ref))
