"""Does any kernel of the shipped step read memory it (or a predecessor) never wrote?  torch.empty is replaced
by a NaN fill for the whole eager step; the first layer whose output holds a NaN read uninitialised memory."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from disconet_amd import Config, DiscoNet, ops, model as M  # noqa: E402
from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats  # noqa: E402

AGENTS, BATCH, HW = 5, 4, 256
torch.manual_seed(0)
net = DiscoNet(Config(map_hw=HW), kd_flag=0, num_agent=AGENTS)
randomize_bn_stats(net)
net.eval().cuda()
indices, offsets, _ = make_sparse_scene_batch(BATCH, AGENTS, HW)
indices, offsets = indices.cuda(), offsets.cuda()
trans = make_trans_matrices(BATCH, AGENTS, jitter_seed=0).cuda()
na = torch.full((BATCH, AGENTS), AGENTS, dtype=torch.int64).cuda()


def step():
    with torch.no_grad():
        return net(ops.scatter_dense_sp(indices, offsets, AGENTS * BATCH, (HW, HW, 13)), trans, na, BATCH)


ref = {k: v.clone() for k, v in step().items()}
real_empty = torch.empty


def nan_empty(*a, **kw):
    t = real_empty(*a, **kw)
    if t.is_cuda and t.dtype in (torch.float32, torch.float16):
        t.fill_(float("nan"))
    elif t.is_cuda and t.dtype == torch.uint8:
        t.fill_(255)
    return t


def report(name, out):
    for i, t in enumerate(out if isinstance(out, tuple) else (out,)):
        if t is None:
            continue
        x = t.nhwc() if isinstance(t, ops.SpTensor) else t
        bad = int(torch.isnan(x).sum())
        if bad:
            idx = torch.isnan(x).nonzero()
            print("   %-14s output %d: %d NaN of %d, first at %s, last at %s" % (name, i, bad, x.numel(), idx[0].tolist(), idx[-1].tolist()))


for cls in (M._ConvLayer, M._ConvPostLayer):
    orig = cls.run

    def run(self, *a, __orig=orig, **kw):
        out = __orig(self, *a, **kw)
        torch.empty = real_empty
        report(self.name, out)
        torch.empty = nan_empty
        return out
    cls.run = run
orig_fuse = DiscoNet.fuse


def fuse(self, *a, **kw):
    out = orig_fuse(self, *a, **kw)
    torch.empty = real_empty
    report("fuse", out)
    torch.empty = nan_empty
    return out


DiscoNet.fuse = fuse
torch.empty = nan_empty
got = step()
torch.empty = real_empty
print("with NaN-filled allocations: %s" % ", ".join("%s %s" % (k, "same" if torch.equal(got[k], ref[k]) else "DIFFERS(%d values, %d NaN)" % (
    int((got[k] != ref[k]).sum()), int(torch.isnan(got[k]).sum()))) for k in ref))
