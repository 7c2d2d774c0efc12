"""Host-side logic of the training step that needs no GPU: the fusion index lists
(disconet_amd.train.fusion_lists) against the reference's loop order, and closed-form
properties of the oracle's loss restatement (oracle/train_ref.py)."""
import math

import pytest
import torch

from disconet_amd.synthetic import make_trans_matrices
from disconet_amd.train import fusion_lists


def _reference_calls(agents, batch, live, only_v2i):
    """(ego image, source image of the neighbour map) for every PixelWeightedFusion call, in the
    order upstream DiscoNet.forward makes them"""
    img = lambda a, b: a * batch + b
    calls = []
    for b in range(batch):
        n = live[b]
        for i in range(n):
            calls.append((img(i, b), img(i, b)))                 # cat[tg_agent, tg_agent]
            for j in range(n):
                if j != i and not (only_v2i and i != 0 and j != 0):
                    calls.append((img(i, b), img(j, b)))         # cat[tg_agent, warp(j -> i)]
    return calls


@pytest.mark.parametrize("agents,batch,live,only_v2i", [
    (2, 1, [2], False), (4, 2, [3, 2], False), (5, 4, [5, 5, 5, 5], False),
    (4, 2, [3, 2], True), (3, 2, [1, 2], False), (3, 1, [0], False)])
def test_fusion_lists_follow_the_reference_loop(agents, batch, live, only_v2i):
    trans = make_trans_matrices(batch, agents, jitter_seed=1)
    F = fusion_lists(agents, only_v2i, trans, torch.tensor(live), batch, torch.device("cpu"))
    NI = agents * batch
    calls = _reference_calls(agents, batch, live, only_v2i)
    n_warps = sum(1 for e, s in calls if e != s)
    assert F["n_warps"] == n_warps and F["n_calls"] == len(calls)
    src = F["src_image"].tolist()
    ego_image = F["ego_image"].tolist()
    # the BatchNorm running-statistics order = the reference's call order
    got = [(ego_image[p], p if p < NI else src[p - NI]) for p in F["order"].tolist()]
    assert got == calls
    # every map (own or warped) sits in exactly one ego's list; padding agents fuse with themselves only
    assert sorted(F["map_image"].tolist()) == list(range(NI + n_warps))
    first, pairs, maps = F["first"].tolist(), F["pair_index"].tolist(), F["map_image"].tolist()
    for e, out_img in enumerate(F["ego_out"].tolist()):
        a, b = out_img // batch, out_img % batch
        entries = list(zip(pairs[first[e]:first[e + 1]], maps[first[e]:first[e + 1]]))
        if a >= live[b]:
            assert entries == [(-1, out_img)]
        else:
            assert entries[0] == (out_img, out_img)
            assert all(p == m and p >= NI and ego_image[p] == out_img for p, m in entries[1:])
    # poses: trans_matrices[b, ego, neighbour] of each warp
    for w_, p in enumerate(range(NI, NI + n_warps)):
        e, s = ego_image[p], src[w_]
        assert torch.equal(F["poses"][w_], trans[e % batch, e // batch, s // batch])
    # dE = sum over the pairs of an ego image
    efirst, epairs = F["efirst"].tolist(), F["epairs"].tolist()
    for im in range(NI):
        assert sorted(epairs[efirst[im]:efirst[im + 1]]) == [p for p in range(NI + n_warps) if ego_image[p] == im]


def test_oracle_losses_closed_forms():
    from oracle.train_ref import focal_loss, smooth_l1_loss
    g = torch.Generator().manual_seed(0)
    cls = torch.randn(50, 2, generator=g, dtype=torch.float64)
    fg = torch.rand(50, generator=g) < 0.3
    labels = torch.stack([(~fg).double(), fg.double()], -1)
    labels[:5] = 0                                     # don't-care rows contribute nothing
    # gamma = 0: alpha-weighted cross entropy
    logp = torch.log_softmax(cls, -1)
    want = -(torch.where(fg, 0.25 * logp[:, 1], 0.75 * logp[:, 0]))[5:].sum()
    assert abs(float(focal_loss(cls, labels, 0.25, 0.0)) - float(want)) < 1e-12
    # gamma = 2 down-weights easy examples: a confidently right anchor costs ~0
    easy = torch.tensor([[10.0, -10.0]], dtype=torch.float64)
    assert float(focal_loss(easy, torch.tensor([[1.0, 0.0]]), 0.25, 2.0)) < 1e-15
    # smooth L1 (sigma 3): quadratic inside |d| <= 1/9, linear outside, continuous at the joint
    d = torch.tensor([[0.05, -0.05, 1.0 / 9.0, 2.0, -3.0, 0.0]], dtype=torch.float64)
    l = smooth_l1_loss(d, torch.zeros_like(d), torch.ones(1, dtype=torch.float64), 3.0)
    want = 2 * 0.5 * 9 * 0.05 ** 2 + 0.5 * 9 / 81 + (2 - 0.5 / 9) + (3 - 0.5 / 9)
    assert abs(float(l) - want) < 1e-12
    assert float(smooth_l1_loss(d, torch.zeros_like(d), torch.zeros(1, dtype=torch.float64), 3.0)) == 0.0


def test_kd_loss_is_zero_for_identical_maps_and_scales_with_weight():
    from oracle.teacher_ref import kd_loss
    g = torch.Generator().manual_seed(1)
    maps = [torch.randn(2, c, 8, 8, generator=g) for c in (256, 128, 64, 256)]
    assert abs(float(kd_loss(maps, maps, 1e5))) < 1e-4          # fp32 log-softmax noise x 1e5
    other = [m + 0.1 * torch.randn(m.shape, generator=g) for m in maps]
    a, b = float(kd_loss(maps, other, 1e5)), float(kd_loss(maps, other, 2e5))
    assert a > 0 and math.isclose(b, 2 * a, rel_tol=1e-6)
