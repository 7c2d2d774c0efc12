"""Agent-sharded seams on ONE GPU: fusing an ego sub-range against all agents'
maps must equal the matching rows of the full fusion (the RCCL all-gather itself
needs > 1 GPU; its host logic is covered by tests/test_sharded_gloo.py)."""
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def test_ego_subranges_match_full_fusion():
    from disconet_amd import Config, DiscoNet
    c = cases.MODEL_CASES["ragged_a4"]
    A, B = c["agents"], c["batch"]
    ref = cases.ref_model(c["map_hw"], A)
    bevs, trans, na = cases.model_inputs("ragged_a4")
    m = DiscoNet(Config(map_hw=c["map_hw"]), kd_flag=1, num_agent=A).eval()
    m.load_state_dict(ref.state_dict())
    m.cuda()
    P = m._get_plan()
    x3 = m.encode(bevs.cuda(), P)[3]
    num_agent = na[:, 0].to(torch.int32).cuda()
    tr = trans.cuda().contiguous()
    full = m.fuse(x3, tr, num_agent, B, P)
    for first, count in ((0, 1), (1, 2), (3, 1), (2, 2)):
        part = m.fuse(x3, tr, num_agent, B, P, ego_first=first, ego_count=count)
        assert part.shape[0] == count * B
        assert torch.equal(part, full[first * B:(first + count) * B]), (first, count)


def test_eight_agents_one_ego_per_rank_slices():
    """BASELINE configs[4]: 8-agent scenes sharded one agent per GPU -- every rank fuses ONE ego
    (ego_count = 1) against the gathered maps of all 8; each slice must be the matching rows of the
    full fusion, bit for bit (8 slots all live, and 8 slots with 5 live)"""
    from disconet_amd import Config, DiscoNet
    from disconet_amd.synthetic import make_scene_batch
    A, B, hw = 8, 2, 128
    torch.manual_seed(3)
    m = DiscoNet(Config(map_hw=hw), kd_flag=0, num_agent=A).eval().cuda()
    P = m._get_plan()
    for live in ([8, 8], [5, 8]):
        bevs, trans, na = make_scene_batch(B, A, hw, live=live, jitter_seed=7)
        x3 = m.encode(bevs.cuda(), P)[3]
        num_agent = na[:, 0].to(torch.int32).cuda()
        tr = trans.cuda().contiguous()
        full = m.fuse(x3, tr, num_agent, B, P)
        assert torch.isfinite(full).all()
        for ego in range(A):
            part = m.fuse(x3, tr, num_agent, B, P, ego_first=ego, ego_count=1)
            assert torch.equal(part, full[ego * B:(ego + 1) * B]), (live, ego)
        # dead slots keep their own (un-fused) map
        assert torch.equal(full[7 * B + 0], x3[7 * B + 0]) == (live[0] < 8)


def test_hip_engine_single_rank_world():
    """world_size 1 (gloo on CPU for the group, tensors on the GPU): the sharded
    forward through HipEngine equals the plain forward."""
    import os
    import torch.distributed as dist
    from disconet_amd import Config, DiscoNet, sharded
    c = cases.MODEL_CASES["cfg1_f1"]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ref = cases.ref_model(c["map_hw"], c["agents"])
        bevs, trans, na = cases.model_inputs("cfg1_f1")
        m = DiscoNet(Config(map_hw=c["map_hw"]), kd_flag=0, num_agent=c["agents"]).eval()
        m.load_state_dict(ref.state_dict())
        m.cuda()
        with torch.no_grad():
            want = m(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
            got, _ = sharded.forward_agent_sharded(sharded.HipEngine(m), bevs.cuda(), trans.cuda(),
                                                   na.cuda(), c["batch"])
        assert torch.equal(got["cls"], want["cls"]) and torch.equal(got["loc"], want["loc"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [8, 4])
def test_graphed_emulated_share_equals_unsharded_rows(world):
    """bench.py --mode agent --emulate-world: one rank's share of a `world`-rank agent-sharded run (its agents
    through graph A, the peers' maps by device copy, its egos through graph B) must be the matching rows of
    the unsharded 8-agent forward, bit for bit -- for every rank of the emulated world."""
    from disconet_amd import Config, DiscoNet, ops, sharded
    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices
    A, B, hw = 8, 2, 128
    torch.manual_seed(5)
    m = DiscoNet(Config(map_hw=hw), kd_flag=0, num_agent=A).eval().cuda()
    engine = sharded.HipEngine(m)
    indices, offsets, _ = make_sparse_scene_batch(B, A, hw)
    trans = make_trans_matrices(B, A, jitter_seed=3).cuda()
    na = torch.full((B, A), A, dtype=torch.int64).cuda()
    dims = (hw, hw, 13)

    def inputs(first, count):
        lo, hi = int(offsets[first * B]), int(offsets[(first + count) * B])
        idx = indices[lo:hi].contiguous().cuda()
        off = (offsets[first * B:(first + count) * B + 1] - lo).to(torch.int32).cuda()
        return lambda: ops.scatter_dense_sp(idx, off, count * B, dims, hi_only=True)

    with torch.no_grad():
        want = m(ops.scatter_dense_sp(indices.cuda(), offsets.cuda(), A * B, dims), trans, na, B)
    full = sharded.GraphedAgentStep(engine, inputs(0, A), trans, na, B, 0, A)      # no process group: local exchange
    res, _ = full()
    torch.cuda.synchronize()
    assert torch.equal(res["cls"], want["cls"]) and torch.equal(res["loc"], want["loc"])
    feat_all = full.feat_all.clone()
    cnt = A // world
    for r in (0, world - 1, world // 2):
        share = sharded.GraphedAgentStep(engine, inputs(r * cnt, cnt), trans, na, B, r * cnt, cnt,
                                         emulate_feat_all=feat_all)
        for _ in range(2):          # replayed: the second replay must reproduce the first
            got, _ = share()
        torch.cuda.synchronize()
        rows = slice(r * cnt * B, (r + 1) * cnt * B)
        assert torch.equal(got["cls"], want["cls"][rows]) and torch.equal(got["loc"], want["loc"][rows]), (world, r)
        if r == 0:      # the opt-in one-graph form (graph A + exchange + graph B in ONE capture): the same bits, replayed twice
            assert share.capture_one_graph() is None
            for _ in range(2):
                got1, _ = share()
            torch.cuda.synchronize()
            assert torch.equal(got1["cls"], want["cls"][rows]) and torch.equal(got1["loc"], want["loc"][rows]), (world, "one graph")
