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
