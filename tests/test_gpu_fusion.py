"""K4-K6: warp / attention / agent-softmax / weighted sum vs the oracle."""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_warp_unit_poses_vs_oracle_and_golden(golden_dir):
    from disconet_amd import ops
    from oracle.disconet_ref import feature_transformation
    g = np.load(os.path.join(golden_dir, "warp_unit.npz"))
    feat = cases.warp_feature()                        # [1, C, H, W]
    # two agents holding the same map; warp agent 1 into agent 0 with each pose
    feat2 = torch.cat([feat, feat], 0)                 # agent-major, B = 1
    for name, pose in cases.WARP_POSES.items():
        trans = torch.eye(4).repeat(1, 2, 2, 1, 1)
        trans[0, 0, 1] = torch.from_numpy(pose)
        na = torch.tensor([2], dtype=torch.int32)
        warped = ops.warp_neighbors(_nhwc(feat2).cuda(), trans.cuda(), na.cuda(), 1, 2)
        got = warped[0, 0, 0].cpu().permute(2, 0, 1).numpy()
        want = feature_transformation(0, 0, feat.unsqueeze(0), torch.from_numpy(pose)[None],
                                      tuple(feat.shape)).numpy()
        assert np.abs(got - want).max() <= 1e-5, name
        assert np.abs(got - g[name]).max() <= 1e-5, name


@pytest.mark.parametrize("case", ["cfg1_f1", "ragged_a4"])
def test_fusion_block_vs_oracle(case):
    from disconet_amd import Config, DiscoNet
    c = cases.MODEL_CASES[case]
    ref = cases.ref_model(c["map_hw"], c["agents"])
    bevs, trans, na = cases.model_inputs(case)
    with torch.no_grad():
        x3 = ref.u_encoder(bevs.permute(0, 1, 4, 2, 3))[3]
        fused_ref = ref(bevs, trans, na, c["batch"])[-1]
    m = DiscoNet(Config(map_hw=c["map_hw"]), kd_flag=1, num_agent=c["agents"]).eval()
    m.load_state_dict(ref.state_dict())
    m.cuda()
    P = m._get_plan()
    num_agent = na[:, 0].to(torch.int32).cuda()
    fused, wts = m.fuse(_nhwc(x3).cuda(), trans.cuda().contiguous(), num_agent, c["batch"], P,
                        want_weights=True)
    got = fused.cpu().permute(0, 3, 1, 2)
    err = (got - fused_ref).abs().max().item()
    assert err <= TOL, "fused max abs err %.3e" % err
    # softmax weights over the live neighbours of every live ego sum to one
    live = c["live"] or [c["agents"]] * c["batch"]
    w = wts.cpu()
    for b, n in enumerate(live):
        for i in range(n):
            s = w[b, i, :n].sum(0)
            assert (s - 1).abs().max() <= 1e-5


def test_identical_agents_identity_pose_fuse_to_themselves():
    """size-independent property: all agents see the same map under identity
    poses -> every neighbour equals the ego, so fused == ego whatever the weights."""
    from disconet_amd import Config, DiscoNet
    torch.manual_seed(0)
    A, B = 5, 4
    m = DiscoNet(Config(), kd_flag=1, num_agent=A).eval().cuda()
    x = torch.randn(1, 32, 32, 256).clamp_(min=0).repeat(A * B, 1, 1, 1).contiguous().cuda()
    trans = torch.eye(4).repeat(B, A, A, 1, 1).cuda()
    na = torch.full((B,), A, dtype=torch.int32).cuda()
    fused = m.fuse(x, trans, na, B, m._get_plan())
    assert (fused - x).abs().max().item() <= 1e-5


def test_fusion_block_at_baseline_map_size_vs_oracle_and_golden(golden_dir):
    """5 agents x [256, 32, 32] (SURVEY.md §8(c) golden (3)): warp + attention + softmax +
    weighted sum through the C ABI vs the oracle's loop and the committed golden"""
    from disconet_amd import Config, DiscoNet
    g = np.load(os.path.join(golden_dir, "fusion_5x256.npz"))
    ref = cases.ref_model(256, 5)
    feat, trans, na = cases.fusion_inputs()
    want = cases.ref_fuse(ref, feat, trans, na)
    m = DiscoNet(Config(), kd_flag=1, num_agent=5).eval()
    m.load_state_dict(ref.state_dict())
    m.cuda()
    fused = m.fuse(_nhwc(feat).cuda(), trans.cuda().contiguous(), na[:, 0].to(torch.int32).cuda(), 1,
                   m._get_plan())
    got = fused.cpu().permute(0, 3, 1, 2)
    assert (got - want).abs().max().item() <= TOL
    assert np.abs(got.numpy()[:, ::4, ::2, ::2] - g["fused"]).max() <= TOL


@pytest.mark.parametrize("case", ["cfg1_f1", "ragged_a4"])
def test_one_launch_fusion_matches_the_three_launch_form(case):
    """dn_disco_fuse_mlp (all MLP layers + softmax + sum in one launch) against the
    dn_conv2d x2 + dn_disco_fuse_tail chain it replaces: same fused maps and weights, and the
    split-planar output is the split of the fp32 one"""
    from disconet_amd import Config, DiscoNet, ops
    c = cases.MODEL_CASES[case]
    ref = cases.ref_model(c["map_hw"], c["agents"])
    bevs, trans, na = cases.model_inputs(case)
    with torch.no_grad():
        x3 = ref.u_encoder(bevs.permute(0, 1, 4, 2, 3))[3]
    feat = _nhwc(x3).cuda()
    num_agent = na[:, 0].to(torch.int32).cuda()
    outs = {}
    for one_launch in (True, False):
        m = DiscoNet(Config(map_hw=c["map_hw"]), kd_flag=1, num_agent=c["agents"]).eval()
        m.load_state_dict(ref.state_dict())
        m.fuse_mlp = one_launch
        m.cuda()
        P = m._get_plan()
        assert ("_fuse_mlp" in P) == one_launch
        outs[one_launch] = m.fuse(feat, trans.cuda().contiguous(), num_agent, c["batch"], P, want_weights=True)
        if one_launch:
            sp = m.fuse(feat, trans.cuda().contiguous(), num_agent, c["batch"], P, sp_out=True)
            assert isinstance(sp, ops.SpTensor)
            assert torch.equal(sp.data, ops.SpTensor.from_nhwc(outs[True][0]).data)
    assert (outs[True][0] - outs[False][0]).abs().max().item() <= 2e-5
    assert (outs[True][1] - outs[False][1]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("A,B,h,w,C,live,v2i", [
    (5, 2, 32, 32, 256, None, False),       # the BASELINE fusion shape: waves get 2, 1, 1, 1 list slots
    (8, 1, 32, 32, 256, None, False),       # the agent-sharded scenes: two slots per wave
    (8, 2, 32, 32, 256, [5, 2], False),     # padded agents pass through; a sample with fewer slots than waves
    (4, 2, 20, 28, 128, [4, 3], True),      # ragged last tile, only_v2i
    (3, 1, 12, 16, 64, None, False),        # one k-step per wave in the weighted sum
    (1, 2, 16, 16, 256, None, False),       # the ego alone
    (5, 4, 32, 32, 256, None, False),       # BASELINE configs[1]: 640 tiles -- workgroups of three tiles in the weight-in-LDS form
    (6, 4, 32, 32, 256, [6, 3, 6, 1], False),   # 768 tiles: workgroups of four; padded agents inside a workgroup
])
def test_four_wave_attention_launch_is_bit_identical_to_the_one_wave_form(A, B, h, w, C, live, v2i):
    """dn_disco_fuse_mlp with the work of a 32-pixel tile split over four waves (small launches) against the
    one-wave chain: fused maps (fp32 and split-planar), softmax weights and the ego sub-range form, bitwise"""
    from disconet_amd import Config, DiscoNet, ops
    from disconet_amd.synthetic import make_trans_matrices
    torch.manual_seed(A * 10 + h)
    layer = {256: 3, 128: 2, 64: 1}[C]
    feat = torch.randn(A * B, h, w, C).clamp_(min=0).cuda()
    trans = make_trans_matrices(B, A, jitter_seed=2).cuda()
    na = torch.tensor(live or [A] * B, dtype=torch.int32).cuda()
    torch.manual_seed(3)
    m = DiscoNet(Config(), layer=layer, kd_flag=1, num_agent=A, only_v2i=v2i).eval()
    for bn in (m.pixel_weighted_fusion.bn1_1, m.pixel_weighted_fusion.bn1_2, m.pixel_weighted_fusion.bn1_3):
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    m.cuda()
    P = m._get_plan()
    assert "_fuse_mlp" in P
    outs = {}
    try:
        for waves in (1, 4, 2):
            ops.set_fuse_mlp_waves(waves)
            fused, weights = m.fuse(feat, trans, na, B, P, want_weights=True)
            sp = m.fuse(feat, trans, na, B, P, sp_out=True)
            part = m.fuse(feat, trans, na, B, P, ego_first=A - 1, ego_count=1)
            torch.cuda.synchronize()
            outs[waves] = (fused, weights, sp.data.clone(), part)
    finally:
        ops.set_fuse_mlp_waves(0)
    assert torch.isfinite(outs[4][0]).all()
    for x, y in zip(outs[1], outs[4]):
        assert torch.equal(x, y)
    # the weight-in-LDS form (one wave per tile, the layer-1 matrices staged once per workgroup of 2-4 tiles): same bits
    for x, y in zip(outs[1], outs[2]):
        assert torch.equal(x, y)
    assert torch.equal(outs[4][3], outs[4][0][(A - 1) * B:])
