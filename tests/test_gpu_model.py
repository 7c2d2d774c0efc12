"""Whole `--com disco` forward through the C ABI vs the CPU oracle and the
committed goldens.  Tolerance: 1e-4 absolute (BASELINE.json north_star) on
kaiming-initialised weights that keep activations O(1)."""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


MATHS = ["f32", "f16x3", "sp"]


def _product(ref, map_hw, agents, kd_flag=1, math="f32", **kw):
    from disconet_amd import Config, DiscoNet
    m = DiscoNet(Config(map_hw=map_hw), kd_flag=kd_flag, num_agent=agents, **kw).eval()
    m.conv_math = math
    m.load_state_dict({"module." + k: v for k, v in ref.state_dict().items()})
    return m.cuda()


def _gpu_outputs(m, bevs, trans, na, batch):
    with torch.no_grad():
        res, x8, x7, x6, x5, fused = m(bevs.cuda(), trans.cuda(), na.cuda(), batch)
    torch.cuda.synchronize()
    return {"cls": res["cls"].cpu(), "loc": res["loc"].cpu(), "x8": x8.cpu(), "x5": x5.cpu(),
            "fused": fused.cpu()}


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("case", list(cases.MODEL_CASES))
def test_model_vs_oracle_and_golden(case, math, golden_dir):
    c = cases.MODEL_CASES[case]
    ref = cases.ref_model(c["map_hw"], c["agents"])
    want = cases.run_ref(case, ref)
    m = _product(ref, c["map_hw"], c["agents"], math=math)
    bevs, trans, na = cases.model_inputs(case)
    got = _gpu_outputs(m, bevs, trans, na, c["batch"])
    g = np.load(os.path.join(golden_dir, "model_cases.npz"))
    for name in want:
        assert got[name].shape == want[name].shape, name
        err = (got[name] - want[name]).abs().max().item()
        assert err <= TOL, "%s/%s max abs err %.3e" % (case, name, err)
        gerr = np.abs(cases.subsample(name, got[name]) - g["%s/%s" % (case, name)]).max()
        assert gerr <= TOL, "%s/%s golden err %.3e" % (case, name, gerr)


@pytest.mark.parametrize("math", MATHS)
def test_model_256_five_agents_default_init(math):
    """BASELINE plane size, torch default init (the bench's weights), batch 1."""
    from disconet_amd.synthetic import make_scene_batch
    ref = cases.ref_model(256, 5, init="torch")
    bevs, trans, na = make_scene_batch(1, 5, 256)
    with torch.no_grad():
        res, x8, x7, x6, x5, fused = ref(bevs, trans, na, 1)
    m = _product(ref, 256, 5, math=math)
    got = _gpu_outputs(m, bevs, trans, na, 1)
    for name, w in (("cls", res["cls"]), ("loc", res["loc"]), ("x8", x8), ("fused", fused)):
        err = (got[name] - w).abs().max().item()
        assert err <= TOL, "%s max abs err %.3e" % (name, err)


@pytest.mark.parametrize("math", MATHS)
def test_benchmarked_path_256_a5_b4_vs_oracle_and_golden(math, golden_dir):
    """The configuration bench.py times, through the form it times (missing #1): BASELINE configs[1] --
    5 agents, batch 4, 256 x 256 x 13 -- kaiming weights (logits of O(1)), the input handed over as sorted sparse voxel lists
    and rebuilt by dn_scatter_dense_bits, so that on the SP engine the stem pair (dn_spconv2d_pre_pair), the K-sliced conv5_1,
    the fragment-major warp and the one-launch attention kernel are the launches that run.  Against the oracle on the same
    inputs and against tests/golden/model_256_a5.npz, 1e-4 absolute.  The fp32-NHWC engines (f32 / f16x3) have no
    occupancy-word source: they get the float32 grid dn_scatter_dense builds from the same lists."""
    from disconet_amd import ops
    c = cases.BENCH_CASE
    want, ref = cases.run_ref_bench_case()
    m = _product(ref, c["map_hw"], c["agents"], math=math)
    indices, offsets, bevs, trans, na = cases.bench_case_inputs()
    n, dims = c["agents"] * c["batch"], (c["map_hw"], c["map_hw"], bevs.shape[-1])
    if math == "sp":
        x = ops.scatter_dense_bits(indices.cuda(), offsets.cuda(), n, dims)
        assert isinstance(x, ops.SpTensor) and x.bits
        P = m._get_plan()
        assert P["conv5_1"].kslices == 4, "conv5_1 must run K-sliced in the benchmarked configuration"
        assert m._stem_pair(x, P) is not None, "the stem pair launch must be the one that runs"
    else:
        x = ops.scatter_dense(indices.cuda(), offsets.cuda(), n, dims)
        assert torch.equal(x.cpu().reshape(bevs.shape), bevs)
    with torch.no_grad():
        res, x8, x7, x6, x5, fused = m(x, trans.cuda(), na.cuda(), c["batch"])
    torch.cuda.synchronize()
    assert ops.sp_range_flags() == 0
    got = {"cls": res["cls"].cpu(), "loc": res["loc"].cpu(), "x8": x8.cpu(), "x5": x5.cpu(), "fused": fused.cpu()}
    g = np.load(os.path.join(golden_dir, "model_256_a5.npz"))
    for name in want:
        assert got[name].shape == want[name].shape, name
        err = (got[name] - want[name]).abs().max().item()
        assert err <= TOL, "%s max abs err %.3e (max |ref| %.2f)" % (name, err, want[name].abs().max())
        gerr = np.abs(cases.subsample_bench(name, got[name]) - g[name]).max()
        assert gerr <= TOL, "%s golden err %.3e" % (name, gerr)


def test_kd_flag_zero_returns_dict_and_only_v2i():
    c = cases.MODEL_CASES["ragged_a4"]
    ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0, only_v2i=True)
    bevs, trans, na = cases.model_inputs("ragged_a4")
    with torch.no_grad():
        want = ref(bevs, trans, na, c["batch"])
    m = _product(ref, c["map_hw"], c["agents"], kd_flag=0, only_v2i=True)
    with torch.no_grad():
        got = m(bevs.cuda(), trans.cuda(), na.cuda(), c["batch"])
    assert isinstance(got, dict) and set(got) == {"loc", "cls"}
    for k in got:
        assert (got[k].cpu() - want[k]).abs().max().item() <= TOL


@pytest.mark.parametrize("math", MATHS)
def test_batch_position_independence_at_baseline_size(math):
    """configs[1] size (5 agents, batch 4, 256x256x13): a scene's result must not
    depend on which batch slot it sits in -- bitwise (fixed summation order)."""
    from disconet_amd import Config, DiscoNet
    from disconet_amd.synthetic import make_scene_batch
    torch.manual_seed(0)
    A, B = 5, 4
    m = DiscoNet(Config(), kd_flag=0, num_agent=A).eval().cuda()
    m.conv_math = math
    bevs, trans, na = make_scene_batch(B, A, 256, jitter_seed=3)
    with torch.no_grad():
        full = m(bevs.cuda(), trans.cuda(), na.cuda(), B)
        sel = torch.tensor([a * B + 2 for a in range(A)])
        one = m(bevs[sel].cuda(), trans[2:3].cuda(), na[2:3].cuda(), 1)
    assert torch.equal(full["cls"][sel.cuda()], one["cls"])
    assert torch.equal(full["loc"][sel.cuda()], one["loc"])
    assert torch.isfinite(full["cls"]).all() and torch.isfinite(full["loc"]).all()


def test_cpu_tensors_fail_loudly():
    from disconet_amd import Config, DiscoNet, _lib
    from disconet_amd.synthetic import make_scene_batch
    m = DiscoNet(Config(map_hw=128), kd_flag=0, num_agent=2).eval()
    bevs, trans, na = make_scene_batch(1, 2, 128)
    with pytest.raises(_lib.DnError):
        m(bevs, trans, na, 1)


@pytest.mark.parametrize("kw", [dict(layer=2), dict(layer=4), dict(layer=1), dict(compress_level=1),
                                dict(compress_level=2, only_v2i=True)])
def test_other_fusion_layers_and_compression(kw):
    """the constructor's other knobs: fusion at another pyramid level (C = 64 / 128 / 512
    maps at 64x64 / 32x32 / 8x8) and the 1x1 compress/decompress of the exchanged map"""
    c = cases.MODEL_CASES["ragged_a4"]
    ref = cases.ref_model(c["map_hw"], c["agents"], **kw)
    bevs, trans, na = cases.model_inputs("ragged_a4")
    with torch.no_grad():
        res, x8, x7, x6, x5, fused = ref(bevs, trans, na, c["batch"])
    m = _product(ref, c["map_hw"], c["agents"], **kw)
    got = _gpu_outputs(m, bevs, trans, na, c["batch"])
    for name, w in (("cls", res["cls"]), ("loc", res["loc"]), ("fused", fused), ("x5", x5)):
        assert got[name].shape == w.shape, (kw, name)
        err = (got[name] - w).abs().max().item()
        assert err <= TOL, "%s %s max abs err %.3e" % (kw, name, err)


@pytest.mark.parametrize("agents,live", [(1, [1]), (6, [6]), (6, [1]), (5, [2])])
def test_agent_count_edges(agents, live):
    """one agent (no neighbours at all), six agents (5 vehicles + RSU), and scenes whose
    live count is 1 or 2 of the padded slots"""
    from disconet_amd.synthetic import make_scene_batch
    ref = cases.ref_model(128, agents)
    bevs, trans, na = make_scene_batch(1, agents, 128, live=live, jitter_seed=11)
    with torch.no_grad():
        res, _, _, _, _, fused = ref(bevs, trans, na, 1)
    m = _product(ref, 128, agents)
    got = _gpu_outputs(m, bevs, trans, na, 1)
    for name, w in (("cls", res["cls"]), ("loc", res["loc"]), ("fused", fused)):
        err = (got[name] - w).abs().max().item()
        assert err <= TOL, "A=%d live=%s %s max abs err %.3e" % (agents, live, name, err)


def test_fused_1x1_layers_match_unfused():
    """split-f16: the launches that fold a 1x1 layer into the preceding conv give the
    same result as the separate launches (same arithmetic up to the order of one sum)"""
    from disconet_amd import Config, DiscoNet
    from disconet_amd.synthetic import make_scene_batch
    ref = cases.ref_model(128, 3)
    bevs, trans, na = make_scene_batch(2, 3, 128, jitter_seed=2)
    outs = []
    for fuse in (True, False, True, False):
        m = DiscoNet(Config(map_hw=128), kd_flag=0, num_agent=3).eval()
        m.conv_math = "sp" if len(outs) < 2 else "f16x3"
        m.load_state_dict(ref.state_dict())
        m.fuse_1x1 = fuse
        m.cuda()
        with torch.no_grad():
            outs.append(m(bevs.cuda(), trans.cuda(), na.cuda(), 2))
        assert ("heads_fused" in m._plan) == fuse
    for k in ("cls", "loc"):
        assert (outs[0][k] - outs[1][k]).abs().max().item() <= 2e-5
        assert (outs[2][k] - outs[3][k]).abs().max().item() <= 2e-5
        assert (outs[0][k] - outs[2][k]).abs().max().item() <= 2e-5    # the two split-f16 engines agree


@pytest.mark.parametrize("live", [[8], [5]])
def test_eight_agent_scenes(live):
    """BASELINE configs[4]: 8-agent scenes (the agent-sharded workload), all live and 8 slots with 5
    live -- the fusion kernels' agent limit (MAX_AGENTS = 8) is reached, whole model vs the oracle"""
    from disconet_amd.synthetic import make_scene_batch
    ref = cases.ref_model(128, 8)
    bevs, trans, na = make_scene_batch(1, 8, 128, live=live, jitter_seed=5)
    with torch.no_grad():
        res, _, _, _, x5, fused = ref(bevs, trans, na, 1)
    m = _product(ref, 128, 8)
    got = _gpu_outputs(m, bevs, trans, na, 1)
    for name, w in (("cls", res["cls"]), ("loc", res["loc"]), ("fused", fused), ("x5", x5)):
        err = (got[name] - w).abs().max().item()
        assert err <= TOL, "8 agents live=%s %s max abs err %.3e" % (live, name, err)


def test_split_planar_bevs_input_equals_dense_input():
    """forward() takes the voxel batch either as the reference's dense float32 tensor or already
    scattered into the conv engine's layout (ops.scatter_dense_sp): bit-identical results"""
    from disconet_amd import Config, DiscoNet, ops
    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices
    A, B, hw = 3, 2, 128
    torch.manual_seed(1)
    m = DiscoNet(Config(map_hw=hw), kd_flag=0, num_agent=A).eval().cuda()
    indices, offsets, _ = make_sparse_scene_batch(B, A, hw)
    indices, offsets = indices.cuda(), offsets.cuda()
    trans = make_trans_matrices(B, A, jitter_seed=2).cuda()
    na = torch.full((B, A), A, dtype=torch.int64).cuda()
    dims = (hw, hw, 13)
    with torch.no_grad():
        dense = m(ops.scatter_dense(indices, offsets, A * B, dims), trans, na, B)
        sp = m(ops.scatter_dense_sp(indices, offsets, A * B, dims), trans, na, B)
        hi = m(ops.scatter_dense_sp(indices, offsets, A * B, dims, hi_only=True), trans, na, B)
    assert torch.equal(dense["cls"], sp["cls"]) and torch.equal(dense["loc"], sp["loc"])
    # hi-only occupancy planes (half the bytes, 2 MFMAs per product in conv_pre_1): the dropped terms are exact zeros
    assert torch.equal(hi["cls"], sp["cls"]) and torch.equal(hi["loc"], sp["loc"])
    # occupancy bit grid (1/32 of the float32 bytes, expanded on its way into LDS): the same operands, the same MFMAs
    with torch.no_grad():
        bits = m(ops.scatter_dense_bits(indices, offsets, A * B, dims), trans, na, B)
    assert torch.equal(bits["cls"], sp["cls"]) and torch.equal(bits["loc"], sp["loc"])
    # ... and with the stem's two layers as two launches instead of one (DN_STEM_PAIR=0): the same bits
    import os
    os.environ["DN_STEM_PAIR"] = "0"
    try:
        with torch.no_grad():
            two = m(ops.scatter_dense_bits(indices, offsets, A * B, dims), trans, na, B)
    finally:
        del os.environ["DN_STEM_PAIR"]
    assert torch.equal(two["cls"], bits["cls"]) and torch.equal(two["loc"], bits["loc"])


def _trained_like(ref):
    """A network with the statistics of a trained checkpoint rather than of an initialiser: BatchNorm gains up to 4,
    a stage whose activations reach ~1e3 and a layer with |w| ~ 1e-4 that brings them back."""
    with torch.no_grad():
        e, d = ref.u_encoder, ref.decoder
        e.bn_pre_2.weight.mul_(4.0)
        e.bn1_1.weight.mul_(3.0)
        e.conv1_1.weight.mul_(60.0); e.conv1_1.bias.mul_(60.0)          # x1 activations up to ~1e3
        e.conv1_2.weight.mul_(1.5e-3); e.conv1_2.bias.mul_(0.1)          # |w| ~ 1e-4
        e.bn1_2.weight.mul_(2.0)
        d.bn7_1.weight.mul_(4.0); d.bn7_1.running_var.mul_(16.0)
        d.conv6_2.weight.mul_(20.0); d.bn6_2.running_var.mul_(400.0)
    return ref


@pytest.mark.parametrize("math", MATHS)
def test_model_with_trained_like_statistics(math):
    """The split-f16 format has the f16 exponent range: parity must hold away from O(1) data as well, and the
    range flags must stay clear below 2^14 (include/disconet_hip.h :: dn_sp_range_flags)."""
    from disconet_amd import ops
    c = cases.MODEL_CASES["cfg1_f1"]
    ref = _trained_like(cases.ref_model(c["map_hw"], c["agents"]))
    bevs, trans, na = cases.model_inputs("cfg1_f1")
    with torch.no_grad():
        res, x8, x7, x6, x5, fused = ref(bevs, trans, na, c["batch"])
        enc = ref.u_encoder(bevs.permute(0, 1, 4, 2, 3))
        big = torch.relu(ref.u_encoder.bn1_1(ref.u_encoder.conv1_1(enc[0])))     # conv1_1's output, the large stage
    assert 200.0 < big.abs().max().item() < 1.6e4, big.abs().max().item()
    assert ref.u_encoder.conv1_2.weight.abs().max().item() < 1e-3
    want = {"cls": res["cls"], "loc": res["loc"], "x8": x8, "x5": x5, "fused": fused}
    ops.sp_range_flags(reset=True)
    m = _product(ref, c["map_hw"], c["agents"], math=math)
    got = _gpu_outputs(m, bevs, trans, na, c["batch"])
    for name in want:
        scale = max(1.0, want[name].abs().max().item())
        err = (got[name] - want[name]).abs().max().item()
        assert err <= TOL * scale, "%s max abs err %.3e (scale %.3g)" % (name, err, scale)
    assert ops.sp_range_flags(reset=True) == 0


def test_range_guard_reports_a_clamped_activation(monkeypatch):
    """an activation beyond 65504 cannot be stored as an f16 hi/lo pair: the sticky flag is raised and the guard --
    on by default, asynchronous: no synchronisation in the forward -- refuses to go on at the next call; with
    DN_SP_CHECK=1 the offending forward itself raises, with DN_SP_CHECK=0 nothing does"""
    from disconet_amd import ops
    from disconet_amd._lib import DnError
    c = cases.MODEL_CASES["cfg1_f1"]
    ref = cases.ref_model(c["map_hw"], c["agents"])
    with torch.no_grad():
        ref.u_encoder.conv_pre_2.weight.mul_(3.0e5)
    bevs, trans, na = cases.model_inputs("cfg1_f1")
    m = _product(ref, c["map_hw"], c["agents"], math="sp")
    monkeypatch.delenv("DN_SP_CHECK", raising=False)
    ops.sp_range_flags(reset=True)
    _gpu_outputs(m, bevs, trans, na, c["batch"])            # clamps; the asynchronous read is in flight
    assert ops.sp_range_flags(reset=False) & 1              # (blocking read: the flag is sticky)
    with pytest.raises(DnError, match="clamped"):           # default mode: reported at the next call
        torch.cuda.synchronize()
        _gpu_outputs(m, bevs, trans, na, c["batch"])
    assert ops.sp_range_flags(reset=True) == 0              # reporting clears the sticky flags
    ops.drain_sp_range()                                    # nothing outstanding
    monkeypatch.setenv("DN_SP_CHECK", "1")
    with pytest.raises(DnError, match="clamped"):
        _gpu_outputs(m, bevs, trans, na, c["batch"])
    assert ops.sp_range_flags(reset=True) == 0
    monkeypatch.setenv("DN_SP_CHECK", "0")
    _gpu_outputs(m, bevs, trans, na, c["batch"])
    _gpu_outputs(m, bevs, trans, na, c["batch"])
    assert ops.sp_range_flags(reset=True) & 1


def test_non_finite_parameters_and_inputs_are_refused():
    """ReLU and the split's clamp turn a NaN into a finite number, so it has to be caught where it enters: a
    non-finite parameter when the plan is packed, a NaN input when it is split (range flag bit 2)"""
    from disconet_amd import ops
    from disconet_amd._lib import DnError
    c = cases.MODEL_CASES["cfg1_f1"]
    ref = cases.ref_model(c["map_hw"], c["agents"])
    with torch.no_grad():
        ref.u_encoder.conv1_1.bias[3] = float("nan")
    bevs, trans, na = cases.model_inputs("cfg1_f1")
    m = _product(ref, c["map_hw"], c["agents"], math="sp")
    with pytest.raises(DnError, match="non-finite values in u_encoder.conv1_1.bias"):
        _gpu_outputs(m, bevs, trans, na, c["batch"])
    ops.sp_range_flags(reset=True)
    x = torch.rand(2, 8, 8, 16)
    x[1, 3, 4, 5] = float("nan")
    ops.SpTensor.from_nhwc(x.cuda())
    assert ops.sp_range_flags(reset=True) & 4


def test_dataparallel_wrapper_on_one_device():
    """the reference tools wrap the model in nn.DataParallel (/root/reference/README.md:54-75): on ONE
    device the wrapper is a pass-through -- forward bitwise equal to the bare model, `module.`-prefixed checkpoints
    load through the wrapper -- and over several devices replication is refused with the one-process-per-GPU message"""
    import torch.nn as nn
    c = cases.MODEL_CASES["cfg1_f1"]
    ref = cases.ref_model(c["map_hw"], c["agents"])
    bevs, trans, na = cases.model_inputs("cfg1_f1")
    bare = _product(ref, c["map_hw"], c["agents"], math="sp")
    want = _gpu_outputs(bare, bevs, trans, na, c["batch"])
    from disconet_amd import Config, DiscoNet
    inner = DiscoNet(Config(map_hw=c["map_hw"]), kd_flag=1, num_agent=c["agents"]).eval().cuda()
    wrapped = nn.DataParallel(inner, device_ids=[0])
    sd = {"module." + k: v for k, v in ref.state_dict().items()}          # a released DataParallel checkpoint
    own = set(wrapped.state_dict().keys())
    wrapped.load_state_dict({k: v for k, v in sd.items() if k in own})    # the wrapper's strict load, module. keys
    got = _gpu_outputs(wrapped, bevs, trans, na, c["batch"])
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert all(k.startswith("module.") for k in wrapped.state_dict())
    with pytest.raises(RuntimeError, match="one process per GPU"):
        inner._replicate_for_data_parallel()
