"""hipGraph replay of the step (disconet_amd/graph.py): every replay must reproduce the eager step bit for bit.
Regression for profiles/r02_hazard_repro.txt part A: a hipMemsetAsync captured into the graph zeroed the voxel
grid on the first replay only and wrote a garbage pattern from the second replay on."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(math, hw=64, batch=2, agents=3):
    from disconet_amd import Config, DiscoNet, ops
    from disconet_amd.synthetic import make_sparse_scene_batch, make_trans_matrices, randomize_bn_stats
    torch.manual_seed(0)
    model = DiscoNet(Config(map_hw=hw), kd_flag=0, num_agent=agents)
    randomize_bn_stats(model)
    model.conv_math = math
    model.eval().cuda()
    indices, offsets, _ = make_sparse_scene_batch(batch, agents, hw)
    indices, offsets = indices.cuda(), offsets.cuda()
    trans = make_trans_matrices(batch, agents, jitter_seed=0).cuda()
    na = torch.full((batch, agents), agents, dtype=torch.int64).cuda()
    scatter = ops.scatter_dense_sp if math == "sp" else ops.scatter_dense

    def step():
        with torch.no_grad():
            return model(scatter(indices, offsets, agents * batch, (hw, hw, 13)), trans, na, batch)
    return step


@pytest.mark.parametrize("math", ["sp", "f32"])
def test_every_replay_equals_the_eager_step(math):
    from disconet_amd.graph import GraphedStep
    step = _setup(math)
    want = {k: v.clone() for k, v in step().items()}
    g = GraphedStep(step)                       # warm-up on a side stream, then capture
    junk = []
    for i in range(5):
        out = g()
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(out[k], want[k]), "replay %d: %s differs from the eager step" % (i, k)
        junk.append(torch.full((1 << (9 + 3 * i),), float("nan"), device="cuda"))   # allocator churn between replays


def test_two_graphs_on_two_streams_replay_identically():
    from disconet_amd.graph import GraphedStep
    step = _setup("sp")
    want = {k: v.clone() for k, v in step().items()}
    slots = [(GraphedStep(step), torch.cuda.Stream()) for _ in range(2)]
    kept = []
    for i in range(8):
        g, st = slots[i % 2]
        with torch.cuda.stream(st):
            kept.append({k: v.clone() for k, v in g().items()})
    torch.cuda.synchronize()
    for i, o in enumerate(kept):
        for k in want:
            assert torch.equal(o[k], want[k]), "replay %d (stream %d): %s differs" % (i, i % 2, k)


def test_a_replayed_step_reports_a_clamped_activation():
    """ops.check_sp_range skips itself under capture, so the replayed step -- the production mode -- never
    polled the range flags.  GraphedStep now ends its capture with the stream-ordered collect + a pinned copy and looks at the
    previous replay's word on every call: a clamp raises at the next call (or at drain()), range_guard=False leaves it alone."""
    from disconet_amd import Config, DiscoNet, ops
    from disconet_amd._lib import DnError
    from disconet_amd.graph import GraphedStep
    from disconet_amd.synthetic import make_scene_batch
    torch.manual_seed(0)
    hw, agents = 64, 2
    model = DiscoNet(Config(map_hw=hw), kd_flag=0, num_agent=agents).eval()
    with torch.no_grad():
        model.u_encoder.conv_pre_2.weight.mul_(3.0e6)       # activations far beyond 65504: the split clamps
    model.cuda()
    bevs, trans, na = make_scene_batch(1, agents, hw)
    bevs, trans, na = bevs.cuda(), trans.cuda(), na.cuda()

    def step():
        with torch.no_grad():
            return model(bevs, trans, na, 1)
    import os
    os.environ["DN_SP_CHECK"] = "0"                          # the eager warm-up inside GraphedStep must not raise first
    try:
        ops.sp_range_flags(reset=True)
        g = GraphedStep(step)
        ops.sp_range_flags(reset=True)                       # (the warm-up's own clamps)
        g()
        torch.cuda.synchronize()
        with pytest.raises(DnError, match="clamped"):
            g()                                              # the previous replay's word has landed
        assert ops.sp_range_flags(reset=True) == 0           # reporting cleared the sticky flags
        g()
        with pytest.raises(DnError, match="clamped"):
            g.drain()
        quiet = GraphedStep(step, range_guard=False)
        ops.sp_range_flags(reset=True)
        quiet(); quiet()
        torch.cuda.synchronize()
        assert ops.sp_range_flags(reset=True) & 1            # nothing raised, the flag is there for whoever reads it
    finally:
        del os.environ["DN_SP_CHECK"]
