"""N>1 path on CPU: world_size-2 gloo run of the agent-sharded forward
(disconet_amd.sharded) with the oracle injected as the compute engine, checked
against the un-sharded oracle forward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disconet_amd import sharded
        from tests.oracle_engine import OracleEngine
        c = cases.MODEL_CASES[case]
        ref = cases.ref_model(c["map_hw"], c["agents"])
        bevs, trans, na = cases.model_inputs(case)
        mine = sharded.local_bevs(bevs, c["agents"], c["batch"], world, rank)
        res, fused = sharded.forward_agent_sharded(OracleEngine(ref), mine, trans, na, c["batch"])
        q.put((rank, res["cls"].numpy(), res["loc"].numpy(), fused.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ragged_a4", "cfg1_f2"])
def test_agent_sharded_forward_matches_unsharded(case):
    world = 2
    c = cases.MODEL_CASES[case]
    want = cases.run_ref(case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, cls, loc, fused = q.get(timeout=300)
        got[rank] = (cls, loc, fused)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    per = c["agents"] // world * c["batch"]
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        cls, loc, fused = got[r]
        assert abs(cls - want["cls"][sl].numpy()).max() <= 1e-5
        assert abs(loc - want["loc"][sl].numpy()).max() <= 1e-5
        assert abs(fused - want["fused"][sl].numpy()).max() <= 1e-5


def test_agent_range_and_layout_helpers():
    from disconet_amd import sharded
    assert sharded.agent_range(8, 8, 3) == (3, 1)
    assert sharded.agent_range(6, 2, 1) == (3, 3)
    with pytest.raises(ValueError):
        sharded.agent_range(5, 2, 0)
    x = torch.arange(12).view(6, 2)             # A=3, B=2 agent-major
    assert torch.equal(sharded.local_bevs(x, 3, 2, 3, 1), x[2:4])


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disconet_amd import sharded
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(7919, generator=g)          # each rank's own gradient
        out = sharded.average_gradients_(flat)
        assert out.data_ptr() == flat.data_ptr()       # in place, one buffer, one collective
        q.put((rank, flat.numpy()))
    finally:
        dist.destroy_process_group()


def test_flat_gradient_average_is_the_mean_over_ranks():
    """the training step's data-parallel exchange (TrainEngine.allreduce_grads)"""
    from disconet_amd import sharded
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = sum(torch.randn(7919, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    for r in range(world):
        assert abs(got[r] - want.numpy()).max() < 1e-6
    t = torch.ones(5)
    assert sharded.average_gradients_(t) is t and float(t.sum()) == 5.0     # no process group: no-op


# ---- agent-parallel TRAINING: all-gather forward, reduce-scatter backward, BatchNorm sums all-reduced ---------------------
_ATRAIN = dict(map_hw=64, agents=4, batch=2, live=[3, 2], jitter=7, lr=0.02)


def _atrain_teacher():
    """float64 oracle teacher + the agents' holistic views (BASELINE configs[2]'s KD term under the agent split)"""
    from disconet_amd.synthetic import make_bevs
    from oracle.disconet_ref import RefConfig
    from oracle.teacher_ref import build_teacher
    c = _ATRAIN
    t = build_teacher(RefConfig(c["map_hw"])).double().eval()
    t.stpn.conv_pre_1.register_forward_pre_hook(lambda m, inp: (inp[0].double(),))
    return t, make_bevs(c["batch"], c["agents"], c["map_hw"], p=0.05), 1e3


def _atrain_setup(only_v2i=False, kd=False):
    """float64 oracle + inputs of the agent-parallel training case (every rank and the parent build the same)"""
    import torch.nn.functional as F
    from disconet_amd.synthetic import make_scene_batch, make_train_targets
    c = _ATRAIN
    ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=1 if kd else 0, only_v2i=only_v2i).double()
    ref.u_encoder.conv_pre_1.register_forward_pre_hook(lambda m, inp: (inp[0].double(),))
    bevs, trans, na = make_scene_batch(c["batch"], c["agents"], c["map_hw"], live=c["live"], jitter_seed=c["jitter"])
    labels, targets, mask = make_train_targets(bevs.shape[0], c["map_hw"], p_fg=0.02)
    if not getattr(F.grid_sample, "_dn_f64", False):      # the oracle builds float32 grids (as upstream): cast to the map's dtype
        orig = F.grid_sample
        def patched(inp, grid, **kw):
            return orig(inp, grid.to(inp.dtype), **kw)
        patched._dn_f64 = True
        F.grid_sample = patched
    return ref, (bevs, trans, na), (labels, targets, mask)


def _state(ref):
    return {k.replace(".bn.", "."): v.detach().clone() for k, v in ref.state_dict().items()}


def _atrain_worker(rank, world, port, q, only_v2i=False, kd=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disconet_amd import sharded
        from tests.oracle_engine import oracle_agent_sharded_train_step
        c = _ATRAIN
        ref, (bevs, trans, na), (labels, targets, mask) = _atrain_setup(only_v2i, kd)
        shard = sharded.AgentShard(c["agents"])
        mine = lambda t: sharded.local_bevs(t, c["agents"], c["batch"], world, rank)
        opt = torch.optim.SGD(ref.parameters(), lr=c["lr"])      # (why not Adam: see the test)
        kd_arg = None
        if kd:
            teacher, bevs_t, kd_weight = _atrain_teacher()
            kd_arg = (teacher, mine(bevs_t), kd_weight)          # the teacher sees THIS rank's agents' holistic views only
        losses = [oracle_agent_sharded_train_step(ref, shard, opt, mine(bevs), trans, na, c["batch"], mine(labels),
                                                  mine(targets), mine(mask), kd=kd_arg) for _ in range(2)]
        q.put((rank, losses, {k: v.numpy() for k, v in _state(ref).items()}))
    finally:
        dist.destroy_process_group()


def _unsharded_kd_step(ref, opt, teacher, bevs_t, kd_weight, bevs, trans, na, batch, labels, targets, mask):
    """CoDetModule.step with kd_flag = 1 on the un-sharded oracle (oracle/train_ref.py :: train_step + teacher_ref.kd_loss)"""
    from oracle.teacher_ref import kd_loss
    from oracle.train_ref import det_loss
    ref.train()
    res, x8, x7, x6, x5, fused = ref(bevs, trans, na, batch)
    with torch.no_grad():
        t8, t7, t6, t5, t3, t2 = teacher(bevs_t.double())
    l_cls, l_loc = det_loss(res, labels, targets, mask, norm=bevs.shape[0])
    l_kd = kd_loss((x5, x6, x7, fused), (t5, t6, t7, t3), kd_weight)
    opt.zero_grad()
    (l_cls + l_loc + l_kd).backward()
    opt.step()
    return float(l_cls.detach()), float(l_loc.detach()), float(l_kd.detach())


@pytest.mark.parametrize("only_v2i,kd", [(False, False), (True, False), (False, True)])
def test_agent_sharded_training_step_matches_unsharded_oracle(only_v2i, kd):
    """"Backward of (ii) is a reduce-scatter".  Two gloo ranks, two agents each of 4-agent scenes (one scene
    with 3 live agents, one with 2: rank 1 holds a padded agent in one scene and nothing live in the other), TWO consecutive
    steps of the oracle twin (tests/oracle_engine.py) through disconet_amd.sharded.AgentShard -- the collectives the HIP
    engine makes -- in float64: the losses, EVERY parameter after the update and EVERY BatchNorm buffer (the attention MLP's
    running statistics replayed in the reference's call order) equal the un-sharded oracle's on both ranks."""
    from oracle.train_ref import train_step
    world, c = 2, _ATRAIN
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_atrain_worker, args=(r, world, port, q, only_v2i, kd)) for r in range(world)]
    for p in procs:
        p.start()
    torch.set_num_threads(2)
    ref, (bevs, trans, na), (labels, targets, mask) = _atrain_setup(only_v2i, kd)
    # Plain SGD on purpose: a conv bias in front of a BatchNorm has a gradient that is exactly zero in mathematics and rounding
    # noise (~1e-17) in any backward; Adam normalises that noise into a +-lr step whose sign depends on the summation order,
    # which would force a loose comparison.  The collectives under test do not depend on the optimizer.
    opt = torch.optim.SGD(ref.parameters(), lr=c["lr"])
    if kd:      # kd_flag = 1 under the agent split (missing #5): the teacher is replicated, the KL means are global
        teacher, bevs_t, kd_weight = _atrain_teacher()
        want_losses = [_unsharded_kd_step(ref, opt, teacher, bevs_t, kd_weight, bevs, trans, na, c["batch"], labels, targets, mask)
                       for _ in range(2)]
        assert want_losses[0][2] > 1e-3 * (want_losses[0][0] + want_losses[0][1])      # the KD term is live in this case
    else:
        want_losses = [train_step(ref, opt, bevs, trans, na, c["batch"], labels, targets, mask) for _ in range(2)]
    want = _state(ref)
    got = {}
    for _ in range(world):
        rank, losses, state = q.get(timeout=600)
        got[rank] = (losses, state)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    moved = 0
    for r in range(world):
        losses, state = got[r]
        for a, b in zip(losses, want_losses):
            assert len(a) == len(b) == (3 if kd else 2)
            for av, bv in zip(a, b):
                assert abs(av - bv) <= 1e-9 * abs(bv), (r, losses, want_losses)
        assert set(state) == set(want)
        for k, w in want.items():
            g = torch.from_numpy(state[k])
            scale = float(w.abs().max()) if w.numel() else 0.0
            assert float((g - w).abs().max()) <= 1e-8 * max(scale, 1e-3), (r, k, float((g - w).abs().max()), scale)
    # the step did move the parameters of every part (encoder, fusion MLP, decoder, heads)
    init = _state(_atrain_setup(only_v2i, kd)[0])
    for key in ("u_encoder.conv1_1.weight", "pixel_weighted_fusion.conv1_1.weight", "decoder.conv5_1.weight",
                "classification.conv2.weight", "pixel_weighted_fusion.bn1_2.running_mean"):
        assert float((want[key] - init[key]).abs().max()) > 0, key


def _shard_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disconet_amd import sharded
        sh = sharded.AgentShard(4)                      # two agents per rank
        B, n = 3, 2 * 3                                 # rows per rank = count * B
        rows = torch.zeros(2 * n, 5)
        rows[rank * n:(rank + 1) * n] = torch.arange(n * 5, dtype=torch.float32).view(n, 5) + 100 * rank
        sh.gather_rows(rows, rank * n, n)               # in place: own slice of the gathered buffer
        contrib = torch.arange(2 * n * 5, dtype=torch.float32).view(2 * n, 5) * (rank + 1)
        mine = sh.reduce_scatter_rows(contrib, rank * n, n)
        pad = sh.gather_padded(torch.full((rank + 1, 2), float(rank + 1)), 3)
        t = sh.sum_(torch.tensor([1.0 + rank, 10.0], dtype=torch.float64))
        q.put((rank, rows.numpy(), mine.numpy(), pad.numpy(), t.numpy()))
    finally:
        dist.destroy_process_group()


def test_agent_shard_collectives_on_two_gloo_ranks():
    """AgentShard's row collectives where gloo has no native form (reduce_scatter, *_into_tensor): rebuilt from all_reduce /
    all_gather, checked value by value"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=120)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n = 6
    own = lambda r: torch.arange(n * 5, dtype=torch.float32).view(n, 5) + 100 * r
    full = torch.cat([own(0), own(1)])
    total = torch.arange(2 * n * 5, dtype=torch.float32).view(2 * n, 5) * 3          # rank 0's + rank 1's contribution
    for r in range(world):
        rows, mine, pad, t = got[r]
        assert torch.equal(torch.from_numpy(rows), full)
        assert torch.equal(torch.from_numpy(mine), total[r * n:(r + 1) * n])
        assert pad.shape == (2, 3, 2) and pad[0, 0, 0] == 1 and pad[0, 1, 0] == 0 and pad[1, 1, 1] == 2 and pad[1, 2, 0] == 0
        assert t.tolist() == [3.0, 20.0]


def test_agent_shard_helpers_without_a_process_group():
    from disconet_amd import sharded
    from disconet_amd.train import fusion_call_counts
    sh = sharded.AgentShard(6)
    assert (sh.world, sh.rank, sh.first, sh.count) == (1, 0, 0, 6)
    x = torch.arange(12.0).view(6, 2)
    assert sh.gather_rows(x, 0, 6) is x and torch.equal(sh.reduce_scatter_rows(x, 0, 6), x) and sh.sum_(x) is x
    assert tuple(sh.gather_padded(x[:4], 5).shape) == (1, 5, 2)
    counts = fusion_call_counts(4, False, torch.tensor([3, 2]), 2)
    assert counts == [[3, 3, 3, 0], [2, 2, 0, 0]]
    assert fusion_call_counts(4, True, torch.tensor([3, 1]), 2) == [[3, 2, 2, 0], [1, 0, 0, 0]]
    order = sh.calls_in_reference_order(counts, 13)
    assert order["total"] == 13 and order["index"].tolist() == list(range(13))
    # two ranks of two agents: per scene the ranks' runs in rank order (rank 1's rows live behind max_per_rank)
    two = sharded.AgentShard.__new__(sharded.AgentShard)
    two.world, two.rank, two.first, two.count, two.group, two.num_agent = 2, 1, 2, 2, None, 4
    o = two.calls_in_reference_order(counts, 3)
    assert o["max_per_rank"] == 10 and o["index"].tolist() == [0, 1, 2, 3, 4, 5, 10, 11, 12, 6, 7, 8, 9]
    with pytest.raises(RuntimeError):
        two.calls_in_reference_order(counts, 4)
