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
