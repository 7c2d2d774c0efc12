"""Agent-parallel TRAINING on the HIP path (train.TrainEngine(shard=...), disconet_amd.sharded.AgentShard; SURVEY.md 8(e):
"Backward of (ii) is a reduce-scatter").  The collectives' structure is pinned on the CPU by tests/test_sharded_gloo.py (oracle
twin, float64, two gloo ranks == the un-sharded oracle step); here the HIP engine makes the same calls:

  * one process, a shard of world size 1: bit for bit the plain CoDetModule.step (the plumbing changes nothing);
  * TWO processes sharing the one GPU of the test box (gloo backend on CUDA tensors: RCCL refuses two ranks on one device),
    two agents each of 4-agent scenes with padded agents: losses, the summed gradient of EVERY parameter, BatchNorm running
    statistics (the attention MLP's in the reference's call order) and the parameters after Adam against the oracle, under
    the criteria of tests/test_gpu_train_step.py (float64 oracle run as the truth), and both ranks end with the same bits.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cases
from tests.test_gpu_train_step import _assert_grads, _fp64_grads, _setup

pytestmark = pytest.mark.gpu


def _data(inputs, targets, sl=slice(None)):
    bevs, trans, na = inputs
    labels, reg, mask = targets
    return {"bev_seq": bevs[sl].cuda(), "trans_matrices": trans.cuda(), "num_agent": na.cuda(),
            "labels": labels[sl].cuda(), "reg_targets": reg[sl].cuda(), "reg_loss_mask": mask[sl].cuda()}


def test_shard_of_world_one_is_the_plain_step():
    from disconet_amd import CoDetModule, sharded
    outs = []
    for use_shard in (False, True):
        c, ref, model, inputs, targets = _setup("ragged_a4", "f32")
        mod = CoDetModule(model, lr=1e-3, shard=sharded.AgentShard(c["agents"]) if use_shard else None)
        out = mod.step(_data(inputs, targets), c["batch"])
        outs.append((out, mod.engine.flat_g.clone(), mod.engine.flat_p.clone(),
                     {k: v.clone() for k, v in model.named_buffers()}))
    for k in outs[0][0]:      # (the reported loss scalars go through one f64 atomic per workgroup: equal to rounding, not to the bit)
        assert abs(outs[0][0][k] - outs[1][0][k]) <= 1e-12 * abs(outs[0][0][k]), k
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    for k, v in outs[0][3].items():
        assert torch.equal(v, outs[1][3][k]), k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, math, q, kw=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disconet_amd import CoDetModule, sharded
        c, ref, model, inputs, targets = _setup(case, math, **(kw or {}))
        shard = sharded.AgentShard(c["agents"])
        per = shard.count * c["batch"]
        sl = slice(shard.first * c["batch"], shard.first * c["batch"] + per)
        mod = CoDetModule(model, lr=1e-3, shard=shard)
        out = mod.step(_data(inputs, targets, sl), c["batch"])
        eng = mod.engine
        # (numpy: pickled by value -- torch tensors travel through the queue as shared-memory handles of a process that exits)
        grads = {n: eng.g(p).cpu().numpy() for n, p in model.named_parameters()}
        params = {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}
        bufs = {n: b.cpu().numpy() for n, b in model.named_buffers()}
        out2 = mod.step(_data(inputs, targets, sl), c["batch"])
        q.put((rank, out, out2, grads, params, bufs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("math,kw", [("f32", {}), ("f16x3", {}),
                                     ("f32", dict(only_v2i=True, compress_level=1))])      # the 1x1 compress / decompress pair around the exchange
def test_two_ranks_agent_parallel_step_matches_oracle(math, kw, monkeypatch):
    from oracle.train_ref import train_step
    case, world = "ragged_a4", 2              # 4 agent slots, batch 2, live agents [3, 2]: rank 1 = agents 2, 3
    c, ref, _, inputs, targets = _setup(case, math, **kw)
    bevs, trans, na = inputs
    g64 = _fp64_grads(ref, inputs, targets, c["batch"], monkeypatch)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    l_ref = train_step(ref, opt, bevs, trans, na, c["batch"], *targets)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, math, q, kw)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=600)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ref_named = dict(ref.named_parameters())
    ref_buf = dict(ref.named_buffers())
    gmax = max(float(g.abs().max()) for g in g64.values())
    for r in range(world):
        out, out2, grads, params, bufs = got[r]
        grads, params, bufs = ({k: torch.from_numpy(v) for k, v in d.items()} for d in (grads, params, bufs))
        # losses of the WHOLE scenes, on every rank
        assert abs(out["cls_loss"] - l_ref[0]) < 2e-5 * abs(l_ref[0]), (r, out, l_ref)
        assert abs(out["loc_loss"] - l_ref[1]) < 2e-5 * abs(l_ref[1]), (r, out, l_ref)
        assert out2["loss"] < out["loss"]
        # the summed gradient of every parameter against the float64 oracle run (criteria of test_gpu_train_step.py)
        rows = {}
        for name, t in g64.items():
            den = max(float(t.abs().max()), 1e-4 * gmax)
            g = grads[name].double()
            e_hip = float((g - t).abs().max()) / den
            e_ora = float((ref_named[name].grad.double() - t).abs().max()) / den
            cos = float((g * t).sum() / (g.norm() * t.norm()).clamp_min(1e-300))
            rows[name] = (e_hip, e_ora, cos, float(t.abs().max()) > 1e-4 * gmax)
        _assert_grads(rows)
        # BatchNorm running statistics: the batch's (all-reduced sums), the attention MLP's in the reference's call order
        for name, b in bufs.items():
            want = ref_buf[name]
            if name.endswith("num_batches_tracked"):
                assert int(b) == int(want), (r, name)
            else:
                assert float((b - want).abs().max()) < 1e-4 * max(float(want.abs().max()), 1.0), (r, name)
        # parameters after Adam where the gradient is above its noise floor
        for name, p in params.items():
            gr = ref_named[name].grad
            if float(gr.abs().max()) < 1e-4 * gmax:
                continue
            sel = gr.abs() > 0.1 * gr.abs().max()
            assert float((p - ref_named[name].detach())[sel].abs().max()) < 2e-5, (r, name)
    # the ranks hold the same replica: identical bits after the step (same summed gradient, same Adam)
    import numpy as np
    for name in got[0][3]:
        assert np.array_equal(got[0][3][name], got[1][3][name]), name
    for name in got[0][2]:
        assert np.array_equal(got[0][2][name], got[1][2][name]), name
