"""Agent-parallel `--com disco` forward: the A agents of a scene are sharded
across the ranks of one node, one process per GPU, and the only exchange is an
all-gather of the layer-`layer` feature maps (RCCL over xGMI) -- the stand-in
for the V2X feature broadcast (BASELINE.json configs[4]).

    rank r owns agents [r*A/G, (r+1)*A/G):
      encode own bevs  ->  all_gather(x_layer)  ->  fuse own egos against all
      agents' maps  ->  decode own  ->  heads

Per sample and agent the exchanged map is 256x32x32 fp32 = 1 MiB, so the
collective is latency-, not bandwidth-bound on the 7 x 153 GB/s xGMI links; one
all-gather per forward, no other data-path collective.  The reference has no
counterpart (it keeps every agent's map in one tensor of one process).

The compute engine is injected: `HipEngine` (the product path, HIP kernels
only) or, in the CPU gloo tests, an oracle-backed engine.  Nothing here falls
back to the CPU on its own.
"""
import torch
import torch.distributed as dist


def agent_range(num_agent, world_size, rank):
    if num_agent % world_size != 0:
        raise ValueError("num_agent (%d) must be a multiple of the number of ranks (%d)"
                         % (num_agent, world_size))
    per = num_agent // world_size
    return rank * per, per


def local_bevs(bevs_all, num_agent, batch_size, world_size, rank):
    """Slice of an agent-major [A*B, ...] stack owned by `rank`."""
    first, count = agent_range(num_agent, world_size, rank)
    return bevs_all[first * batch_size:(first + count) * batch_size]


def all_gather_agent_major(x_local, group=None, out=None):
    """[A_local*B, ...] per rank -> [A*B, ...] agent-major on every rank.  Rank
    order == agent order, so the concatenation IS the agent-major layout.
    `out`: a preallocated result (a static buffer of a captured step)."""
    world = dist.get_world_size(group)
    if out is None:
        out = x_local.new_empty((x_local.shape[0] * world,) + tuple(x_local.shape[1:]))
    x_local = x_local.contiguous()
    # the form follows the group's backend (gloo has no *_into_tensor collectives); an RCCL error is never caught here
    if str(dist.get_backend(group)).lower() != "gloo":
        dist.all_gather_into_tensor(out, x_local, group=group)
    else:
        parts = list(out.chunk(world, 0))
        dist.all_gather(parts, x_local, group=group)
    return out


class HipEngine:
    """Product compute engine: the HIP kernels behind disconet_amd.DiscoNet."""

    def __init__(self, model):
        self.model = model
        self.layer = model.layer
        self.agent_num = model.agent_num

    def encode(self, bevs_local):
        return self.model.encode(bevs_local, self.model._get_plan())

    def fuse(self, feat_all, trans, num_agent, batch_size, ego_first, ego_count):
        m = self.model
        return m.fuse(feat_all, trans, num_agent, batch_size, m._get_plan(),
                      ego_first=ego_first, ego_count=ego_count, sp_out=m.conv_math == "sp")

    def decode_heads(self, enc_local):
        m = self.model
        P = m._get_plan()
        x8 = m.decode(enc_local, P)[0]
        return m.heads(x8, P)


def forward_agent_sharded(engine, bevs_local, trans_matrices, num_agent_tensor, batch_size=1,
                          group=None):
    """One rank's part of the forward.  bevs_local: this rank's agents,
    agent-major [A_local*B, 1, H, W, Z]; trans_matrices / num_agent_tensor are
    replicated.  Returns (result for the local agents, fused local maps)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = agent_range(engine.agent_num, world, rank)
    if bevs_local.shape[0] != count * batch_size:
        raise ValueError("rank %d expects %d local images, got %d"
                         % (rank, count * batch_size, bevs_local.shape[0]))
    enc = engine.encode(bevs_local)
    feat_all = all_gather_agent_major(enc[engine.layer], group)      # the V2X exchange
    dev = feat_all.device
    trans = trans_matrices.to(device=dev, dtype=torch.float32).contiguous()
    num_agent = num_agent_tensor[:, 0].to(device=dev, dtype=torch.int32).contiguous()
    fused = engine.fuse(feat_all, trans, num_agent, batch_size, first, count)
    enc = list(enc)
    enc[engine.layer] = fused
    return engine.decode_heads(enc), fused


class GraphedAgentStep:
    """One rank's agent-sharded forward as TWO captured hipGraphs around the exchange:

        graph A: dense rebuild of this rank's voxel lists + encoder        (make_bevs, engine.encode)
        exchange: RCCL all-gather of the layer-`layer` maps, eager, in stream order between the graphs
                  (the collective's kernel never runs beside one of ours: the compute stream waits for it)
        graph B: fusion of this rank's egos against all maps + decoder + heads

    Per-rank work at 8 GPUs is 4 images -- ~50 launches of 5-30 us each -- so eager launches from Python
    would cost more than the kernels; the two replays leave one host call per phase.

    `emulate_feat_all`: single-process emulation of one rank's share (bench.py --emulate-world): a tensor
    holding ALL agents' layer maps; the exchange becomes a device copy of the peers' maps into the gathered
    buffer (same bytes landing in HBM, no link latency) and this rank's own rows are overwritten by what
    graph A just computed.
    """

    def __init__(self, engine, make_bevs, trans_matrices, num_agent_tensor, batch_size, first, count,
                 group=None, emulate_feat_all=None, emulate_collective=False, range_guard=True):
        """emulate_collective (with emulate_feat_all, needs an initialised process group -- one rank is enough): the
        emulated exchange ALSO runs the real RCCL all_gather_into_tensor of this rank's maps, so that the collective's
        launch + kernel latency sits between the two graphs as it will on 8 ranks (link time does not).
        range_guard: graph B ends with the captured poll of the split-f16 range flags (graph.GraphedStep); False leaves the
        guard to the caller (bench.py reads the flags once after its timed regions)."""
        from .graph import GraphedStep
        self.engine, self.group, self.first, self.count, self.batch = engine, group, first, count, batch_size
        self.emulated = emulate_feat_all is not None
        self.emulate_collective = bool(emulate_collective and self.emulated and dist.is_available() and dist.is_initialized())
        self._own_gather = None
        layer = engine.layer
        self.graph_all = None
        with torch.no_grad():
            # (the range flags are sticky: graph B's captured poll covers graph A's launches too)
            self.graph_a = GraphedStep(lambda: engine.encode(make_bevs()), range_guard=False)
        self.enc = list(self.graph_a.outputs)
        x_local = self.enc[layer]
        world = 1 if (self.emulated or not (dist.is_available() and dist.is_initialized())) else dist.get_world_size(group)
        n_all = engine.agent_num * batch_size
        assert x_local.shape[0] * (engine.agent_num // count) == n_all
        self.feat_all = x_local.new_empty((n_all,) + tuple(x_local.shape[1:]))
        self.peers = emulate_feat_all
        self.world = world
        dev = x_local.device
        trans = trans_matrices.to(device=dev, dtype=torch.float32).contiguous()
        num_agent = num_agent_tensor[:, 0].to(device=dev, dtype=torch.int32).contiguous()
        self.exchange()                       # the gathered buffer holds real maps before graph B's warm-up

        def fuse_decode():
            fused = engine.fuse(self.feat_all, trans, num_agent, batch_size, first, count)
            enc = list(self.enc)
            enc[layer] = fused
            return engine.decode_heads(enc), fused

        with torch.no_grad():
            self.graph_b = GraphedStep(fuse_decode, range_guard=range_guard)

        def whole():      # the same step from the functions, for capture_one_graph(): its own encoder outputs and gathered buffer
            enc = list(engine.encode(make_bevs()))
            self.exchange(enc[layer])
            fused = engine.fuse(self.feat_all, trans, num_agent, batch_size, first, count)
            enc[layer] = fused
            return engine.decode_heads(enc), fused
        self._whole_fn = whole

    def exchange(self, x_local=None):
        x_local = self.enc[self.engine.layer] if x_local is None else x_local
        if self.emulated:
            if self.emulate_collective:                          # the collective's own launch + kernel, on one rank
                if self._own_gather is None:
                    self._own_gather = x_local.new_empty((x_local.shape[0] * dist.get_world_size(self.group),)
                                                         + tuple(x_local.shape[1:]))
                all_gather_agent_major(x_local, self.group, out=self._own_gather)
            self.feat_all.copy_(self.peers)                      # the peers' maps arriving
            lo = self.first * self.batch
            self.feat_all[lo:lo + x_local.shape[0]].copy_(x_local)
        elif self.count == self.engine.agent_num and not (dist.is_available() and dist.is_initialized()):
            self.feat_all.copy_(x_local)                         # one process owns every agent: nothing to exchange
        else:
            all_gather_agent_major(x_local, self.group, out=self.feat_all)

    def __call__(self):
        if self.graph_all is not None:
            return self.graph_all()
        self.graph_a()
        self.exchange()
        return self.graph_b()

    def capture_one_graph(self, range_guard=False):
        """OPT-IN measurement form (bench.py: DN_AGENT_ONE_GRAPH=1): graph A, the exchange -- the collective included -- and graph B
        captured as ONE hipGraph, one host call per step instead of three.  Whether RCCL's kernel can be captured depends on the
        runtime; on failure the three-launch form stays and the error is returned.  Not the default anywhere: a captured
        collective has never run with N > 1 ranks here."""
        from .graph import GraphedStep
        try:
            with torch.no_grad():
                # (graph A's outputs live in ITS private pool and graph B's closure reads those: the one-graph form does not
                # reuse either -- it captures the whole step again from the functions, with its own buffers)
                g = GraphedStep(self._whole_fn, range_guard=range_guard)
            self.graph_all = g
            return None
        except Exception as e:      # noqa: BLE001 -- a runtime that refuses the capture keeps the three-launch form
            torch.cuda.synchronize()
            self.graph_all = None
            return repr(e)


class AgentShard:
    """One rank's place in an AGENT-PARALLEL TRAINING step (SURVEY.md 8(e)(ii) + its backward, "Backward of (ii) is a
    reduce-scatter"): rank r owns the agents [first, first + count) of every scene and holds the replicated parameters.
    The collectives of a step, all through this object so that the HIP engine (train.TrainEngine(shard=...)) and the CPU
    oracle twin of the gloo tests (tests/oracle_engine.py) make the same calls:

      forward   sum_()                per BatchNorm layer: this rank's sum z, sum z^2 (float64) -> the batch's
                gather_rows()         the layer-`layer` maps of the own agents -> every agent's (the V2X exchange)
                gather_padded()       per-call statistics of the attention MLP's BatchNorms (running-stat replay order)
      backward  sum_()                per BatchNorm layer: sum g, sum g * zhat
                reduce_scatter_rows() d(loss terms of this rank) / d(every agent's map) -> d(loss) / d(own maps)
                sum_()                the flat parameter gradient (ranks hold disjoint terms of ONE loss: summed, not averaged)

    Backends: RCCL ("nccl") on the GPUs; gloo in the CPU tests (its missing reduce_scatter / *_into_tensor forms are
    rebuilt from all_reduce / all_gather here).  world == 1 (or no process group): every call is the identity."""

    def __init__(self, num_agent, group=None):
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.first, self.count = agent_range(num_agent, self.world, self.rank)
        self.num_agent = num_agent
        # the collective forms are chosen ONCE from the group's backend -- not by catching the other form's error: a genuine
        # RCCL failure on one rank (timeout, abort) must propagate, not turn into a collective its peers never posted
        self.backend = str(dist.get_backend(group)).lower() if on else "none"
        self.tensor_collectives = self.backend != "gloo"       # gloo: no reduce_scatter, no *_into_tensor forms

    # -- sums --------------------------------------------------------------------------------------------------
    def sum_(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def max_(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    # -- rows of an agent-major [A*B, ...] buffer ---------------------------------------------------------------
    def gather_rows(self, all_rows, lo, n):
        """all_rows[lo:lo + n] holds this rank's rows; afterwards all_rows holds every rank's (rank order == agent order).
        In place: the input is the rank's own slice of the output (NCCL's in-place all-gather)."""
        if self.world == 1:
            return all_rows
        assert all_rows.shape[0] == n * self.world and lo == self.rank * n
        if self.tensor_collectives:
            dist.all_gather_into_tensor(all_rows, all_rows[lo:lo + n], group=self.group)
        else:
            dist.all_gather(list(all_rows.chunk(self.world, 0)), all_rows[lo:lo + n].clone(), group=self.group)
        return all_rows

    def reduce_scatter_rows(self, all_rows, lo, n):
        """sum over the ranks of all_rows, of which this rank keeps rows [lo, lo + n) -> [n, ...]"""
        if self.world == 1:
            return all_rows[lo:lo + n]
        assert all_rows.shape[0] == n * self.world and lo == self.rank * n
        out = all_rows.new_empty((n,) + tuple(all_rows.shape[1:]))
        if self.tensor_collectives:
            dist.reduce_scatter_tensor(out, all_rows.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
        else:                                            # gloo: no reduce-scatter
            full = all_rows.clone()
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(full[lo:lo + n])
        return out

    def gather_padded(self, x_local, max_rows):
        """[n_r, C] per rank (n_r <= max_rows) -> [world, max_rows, C] on every rank (rows past n_r are zero)"""
        pad = x_local.new_zeros((max_rows,) + tuple(x_local.shape[1:]))
        pad[:x_local.shape[0]] = x_local
        if self.world == 1:
            return pad.unsqueeze(0)
        out = pad.new_empty((self.world * max_rows,) + tuple(pad.shape[1:]))
        if self.tensor_collectives:
            dist.all_gather_into_tensor(out, pad, group=self.group)
        else:
            dist.all_gather(list(out.chunk(self.world, 0)), pad, group=self.group)
        return out.view((self.world, max_rows) + tuple(pad.shape[1:]))

    def calls_in_reference_order(self, counts, n_local_calls):
        """counts[b][i] = attention-MLP calls of ego i in scene b (train.fusion_call_counts; replicated knowledge).  Every rank
        lists its calls scene by scene, ego by ego; the reference's order is scene, then ego over ALL agents -> for each
        scene the ranks' runs in rank order.  Returns {"total", "max_per_rank", "index"}: `index` (int32) lists, in the
        reference's call order, the rows of the [world * max_per_rank, C] buffer gather_padded() builds."""
        per = [[sum(row[r * self.count:(r + 1) * self.count]) for row in counts] for r in range(self.world)]   # [rank][scene]
        if sum(per[self.rank]) != n_local_calls:
            raise RuntimeError("agent shard: this rank lists %d attention calls, the scenes' agent counts give %d"
                               % (n_local_calls, sum(per[self.rank])))
        max_rows = max(1, max(sum(p) for p in per))
        index, offs = [], [0] * self.world
        for b in range(len(counts)):
            for r in range(self.world):
                index.extend(r * max_rows + offs[r] + k for k in range(per[r][b]))
                offs[r] += per[r][b]
        return {"total": len(index), "max_per_rank": max_rows, "index": torch.tensor(index, dtype=torch.int32)}


def average_gradients_(flat_grad):
    """Data-parallel training (upstream: nn.DataParallel / DDP around CoDetModule.step): every rank
    holds the gradient of its own scenes in ONE flat buffer (train.TrainEngine.flat_g, 7.9 M fp32 =
    31.5 MB), so the exchange is a single all-reduce -- RCCL rings over xGMI are per-link bound, one
    large message is the efficient shape -- followed by the 1 / world scale.  In place; returns the
    tensor.  No process group or a world of 1: nothing to do."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    flat_grad.mul_(1.0 / dist.get_world_size())
    return flat_grad
