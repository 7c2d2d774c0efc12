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


def all_gather_agent_major(x_local, group=None):
    """[A_local*B, ...] per rank -> [A*B, ...] agent-major on every rank.  Rank
    order == agent order, so the concatenation IS the agent-major layout."""
    world = dist.get_world_size(group)
    out = x_local.new_empty((x_local.shape[0] * world,) + tuple(x_local.shape[1:]))
    x_local = x_local.contiguous()
    try:
        dist.all_gather_into_tensor(out, x_local, group=group)
    except (RuntimeError, NotImplementedError):
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
                      ego_first=ego_first, ego_count=ego_count)

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
