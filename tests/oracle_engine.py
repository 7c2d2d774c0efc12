"""TEST-ONLY compute engine for disconet_amd.sharded: the CPU oracle cut at the
same three seams (encode / fuse own egos / decode+heads) so the N>1 host logic
(sharding, agent-major all-gather, ego ranges) can run under gloo without a GPU."""
import torch

from oracle.disconet_ref import LAYER_CHANNEL, feature_transformation


class OracleEngine:
    def __init__(self, ref):
        self.ref = ref
        self.layer = ref.layer
        self.agent_num = ref.agent_num

    def encode(self, bevs_local):
        with torch.no_grad():
            return self.ref.u_encoder(bevs_local.permute(0, 1, 4, 2, 3))

    def fuse(self, feat_all, trans, num_agent, batch_size, ego_first, ego_count):
        ref, A, B = self.ref, self.agent_num, batch_size
        com = ref.build_local_communication_matrix(feat_all, B)           # [B, A, C, H, W]
        size = (1,) + tuple(feat_all.shape[1:])
        out = torch.empty((ego_count * B,) + tuple(feat_all.shape[1:]))
        with torch.no_grad():
            for b in range(B):
                n = int(num_agent[b])
                for il in range(ego_count):
                    i = ego_first + il
                    if i >= n:
                        out[il * B + b] = com[b, i]
                        continue
                    nbrs = [com[b, i]]
                    for j in range(n):
                        if j != i and not (ref.only_v2i and i != 0 and j != 0):
                            nbrs.append(feature_transformation(b, j, com, trans[b, i], size))
                    e = [torch.exp(torch.squeeze(ref.pixel_weighted_fusion(
                        torch.cat([com[b, i], nb], 0).unsqueeze(0)))) for nb in nbrs]
                    s = sum(e)
                    out[il * B + b] = sum((ek / s) * nb for ek, nb in zip(e, nbrs))
        return out

    def decode_heads(self, enc_local):
        ref = self.ref
        with torch.no_grad():
            x = ref.decoder(*enc_local, 1, kd_flag=False)[0]
            cls = ref.classification(x).permute(0, 2, 3, 1).contiguous()
            loc = ref.regression(x).permute(0, 2, 3, 1).contiguous()
        return {"cls": cls.view(cls.shape[0], -1, ref.category_num),
                "loc": loc.view(-1, loc.size(1), loc.size(2), ref.anchor_num_per_loc,
                                ref.out_seq_len, ref.box_code_size)}


# ---------------------------------------------------------------------------------------------------------------
# Agent-parallel TRAINING, oracle twin (TEST ONLY): torch autograd of the oracle model with the SAME collectives, through
# the same disconet_amd.sharded.AgentShard object, as the HIP engine (train.TrainEngine(shard=...)) makes --
#   per BatchNorm layer (encoder / decoder / heads): all-reduce of (sum z, sum z^2) forward, of (sum g, sum g zhat) backward;
#   all-gather of the layer-`layer` maps forward, reduce-scatter of their gradient backward;
#   per-call statistics of the attention MLP's BatchNorms gathered and replayed in the reference's call order;
#   one summed all-reduce of the parameter gradients.
# tests/test_sharded_gloo.py runs it on two gloo ranks in float64 and compares every parameter and buffer after the step
# with the un-sharded oracle step.
# ---------------------------------------------------------------------------------------------------------------
import torch.nn as nn
import torch.nn.functional as F


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, shard):
        C = x.shape[1]
        dims = [d for d in range(x.dim()) if d != 1]
        shape = [1, C] + [1] * (x.dim() - 2)
        s = torch.stack([x.sum(dims), (x * x).sum(dims)]).double()
        shard.sum_(s)                                              # this rank's images -> the whole batch's
        N = (x.numel() // C) * shard.world
        mean = s[0] / N
        var = (s[1] / N - mean * mean).clamp_min(0.0)
        rstd = (var + eps).rsqrt().to(x.dtype)
        xhat = (x - mean.to(x.dtype).view(shape)) * rstd.view(shape)
        ctx.save_for_backward(xhat, weight, rstd)
        ctx.shard, ctx.N, ctx.dims, ctx.shape = shard, N, dims, shape
        ctx.mark_non_differentiable(mean, var)
        return xhat * weight.view(shape) + bias.view(shape), mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        xhat, weight, rstd = ctx.saved_tensors
        s = torch.stack([g.sum(ctx.dims), (g * xhat).sum(ctx.dims)]).double()
        dbeta, dgamma = s[0].to(g.dtype).clone(), s[1].to(g.dtype).clone()      # THIS rank's share (summed with the gradients)
        ctx.shard.sum_(s)
        m1, m2 = (s[0] / ctx.N).to(g.dtype).view(ctx.shape), (s[1] / ctx.N).to(g.dtype).view(ctx.shape)
        dx = weight.view(ctx.shape) * rstd.view(ctx.shape) * (g - m1 - xhat * m2)
        return dx, dgamma, dbeta, None, None


class _SyncBN(nn.Module):
    """stands in for one nn.BatchNorm2d / 3d of the oracle in train() mode; shares its parameters and buffers"""

    def __init__(self, bn, shard):
        super().__init__()
        self.bn, self.shard = bn, shard

    def forward(self, x):
        bn = self.bn
        y, mean, var = _SyncBNFn.apply(x, bn.weight, bn.bias, bn.eps, self.shard)
        with torch.no_grad():
            N = (x.numel() // x.shape[1]) * self.shard.world
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(m * mean.to(bn.running_mean.dtype))
            bn.running_var.mul_(1 - m).add_(m * (var * (N / (N - 1.0))).to(bn.running_var.dtype))
            bn.num_batches_tracked += 1
        return y


def _swap_bn(module, shard):
    for name, child in list(module.named_children()):
        if isinstance(child, (nn.BatchNorm2d, nn.BatchNorm3d)):
            setattr(module, name, _SyncBN(child, shard))
        else:
            _swap_bn(child, shard)


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, shard, lo):
        n = x_local.shape[0]
        out = x_local.new_empty((n * shard.world,) + tuple(x_local.shape[1:]))
        out[lo:lo + n] = x_local
        shard.gather_rows(out, lo, n)
        ctx.shard, ctx.lo, ctx.n = shard, lo, n
        return out

    @staticmethod
    def backward(ctx, g):          # the backward of the all-gather is a reduce-scatter
        return ctx.shard.reduce_scatter_rows(g.contiguous(), ctx.lo, ctx.n), None, None


def oracle_agent_sharded_train_step(ref, shard, optimizer, bevs_local, trans, num_agent_tensor, batch_size, labels,
                                    reg_targets, reg_loss_mask, kd=None):
    """One CoDetModule.step of the oracle `ref` (already .double() or float) on THIS rank's agents.  ref is modified in
    place (its BatchNorms of encoder / decoder / heads are wrapped once).  Returns (loss_cls, loss_loc) of the whole scenes --
    (loss_cls, loss_loc, loss_kd) with kd = (teacher, bevs_teacher_local, kd_weight): the frozen teacher (replicated, no
    communication) on this rank's agents' holistic views, the KL means over the GLOBAL row count."""
    from oracle.train_ref import det_loss
    from disconet_amd.train import fusion_call_counts
    A, B = ref.agent_num, batch_size
    if not getattr(ref, "_sync_bn_swapped", False):
        for part in (ref.u_encoder, ref.decoder, ref.classification, ref.regression):
            _swap_bn(part, shard)
        ref._sync_bn_swapped = True
    ref.train()
    f = ref.pixel_weighted_fusion
    mlp_bns = (f.bn1_1, f.bn1_2, f.bn1_3)
    # the MLP's BatchNorms see one pair per call: statistics are rank-local, the RUNNING statistics are replayed afterwards
    calls = [[] for _ in mlp_bns]
    saved = [(bn.momentum, int(bn.num_batches_tracked)) for bn in mlp_bns]
    hooks = []
    for k, bn in enumerate(mlp_bns):
        bn.momentum = 0.0
        hooks.append(bn.register_forward_pre_hook(
            lambda mod, inp, k=k: calls[k].append((inp[0].detach().mean((0, 2, 3)), inp[0].detach().var((0, 2, 3), unbiased=False)))))
    try:
        enc = list(ref.u_encoder(bevs_local.permute(0, 1, 4, 2, 3)))
        lo = shard.first * B
        feat_all = _GatherRowsFn.apply(enc[ref.layer], shard, lo)
        com = ref.build_local_communication_matrix(feat_all, B)            # [B, A, C, H, W]
        size = (1,) + tuple(feat_all.shape[1:])
        fused = [None] * (shard.count * B)
        for b in range(B):
            n = int(num_agent_tensor[b, 0])
            for il in range(shard.count):
                i = shard.first + il
                if i >= n:
                    fused[il * B + b] = com[b, i]
                    continue
                nbrs = [com[b, i]]
                for j in range(n):
                    if j != i and not (ref.only_v2i and i != 0 and j != 0):
                        nbrs.append(feature_transformation(b, j, com, trans[b, i], size))
                e = [torch.exp(torch.squeeze(f(torch.cat([com[b, i], nb], 0).unsqueeze(0)))) for nb in nbrs]
                s = sum(e)
                fused[il * B + b] = sum((ek / s) * nb for ek, nb in zip(e, nbrs))
        enc[ref.layer] = torch.stack(fused, 0)
        dec = ref.decoder(*enc, B, kd_flag=kd is not None)
        x = dec[0]
        cls = ref.classification(x).permute(0, 2, 3, 1).contiguous()
        loc = ref.regression(x).permute(0, 2, 3, 1).contiguous()
        result = {"cls": cls.view(cls.shape[0], -1, ref.category_num),
                  "loc": loc.view(-1, loc.size(1), loc.size(2), ref.anchor_num_per_loc, ref.out_seq_len, ref.box_code_size)}
        l_cls, l_loc = det_loss(result, labels, reg_targets, reg_loss_mask, norm=A * B)      # the reference's N: every image
        l_kd = None
        if kd is not None:
            import torch.nn.functional as F
            teacher, bevs_t, kd_weight = kd
            with torch.no_grad():
                t8, t7, t6, t5, t3, t2 = teacher(bevs_t.to(x.dtype))
            x8, x7, x6, x5 = dec
            l_kd = 0.0
            for s_map, t_map in ((x5, t5), (x6, t6), (x7, t7), (enc[ref.layer], t3)):
                C = s_map.shape[1]
                s_rows = s_map.permute(0, 2, 3, 1).reshape(-1, C)
                t_rows = t_map.permute(0, 2, 3, 1).reshape(-1, C)
                # KLDivLoss(reduction="mean") of the un-sharded step divides by (all rows) * C: this rank's rows are 1 / world of them
                l_kd = l_kd + F.kl_div(F.log_softmax(s_rows, 1), F.softmax(t_rows, 1), reduction="sum") / (
                    s_rows.shape[0] * shard.world * C)
            l_kd = kd_weight * l_kd
        optimizer.zero_grad()
        (l_cls + l_loc + (l_kd if l_kd is not None else 0.0)).backward()
    finally:
        for h in hooks:
            h.remove()
        for bn, (mom, _) in zip(mlp_bns, saved):
            bn.momentum = mom
    # running statistics of the MLP's BatchNorms: every rank replays EVERY call, in the reference's order
    order = shard.calls_in_reference_order(fusion_call_counts(A, ref.only_v2i, num_agent_tensor[:, 0], B), len(calls[0]))
    hw = feat_all.shape[-1] * feat_all.shape[-2]
    with torch.no_grad():
        for bn, lst, (mom, tracked) in zip(mlp_bns, calls, saved):
            C = bn.num_features
            mean = torch.stack([m for m, _ in lst]) if lst else torch.zeros(0, C, dtype=bn.running_mean.dtype)
            var = torch.stack([v for _, v in lst]) if lst else torch.zeros(0, C, dtype=bn.running_mean.dtype)
            allm = shard.gather_padded(mean, order["max_per_rank"]).reshape(-1, C)
            allv = shard.gather_padded(var, order["max_per_rank"]).reshape(-1, C)
            for row in order["index"].tolist():
                bn.running_mean.mul_(1 - mom).add_(mom * allm[row])
                bn.running_var.mul_(1 - mom).add_(mom * allv[row] * (hw / (hw - 1.0)))
            bn.num_batches_tracked.fill_(tracked + order["total"])
    # ONE summed all-reduce of the gradients (each rank holds its own terms of the one loss)
    params = [p for p in ref.parameters()]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    shard.sum_(flat)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    optimizer.step()
    losses = torch.stack([l_cls.detach(), l_loc.detach()] + ([l_kd.detach()] if l_kd is not None else [])).double()
    shard.sum_(losses)
    return tuple(float(v) for v in losses)
