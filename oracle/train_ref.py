"""torch-CPU restatement of the training step around the hot path.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no source / vectors in /root/reference).

Follows, from recollection (SURVEY.md §3 call stack, §8(f) next #1):
  upstream:coperception/utils/CoDetModule.py :: CoDetModule.step / loss_calculator
      result = model(bev_seq, trans_matrices, num_agent, batch_size)   (model.train())
      loss = sum(focal(cls, labels)) / N + sum(smooth_l1(loc, reg_targets)[mask]) / N,
      N = bev_seq.shape[0];  optimizer.zero_grad(); loss.backward(); optimizer.step()
  upstream:coperception/utils/loss.py :: SoftmaxFocalClassificationLoss (alpha 0.25, gamma 2),
      WeightedSmoothL1LocalizationLoss (sigma 3)      (second.pytorch lineage)
The exact weighting details of the pinned commit are unknown; what this module pins is the
arithmetic the HIP loss / backward / Adam kernels are checked against: autograd of the oracle
model in train() mode through these two losses, then torch.optim.Adam.
"""
import torch
import torch.nn.functional as F


def focal_loss(cls, labels, alpha=0.25, gamma=2.0):
    """cls [n, 2] logits, labels [n, 2] one-hot (an all-zero row is ignored) -> summed loss.
    FL = -alpha_t (1 - p_t)^gamma log p_t with alpha_t = alpha for the foreground class."""
    logp = F.log_softmax(cls, dim=-1)
    fg = labels[:, 1] > 0.5
    care = fg | (labels[:, 0] > 0.5)
    logq = torch.where(fg, logp[:, 1], logp[:, 0])
    q = logq.exp()
    a = torch.where(fg, torch.full_like(q, alpha), torch.full_like(q, 1.0 - alpha))
    return (-(a * (1.0 - q) ** gamma * logq) * care).sum()


def smooth_l1_loss(loc, targets, mask, sigma=3.0):
    """loc, targets [n, code]; mask [n] -> summed masked loss"""
    d = loc - targets
    ad = d.abs()
    s2 = sigma * sigma
    l = torch.where(ad <= 1.0 / s2, 0.5 * s2 * d * d, ad - 0.5 / s2)
    return (l.sum(-1) * mask).sum()


def det_loss(result, labels, reg_targets, reg_loss_mask, norm, alpha=0.25, gamma=2.0, sigma=3.0):
    """result: the model's {"cls": [N, H*W*A, 2], "loc": [N, H, W, A, 1, code]} ->
    (loss_cls, loss_loc), each already divided by `norm`."""
    cls = result["cls"].reshape(-1, 2)
    code = result["loc"].shape[-1]
    loc = result["loc"].reshape(-1, code)
    l_cls = focal_loss(cls, labels.reshape(-1, 2).to(cls.dtype), alpha, gamma) / norm
    l_loc = smooth_l1_loss(loc, reg_targets.reshape(-1, code).to(loc.dtype),
                           reg_loss_mask.reshape(-1).to(loc.dtype), sigma) / norm
    return l_cls, l_loc


def train_step(model, optimizer, bevs, trans, num_agent, batch_size, labels, reg_targets,
               reg_loss_mask):
    """one CoDetModule.step (no KD): returns (loss_cls, loss_loc) as floats; grads stay on the
    parameters for inspection."""
    model.train()
    out = model(bevs, trans, num_agent, batch_size)
    result = out[0] if isinstance(out, tuple) else out
    l_cls, l_loc = det_loss(result, labels, reg_targets, reg_loss_mask, norm=bevs.shape[0])
    optimizer.zero_grad()
    (l_cls + l_loc).backward()
    optimizer.step()
    return float(l_cls.detach()), float(l_loc.detach())
