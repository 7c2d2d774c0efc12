"""torch-CPU restatement of the segmentation training step around the hot path.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no source / vectors in /root/reference).

Follows, from recollection (mounted mentions: /root/reference/README.md:15, :37, :49):
  upstream:coperception/utils/SegModule.py :: SegModule.step
      pred = model(bev, trans_matrices, num_agent)          (model.train())
      with `com` on: pred / labels of every image whose BEV is EMPTY (torch.sum(bev[i]) <= 1e-4: the padded slots of
      scenes with fewer than num_agent live agents) are dropped before the criterion
      loss = nn.CrossEntropyLoss()(pred, labels.long());  optimizer.zero_grad();  loss.backward();  optimizer.step()
What this module pins is the arithmetic the HIP forward / loss / reverse pass / Adam are checked against: autograd
of oracle/seg_ref.py :: SegDiscoNetRef in train() mode through F.cross_entropy, then torch.optim.Adam.
"""
import torch

from .seg_ref import seg_train_loss


def seg_train_step(model, optimizer, x, trans, num_agent, batch_size, labels):
    """one SegModule.step: returns the loss as a float; grads stay on the parameters for inspection."""
    model.train()
    out = model(x, trans, num_agent, batch_size)
    logits = out[0] if isinstance(out, tuple) else out
    loss = seg_train_loss(logits, labels, x)      # upstream's empty-image filter, then the criterion
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return float(loss.detach())
