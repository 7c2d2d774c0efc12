"""What the reference's CoDetModule.predict_all does after the forward
(upstream:coperception/utils/CoDetModule.py, postprocess.py; SURVEY.md §8(f)
next #3): softmax + box decode for every anchor (one HIP kernel), candidate
selection, then rotated NMS on the host -- the reference runs NMS on the CPU too,
so this is its placement, not a fallback.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .ops import _need_gpu, _ptr, _stream


def make_anchors(config, map_hw=None, device="cuda"):
    """[H, W, A, 6] = (x, y, w, h, sin, cos): one anchor set per BEV cell centre."""
    h = w = map_hw or config.map_dims[0]
    vs = config.voxel_size
    xs = config.area_extents[0][0] + (torch.arange(h, dtype=torch.float64) + 0.5) * vs[0]
    ys = config.area_extents[1][0] + (torch.arange(w, dtype=torch.float64) + 0.5) * vs[1]
    a = torch.as_tensor(np.asarray(config.anchor_size), dtype=torch.float64)
    out = torch.zeros((h, w, a.shape[0], 6), dtype=torch.float64)
    out[..., 0] = xs[:, None, None]
    out[..., 1] = ys[None, :, None]
    out[..., 2] = a[None, None, :, 0]
    out[..., 3] = a[None, None, :, 1]
    out[..., 4] = torch.sin(a[None, None, :, 2])
    out[..., 5] = torch.cos(a[None, None, :, 2])
    return out.to(torch.float32).to(device)


def decode(result, anchors):
    """result = {"cls": [N, H*W*A, 2], "loc": [N, H, W, A, 1, 6]} from DiscoNet.forward ->
    (scores [N, H*W*A], boxes [N, H*W*A, 6]) on the GPU."""
    cls, loc = result["cls"], result["loc"]
    _need_gpu(cls, loc, anchors)
    n, apl = cls.shape[0], cls.shape[1]
    cls = cls.contiguous()
    loc = loc.reshape(n, apl, 6).contiguous()
    anchors = anchors.reshape(apl, 6).contiguous()
    scores = torch.empty((n, apl), dtype=torch.float32, device=cls.device)
    boxes = torch.empty((n, apl, 6), dtype=torch.float32, device=cls.device)
    _lib.check(_lib.load().dn_decode_boxes(_ptr(cls), _ptr(loc), _ptr(anchors), n, apl, _ptr(scores),
                                           _ptr(boxes), _stream()), "dn_decode_boxes")
    return scores, boxes


# ---------------------------------------------------------------------------
# host-side rotated NMS (vectorised over the candidates still alive)
# ---------------------------------------------------------------------------
def _corners(b):
    """[K, 6] boxes -> [K, 4, 2] corners, counter-clockwise."""
    n = np.maximum(np.hypot(b[:, 4], b[:, 5]), 1e-12)
    s, c = b[:, 4] / n, b[:, 5] / n
    dx, dy = b[:, 2] / 2.0, b[:, 3] / 2.0
    loc = np.stack([np.stack([-dx, -dy], -1), np.stack([dx, -dy], -1),
                    np.stack([dx, dy], -1), np.stack([-dx, dy], -1)], 1)          # [K, 4, 2]
    x = loc[..., 0] * c[:, None] - loc[..., 1] * s[:, None] + b[:, None, 0]
    y = loc[..., 0] * s[:, None] + loc[..., 1] * c[:, None] + b[:, None, 1]
    return np.stack([x, y], -1)


def _intersection_area(ca, cb):
    """area of the intersection of two convex quads (Sutherland-Hodgman)."""
    poly = [ca[i] for i in range(4)]
    for i in range(4):
        a, b = cb[i], cb[(i + 1) % 4]
        nxt = []
        for k in range(len(poly)):
            p, q = poly[k], poly[(k + 1) % len(poly)]
            sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
            sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
            if sp >= 0:
                nxt.append(p)
            if sp * sq < 0:
                nxt.append(p + (sp / (sp - sq)) * (q - p))
        poly = nxt
        if len(poly) < 3:
            return 0.0
    p = np.asarray(poly)
    return 0.5 * abs(np.dot(p[:, 0], np.roll(p[:, 1], -1)) - np.dot(p[:, 1], np.roll(p[:, 0], -1)))


def nms_rotated(boxes, scores, iou_thr=0.01):
    """boxes [K, 6], scores [K] (numpy) -> indices kept, best score first; ties broken by
    the lower index."""
    order = np.argsort(-scores, kind="stable")
    boxes = np.asarray(boxes, dtype=np.float64)[order]
    corners = _corners(boxes)
    radius = 0.5 * np.hypot(boxes[:, 2], boxes[:, 3])
    area = boxes[:, 2] * boxes[:, 3]
    alive = np.ones(len(order), dtype=bool)
    keep = []
    for i in range(len(order)):
        if not alive[i]:
            continue
        keep.append(int(order[i]))
        rest = np.nonzero(alive[i + 1:])[0] + i + 1
        if len(rest) == 0:
            continue
        near = rest[np.hypot(boxes[rest, 0] - boxes[i, 0], boxes[rest, 1] - boxes[i, 1])
                    < radius[rest] + radius[i]]
        for j in near:
            inter = _intersection_area(corners[i], corners[j])
            union = area[i] + area[j] - inter
            if union > 0 and inter / union > iou_thr:
                alive[j] = False
    return np.asarray(keep, dtype=np.int64)


def predict_all(model, anchors, bevs, trans_matrices, num_agent_tensor, batch_size=1,
                pre_nms_top_k=300, iou_thr=0.01, score_thr=None):
    """Per-image detections [(boxes [K, 6], scores [K])] for the agent-major batch: forward
    (HIP), decode (HIP), top-k by score (torch on the GPU), rotated NMS (host)."""
    with torch.no_grad():
        out = model(bevs, trans_matrices, num_agent_tensor, batch_size)
    result = out[0] if isinstance(out, tuple) else out
    scores, boxes = decode(result, anchors)
    dets = []
    for i in range(scores.shape[0]):
        s, b = scores[i], boxes[i]
        if score_thr is not None:
            s = torch.where(s > score_thr, s, torch.full_like(s, -1.0))
        k = min(pre_nms_top_k, s.numel())
        # stable: descending score, ascending index among equals
        top = torch.sort(s, descending=True, stable=True)[1][:k]
        sb, bb = s[top].cpu().numpy(), b[top].cpu().numpy()
        valid = sb >= 0
        sb, bb = sb[valid], bb[valid]
        keep = nms_rotated(bb, sb, iou_thr)
        dets.append((bb[keep], sb[keep]))
    return dets
