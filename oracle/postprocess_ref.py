"""numpy restatement of the detection post-processing that follows the hot path.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no source / vectors in /root/reference).

Follows, from recollection (Appx A.1):
  upstream:coperception/utils/obj_util.py   -- anchor layout, box code (x, y, w, h, sin, cos)
  upstream:coperception/utils/postprocess.py -- softmax, box decode vs anchors, rotated NMS
  upstream:coperception/utils/mean_ap.py    -- mmdetection-style AP at IoU 0.5 / 0.7
                                               (credited at /root/reference/README.md:105)
The exact anchor sizes and thresholds of the pinned commit are unknown; what this
module pins is the ARITHMETIC (decode formula, rotated IoU, greedy NMS, area-under-
curve AP), which is what the HIP decode kernel and the mAP-parity test are checked
against.
"""
import math

import numpy as np


def make_anchors(config, map_hw=None):
    """[H, W, A, 6] = (x, y, w, h, sin, cos) in metres, one set per BEV cell centre."""
    h = w = map_hw or config.map_dims[0]
    vs = config.voxel_size
    x0, y0 = config.area_extents[0][0], config.area_extents[1][0]
    xs = x0 + (np.arange(h) + 0.5) * vs[0]          # first BEV axis = x
    ys = y0 + (np.arange(w) + 0.5) * vs[1]
    a = np.asarray(config.anchor_size, dtype=np.float64)       # [A, 3] = (w, l, yaw)
    out = np.zeros((h, w, len(a), 6), dtype=np.float32)
    out[..., 0] = xs[:, None, None]
    out[..., 1] = ys[None, :, None]
    out[..., 2] = a[None, None, :, 0]
    out[..., 3] = a[None, None, :, 1]
    out[..., 4] = np.sin(a[None, None, :, 2])
    out[..., 5] = np.cos(a[None, None, :, 2])
    return out


def softmax_fg(cls_logits):
    """cls [..., 2] logits -> foreground probability (class 1)."""
    z = cls_logits - cls_logits.max(-1, keepdims=True)
    e = np.exp(z)
    return e[..., 1] / e.sum(-1)


def decode_boxes(loc, anchors):
    """Residual code -> boxes.  loc, anchors [..., 6]:
       x = xa + tx*wa, y = ya + ty*ha, w = wa*exp(tw), h = ha*exp(th),
       (sin, cos) = rotation of the anchor's angle by the predicted one."""
    xa, ya, wa, ha, sa, ca = [anchors[..., i] for i in range(6)]
    tx, ty, tw, th, ts, tc = [loc[..., i] for i in range(6)]
    out = np.empty(np.broadcast(loc, anchors).shape, dtype=np.float32)
    out[..., 0] = xa + tx * wa
    out[..., 1] = ya + ty * ha
    out[..., 2] = wa * np.exp(tw)
    out[..., 3] = ha * np.exp(th)
    out[..., 4] = sa * tc + ca * ts
    out[..., 5] = ca * tc - sa * ts
    return out


def box_corners(b):
    """[x, y, w, h, sin, cos] -> 4x2 corner array (counter-clockwise)."""
    x, y, w, h, s, c = [float(v) for v in b]
    n = math.hypot(s, c) or 1.0
    s, c = s / n, c / n
    dx, dy = w / 2.0, h / 2.0
    pts = np.array([[-dx, -dy], [dx, -dy], [dx, dy], [-dx, dy]])
    rot = np.array([[c, -s], [s, c]])
    return pts @ rot.T + np.array([x, y])


def _clip(poly, a, b):
    """Sutherland-Hodgman: keep the part of `poly` left of the directed edge a->b."""
    out = []
    n = len(poly)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if sp * sq < 0:
            t = sp / (sp - sq)
            out.append(p + t * (q - p))
    return out


def _area(poly):
    if len(poly) < 3:
        return 0.0
    p = np.asarray(poly)
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def rotated_iou(b1, b2):
    c1, c2 = box_corners(b1), box_corners(b2)
    poly = [p for p in c1]
    for i in range(4):
        poly = _clip(poly, c2[i], c2[(i + 1) % 4])
        if not poly:
            return 0.0
    inter = _area(poly)
    union = float(b1[2]) * float(b1[3]) + float(b2[2]) * float(b2[3]) - inter
    return inter / union if union > 0 else 0.0


def _maybe_overlap(b, others):
    """cheap circumscribed-circle test before the polygon clip"""
    r = 0.5 * np.hypot(others[:, 2], others[:, 3]) + 0.5 * math.hypot(b[2], b[3])
    return np.hypot(others[:, 0] - b[0], others[:, 1] - b[1]) < r


def nms_rotated(boxes, scores, iou_thr=0.01, top_k=None):
    """greedy NMS on rotated boxes; returns kept indices in score order.  Ties in
    score are broken by the lower index (stable sort), so the result is deterministic."""
    order = np.argsort(-scores, kind="stable")
    if top_k is not None:
        order = order[:top_k]
    keep = []
    alive = np.ones(len(order), dtype=bool)
    for oi, i in enumerate(order):
        if not alive[oi]:
            continue
        keep.append(int(i))
        rest = np.nonzero(alive[oi + 1:])[0] + oi + 1
        if len(rest) == 0:
            continue
        cand = rest[_maybe_overlap(boxes[i], boxes[order[rest]])]
        for cj in cand:
            if rotated_iou(boxes[i], boxes[order[cj]]) > iou_thr:
                alive[cj] = False
    return np.asarray(keep, dtype=np.int64)


def average_precision(det_boxes, det_scores, gt_boxes, iou_thr):
    """mmdetection-style AP (area under the interpolated precision/recall curve) of one
    class over a list of images."""
    records, n_gt = [], 0
    for boxes, scores, gts in zip(det_boxes, det_scores, gt_boxes):
        n_gt += len(gts)
        taken = np.zeros(len(gts), dtype=bool)
        for i in np.argsort(-scores, kind="stable"):
            best, best_j = 0.0, -1
            if len(gts):
                for j in np.nonzero(_maybe_overlap(boxes[i], gts))[0]:
                    iou = rotated_iou(boxes[i], gts[j])
                    if iou > best:
                        best, best_j = iou, j
            tp = best >= iou_thr and not taken[best_j]
            if tp:
                taken[best_j] = True
            records.append((float(scores[i]), tp))
    if n_gt == 0:
        return 0.0
    records.sort(key=lambda r: -r[0])
    tps = np.cumsum([r[1] for r in records])
    fps = np.cumsum([not r[1] for r in records])
    recall = tps / n_gt
    precision = tps / np.maximum(tps + fps, 1)
    mrec = np.concatenate([[0.0], recall, [1.0]])
    mpre = np.concatenate([[0.0], precision, [0.0]])
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def detections_from_logits(cls, loc, anchors, score_thr=None, pre_nms_top_k=300, iou_thr=0.01):
    """predict_all's per-agent tail: softmax -> decode -> (score filter / top-k) -> rotated NMS.
    cls [H*W*A, 2], loc [H, W, A, 1, 6], anchors [H, W, A, 6] -> (boxes [K, 6], scores [K])."""
    scores = softmax_fg(cls.reshape(-1, 2))
    boxes = decode_boxes(loc.reshape(-1, 6), anchors.reshape(-1, 6))
    idx = np.arange(len(scores))
    if score_thr is not None:
        idx = idx[scores > score_thr]
    order = idx[np.argsort(-scores[idx], kind="stable")][:pre_nms_top_k]
    keep = nms_rotated(boxes[order], scores[order], iou_thr)
    return boxes[order][keep], scores[order][keep]
