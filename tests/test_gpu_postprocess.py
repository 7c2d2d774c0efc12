"""Detection decode on the GPU vs the oracle, and the mAP-parity statement of
BASELINE.json's metric: detections from the HIP path scored against the detections
the CPU oracle produces on the same synthetic scene."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def test_decode_kernel_vs_oracle():
    from disconet_amd import Config, postprocess
    from oracle import postprocess_ref as R
    g = torch.Generator().manual_seed(0)
    cfg = Config(map_hw=64)
    n, h, a = 3, 64, 6
    cls = torch.randn(n, h * h * a, 2, generator=g)
    loc = torch.randn(n, h, h, a, 1, 6, generator=g) * 0.3
    anchors = postprocess.make_anchors(cfg, device="cuda")
    assert np.allclose(anchors.cpu().numpy(), R.make_anchors(cfg), atol=1e-6)
    scores, boxes = postprocess.decode({"cls": cls.cuda(), "loc": loc.cuda()}, anchors)
    want_s = R.softmax_fg(cls.numpy())
    want_b = R.decode_boxes(loc.numpy().reshape(n, -1, 6), R.make_anchors(cfg).reshape(1, -1, 6))
    assert np.abs(scores.cpu().numpy() - want_s).max() <= 1e-6
    assert np.abs(boxes.cpu().numpy() - want_b).max() <= 1e-5


def test_map_parity_hip_vs_oracle_detections():
    """mAP@0.5 / @0.7 of the HIP path's detections against the oracle's detections on
    identical inputs and weights (the oracle plays ground truth): the north star asks
    for mAP@0.7 within 0.5 pt."""
    from disconet_amd import Config, DiscoNet, postprocess
    from oracle import postprocess_ref as R
    c = cases.MODEL_CASES["cfg1_f1"]
    ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0)
    bevs, trans, na = cases.model_inputs("cfg1_f1")
    with torch.no_grad():
        want = ref(bevs, trans, na, c["batch"])
    cfg = Config(map_hw=c["map_hw"])
    anchors_np = R.make_anchors(cfg)
    gt_boxes, gt_scores = [], []
    for i in range(want["cls"].shape[0]):
        b, s = R.detections_from_logits(want["cls"][i].numpy(), want["loc"][i].numpy(), anchors_np)
        gt_boxes.append(b)
        gt_scores.append(s)
    m = DiscoNet(cfg, kd_flag=0, num_agent=c["agents"]).eval()
    m.load_state_dict(ref.state_dict())
    m.cuda()
    dets = postprocess.predict_all(m, postprocess.make_anchors(cfg), bevs.cuda(), trans.cuda(),
                                   na.cuda(), c["batch"])
    assert all(len(b) > 10 for b in gt_boxes)
    for thr in (0.5, 0.7):
        ap = R.average_precision([d[0] for d in dets], [d[1] for d in dets], gt_boxes, thr)
        assert ap >= 0.995, "mAP@%.1f of HIP detections vs oracle detections = %.4f" % (thr, ap)
