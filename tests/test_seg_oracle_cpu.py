"""Segmentation variant, CPU side: the oracle (oracle/seg_ref.py) against its committed golden
vectors (tests/golden/seg_cases.npz, written by tests/golden/make_golden.py -- they pin the oracle
against drift, not the reference: parity unpinned, see the oracle's header)."""
import os

import numpy as np
import pytest
import torch

from tests import cases


@pytest.mark.parametrize("case", list(cases.SEG_CASES))
def test_seg_oracle_golden(case, golden_dir):
    g = np.load(os.path.join(golden_dir, "seg_cases.npz"))
    outs, loss = cases.run_seg_ref(case)
    for name, t in outs.items():
        assert np.abs(cases.seg_subsample(name, t) - g["%s/%s" % (case, name)]).max() <= 1e-5, name
    assert abs(loss - float(g["%s/loss" % case])) <= 1e-6 * abs(loss)


def test_seg_oracle_structure():
    """the UNet of the recollected upstream: 512-channel bottleneck at H/8, dead agents keep their map,
    one agent alone is not fused at all"""
    m = cases.seg_ref_model(3, kd_flag=True)
    from disconet_amd.synthetic import make_scene_batch
    bevs, trans, na = make_scene_batch(1, 3, 128, live=[1])
    x = bevs[:, 0].permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        logits, x9, x8, x7, x6, x5, fused = m(x, trans, na, 1)
        x4 = m.down3(m.down2(m.down1(m.inc(x))))
    assert logits.shape == (3, 8, 128, 128) and fused.shape == (3, 512, 16, 16) and x5.shape == (3, 512, 8, 8)
    assert torch.allclose(fused, x4, atol=1e-6)            # one live agent: softmax over itself
    names = set(m.state_dict())
    assert {"inc.double_conv.0.weight", "down4.maxpool_conv.1.double_conv.4.running_var",
            "up1.conv.double_conv.0.weight", "outc.conv.bias", "pixel_weighted_fusion.conv1_1.weight"} <= names
    assert m.up1.conv.double_conv[0].weight.shape == (512, 1024, 3, 3)
