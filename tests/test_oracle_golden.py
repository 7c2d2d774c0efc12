"""The CPU oracle against the committed golden fixtures (tests/golden/*.npz,
made by tests/golden/make_golden.py).  Parity is UNPINNED against the reference
(no source / vectors in /root/reference): these goldens pin the oracle itself."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle.disconet_ref import feature_transformation
from oracle.voxel_ref import dense_from_indices, voxelize_occupy
from tests import cases


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_voxelizer_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "voxel_20k.npz"))
    pts = cases.voxel_cloud()
    assert _sha(pts) == str(g["pts_sha256"])
    dense, idx = voxelize_occupy(pts, cases.VOXEL_SIZE, cases.EXTENTS, return_indices=True)
    assert dense.shape == cases.DIMS and dense.dtype == np.float32
    assert np.array_equal(idx.astype(np.int32), g["indices"])
    assert _sha(dense) == str(g["dense_sha256"])
    assert int(dense.sum()) == int(g["n_occupied"]) == len(idx)
    # dense rebuild (V2XSimDet.__getitem__) round-trips the sparse list
    assert np.array_equal(dense_from_indices(idx, cases.DIMS)[0], dense)


def test_voxelizer_boundary_semantics():
    vs, ext = cases.VOXEL_SIZE, cases.EXTENTS
    pts = np.array([
        [-32.0, 0.0, 0.0, 0],       # on the low extent: rejected (strict <)
        [32.0, 0.0, 0.0, 0],        # on the high extent: rejected
        [0.0, 0.0, 2.0, 0],         # z on the high extent: rejected
        [0.0, 0.0, -3.0, 0],        # z on the low extent: rejected
        [0.25, 0.5, 0.4, 0],        # exactly on voxel boundaries -> floor
        [-0.25, -0.5, -0.4, 0],
        [31.99, 31.99, 1.99, 0],
        [-31.99, -31.99, -2.99, 0],
    ], dtype=np.float32)
    dense, idx = voxelize_occupy(pts, vs, ext, return_indices=True)
    want = np.array([[0, 0, 0], [127, 126, 6], [129, 130, 9], [255, 255, 12]])
    # float32(0.4)/0.4 in float64 is 1.0000000149 -> floor 1 -> z index 9;
    # float32(-0.4)/0.4 -> -1.0000000149 -> floor -2 -> z index 6
    assert np.array_equal(idx, want)
    assert dense.sum() == 4


def test_voxelizer_empty_cloud():
    dense, idx = voxelize_occupy(np.zeros((0, 4), np.float32), cases.VOXEL_SIZE, cases.EXTENTS,
                                 return_indices=True)
    assert dense.shape == cases.DIMS and dense.sum() == 0 and idx.shape == (0, 3)


def test_warp_unit_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "warp_unit.npz"))
    feat = cases.warp_feature()
    for name, pose in cases.WARP_POSES.items():
        w = feature_transformation(0, 0, feat.unsqueeze(0), torch.from_numpy(pose)[None],
                                   tuple(feat.shape)).numpy()
        assert np.allclose(w, g[name], atol=1e-6), name
    assert np.allclose(g["identity"], feat[0].numpy(), atol=1e-5)
    assert np.abs(g["out_of_frame"]).max() == 0.0
    # whole-pixel shift: pure index shift with zero fill. x_trans = 4*4/128 -> +2 px
    # sample offset in x, y_trans = -(4*-6)/128 -> +3 px in y.
    src = feat[0].numpy()
    want = np.zeros_like(src)
    want[:, :-3, :-2] = src[:, 3:, 2:]
    assert np.allclose(g["shift_whole_px"], want, atol=1e-5)


@pytest.mark.parametrize("case", list(cases.MODEL_CASES))
def test_model_golden(case, golden_dir):
    g = np.load(os.path.join(golden_dir, "model_cases.npz"))
    outs = cases.run_ref(case)
    for name, t in outs.items():
        want = g["%s/%s" % (case, name)]
        got = cases.subsample(name, t)
        assert got.shape == want.shape
        err = np.abs(got - want).max()
        assert err <= 2e-5, (case, name, err)


def test_benchmarked_configuration_golden(golden_dir):
    """the 256 x 256, 5-agent, batch-4 whole-model golden of the configuration bench.py times
    (tests/golden/model_256_a5.npz).  The inputs regenerate bit for bit from their seeds (SHA-256); the oracle's outputs
    agree with the committed strided slices to 2e-5 (summation order moves the last bits with the thread count, so the
    full-tensor SHA-256 in the file is a record of the generating run, not an assertion)."""
    g = np.load(os.path.join(golden_dir, "model_256_a5.npz"))
    indices, offsets, bevs, trans, na = cases.bench_case_inputs()
    assert _sha(indices.numpy()) == str(g["indices_sha256"])
    assert _sha(offsets.numpy()) == str(g["offsets_sha256"])
    assert _sha(trans.numpy()) == str(g["trans_sha256"])
    # the sparse lists ARE the dense grid the oracle reads
    dense = torch.zeros(bevs.shape[0], *bevs.shape[2:])
    img = torch.repeat_interleave(torch.arange(bevs.shape[0]), (offsets[1:] - offsets[:-1]).long())
    dense[img, indices[:, 0].long(), indices[:, 1].long(), indices[:, 2].long()] = 1.0
    assert torch.equal(dense, bevs[:, 0])
    outs, _ = cases.run_ref_bench_case()
    for name, t in outs.items():
        want = g[name]
        got = cases.subsample_bench(name, t)
        assert got.shape == want.shape
        assert float(t.abs().max()) > 1.0, name                # O(1) activations: the error bar means something
        err = np.abs(got - want).max()
        assert err <= 2e-5, (name, err)


def test_ragged_scene_dead_agents_pass_through():
    """num_agent_tensor[b, 0] live agents are fused; padded agents keep their map."""
    c = cases.MODEL_CASES["ragged_a4"]
    model = cases.ref_model(c["map_hw"], c["agents"])
    bevs, trans, na = cases.model_inputs("ragged_a4")
    with torch.no_grad():
        enc = model.u_encoder(bevs.permute(0, 1, 4, 2, 3))
        fused = model(bevs, trans, na, c["batch"])[-1]
    x3 = enc[3]
    B = c["batch"]
    for b, live in enumerate(c["live"]):
        for a in range(c["agents"]):
            same = torch.equal(fused[a * B + b], x3[a * B + b])
            assert same == (a >= live), (b, a)


def test_fusion_block_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "fusion_5x256.npz"))
    model = cases.ref_model(256, 5)
    feat, trans, na = cases.fusion_inputs()
    fused = cases.ref_fuse(model, feat, trans, na).numpy()
    assert np.abs(fused[:, ::4, ::2, ::2] - g["fused"]).max() <= 2e-5
    # and the helper agrees with the oracle's own forward loop: warped neighbours matter
    assert np.abs(fused - feat.numpy()).max() > 0.1


@pytest.mark.parametrize("case,tag", [("cfg1", "det"), ("cfg1", "kd"), ("ragged_a4", "kd"), ("lonely_a3", "det")])
def test_training_step_golden(golden_dir, case, tag):
    """the oracle's float64 training forward / backward (oracle/train_ref.py, oracle/teacher_ref.py)
    against tests/golden/train_step.npz: losses and strided slices of ten parameters' gradients.
    float64 on purpose -- in float32 this backward moves by ~1 % with the summation order
    (thread count) alone, see tests/test_gpu_train_step.py."""
    from oracle.disconet_ref import RefConfig
    from oracle.teacher_ref import build_teacher
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    c = cases.TRAIN_CASES[case]
    ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0)
    teacher = build_teacher(RefConfig(c["map_hw"])) if tag == "kd" else None
    losses, grads = cases.oracle_train_fp64(case, ref, teacher)
    assert np.allclose(losses, g["%s/%s/losses" % (case, tag)], rtol=1e-9)
    for n in cases.GOLDEN_GRAD_TENSORS:
        want = g["%s/%s/%s" % (case, tag, n)]
        scale = float(g["%s/%s/%s/absmax" % (case, tag, n)])
        assert abs(cases.grad_slice(grads[n]) - want).max() <= 1e-7 * scale, n


def test_seg_training_step_golden(golden_dir):
    """the seg oracle's float64 training forward / backward (oracle/seg_ref.py through F.cross_entropy; what
    oracle/seg_train_ref.py steps) against tests/golden/seg_train_step.npz: loss and gradient slices"""
    g = np.load(os.path.join(golden_dir, "seg_train_step.npz"))
    case = "seg_a2"
    ref = cases.seg_ref_model(cases.SEG_CASES[case]["agents"])
    loss, grads = cases.oracle_seg_train_fp64(case, ref)
    assert abs(loss - float(g["%s/loss" % case])) <= 1e-9 * abs(loss)
    for n in cases.SEG_GOLDEN_GRAD_TENSORS:
        scale = float(g["%s/%s/absmax" % (case, n)])
        assert abs(cases.grad_slice(grads[n]) - g["%s/%s" % (case, n)]).max() <= 1e-7 * scale, n
