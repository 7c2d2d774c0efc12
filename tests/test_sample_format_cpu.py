"""The sparse-voxel sample reader (disconet_amd/sample_format.py; SURVEY.md §8(f) #4, second half): files written in the
recalled on-disk form round-trip into the hot path's batch inputs, and the dense rebuild of those inputs (numpy here; the
GPU scatter is covered by tests/test_gpu_voxel.py) equals the grid the files were made from."""
import numpy as np
import pytest
import torch

from disconet_amd import sample_format
from disconet_amd.synthetic import make_bevs, make_trans_matrices


def test_round_trip_through_npy_files(tmp_path):
    A, B, hw = 3, 2, 32
    bevs = make_bevs(B, A, hw)                       # [A*B, 1, hw, hw, 13]
    trans = make_trans_matrices(B, A).numpy()
    samples = [[None] * A for _ in range(B)]
    for a in range(A):
        for b in range(B):
            grid = bevs[a * B + b, 0].numpy()
            idx = np.argwhere(grid > 0).astype(np.int64)             # sorted unique, as voxelize_occupy leaves it
            path = tmp_path / ("scene%d_agent%d.npy" % (b, a))
            np.save(path, {"voxel_indices_0": idx, "trans_matrices": trans[b, a], "num_sensor": A,
                           "reg_target_sparse": np.zeros((1, 6))}, allow_pickle=True)
            samples[b][a] = sample_format.load_sample(str(path), (hw, hw, 13))
            assert samples[b][a]["indices"].dtype == np.int32 and "reg_target_sparse" in samples[b][a]["rest"]
    indices, offsets, tr, na = sample_format.batch_from_samples(samples, A, device="cpu")
    assert offsets.dtype == torch.int32 and offsets.shape == (A * B + 1,) and int(offsets[-1]) == indices.shape[0]
    assert torch.equal(tr, torch.from_numpy(trans)) and torch.equal(na, torch.full((B, A), A))
    dense = torch.zeros(A * B, hw, hw, 13)
    for g in range(A * B):
        rows = indices[int(offsets[g]):int(offsets[g + 1])].long()
        dense[g, rows[:, 0], rows[:, 1], rows[:, 2]] = 1.0
    # the model's frame is the stored grid after the loader's np.rot90(grid, 3) over (x, y)
    want = torch.from_numpy(np.stack([np.rot90(bevs[g, 0].numpy(), 3).copy() for g in range(A * B)]))
    assert torch.equal(dense, want)
    # ... and with rotate=False the stored list is handed through
    raw = sample_format.load_sample(str(tmp_path / "scene0_agent0.npy"), hw, rotate=False)["indices"]
    assert np.array_equal(raw, np.argwhere(bevs[0, 0].numpy() > 0).astype(np.int32))


def test_rot90_index_map_equals_numpy_rot90_of_the_dense_rebuild():
    rng = np.random.default_rng(5)
    for (X, Y, Z) in ((8, 8, 3), (16, 12, 13), (5, 9, 2)):
        dense = rng.random((X, Y, Z)) < 0.2
        idx = np.argwhere(dense).astype(np.int32)
        got = sample_format.rot90_indices(idx, X)
        want = np.argwhere(np.rot90(dense, 3)).astype(np.int32)       # argwhere is row-major: the sorted-unique order
        assert np.array_equal(got, want)


def test_missing_key_is_reported(tmp_path):
    path = tmp_path / "bad.npy"
    np.save(path, {"voxel_indices": np.zeros((0, 3))}, allow_pickle=True)
    with pytest.raises(KeyError, match="voxel_indices_0"):
        sample_format.load_sample(str(path), 256)


def test_grid_dims_are_required_and_checked(tmp_path):
    """a stored grid smaller than an assumed 256 rows mapped x to 255 - x, outside the real grid, and the scatter's
    bounds check dropped the voxels silently.  dims is required now, every index is checked, the rotated dims travel with the
    sample and batch_from_samples refuses a batch built for another grid."""
    idx = np.array([[0, 1, 2], [31, 47, 12]], dtype=np.int64)           # a 32 x 48 x 13 stored grid
    path = tmp_path / "s.npy"
    np.save(path, {"voxel_indices_0": idx, "trans_matrices": np.eye(4)[None], "num_sensor": 1}, allow_pickle=True)
    with pytest.raises(TypeError):
        sample_format.load_sample(str(path))                             # no default grid any more
    with pytest.raises(ValueError, match="outside the stored grid"):
        sample_format.load_sample(str(path), 32)                         # y = 47 does not fit a square 32 grid
    with pytest.raises(ValueError, match="outside the stored grid"):
        sample_format.load_sample(str(path), (32, 48, 12))               # z = 12 does not fit 12 bins
    s = sample_format.load_sample(str(path), (32, 48, 13))
    assert s["dims"] == (48, 32, 13)                                     # rot90 swaps X and Y
    assert np.array_equal(s["indices"], np.array([[1, 31, 2], [47, 0, 12]], dtype=np.int32))
    assert sample_format.load_sample(str(path), (32, 48, 13), rotate=False)["dims"] == (32, 48, 13)
    sample_format.batch_from_samples([[s]], 1, device="cpu", dims=(48, 32, 13))
    with pytest.raises(ValueError, match="swap X and Y"):
        sample_format.batch_from_samples([[s]], 1, device="cpu", dims=(32, 48, 13))
