"""K1 on the MI355X vs the numpy oracle: bit-exact integer/occupancy results."""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def _ops():
    from disconet_amd import ops
    return ops


def test_voxelize_matches_oracle_bit_exact(golden_dir):
    from oracle.voxel_ref import voxelize_occupy
    ops = _ops()
    pts = cases.voxel_cloud()
    dense_ref, idx_ref = voxelize_occupy(pts, cases.VOXEL_SIZE, cases.EXTENTS, return_indices=True)
    dense = ops.voxelize_occupy(torch.from_numpy(pts).cuda(), cases.VOXEL_SIZE, cases.EXTENTS, cases.DIMS)
    assert np.array_equal(dense.cpu().numpy(), dense_ref)
    idx, m = ops.voxel_compact(dense)
    assert m == len(idx_ref)
    assert np.array_equal(idx.cpu().numpy(), idx_ref.astype(np.int32))
    g = np.load(os.path.join(golden_dir, "voxel_20k.npz"))
    assert np.array_equal(idx.cpu().numpy(), g["indices"])


def test_voxelize_xyz_only_and_empty():
    from oracle.voxel_ref import voxelize_occupy
    ops = _ops()
    pts = cases.voxel_cloud()[:, :3].copy()
    dense_ref = voxelize_occupy(pts, cases.VOXEL_SIZE, cases.EXTENTS)
    dense = ops.voxelize_occupy(torch.from_numpy(pts).cuda(), cases.VOXEL_SIZE, cases.EXTENTS, cases.DIMS)
    assert np.array_equal(dense.cpu().numpy(), dense_ref)
    empty = ops.voxelize_occupy(torch.zeros((0, 4), device="cuda"), cases.VOXEL_SIZE, cases.EXTENTS, cases.DIMS)
    assert float(empty.sum()) == 0.0
    idx, m = ops.voxel_compact(empty)
    assert m == 0 and idx.shape == (0, 3)


def test_voxelize_boundary_points():
    from oracle.voxel_ref import voxelize_occupy
    ops = _ops()
    pts = np.array([[-32.0, 0, 0, 0], [32.0, 0, 0, 0], [0, 0, 2.0, 0], [0, 0, -3.0, 0],
                    [0.25, 0.5, 0.4, 0], [-0.25, -0.5, -0.4, 0], [31.99, 31.99, 1.99, 0],
                    [-31.99, -31.99, -2.99, 0]], dtype=np.float32)
    _, idx_ref = voxelize_occupy(pts, cases.VOXEL_SIZE, cases.EXTENTS, return_indices=True)
    dense = ops.voxelize_occupy(torch.from_numpy(pts).cuda(), cases.VOXEL_SIZE, cases.EXTENTS, cases.DIMS)
    idx, m = ops.voxel_compact(dense)
    assert np.array_equal(idx.cpu().numpy(), idx_ref.astype(np.int32))


def test_full_size_properties():
    """BASELINE size (60k points / agent): idempotence, sortedness, round trip
    sparse -> dense -> sparse, batched dense rebuild == per-agent voxelize."""
    from disconet_amd.synthetic import make_point_cloud
    ops = _ops()
    denses, lists = [], []
    for a in range(5):
        pts = torch.from_numpy(make_point_cloud(60000, seed=40 + a)).cuda()
        d1 = ops.voxelize_occupy(pts, cases.VOXEL_SIZE, cases.EXTENTS, cases.DIMS)
        d2 = ops.voxelize_occupy(torch.cat([pts, pts.flip(0)]), cases.VOXEL_SIZE, cases.EXTENTS, cases.DIMS)
        assert torch.equal(d1, d2)                       # duplicates / order are irrelevant
        assert set(torch.unique(d1).tolist()) <= {0.0, 1.0}
        idx, m = ops.voxel_compact(d1)
        assert m == int(d1.sum().item())
        lin = (idx[:, 0].long() * 256 + idx[:, 1].long()) * 13 + idx[:, 2].long()
        assert bool((lin[1:] > lin[:-1]).all())          # strictly increasing = sorted + unique
        denses.append(d1)
        lists.append(idx)
    offsets = torch.tensor([0] + list(np.cumsum([len(l) for l in lists])), dtype=torch.int32).cuda()
    bevs = ops.scatter_dense(torch.cat(lists), offsets, 5, cases.DIMS)
    assert bevs.shape == (5, 1, 256, 256, 13)
    for a in range(5):
        assert torch.equal(bevs[a, 0], denses[a])


def test_scatter_dense_sp_matches_dense():
    """the split-planar rebuild is bit for bit the dense rebuild: 1.0 -> (hi 1.0, lo 0), and the
    conversion kernel of the dense grid gives the same bytes (padding channels 13..15 zero)"""
    from disconet_amd.synthetic import make_sparse_scene_batch
    ops = _ops()
    indices, offsets, _ = make_sparse_scene_batch(2, 3, 128)
    indices, offsets = indices.cuda(), offsets.cuda()
    dims = (128, 128, 13)
    dense = ops.scatter_dense(indices, offsets, 6, dims)
    sp = ops.scatter_dense_sp(indices, offsets, 6, dims)
    assert sp.shape == (6, 128, 128, 13) and sp.data.shape == (6, 1, 4, 128, 128, 8)
    assert torch.equal(sp.nhwc(), dense.view(6, 128, 128, 13))
    assert torch.equal(sp.data, ops.SpTensor.from_nhwc(dense.view(6, 128, 128, 13)).data)
    hi = ops.scatter_dense_sp(indices, offsets, 6, dims, hi_only=True)
    assert hi.data.shape == (6, 1, 2, 128, 128, 8) and torch.equal(hi.data, sp.data[:, :, :2])
    assert float(sp.data[:, :, 2:].float().abs().sum()) == 0.0          # what the hi-only form leaves out
    assert torch.equal(hi.nhwc(), dense.view(6, 128, 128, 13))
    # empty batch entry / no voxels at all
    empty = ops.scatter_dense_sp(indices[:0], torch.zeros(3, dtype=torch.int32, device="cuda"), 2, dims)
    assert float(empty.data.float().abs().sum()) == 0.0


def test_scatter_dense_bits_matches_dense():
    """one occupancy word per pixel (bit z = bin z): the same cells as the dense rebuild, duplicates, out-of-range rows
    and empty images included; more than 32 bins refused"""
    from disconet_amd.synthetic import make_sparse_scene_batch
    ops = _ops()
    indices, offsets, _ = make_sparse_scene_batch(2, 3, 128)
    dims = (128, 128, 13)
    # image 2 sees its rows twice (duplicates), two rows fall outside the grid, image 5's list is emptied
    o = offsets.tolist()
    dup = indices[o[2]:o[3]]
    bad = torch.tensor([[128, 0, 0], [5, 7, 13]], dtype=torch.int32)
    rows = torch.cat([indices[:o[3]], dup, bad, indices[o[3]:o[5]]])
    extra = dup.shape[0] + 2
    offs = torch.tensor(o[:3] + [o[3] + extra, o[4] + extra, o[5] + extra, o[5] + extra], dtype=torch.int32)
    rows, offs = rows.cuda(), offs.cuda()
    dense = ops.scatter_dense(rows, offs, 6, dims)
    bits = ops.scatter_dense_bits(rows, offs, 6, dims)
    assert bits.bits and bits.data.shape == (6, 128, 128) and bits.data.dtype == torch.int32
    want = (dense.view(6, 128, 128, 13).to(torch.int32) << torch.arange(13, device="cuda", dtype=torch.int32)).sum(-1)
    assert torch.equal(bits.data, want.to(torch.int32))
    assert torch.equal(bits.nhwc(), dense.view(6, 128, 128, 13))
    assert int(bits.data[5].abs().sum()) == 0
    with pytest.raises(ops._lib.DnError):
        ops.scatter_dense_bits(rows, offs, 6, (128, 128, 33))
