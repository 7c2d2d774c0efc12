"""numpy restatement of the point-cloud -> BEV occupancy voxelizer.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no source / golden vectors in
/root/reference; see oracle/__init__.py).

Follows upstream:coperception/utils/data_util.py :: voxelize_occupy (MotionNet
lineage, credited at /root/reference/README.md:104) as specified in SURVEY.md
Appx A.2, and upstream:coperception/datasets/V2XSimDet.py :: __getitem__
(dense rebuild) as specified in SURVEY.md §8 a2.
"""
import numpy as np


def voxelize_occupy(pts, voxel_size, extents=None, return_indices=False):
    """pts [N, 3|4] float32 -> dense float32 occupancy (+ sorted unique indices).

    * strict extent filter on the raw float coordinates;
    * floor(pts / voxel_size): voxel_size is a Python tuple, so numpy promotes
      the divide to float64;
    * lexsort by (x, then y, then z) + unique rows;
    * indices shifted by floor(extent_lo / voxel).
    """
    VOXEL_EMPTY, VOXEL_FILLED = 0, 1
    if pts.shape[1] < 3 or pts.shape[1] > 4:
        raise ValueError("Points have the wrong shape: {}".format(pts.shape))
    if extents is not None:
        if extents.shape != (3, 2):
            raise ValueError("Extents are the wrong shape {}".format(extents.shape))
        filter_idx = np.where((extents[0, 0] < pts[:, 0]) & (pts[:, 0] < extents[0, 1]) &
                              (extents[1, 0] < pts[:, 1]) & (pts[:, 1] < extents[1, 1]) &
                              (extents[2, 0] < pts[:, 2]) & (pts[:, 2] < extents[2, 1]))[0]
        pts = pts[filter_idx]

    voxel_size = np.asarray(voxel_size, dtype=np.float64)
    discrete_pts = np.floor(pts[:, :3] / voxel_size).astype(np.int32)

    x_col, y_col, z_col = discrete_pts[:, 0], discrete_pts[:, 1], discrete_pts[:, 2]
    sorted_order = np.lexsort((z_col, y_col, x_col))
    discrete_pts = discrete_pts[sorted_order]

    contiguous_array = np.ascontiguousarray(discrete_pts).view(
        np.dtype((np.void, discrete_pts.dtype.itemsize * discrete_pts.shape[1])))
    _, unique_indices = np.unique(contiguous_array, return_index=True)
    unique_indices.sort()
    voxel_coords = discrete_pts[unique_indices]

    if extents is not None:
        min_voxel_coord = np.floor(extents.T[0] / voxel_size)
        max_voxel_coord = np.ceil(extents.T[1] / voxel_size) - 1
    else:
        min_voxel_coord = np.amin(voxel_coords, axis=0)
        max_voxel_coord = np.amax(voxel_coords, axis=0)

    num_divisions = ((max_voxel_coord - min_voxel_coord) + 1).astype(np.int32)
    voxel_indices = (voxel_coords - min_voxel_coord).astype(int)

    leaf_layout = VOXEL_EMPTY * np.ones(num_divisions.astype(int), dtype=np.float32)
    leaf_layout[voxel_indices[:, 0], voxel_indices[:, 1], voxel_indices[:, 2]] = VOXEL_FILLED

    if return_indices:
        return leaf_layout, voxel_indices
    return leaf_layout


def dense_from_indices(voxel_indices, map_dims):
    """V2XSimDet.__getitem__ dense rebuild: zeros(map_dims); dense[ix,iy,iz]=1;
    leading frame dim added -> [1, X, Y, Z] float32."""
    dense = np.zeros(tuple(map_dims), dtype=np.float32)
    if len(voxel_indices):
        dense[voxel_indices[:, 0], voxel_indices[:, 1], voxel_indices[:, 2]] = 1
    return dense[None]
