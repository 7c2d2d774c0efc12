"""On-disk sample format of the reference's V2X-Sim detection set -> the inputs of the hot path (SURVEY.md §8(f) #4,
second half; /root/reference/README.md:33 names the dataset, the loader is upstream:coperception/datasets/V2XSimDet.py).

RECOLLECTION, like the rest of the boundary: the loader's source is not in the mount.  What is recalled: one `.npy` per
(scene, frame, agent) holding a pickled dict whose entry `voxel_indices_0` is the sorted-unique [M, 3] integer list of
occupied voxels of the current sweep (what `voxelize_occupy` produced offline, §8 a1), next to `trans_matrices`
[A, 4, 4] (this agent's row of the pose table), `num_sensor` / the live-agent count and the detection targets.
`V2XSimDet.__getitem__` rebuilds the dense grid from that list (§8 a2) and then applies `np.rot90(grid, 3)` over the
(x, y) axes before the grid reaches the model -- the frame the checkpoints were trained in and the one
`trans_matrices` warps assume.  Here `dn_scatter_dense_sp(_hi)` does the rebuild for the whole batch, straight into the
conv engine's layout, and the rotation is an INDEX map applied at load time: voxel (x, y, z) of an [X, Y, Z] grid
lands at (y, X - 1 - x, z), the list is re-sorted into the sorted-unique (row-major) order the scatter's contract asks
for (`rot90_indices`; `rotate=False` hands the stored list through).  Everything else in the dict (targets, visualisation maps) is
handed through untouched.  The key names live in KEYS so that a real checkout can correct them in one place
(SURVEY.md Appx C).

This is host-side plumbing (numpy), not a data loader: datasets are out of scope.
"""
import numpy as np
import torch

KEYS = {"indices": "voxel_indices_0", "trans": "trans_matrices", "num_agent": "num_sensor"}


def rot90_indices(idx, grid_x):
    """np.rot90(grid, 3) (axes (0, 1)) of an [X, Y, Z] occupancy grid as a map of its sorted-unique index list:
    (x, y, z) -> (y, X - 1 - x, z), re-sorted row-major.  Equals np.argwhere(np.rot90(dense, 3))."""
    out = np.stack([idx[:, 1], grid_x - 1 - idx[:, 0], idx[:, 2]], 1).astype(np.int32)
    order = np.lexsort((out[:, 2], out[:, 1], out[:, 0]))
    return np.ascontiguousarray(out[order])


def load_sample(path, grid_x=256, rotate=True):
    """One `.npy` file -> dict with `indices` [M, 3] int32 (sorted unique, in the MODEL's frame: after the loader's
    rot90(k=3) when `rotate`; `grid_x` = first dimension of the stored grid), `trans_matrices` [A, 4, 4] float32,
    `num_agent` int, and `rest` (every other entry, untouched)."""
    raw = np.load(path, allow_pickle=True)
    d = raw.item() if raw.dtype == object and raw.shape == () else dict(raw)
    for k in KEYS.values():
        if k not in d:
            raise KeyError("sample %s has no %r entry (keys: %s); see disconet_amd/sample_format.py :: KEYS"
                           % (path, k, sorted(d)[:12]))
    idx = np.ascontiguousarray(np.asarray(d[KEYS["indices"]]).reshape(-1, 3).astype(np.int32))
    if rotate:
        if idx.size and (idx[:, 0].max() >= grid_x or idx.min() < 0):
            raise ValueError("sample %s: voxel x index %d outside the %d-row grid (pass grid_x)" % (path, int(idx[:, 0].max()), grid_x))
        idx = rot90_indices(idx, grid_x)
    rest = {k: v for k, v in d.items() if k not in KEYS.values()}
    return {"indices": idx, "trans_matrices": np.asarray(d[KEYS["trans"]], dtype=np.float32),
            "num_agent": int(np.asarray(d[KEYS["num_agent"]]).reshape(-1)[0]), "rest": rest}


def batch_from_samples(samples, num_agent, device="cuda"):
    """samples[b][a] = load_sample(...) of agent a of scene b (None for an absent agent) -> the hot path's inputs in
    the agent-major image order the reference's tools build (image = a * B + b):
        indices [Mtot, 3] int32, offsets [A*B + 1] int32   (for ops.scatter_dense_sp / scatter_dense)
        trans_matrices [B, A, A, 4, 4] float32, num_agent_tensor [B, A] int64"""
    B = len(samples)
    lists, offsets = [], [0]
    trans = np.tile(np.eye(4, dtype=np.float32), (B, num_agent, num_agent, 1, 1))
    na = np.zeros((B, num_agent), dtype=np.int64)
    for a in range(num_agent):
        for b in range(B):
            s = samples[b][a] if a < len(samples[b]) else None
            idx = s["indices"] if s is not None else np.zeros((0, 3), np.int32)
            lists.append(idx)
            offsets.append(offsets[-1] + idx.shape[0])
            if s is not None:
                t = s["trans_matrices"]
                trans[b, a, :t.shape[0]] = t[:num_agent]
                na[b, :] = max(na[b, 0], s["num_agent"])
    indices = np.concatenate(lists, 0) if lists else np.zeros((0, 3), np.int32)
    to = lambda x: torch.from_numpy(x).to(device)
    return to(indices), to(np.asarray(offsets, dtype=np.int32)), to(trans), to(na)
