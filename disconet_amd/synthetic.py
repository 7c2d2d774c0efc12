"""Seeded synthetic collaborative-perception scenes.

There is no V2X-Sim data in this environment, so bench.py and the parity tests
feed both the HIP path and the oracle from this generator:

* voxels: per agent Bernoulli(p) occupancy on (H, W, 13), seed 1234 + agent,
  or a seeded point cloud xyz ~ U(extents) for the voxelizer row;
* poses: agent i at (x, y, yaw) = (6i cos 0.7i, 6i sin 0.7i, 0.15 i);
  trans_matrices[b, i, j] = T_i^-1 T_j (4x4 float32, maps agent j's frame into
  agent i's frame);
* num_agent_tensor[b, :] = number of live agents (padded agents are all-zero).
"""
import math

import numpy as np
import torch


def agent_pose(i):
    x = 6.0 * i * math.cos(0.7 * i)
    y = 6.0 * i * math.sin(0.7 * i)
    yaw = 0.15 * i
    c, s = math.cos(yaw), math.sin(yaw)
    T = np.eye(4, dtype=np.float64)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = c, -s, s, c
    T[0, 3], T[1, 3] = x, y
    return T


def make_trans_matrices(batch_size, num_agent, jitter_seed=None):
    """[B, A, A, 4, 4] float32; entry [b, i, j] = T_i^-1 T_j."""
    poses = [agent_pose(i) for i in range(num_agent)]
    out = np.zeros((batch_size, num_agent, num_agent, 4, 4), dtype=np.float32)
    rng = np.random.RandomState(jitter_seed) if jitter_seed is not None else None
    for b in range(batch_size):
        ps = poses
        if rng is not None:
            ps = []
            for T in poses:
                J = np.eye(4)
                a = rng.uniform(-0.05, 0.05)
                J[0, 0], J[0, 1], J[1, 0], J[1, 1] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
                J[0, 3], J[1, 3] = rng.uniform(-1, 1, size=2)
                ps.append(T @ J)
        for i in range(num_agent):
            Ti_inv = np.linalg.inv(ps[i])
            for j in range(num_agent):
                out[b, i, j] = (Ti_inv @ ps[j]).astype(np.float32)
    return torch.from_numpy(out)


def make_bevs(batch_size, num_agent, map_hw=256, z=13, p=0.02, live=None):
    """Agent-major occupancy stack [A*B, 1, H, W, Z] float32 (image index =
    agent * B + b, the layout the reference's tools build with torch.cat)."""
    per_agent = []
    for a in range(num_agent):
        g = torch.Generator().manual_seed(1234 + a)
        occ = (torch.rand(batch_size, 1, map_hw, map_hw, z, generator=g) < p).to(torch.float32)
        if live is not None:
            for b in range(batch_size):
                if a >= int(live[b]):
                    occ[b].zero_()
        per_agent.append(occ)
    return torch.cat(per_agent, 0)


def make_scene_batch(batch_size=4, num_agent=5, map_hw=256, live=None, jitter_seed=None):
    """Returns (bevs, trans_matrices, num_agent_tensor) shaped like the
    reference's CoDetModule.step inputs."""
    if live is None:
        live = [num_agent] * batch_size
    bevs = make_bevs(batch_size, num_agent, map_hw, live=live)
    trans = make_trans_matrices(batch_size, num_agent, jitter_seed)
    num_agent_tensor = torch.tensor([[int(n)] * num_agent for n in live], dtype=torch.int64)
    return bevs, trans, num_agent_tensor


def make_point_cloud(n_points=60000, seed=0, extents=((-32.0, 32.0), (-32.0, 32.0), (-3.0, 2.0)),
                     boundary_cases=True):
    """Seeded cloud [N, 4] float32 (x, y, z, intensity).  Slightly over-scans
    the extents so the strict range filter has something to reject, and
    appends points lying exactly on voxel and extent boundaries."""
    rng = np.random.RandomState(seed)
    lo = np.array([e[0] for e in extents]) - 1.0
    hi = np.array([e[1] for e in extents]) + 1.0
    pts = rng.uniform(lo, hi, size=(n_points, 3))
    if boundary_cases:
        vs = np.array([0.25, 0.25, 0.4])
        k = rng.randint(-130, 130, size=(512, 3))
        on_edges = k * vs                      # exactly on voxel boundaries (in fp64)
        ext = np.array([[extents[0][0], 0.1, 0.1], [extents[0][1], 0.1, 0.1],
                        [0.1, extents[1][0], 0.1], [0.1, extents[1][1], 0.1],
                        [0.1, 0.1, extents[2][0]], [0.1, 0.1, extents[2][1]],
                        [np.nextafter(np.float32(extents[0][1]), np.float32(0)), 0.0, 0.0],
                        [np.nextafter(np.float32(extents[0][0]), np.float32(0)), 0.0, 0.0],
                        [0.0, 0.0, np.nextafter(np.float32(extents[2][1]), np.float32(0))],
                        [0.0, 0.0, np.nextafter(np.float32(extents[2][0]), np.float32(0))]])
        pts = np.concatenate([pts, on_edges, ext], 0)
    inten = rng.uniform(0, 1, size=(pts.shape[0], 1))
    return np.concatenate([pts, inten], 1).astype(np.float32)


def randomize_bn_stats(model, seed=7):
    """Random-init checkpoints have trivial BatchNorm statistics; give eval-mode
    BN something to do: mean ~ N(0, 0.1), var ~ U(0.5, 1.5),
    gamma ~ U(0.8, 1.2), beta ~ N(0, 0.05)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            n = m.num_features
            m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(n, generator=g) + 0.5)
            with torch.no_grad():
                m.weight.copy_(torch.rand(n, generator=g) * 0.4 + 0.8)
                m.bias.copy_(torch.randn(n, generator=g) * 0.05)


def make_sparse_scene_batch(batch_size=4, num_agent=5, map_hw=256, z=13, p=0.02):
    """The on-disk form of the reference's samples: per image a sorted [M, 3]
    int32 voxel index list (what V2XSimDet.__getitem__ scatters into the dense
    grid).  Returns (indices [Mtot, 3] int32, offsets [A*B + 1] int32) in the
    agent-major image order, plus the matching dense bevs for checking."""
    bevs = make_bevs(batch_size, num_agent, map_hw, z, p)
    lists, offsets = [], [0]
    for g in range(bevs.shape[0]):
        idx = torch.nonzero(bevs[g, 0]).to(torch.int32)      # row-major = lexsort(x, y, z)
        lists.append(idx)
        offsets.append(offsets[-1] + idx.shape[0])
    return torch.cat(lists, 0).contiguous(), torch.tensor(offsets, dtype=torch.int32), bevs


def make_train_targets(n_images, map_hw=256, anchors=6, code=6, seed=17, p_fg=0.01, p_ignore=0.005):
    """Seeded stand-ins for what V2XSimDet yields next to the voxels (upstream
    V2XSimDet.__getitem__: labels, reg_targets, reg_loss_mask): one-hot (bg, vehicle) labels per
    anchor [N, H*W*A, 2] (an all-zero row = don't care), box-code regression targets
    [N, H, W, A, 1, code] and the positive-anchor mask [N, H, W, A, 1]."""
    g = torch.Generator().manual_seed(seed)
    n = n_images * map_hw * map_hw * anchors
    fg = torch.rand(n, generator=g) < p_fg
    ignore = torch.rand(n, generator=g) < p_ignore
    labels = torch.stack([(~fg).float(), fg.float()], -1)
    labels[ignore & ~fg] = 0
    reg_targets = (torch.randn(n, code, generator=g) * 0.5) * fg[:, None]
    return (labels.view(n_images, -1, 2),
            reg_targets.view(n_images, map_hw, map_hw, anchors, 1, code),
            fg.view(n_images, map_hw, map_hw, anchors, 1))
