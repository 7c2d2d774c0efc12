"""Regenerates the golden fixtures from the CPU oracle (oracle/).

    python tests/golden/make_golden.py

PARITY UNPINNED: /root/reference contains no source, tests or vectors for this
path, so these vectors pin the *oracle* (this repo's
restatement of SURVEY.md Appendix A) against drift -- torch version, refactors
-- not the reference itself.  Fixtures are data only: expected outputs (small
strided slices) for inputs that tests/cases.py re-creates from seeds.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.disconet_ref import feature_transformation  # noqa: E402
from oracle.voxel_ref import voxelize_occupy  # noqa: E402
from tests import cases  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    # 1. voxelizer
    pts = cases.voxel_cloud()
    dense, idx = voxelize_occupy(pts, cases.VOXEL_SIZE, cases.EXTENTS, return_indices=True)
    np.savez_compressed(os.path.join(HERE, "voxel_20k.npz"),
                        pts_sha256=sha(pts), indices=idx.astype(np.int32),
                        dense_sha256=sha(dense), n_occupied=int(dense.sum()))
    print("voxel:", pts.shape, "->", idx.shape, "occupied", int(dense.sum()))

    # 2. warp unit cases
    feat = cases.warp_feature()
    out = {}
    for name, pose in cases.WARP_POSES.items():
        com = feat.unsqueeze(0)                      # [B=1, A=1, C, H, W]
        all_warp = torch.from_numpy(pose)[None]      # [A=1, 4, 4]
        w = feature_transformation(0, 0, com, all_warp, tuple(feat.shape))
        out[name] = w.numpy()
        print("warp", name, float(np.abs(out[name]).max()))
    np.savez_compressed(os.path.join(HERE, "warp_unit.npz"), **out)

    # 3. model cases
    models = {}
    store = {}
    for case, c in cases.MODEL_CASES.items():
        key = (c["map_hw"], c["agents"])
        if key not in models:
            models[key] = cases.ref_model(*key)
        outs = cases.run_ref(case, models[key])
        for name, t in outs.items():
            store["%s/%s" % (case, name)] = cases.subsample(name, t)
            store["%s/%s_absmax" % (case, name)] = np.float32(t.abs().max())
        print("model", case, {k: tuple(v.shape) for k, v in outs.items()})
    np.savez_compressed(os.path.join(HERE, "model_cases.npz"), **store)

    # 4. fusion block alone at the BASELINE map size (5 agents x [256, 32, 32])
    model = cases.ref_model(256, 5)
    feat, trans, na = cases.fusion_inputs()
    fused = cases.ref_fuse(model, feat, trans, na)
    np.savez_compressed(os.path.join(HERE, "fusion_5x256.npz"),
                        fused=fused.numpy()[:, ::4, ::2, ::2], absmax=np.float32(fused.abs().max()))
    print("fusion", tuple(fused.shape), float(fused.abs().max()))
    # 4b. the benchmarked configuration, whole model: strided slices + SHA-256 of inputs / full outputs
    outs, _ = cases.run_ref_bench_case()
    indices, offsets, bevs, trans, na = cases.bench_case_inputs()
    store = {"indices_sha256": sha(indices.numpy()), "offsets_sha256": sha(offsets.numpy()), "trans_sha256": sha(trans.numpy()),
             "threads": torch.get_num_threads()}
    for name, t in outs.items():
        store[name] = cases.subsample_bench(name, t)
        store[name + "_absmax"] = np.float32(t.abs().max())
        store[name + "_sha256"] = sha(t.numpy())        # of THIS run (thread count above): informational, not asserted
    np.savez_compressed(os.path.join(HERE, "model_256_a5.npz"), **store)
    print("bench case", {k: (tuple(v.shape), float(v.abs().max())) for k, v in outs.items()})
    # 5. training step (float64 oracle): losses and strided gradient slices, with and without KD
    from oracle.disconet_ref import RefConfig
    from oracle.teacher_ref import build_teacher
    store = {}
    for case, c in cases.TRAIN_CASES.items():
        ref = cases.ref_model(c["map_hw"], c["agents"], kd_flag=0)
        teacher = build_teacher(RefConfig(c["map_hw"]))
        for tag, t in (("det", None), ("kd", teacher)):
            losses, grads = cases.oracle_train_fp64(case, ref, t)
            store["%s/%s/losses" % (case, tag)] = np.asarray(losses)
            for n in cases.GOLDEN_GRAD_TENSORS:
                store["%s/%s/%s" % (case, tag, n)] = cases.grad_slice(grads[n])
                store["%s/%s/%s/absmax" % (case, tag, n)] = np.float64(grads[n].abs().max())
            print("train", case, tag, losses)
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **store)

    # 6. segmentation variant: logits / intermediate maps (strided) and the cross-entropy value
    store = {}
    for case in cases.SEG_CASES:
        outs, loss = cases.run_seg_ref(case)
        for name, t in outs.items():
            store["%s/%s" % (case, name)] = cases.seg_subsample(name, t)
        store["%s/loss" % case] = np.float64(loss)
        print("seg", case, {k: tuple(v.shape) for k, v in outs.items()}, "loss", loss)
    np.savez_compressed(os.path.join(HERE, "seg_cases.npz"), **store)

    seg_train_golden()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


def seg_train_golden():
    """7. segmentation training step (float64 oracle): loss and strided gradient slices.
    `python tests/golden/make_golden.py seg_train` regenerates this fixture alone."""
    store = {}
    for case, c in cases.SEG_CASES.items():
        ref = cases.seg_ref_model(c["agents"])
        loss, grads = cases.oracle_seg_train_fp64(case, ref)
        store["%s/loss" % case] = np.float64(loss)
        for n in cases.SEG_GOLDEN_GRAD_TENSORS:
            store["%s/%s" % (case, n)] = cases.grad_slice(grads[n])
            store["%s/%s/absmax" % (case, n)] = np.float64(grads[n].abs().max())
        print("seg train", case, loss)
    np.savez_compressed(os.path.join(HERE, "seg_train_step.npz"), **store)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "seg_train":
        seg_train_golden()
    else:
        main()
