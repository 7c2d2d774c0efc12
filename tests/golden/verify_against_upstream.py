#!/usr/bin/env python
"""Re-verification of the oracle against the REAL reference, for the day its source is available.

    COPERCEPTION_SRC=/path/to/coperception/checkout python tests/golden/verify_against_upstream.py
    python tests/golden/verify_against_upstream.py --self-test        # plumbing check, see below

Why this exists: /root/reference holds no source for the `--com disco` path (the `coperception`
submodule directory is empty, /root/reference/.gitmodules:1-3), so `oracle/` is this repo's
restatement of SURVEY.md Appendix A and the goldens under tests/golden/ pin the ORACLE, not the
reference -- parity is unpinned (header).  This script is what turns that into a pinned
parity: it imports the upstream package IN THE BUILD CONTAINER ONLY (nothing of it travels: the
outputs are one PASS/FAIL line per item), loads the oracle's seeded weights into the upstream
modules through their own state_dict names, runs the seeded inputs of tests/cases.py through the
upstream code and diffs against the committed golden vectors -- one line per item of SURVEY.md
Appendix C (the ten recollection risks) plus the whole-model, fusion and loss comparisons.

`--self-test` runs the same checks with the oracle standing in for the upstream package: it proves
the harness, the name map and the goldens are consistent (every line must PASS) -- it says nothing
about the reference.

Exit code: 0 all PASS, 1 any FAIL, 2 upstream not importable.
"""
import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TOL = 1e-4            # BASELINE.json north_star: fp32 maps / logits within 1e-4
RESULTS = []


def report(item, ok, detail):
    RESULTS.append((item, ok))
    print("%-6s %-5s %s" % (item, "PASS" if ok else ("SKIP" if ok is None else "FAIL"), detail), flush=True)


def check(item, what):
    """decorator: run one item, turn exceptions into a FAIL line with the reason"""
    def deco(fn):
        def run(*a, **kw):
            try:
                ok, detail = fn(*a, **kw)
                report(item, ok, "%s: %s" % (what, detail))
            except Exception as e:                                   # noqa: BLE001 - one line per item
                report(item, False, "%s: %s: %s" % (what, type(e).__name__, e))
        return run
    return deco


# ------------------------------------------------------------------------------------------------
# upstream adapter
# ------------------------------------------------------------------------------------------------
_STUBS = ["cv2", "shapely", "shapely.geometry", "mmcv", "numba", "nuscenes", "nuscenes.nuscenes",
          "nuscenes.utils", "nuscenes.utils.data_classes", "pyquaternion", "terminaltables", "matplotlib",
          "matplotlib.pyplot", "matplotlib.patches", "tqdm", "filterpy", "filterpy.kalman", "seaborn", "imageio"]


def _stub_missing_dependencies():
    """the hot path needs torch + numpy only; the package's __init__ chain may import its data /
    plotting dependencies, absent here: stand-ins that fail on USE, not on import"""
    class _Missing(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Missing(self.__name__ + "." + name)

        def __call__(self, *a, **kw):
            raise RuntimeError("%s is not installed (stubbed by verify_against_upstream.py)" % self.__name__)
    for name in _STUBS:
        try:
            importlib.import_module(name)
        except Exception:                                            # noqa: BLE001
            sys.modules[name] = _Missing(name)


def _first(mod_names, attr):
    err = None
    for m in mod_names:
        try:
            return getattr(importlib.import_module(m), attr)
        except Exception as e:                                       # noqa: BLE001
            err = e
    raise ImportError("%s not found in %s (%s)" % (attr, mod_names, err))


def load_upstream(src):
    sys.path.insert(0, src)
    _stub_missing_dependencies()
    up = {"src": src}
    up["DiscoNet"] = _first(["coperception.models.det", "coperception.models.det.DiscoNet"], "DiscoNet")
    up["Config"] = _first(["coperception.configs", "coperception.configs.Config"], "Config")
    up["voxelize_occupy"] = _first(["coperception.utils.data_util"], "voxelize_occupy")
    for name, mods in (("TeacherNet", ["coperception.models.det", "coperception.models.det.TeacherNet"]),
                       ("CoDetModule", ["coperception.utils.CoDetModule"]),
                       ("FusionBase", ["coperception.models.det.base", "coperception.models.det.base.FusionBase"])):
        try:
            up[name] = _first(mods, name)
        except ImportError:
            up[name] = None
    up["make_config"] = lambda hw: up["Config"]("train", binary=True, only_det=True)
    return up


def load_self_test():
    """the oracle under the adapter's names: harness / golden consistency only"""
    from oracle import disconet_ref, teacher_ref, voxel_ref
    return {"src": None, "DiscoNet": disconet_ref.DiscoNetRef, "Config": disconet_ref.RefConfig,
            "voxelize_occupy": voxel_ref.voxelize_occupy, "TeacherNet": teacher_ref.TeacherNetRef,
            "CoDetModule": None, "FusionBase": None, "make_config": lambda hw: disconet_ref.RefConfig(hw)}


# ------------------------------------------------------------------------------------------------
def build_pair(up, hw, agents, **kw):
    """(oracle model, upstream model with the oracle's weights) for one configuration"""
    from tests import cases
    ref = cases.ref_model(hw, agents, **kw)
    cfg = up["make_config"](hw)
    if hw != 256 and hasattr(cfg, "map_dims") and up["src"] is not None:
        # upstream's Config fixes the V2X-Sim extents (256 x 256): shrink them the way RefConfig does
        half = hw * 0.25 / 2.0
        cfg.area_extents = np.array([[-half, half], [-half, half], [-3.0, 2.0]])
        cfg.map_dims = [hw, hw, 13]
    model = up["DiscoNet"](cfg, layer=kw.get("layer", 3), kd_flag=kw.get("kd_flag", 1), num_agent=agents,
                           **{k: v for k, v in kw.items() if k in ("compress_level", "only_v2i")})
    want, have = ref.state_dict(), model.state_dict()
    missing = sorted(set(want) - set(have))
    extra = sorted(k for k in set(have) - set(want)
                   if not k.endswith("num_batches_tracked"))
    model.load_state_dict({k: v for k, v in want.items() if k in have}, strict=False)
    return ref, model.eval(), missing, extra


def run_items(up):
    from tests import cases
    g_model = np.load(os.path.join(HERE, "model_cases.npz"))
    g_warp = np.load(os.path.join(HERE, "warp_unit.npz"))
    g_vox = np.load(os.path.join(HERE, "voxel_20k.npz"))
    g_fuse = np.load(os.path.join(HERE, "fusion_5x256.npz"))

    @check("C.1", "anchor count / sizes, head channels")
    def c1():
        from disconet_amd import Config as OurConfig
        cfg, ours = up["make_config"](256), OurConfig()
        a, b = np.asarray(cfg.anchor_size, dtype=np.float64), np.asarray(ours.anchor_size, dtype=np.float64)
        same = a.shape == b.shape and np.allclose(a, b)
        ok = len(a) == 6 and cfg.category_num == 2 and cfg.box_code_size == 6 and same
        return ok, "anchors %d (ours 6), category_num %s, box_code_size %s, sizes %s" % (
            len(a), cfg.category_num, cfg.box_code_size, "equal" if same else "DIFFER: %s" % a.tolist())

    @check("C.7", "state_dict names (incl. duplicated Backbone parameters, module. prefix)")
    def c7():
        _, model, missing, extra = build_pair(up, 128, 2)
        ok = not missing
        return ok, "%d oracle names missing upstream %s; %d upstream-only names (duplicates the " \
                   "product drops on load) e.g. %s" % (len(missing), missing[:4], len(extra), extra[:3])

    @check("C.6", "Conv3D (1,1,1) + BatchNorm3d after encoder stages 1 and 2")
    def c6():
        _, model, _, _ = build_pair(up, 128, 2)
        sd = model.state_dict()
        w = sd.get("u_encoder.conv3d_1.conv3d.weight")
        bn3 = any(isinstance(m, torch.nn.BatchNorm3d) for m in model.modules())
        ok = w is not None and tuple(w.shape[2:]) == (1, 1, 1) and bn3
        return ok, "conv3d_1 weight %s, BatchNorm3d present: %s" % (None if w is None else tuple(w.shape), bn3)

    @check("C.8", "voxelize_occupy: strict extent filter, float64 floor-divide, sorted unique indices")
    def c8():
        pts = cases.voxel_cloud()
        out = up["voxelize_occupy"](pts, voxel_size=np.asarray(cases.VOXEL_SIZE), extents=cases.EXTENTS,
                                    return_indices=True) \
            if up["src"] is None else up["voxelize_occupy"](pts, voxel_size=cases.VOXEL_SIZE, extents=cases.EXTENTS,
                                                           return_indices=True)
        idx = np.asarray(out[1] if isinstance(out, tuple) else out)
        want = g_vox["indices"]
        ok = idx.shape == want.shape and np.array_equal(np.asarray(idx, dtype=np.int64), want.astype(np.int64))
        return ok, "%d voxels (golden %d), bit-exact: %s" % (len(idx), len(want), ok)

    @check("C.3", "two-pass warp: rotation then translation by (4*tx/128, -4*ty/128), zero pad between")
    def c3():
        if up["src"] is None:
            from oracle.disconet_ref import feature_transformation as ft
        else:
            base = up["FusionBase"] or _first(["coperception.models.det.base.IntermediateModelBase",
                                               "coperception.models.det.base"], "IntermediateModelBase")
            ft = getattr(base, "feature_transformation")
        feat = cases.warp_feature()
        worst = 0.0
        for name, pose in cases.WARP_POSES.items():
            got = ft(0, 0, feat.unsqueeze(0), torch.from_numpy(pose)[None], tuple(feat.shape))
            worst = max(worst, float(np.abs(np.asarray(got.detach()) - g_warp[name]).max()))
        return worst <= 1e-5, "max abs diff vs golden over %d poses: %.2e" % (len(cases.WARP_POSES), worst)

    @check("C.2", "num_agent_tensor[b, 0] = live agents; padded agents skipped (ragged scene)")
    def c2():
        return model_case("ragged_a4")

    def model_case(case):
        c = cases.MODEL_CASES[case]
        _, model, missing, _ = build_pair(up, c["map_hw"], c["agents"])
        if missing:
            return False, "cannot load the oracle's weights: %d names missing upstream" % len(missing)
        bevs, trans, na = cases.model_inputs(case)
        with torch.no_grad():
            res, x8, x7, x6, x5, fused = model(bevs, trans, na, c["batch"])
        outs = {"cls": res["cls"], "loc": res["loc"], "x8": x8, "x5": x5, "fused": fused}
        worst, where = 0.0, ""
        for name, t in outs.items():
            e = float(np.abs(cases.subsample(name, t) - g_model["%s/%s" % (case, name)]).max())
            if e > worst:
                worst, where = e, name
        return worst <= TOL, "%s: max abs diff vs golden %.2e (%s)" % (case, worst, where)

    @check("C.9", "kd_flag == 1 returns (result, x8, x7, x6, x5, fused)")
    def c9():
        c = cases.MODEL_CASES["cfg1_f0"]
        _, model, _, _ = build_pair(up, c["map_hw"], c["agents"])
        bevs, trans, na = cases.model_inputs("cfg1_f0")
        with torch.no_grad():
            out = model(bevs, trans, na, c["batch"])
        chans = [t.shape[1] for t in out[1:]]
        ok = isinstance(out, tuple) and len(out) == 6 and isinstance(out[0], dict) and chans == [32, 64, 128, 256, 256]
        return ok, "tuple of %d, channel counts after the dict: %s (want [32, 64, 128, 256, 256])" % (len(out), chans)

    @check("C.4/5", "MLP last layer ReLU without BN, exp without max-shift, self pair in the softmax")
    def c45():
        ref, model, missing, _ = build_pair(up, 256, 5)
        if missing:
            return False, "cannot load the oracle's weights"
        feat, trans, na = cases.fusion_inputs()
        # upstream has no fusion-only entry point: run its forward loop body on the given maps
        want = g_fuse["fused"]
        got = cases.ref_fuse(model, feat, trans, na) if hasattr(model, "build_local_communication_matrix") \
            else None
        if got is None:
            return None, "upstream model exposes no build_local_communication_matrix: covered by WHOLE-MODEL"
        e = float(np.abs(got.numpy()[:, ::4, ::2, ::2] - want).max())
        return e <= TOL, "fusion block at 5 x [256, 32, 32]: max abs diff vs golden %.2e" % e

    @check("C.10", "tools wrap the model in nn.DataParallel (no torch.distributed)")
    def c10():
        if up["src"] is None:
            return None, "self-test: no tool sources"
        found = {}
        for rel in ("tools/det/train_codet.py", "tools/det/test_codet.py"):
            p = os.path.join(up["src"], rel)
            txt = open(p).read() if os.path.exists(p) else ""
            found[rel] = ("DataParallel" in txt, "DistributedDataParallel" in txt or "init_process_group" in txt)
        ok = all(dp and not ddp for dp, ddp in found.values())
        return ok, str(found)

    @check("MODEL", "whole `--com disco` forward on the seeded cases vs tests/golden/model_cases.npz")
    def whole():
        lines, ok = [], True
        for case in cases.MODEL_CASES:
            o, d = model_case(case)
            ok = ok and bool(o)
            lines.append(d)
        return ok, "; ".join(lines)

    @check("LOSS", "CoDetModule losses (focal cls + masked smooth-L1 loc) vs tests/golden/train_step.npz")
    def loss():
        g = np.load(os.path.join(HERE, "train_step.npz"))
        case = "cfg1"
        c = cases.TRAIN_CASES[case]
        (bevs, trans, na), (labels, targets, mask) = cases.train_inputs(case)
        ref, model, missing, _ = build_pair(up, c["map_hw"], c["agents"], kd_flag=0)
        if up["CoDetModule"] is None:
            if up["src"] is not None:
                return None, "coperception.utils.CoDetModule not importable"
            from oracle.train_ref import det_loss
            model.train()
            out = model(bevs, trans, na, c["batch"])
            l_cls, l_loc = det_loss(out, labels, targets, mask, norm=bevs.shape[0])
            got = [float(l_cls), float(l_loc)]
        else:
            cfg = up["make_config"](c["map_hw"])
            mod = up["CoDetModule"](model.train(), None, cfg, torch.optim.Adam(model.parameters(), lr=0.0), 0)
            data = {"bev_seq": bevs, "labels": labels, "reg_targets": targets, "reg_loss_mask": mask,
                    "anchors": None, "vis_maps": None, "trans_matrices": trans, "num_agent": na}
            out = mod.step(data, c["batch"])
            got = [float(out[1]), float(out[2])] if isinstance(out, (tuple, list)) else \
                  [float(out["cls_loss"]), float(out["loc_loss"])]
        want = g["%s/det/losses" % case][:2]
        rel = max(abs(a - b) / abs(b) for a, b in zip(got, want))
        return rel <= 1e-4, "cls %.6g loc %.6g vs golden %.6g %.6g (rel %.1e)" % (got[0], got[1], want[0], want[1], rel)

    for fn in (c1, c2, c3, c45, c6, c7, c8, c9, c10, whole, loss):
        fn()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--src", default=os.environ.get("COPERCEPTION_SRC", ""),
                    help="checkout of https://github.com/coperception/coperception (default $COPERCEPTION_SRC)")
    ap.add_argument("--self-test", action="store_true", help="run the harness on the oracle itself")
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.self_test:
        up = load_self_test()
        print("SELF-TEST: the oracle stands in for upstream -- this checks the harness, not the reference")
    else:
        if not args.src or not os.path.isdir(args.src):
            print("upstream source not available: set COPERCEPTION_SRC=/path/to/coperception "
                  "(/root/reference/coperception is an empty submodule directory)")
            return 2
        try:
            up = load_upstream(args.src)
        except ImportError as e:
            print("cannot import upstream from %s: %s" % (args.src, e))
            return 2
    run_items(up)
    fails = [i for i, ok in RESULTS if ok is False]
    print("%d items, %d PASS, %d FAIL, %d SKIP" % (len(RESULTS), sum(ok is True for _, ok in RESULTS), len(fails),
                                                  sum(ok is None for _, ok in RESULTS)))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
