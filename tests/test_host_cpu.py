"""Host-side logic that needs no GPU: the C-ABI library loads and exports every
symbol include/disconet_hip.h declares, the class surface / checkpoint names
match the reference's, and the product path fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import ROOT


def _header_functions():
    text = "".join(open(os.path.join(ROOT, "include", f)).read()
                   for f in ("disconet_hip.h", "disconet_train.h", "disconet_seg.h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from disconet_amd import _lib
    names = _header_functions()
    assert len(names) >= 12
    lib = _lib.load()
    for n in names:
        assert n in _lib.SIGNATURES, "binding missing for %s" % n
        assert getattr(lib, n) is not None
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.dn_version() >= 100


def test_library_build_id_is_the_trees_hash(tmp_path):
    """the shipped binary must be what HEAD builds.  The id baked into the library is the
    SHA-256 over every csrc/*.hip|*.inl|*.h, include/*.h and the compiler flags; a stale library refuses to load."""
    import shutil
    import subprocess
    import sys
    from disconet_amd import _lib
    from disconet_amd.csrc import build as _build
    files = _build.tree_files()
    for must in ("disconet_amd/csrc/conv_pre_pair.inl", "disconet_amd/csrc/sp_device.h", "disconet_amd/csrc/conv_sp.hip",
                 "include/disconet_hip.h", "include/disconet_train.h", "include/disconet_seg.h"):
        assert must in files, must
    want = _build.tree_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", want)
    assert _lib.load().dn_build_id().decode() == want        # through the C ABI
    assert _build.built_id() == want                         # from the file, without loading it
    assert _build.tree_hash(("-DX=1",)) != want              # a variant's flags are part of its id
    # a library whose id is not the tree's never runs: DISCONET_NO_AUTOBUILD=1 -> load() raises; a failing rebuild raises too
    # (never the stale binary); DISCONET_ALLOW_STALE_LIB=1 / DISCONET_HIP_LIB opt in to a variant
    pkg = tmp_path / "disconet_amd"
    shutil.copytree(os.path.join(ROOT, "disconet_amd"), pkg, ignore=shutil.ignore_patterns("build", "__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    with open(pkg / "csrc" / "conv_pre_pair.inl", "a") as f:
        f.write("\n// edited after the build\n")
    code = "from disconet_amd import _lib; _lib.load(); print('loaded')"
    env = {k: v for k, v in os.environ.items() if k not in ("DISCONET_HIP_LIB", "DISCONET_ALLOW_STALE_LIB", "DISCONET_NO_AUTOBUILD")}
    run = lambda extra: subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=dict(env, **extra), capture_output=True, text=True)
    r = run({"DISCONET_NO_AUTOBUILD": "1"})
    assert r.returncode != 0 and "stale" in r.stderr and "loaded" not in r.stdout, r.stderr[-400:]
    r = run({"HIPCC": "/bin/false"})                     # the automatic rebuild cannot succeed: still no stale binary
    assert r.returncode != 0 and "rebuild failed" in r.stderr and "loaded" not in r.stdout, r.stderr[-400:]
    r = run({"DISCONET_ALLOW_STALE_LIB": "1"})
    assert r.returncode == 0 and "loaded" in r.stdout, r.stderr[-400:]
    # a deployment that ships the built library WITHOUT csrc/*.hip: nothing to hash or rebuild from -- the
    # library's baked id stands and the import neither fails nor spawns hipcc; without a library it still refuses
    for f in (pkg / "csrc").glob("*.hip"):
        f.unlink()
    r = run({"HIPCC": "/bin/false"})
    assert r.returncode == 0 and "loaded" in r.stdout and "rebuilding" not in r.stderr, r.stderr[-400:]
    (pkg / "libdisconet_hip.so").unlink()
    r = run({})
    assert r.returncode != 0 and "no sources" in r.stderr, r.stderr[-400:]


def test_struct_layouts_match_header():
    from disconet_amd import _lib
    assert ctypes.sizeof(_lib.ConvDesc) == 14 * 4
    assert ctypes.sizeof(_lib.MlpTailParams) == 10 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.Post1x1Desc) == 6 * 4
    assert ctypes.sizeof(_lib.FuseMlpParams) == 9 * ctypes.sizeof(ctypes.c_void_p)


def test_argument_errors_are_reported_not_thrown():
    """Error behaviour of the C ABI: negative return + dn_last_error(), no abort."""
    from disconet_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.n_images, d.h_in, d.w_in, d.c0, d.c_out, d.ksize, d.stride = 1, 8, 8, 32, 32, 5, 1
    d.ld0, d.ldo = 32, 32
    assert lib.dn_conv_packed_weight_floats(ctypes.byref(d)) == 0
    assert b"ksize" in lib.dn_last_error()
    rc = lib.dn_conv2d(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc == -1
    rc = lib.dn_warp_neighbors(None, None, None, 1, 2, 32, 32, 256, 0, 0, 2, None, None)
    assert rc == -1 and b"null" in lib.dn_last_error()
    dims = (ctypes.c_int * 3)(256, 256, 13)
    assert lib.dn_voxel_compact_workspace(dims) == ((256 * 256 * 13 + 1023) // 1024) * 4
    vs = (ctypes.c_double * 3)(0.25, 0.25, 0.4)
    ext = (ctypes.c_double * 6)(-32, 32, -32, 32, -3, 2)
    bad = (ctypes.c_int * 3)(256, 256, 12)
    assert lib.dn_voxelize_occupy(None, 0, 4, vs, ext, bad, ctypes.c_void_p(16), None) == -1
    assert b"dims" in lib.dn_last_error()


def test_packed_weight_image_sizes():
    """size queries are host arithmetic (no GPU): 9 tap blocks per 16-channel chunk, chunks padded to a
    multiple of 4; a 3x3 stride-1 layer over a nearest-upsampled first source packs that source's chunks
    tap-merged: 16 blocks [merged tap 4][parity class 4] (csrc/conv_spq.hip; mode 1 = the row-merged form of
    csrc/conv_sp.hip UPM: 12 blocks [row-tap 2][parity 2][tx 3]; mode 0 = plain)"""
    from disconet_amd import _lib, ops
    lib = _lib.load()
    block = lambda cout: 4 * ((cout + 63) // 64 * 64) * 16
    d = ops.conv_desc(20, 256, 256, 32, 32, 3, math="sp")
    assert lib.dn_spconv_packed_weight_bytes(ctypes.byref(d)) == 4 * 9 * block(32)
    d = ops.conv_desc(20, 32, 32, 512, 256, 3, c1=256, up0=True, math="sp")      # conv5_1
    assert lib.dn_spconv_packed_weight_bytes(ctypes.byref(d)) == (32 * 16 + 16 * 9) * block(256)
    try:
        lib.dn_spconv_set_upmode(1)
        assert lib.dn_spconv_packed_weight_bytes(ctypes.byref(d)) == (32 * 12 + 16 * 9) * block(256)
        lib.dn_spconv_set_upmode(0)
        assert lib.dn_spconv_packed_weight_bytes(ctypes.byref(d)) == 48 * 9 * block(256)
    finally:
        lib.dn_spconv_set_upmode(-1)
    d = ops.conv_desc(20, 33, 32, 512, 256, 3, c1=256, up0=True, math="sp")      # odd height: rejected
    assert lib.dn_spconv_packed_weight_bytes(ctypes.byref(d)) == 0 and b"even" in lib.dn_last_error()
    d = ops.conv_desc(20, 32, 32, 24, 64, 3, c1=8, up0=True, math="sp")          # c0 not whole chunks: rejected at pack
    assert lib.dn_spconv_packed_weight_bytes(ctypes.byref(d)) in (0, 4 * 9 * block(64))
    d = ops.conv_desc(20, 64, 64, 128, 128, 1, math="sp")
    assert lib.dn_spconv_packed_weight_bytes(ctypes.byref(d)) == 8 * 1 * block(128)


def test_reference_state_dict_names_load():
    from disconet_amd import Config, DiscoNet
    from oracle.disconet_ref import RefConfig, build_ref_model
    ref = build_ref_model(RefConfig(128), kd_flag=1, num_agent=2)
    m = DiscoNet(Config(map_hw=128), kd_flag=1, num_agent=2).eval()
    # DataParallel-style prefix + the reference's duplicate backbone parameters
    sd = {"module." + k: v for k, v in ref.state_dict().items()}
    assert "module.decoder.conv_pre_1.weight" in sd and "module.u_encoder.conv5_1.weight" in sd
    res = m.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref.state_dict()[k]), k
    with pytest.raises(RuntimeError):
        m.load_state_dict({"classification.conv9.weight": torch.zeros(1)}, strict=True)
    # the duplicate-Backbone key list is recollection (no reference source in the mount): a Backbone key it
    # does not name is dropped with a WARNING, never silently and never fatally (a real checkpoint must
    # load); a key this model actually runs that is missing still fails the strict check
    for bad in ("decoder.conv_pre_3.weight", "u_encoder.conv9_1.weight", "decoder.typo.bias"):
        with pytest.warns(UserWarning, match="dropped 1 Backbone keys"):
            res = m.load_state_dict(dict(sd, **{bad: torch.zeros(1)}), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    short = {k: v for k, v in sd.items() if k != "module.u_encoder.conv2_1.weight"}
    with pytest.raises(RuntimeError):
        m.load_state_dict(short, strict=True)
    for key in ("u_encoder.conv3d_1.conv3d.weight", "u_encoder.conv3d_2.bn3d.running_var",
                "decoder.bn8_2.weight", "pixel_weighted_fusion.conv1_4.bias",
                "classification.conv2.weight", "regression.box_prediction.3.bias"):
        assert key in m.state_dict()


def test_constructor_surface_and_config_values():
    import inspect
    from disconet_amd import Config, DiscoNet
    sig = inspect.signature(DiscoNet.__init__)
    assert list(sig.parameters)[1:] == ["config", "layer", "in_channels", "kd_flag", "num_agent",
                                        "compress_level", "only_v2i"]
    assert [sig.parameters[k].default for k in list(sig.parameters)[2:]] == [3, 13, True, 5, 0, False]
    fsig = inspect.signature(DiscoNet.forward)
    assert list(fsig.parameters)[1:] == ["bevs", "trans_matrices", "num_agent_tensor", "batch_size"]
    c = Config("train", binary=True, only_det=True)
    assert c.map_dims == [256, 256, 13] and c.voxel_size == (0.25, 0.25, 0.4)
    assert len(c.anchor_size) == 6 and c.category_num == 2 and c.box_code_size == 6
    assert np.array_equal(c.area_extents, np.array([[-32, 32], [-32, 32], [-3, 2]]))


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch):
    from disconet_amd import Config, DiscoNet, _lib, ops
    from disconet_amd.synthetic import make_scene_batch
    m = DiscoNet(Config(map_hw=128), kd_flag=0, num_agent=2).eval()
    bevs, trans, na = make_scene_batch(1, 2, 128)
    with pytest.raises(_lib.DnError):
        m(bevs, trans, na, 1)
    with pytest.raises(_lib.DnError):
        ops.voxelize_occupy(torch.zeros(4, 4), (0.25, 0.25, 0.4), np.array([[-32, 32]] * 2 + [[-3, 2]]),
                            (256, 256, 13))
    m.train()                                  # training mode: same rule, no CPU path
    with pytest.raises(_lib.DnError):
        m(bevs, trans, na, 1)
    m.eval()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdisconet_hip.so")
    with pytest.raises(_lib.DnError, match="not built"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "disconet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_synthetic_generator_is_deterministic_and_agent_major():
    from disconet_amd.synthetic import make_scene_batch, make_sparse_scene_batch
    b1, t1, n1 = make_scene_batch(2, 3, 64, live=[3, 2], jitter_seed=5)
    b2, t2, n2 = make_scene_batch(2, 3, 64, live=[3, 2], jitter_seed=5)
    assert torch.equal(b1, b2) and torch.equal(t1, t2) and torch.equal(n1, n2)
    assert b1.shape == (6, 1, 64, 64, 13) and t1.shape == (2, 3, 3, 4, 4)
    assert b1[2 * 2 + 1].sum() == 0            # agent 2 of sample 1 is padded (image = a*B + b)
    assert b1[2 * 2 + 0].sum() > 0
    # trans[b, i, j] = T_i^-1 T_j  =>  trans[b, i, j] @ trans[b, j, i] = I
    prod = t1[0, 0, 2].double() @ t1[0, 2, 0].double()
    assert torch.allclose(prod, torch.eye(4, dtype=torch.float64), atol=1e-5)
    idx, off, bevs = make_sparse_scene_batch(1, 2, 64)
    assert off[-1] == idx.shape[0] == int(bevs.sum())


def test_bench_stdout_carries_only_the_result_line(tmp_path):
    """bench.py's contract with the driver: ONE JSON line on stdout.  Libraries (RCCL's version banner at
    process-group teardown) write to the process's stdout behind Python's back, so bench.py points file
    descriptor 1 at stderr and emits the result through a saved duplicate (claim_stdout / emit)."""
    import json
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "os.write(1, b'banner from a C library\\n'); print('python noise'); bench.emit({'metric': 'x', 'value': 1.5})"
            % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == {"metric": "x", "value": 1.5}
    assert r.stdout.count("\n") == 1
    assert "banner from a C library" in r.stderr and "python noise" in r.stderr


def test_bench_gpus_n_launches_itself_under_torch_distributed_run():
    """`python bench.py --gpus N` launched bare (the driver's N = 1 command form with another N) must not stop at
    "needs torch.distributed.run": it re-runs itself as N ranks of one node and relays rank 0's line.  --dry-launch
    prints the launcher command instead of running it (missing #1)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    cmd = d["cmd"]
    assert d["dry_launch"] is True and d["gpus"] == 2
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--standalone" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and "127.0.0.1" in cmd
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "2", "--steps", "7"]            # the ranks get the caller's own arguments, minus --dry-launch
    # a rank (WORLD_SIZE set) never launches again; a rank count that contradicts --gpus is refused
    import bench
    assert bench.launcher_command(8, ["--gpus", "8"])[-2:] == ["--gpus", "8"]


def test_bench_summary_is_short_and_last():
    """the driver keeps the last 2000 characters of the line: the figures (alt_math, train_step, agent_sharded) are repeated
    without their notes in `summary`, the line's last key"""
    import json
    import bench
    losses = [71933.6816] * 10
    r = {"value": 2287.6, "n_gpus": 1, "ms_per_step": 1.7486,
         "roofline": {"frac": 0.1528, "frac_executed": 0.378, "kernel_ms_per_step": 1.6418, "achieved": 382.0, "note": "x" * 500,
                      "other_kernels_ms_per_step": {"disco_fuse_mlp": 0.075, "warp_neighbors": 0.037, "scatter": 0.013}},
         "alt_math": {"conv_math": "f32", "value": 718.0, "ms_per_step": 5.57, "dtype": "y" * 300,
                      "roofline": {"frac": 0.73, "peak": 157.3, "achieved": 114.8, "note": "z" * 500}},
         "agent_sharded": {"value": 400.0, "n_gpus": 1, "ms_per_step": 2.7, "rccl_ranks": 1, "exchanged_bytes_per_rank_per_step": 0,
                           "phases_us": {"graph_a_encode": 1.0, "allgather": 2.0, "graph_b_fuse_decode_heads": 3.0},
                           "emulated_share": {"world": 8, "ms_per_step": 0.625, "projected_speedup": 4.33,
                                              "outputs_equal_unsharded_rows": True, "note": "n" * 400}},
         "train_step": {"ms_per_step": 15.27, "losses": losses, "note": "t" * 900,
                        "all_gradients_f32": {"ms_per_step": 21.3, "losses": losses}, "with_kd": {"ms_per_step": 17.6, "kd_loss": 1.0}},
         "cpu_baseline": {"value": 1.64, "cores": 16, "parity_max_abs_err": {"cls": 1.2e-5, "loc": 1.6e-5}, "sample": "s" * 300}}
    s = bench.line_summary(r)
    assert len(json.dumps(s)) < 1900
    assert s["alt_math"]["roofline"]["frac"] == 0.73 and s["train_step"]["all_gradients_f32"]["losses"] == losses
    assert s["agent_sharded"]["emulated_share"]["projected_speedup"] == 4.33
