"""Build libdisconet_hip.so for gfx950, in-tree (it ships to the GPU box as a
built artefact; it is git-ignored).  No torch headers are needed: the library
is plain HIP behind a C ABI (include/disconet_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["common.hip", "conv_mfma.hip", "conv_sp.hip", "conv_spq.hip", "voxel.hip", "warp.hip", "fuse_tail.hip", "fuse_mlp.hip", "decode.hip",
           "conv_wgrad.hip", "train_ops.hip", "seg_ops.hip"]
LIB_PATH = os.path.join(os.path.dirname(HERE), "libdisconet_hip.so")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [
        os.path.join(HERE, "dn_internal.h"), os.path.join(HERE, "sp_layout.h"), os.path.join(HERE, "sp_device.h"), os.path.join(HERE, "warp_device.h"), os.path.join(ROOT, "include", "disconet_hip.h"),
        os.path.join(ROOT, "include", "disconet_train.h"), os.path.join(ROOT, "include", "disconet_seg.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    for s in SOURCES:
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment",
               "-I", os.path.join(ROOT, "include"), "-I", HERE, "-c", os.path.join(HERE, s),
               "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
