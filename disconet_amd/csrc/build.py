"""Build libdisconet_hip.so for gfx950, in-tree (it ships to the GPU box as a
built artefact; it is git-ignored).  No torch headers are needed: the library
is plain HIP behind a C ABI (include/disconet_hip.h).

Reproducibility (round 5).  What decides a rebuild is CONTENT, not mtimes and not a hand-kept
dependency list: `tree_hash()` is the SHA-256 over every `csrc/*.hip|*.inl|*.h`, every `include/*.h`
and the compiler flags.  It is
  * compiled into the library (`-DDN_BUILD_ID=...` on common.hip -> `dn_build_id()`),
  * the basis of every object file's key (`build/<name>.o.key` = hash of the unit's own .hip + every header /
    .inl + flags: any header or .inl edit recompiles every unit; common.o, which carries the id, is keyed by
    the whole tree),
  * checked by `_lib.load()` BEFORE the library is mapped: a library whose id is not the tree's never runs -- it is rebuilt
    (loudly, under a file lock) when hipcc is there, else the import raises; `DISCONET_NO_AUTOBUILD=1` always raises;
    `DISCONET_ALLOW_STALE_LIB=1` / `DISCONET_HIP_LIB` are for A/B runs of a variant library only.
Round 4 shipped a conv_pre_pair_kernel that HEAD's source did not build because conv_pre_pair.inl was
missing from a dependency list; that cannot happen with this scheme.
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["common.hip", "conv_mfma.hip", "conv_sp.hip", "conv_spq.hip", "voxel.hip", "warp.hip", "fuse_tail.hip", "fuse_mlp.hip", "decode.hip",
           "conv_wgrad.hip", "train_ops.hip", "seg_ops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"]
LIB_PATH = os.path.join(os.path.dirname(HERE), "libdisconet_hip.so")


def tree_files():
    """Every file whose content is part of the library: sources, headers, .inl pieces (sorted, repo-relative)."""
    files = []
    for pat in ("*.hip", "*.inl", "*.h"):
        files += glob.glob(os.path.join(HERE, pat))
    files += glob.glob(os.path.join(ROOT, "include", "*.h"))
    return sorted(os.path.relpath(f, ROOT) for f in files)


def sources_present():
    """Is this a SOURCE tree?  A deployment may ship the built library without csrc/*.hip (the .so is a build artefact):
    there is then nothing to hash and nothing to rebuild from -- _lib.load() accepts the library's baked id."""
    return all(os.path.exists(os.path.join(HERE, s)) for s in SOURCES)


_HASH_CACHE = {}


def tree_hash(extra_flags=(), unit=None):
    """16 hex digits of SHA-256(flags, (name, content) of every tree file): the library's build id.
    With `unit` (a .hip name): the key of that object file -- its own source plus every header / .inl.
    Cached per process by the files' (name, size, mtime): a second import / call does not re-read 0.5 MB of sources."""
    try:
        stamp = (tuple(extra_flags), unit, tuple((rel,) + tuple(os.stat(os.path.join(ROOT, rel))[k] for k in (6, 8))
                                                  for rel in tree_files()))
    except OSError:
        stamp = None
    if stamp is not None and stamp in _HASH_CACHE:
        return _HASH_CACHE[stamp]
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + list(extra_flags)).encode())
    for rel in tree_files():
        if unit is not None and rel.endswith(".hip") and os.path.basename(rel) != unit:
            continue
        h.update(b"\0" + rel.encode() + b"\0")
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    if stamp is not None:
        _HASH_CACHE[stamp] = h.hexdigest()[:16]
    return h.hexdigest()[:16]


def built_id(path=None):
    """The build id baked into a built library (None if it is missing or predates the id).  Read from the
    FILE (the marker string "dn-build-id:<16 hex>" of common.hip), not through dlopen: a process that already
    mapped an older build of the same path would be handed that mapping again."""
    import re
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = re.search(rb"dn-build-id:([0-9a-f]{16})", f.read())
    return m.group(1).decode() if m else None


def needs_build():
    return built_id() != tree_hash()


def _compile(job):
    cmd, obj, key, verbose = job
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(obj + ".key", "w") as f:
        f.write(key)


def build(force=False, verbose=True, extra_flags=(), lib_path=None, objdir=None):
    """Compile what is out of date and link.  `extra_flags` / `lib_path` / `objdir` build a VARIANT library
    (tools/ab): its id covers the flags, so it never passes for the tree's default build."""
    lib_path = lib_path or LIB_PATH
    key = tree_hash(extra_flags)
    if not force and built_id(lib_path) == key:
        return lib_path
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = objdir or os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for s in SOURCES:
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        try:
            with open(obj + ".key") as f:
                have = f.read().strip()
        except OSError:
            have = None
        okey = key if s == "common.hip" else tree_hash(extra_flags, unit=s)
        if not force and have == okey and os.path.exists(obj):
            continue
        cmd = [hipcc] + FLAGS + list(extra_flags) + (["-DDN_BUILD_ID=\"%s\"" % key] if s == "common.hip" else []) + [
               "-I", os.path.join(ROOT, "include"), "-I", HERE, "-c", os.path.join(HERE, s), "-o", obj]
        jobs.append((cmd, obj, okey, verbose))
    workers = max(1, min(len(jobs), int(os.environ.get("DN_BUILD_JOBS", os.cpu_count() or 4))))
    with ThreadPoolExecutor(workers) as pool:
        list(pool.map(_compile, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    got = built_id(lib_path)
    if got != key:
        raise RuntimeError("built %s reports id %r, the tree is %r" % (lib_path, got, key))
    return lib_path


def build_locked(**kw):
    """build() under an exclusive file lock: several processes (pytest workers, the ranks of a multi-process test) may find
    the library stale at once; the first one builds, the others wait and find it current."""
    import fcntl
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return build(**kw)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH, tree_hash())
