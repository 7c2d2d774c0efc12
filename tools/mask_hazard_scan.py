"""Static list of the places where a kernel turns a VALU compare into control flow: `v_cmp* vcc/s[..]`
whose result is read by a scalar instruction (s_and_saveexec / s_or / s_andn2 ... exec) within a few
instructions.  These are the candidates for the co-residency mask hazard of DESIGN.md 3.6 (a
lane-dependent loop exit in the warp kernel let lanes 48..63 run an extra iteration beside MFMA-dense
waves of another kernel).  Needs only hipcc (no GPU):

    python tools/mask_hazard_scan.py > profiles/r01_mask_hazard_scan.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "disconet_amd", "csrc")
SOURCES = ["conv_mfma.hip", "voxel.hip", "warp.hip", "fuse_tail.hip", "decode.hip", "conv_wgrad.hip",
           "train_ops.hip"]
WINDOW = 4       # scalar reader within this many instructions of the compare


def kernels_of(asm):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(line.strip())
            if line.strip().startswith("s_endpgm"):
                yield cur, body
                cur = None


def scan(body):
    ins = [l for l in body if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    hits, loops = 0, 0
    for i, l in enumerate(ins):
        m = re.match(r"v_cmpx?_\w+\s+(vcc|s\[\d+:\d+\])", l)
        if not m:
            continue
        dst = m.group(1)
        for j in range(i + 1, min(i + 1 + WINDOW, len(ins))):
            r = ins[j]
            if r.startswith("s_") and dst in r.split(None, 1)[-1] and ("exec" in r or r.startswith(("s_or_b64", "s_and_b64", "s_andn2", "s_orn2"))):
                hits += 1
                # a loop exit: a backward conditional branch on exec follows soon
                if any(x.startswith("s_cbranch_exec") for x in ins[j:j + 6]):
                    loops += 1
                break
    mfma = sum(1 for l in ins if l.startswith("v_mfma"))
    return len(ins), hits, loops, mfma


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    print("%-100s %7s %9s %11s %6s" % ("kernel", "instrs", "cmp->salu", "..+branch", "mfma"))
    with tempfile.TemporaryDirectory() as tmp:
        for src in SOURCES:
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-comment", "-I",
                                   os.path.join(ROOT, "include"), "-I", CSRC, "-c", os.path.join(CSRC, src),
                                   "-o", os.path.join(tmp, "x.o"), "-save-temps=obj"], cwd=tmp,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm = open(os.path.join(tmp, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
            for name, body in kernels_of(asm):
                n, hits, loops, mfma = scan(body)
                short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                short = re.sub(r"\(anonymous namespace\)::", "", short).split("(")[0][:98]
                print("%-100s %7d %9d %11d %6d" % (short, n, hits, loops, mfma))


if __name__ == "__main__":
    main()
