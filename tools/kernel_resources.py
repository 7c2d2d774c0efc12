"""Register / scratch / LDS use of every kernel of one csrc unit, from hipcc's own remarks.

    python tools/kernel_resources.py conv_sp [filter] [-DMACRO=v ...]

Compiles disconet_amd/csrc/<unit>.hip for gfx950 with -Rpass-analysis=kernel-resource-usage (into /tmp, the
in-tree build is untouched) and prints one line per kernel: template arguments, VGPRs, AGPRs, spills, scratch,
waves per SIMD.  Used before / after a kernel edit to see whether it moved an instantiation over a register step
(MI355X_MICROARCH.md, register files: 128 -> 4 waves, 168 -> 3, 256 -> 2) or into scratch.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(unit, extra=()):
    src = os.path.join(ROOT, "disconet_amd", "csrc", unit + ".hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(src), "-Rpass-analysis=kernel-resource-usage",
           "-c", src, "-o", "/tmp/kernel_resources_%s.o" % unit] + list(extra)
    text = subprocess.run(cmd, capture_output=True, text=True).stderr
    out = []
    for block in re.split(r"remark: Function Name: ", text)[1:]:
        name = block.split()[0]
        fields = dict(re.findall(r"remark:\s+([A-Za-z][\w \[\]/]*?): (\w+)", block))
        out.append((name, fields))
    names = subprocess.run(["c++filt"], input="\n".join(n for n, _ in out), capture_output=True, text=True).stdout.split("\n")
    return [(d, f) for d, (_, f) in zip(names, out)]


def main():
    unit = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith("-")]
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    for name, f in resources(unit, extra):
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*$", "", short).replace("void ", "")
        if flt and not any(x in short for x in flt):
            continue
        print("%-88s VGPR %3s AGPR %3s spill %3s scratch %4s waves/SIMD %s SGPR %s" % (
            short[:88], f.get("VGPRs"), f.get("AGPRs"), f.get("VGPRs Spill"), f.get("ScratchSize [bytes/lane]"),
            f.get("Occupancy [waves/SIMD]"), f.get("TotalSGPRs")))


if __name__ == "__main__":
    main()
