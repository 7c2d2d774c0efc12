"""Static guard for the store-data hazard behind DESIGN.md 3.6 (C).

gfx950 needs two wait states between a buffer store of more than 64 bits of data and a VALU write of the
registers holding that data.  hipcc (ROCm 7.2, GCNHazardRecognizer::createsVALUHazard) inserts them -- except
when the store's soffset operand is an SGPR, a form the GCN3/Vega documentation exempts.  Round 3's "plane
offset in the scalar operand" epilogue was exactly that form and wrote ~1e-4 of its `lo` pieces wrong.

This script disassembles every gfx950 code object inside libdisconet_hip.so (or the .so given) and lists
every buffer_store_dwordx3/x4 (and format_xyz/xyzw) with an SGPR soffset whose data registers are written
by a VALU instruction fewer than two wait states later.  Exit code 1 if any site is found.
"""
import re
import struct
import subprocess
import sys
import tempfile
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([OBJCOPY, "--dump-section", ".hip_fatbin=" + fat, so_path])
        blob = open(fat, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan_text(text, need=2):
    sites = []
    kernel = "?"
    lines = []
    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw)
        if m:
            kernel = m.group(1)
            continue
        ins = raw.split("//")[0].strip()
        if ins:
            lines.append((kernel, ins))
    for i, (kern, ins) in enumerate(lines):
        m = re.match(r"buffer_store_(dwordx3|dwordx4|format_xyz|format_xyzw)\s+(\S+),\s*(\S+),\s*(\S+),\s*(\S+)", ins)
        if not m:
            continue
        data, soff = vregs(m.group(2).rstrip(",")), m.group(5).rstrip(",")
        if not re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi|ttmp\d+", soff):
            continue        # literal / inline-constant soffset: the compiler guards this form itself
        waited = 0
        for kern2, nxt in lines[i + 1:i + 6]:
            if kern2 != kern or waited >= need:
                break
            op = nxt.split()[0]
            if op == "s_nop":
                waited += int(nxt.split()[1], 0) + 1
                continue
            if op.startswith("v_") and not op.startswith("v_cmp") and len(nxt.split()) > 1:
                dst = nxt.split()[1].rstrip(",")
                if vregs(dst) & data:
                    sites.append((kern, ins, nxt, waited))
                    break
            waited += 1
    return sites


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..",
                                                            "disconet_amd", "libdisconet_hip.so")
    total, nobj = [], 0
    for co in code_objects(so):
        nobj += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text = subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", f.name], text=True)
        total += scan_text(text)
    print(f"{so}: {nobj} gfx950 code objects, {len(total)} unguarded store-data sites (SGPR soffset)")
    for kern, ins, nxt, waited in total[:40]:
        print(f"  {kern[:110]}\n     {ins}\n     {nxt}    ({waited} wait states between)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
