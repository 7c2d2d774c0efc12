"""Static guard for the store-data hazard of DESIGN.md 3.6 (C): gfx950 needs wait states between a buffer store of
more than 64 bits and a VALU write of its data registers; hipcc (ROCm 7.2) omits them when the store's soffset operand
is an SGPR.  Round 3's scalar-offset epilogue hit exactly that (lo pieces wrong in lanes 12-15 / 28-31); round 4 showed it
on the GPU with four variants (tools/soff, profiles/r04_soffset_hazard.txt).  The shipped library must not contain such
a site: tools/soff/scan_isa.py disassembles every gfx950 code object of libdisconet_hip.so and looks for them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    sys.path.insert(0, os.path.join(ROOT, "tools", "soff"))
    import scan_isa
    return scan_isa


def test_scanner_finds_the_round3_pattern_and_accepts_the_guarded_forms():
    scan = _scanner()
    bad = """
0000000000001000 <kernel_a>:
	buffer_store_dwordx4 v[38:41], v137, s[44:47], s36 offen
	v_add_u32_e32 v38, s91, v156
"""
    guarded = """
0000000000001000 <kernel_b>:
	buffer_store_dwordx4 v[38:41], v137, s[44:47], s36 offen
	s_nop 1
	v_add_u32_e32 v38, s91, v156
	buffer_store_dwordx4 v[50:53], v54, s[44:47], 0 offen
	v_add_u32_e32 v50, s95, v54
	buffer_store_dwordx2 v[60:61], v54, s[44:47], s3 offen
	v_mov_b32_e32 v60, 0
"""
    assert len(scan.scan_text(bad)) == 1
    assert scan.scan_text(guarded) == []        # wait states present / literal soffset (compiler's job) / 64-bit data


def test_shipped_library_has_no_unguarded_store_data_site():
    from disconet_amd.csrc import build as _build
    scan = _scanner()
    lib = _build.build(verbose=False)
    sites, nobj = [], 0
    import subprocess
    import tempfile
    for co in scan.code_objects(lib):
        nobj += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            sites += scan.scan_text(subprocess.check_output([scan.OBJDUMP, "-d", "--no-show-raw-insn", f.name], text=True))
    assert nobj >= 10
    assert sites == [], sites[:3]
