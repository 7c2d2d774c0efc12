"""Per-launch instruction mix from the two counter passes of tools/leases/r04_lease_inst_mix.sh (last 24 product launches)."""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_table import short, product_only, load   # noqa: E402


def main(d, last=24):
    p1, p2 = (product_only(load("%s/pmc%d.csv" % (d, i)))[-last:] for i in (1, 2))
    print("%-34s %7s %8s | per wave: %7s %6s %6s %6s %6s %6s | issue share of wave cycles: %5s %5s %5s %5s | %8s" % (
        "kernel", "waves", "us", "VALU", "MFMA", "SALU", "LDS", "VMrd", "VMwr", "VALU", "LDS", "VMEM", "SCA", "LDSconf%"))
    for a, b in zip(p1, p2):
        assert a["name"] == b["name"]
        c, e = a["c"], b["c"]
        wv = c.get("SQ_WAVES", 1) or 1
        wc = c.get("SQ_WAVE_CYCLES", 1) or 1
        idx = e.get("SQ_LDS_IDX_ACTIVE", 0) or 1
        print("%-34s %7d %8.1f | %16.0f %6.0f %6.0f %6.0f %6.0f %6.0f | %33.1f %5.1f %5.1f %5.1f | %8.1f" % (
            short(a["name"]), wv, a["t"], (c.get("SQ_INSTS_VALU", 0) - c.get("SQ_INSTS_MFMA", 0)) / wv, c.get("SQ_INSTS_MFMA", 0) / wv,
            c.get("SQ_INSTS_SALU", 0) / wv, c.get("SQ_INSTS_LDS", 0) / wv, c.get("SQ_INSTS_VMEM_RD", 0) / wv, c.get("SQ_INSTS_VMEM_WR", 0) / wv,
            100 * e.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * e.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * e.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
            100 * e.get("SQ_ACTIVE_INST_SCA", 0) / wc, 100 * e.get("SQ_LDS_BANK_CONFLICT", 0) / idx))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24)
