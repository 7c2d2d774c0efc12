"""Per-KERNEL-NAME summary of rocprofv3 --pmc CSVs of the training step (tools/leases/r05_wgrad_pmc.sh: pmc1 = busy counters, pmc2 =
FETCH_SIZE, pmc3 = WRITE_SIZE; the last `steps` steady steps of tools/train_step_probe.py are what the dispatches of the split-f16
kernels belong to -- the calibration step runs their fp32 counterparts).  Per name: launches, mean microseconds, per-XCD clock, MFMA
pipe busy share, wave wait / active shares, LDS bank-conflict share of the LDS-active cycles, HBM fetch (x2: gfx950) and write MB per
launch.      python tools/train_pmc_table.py <dir>"""
import collections
import csv
import re
import sys


def load(f):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = d.setdefault(k, dict(name=r["Kernel_Name"], t=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, c={}))
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    return d


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "").replace(" ", ""))[:64]


def by_name(d):
    out = collections.OrderedDict()
    for e in d.values():
        out.setdefault(short(e["name"]), []).append(e)
    return out


def main(d):
    p1, p2, p3 = (by_name(load("%s/pmc%d.csv" % (d, i))) for i in (1, 2, 3))
    rows = []
    for name, es in p1.items():
        n = len(es)
        t = sum(e["t"] for e in es) / n
        c = collections.Counter()
        for e in es:
            c.update(e["c"])
        gui = c["GRBM_GUI_ACTIVE"] / 8.0 / n
        wc = c["SQ_WAVE_CYCLES"] or 1
        f = 2 * sum(e["c"].get("FETCH_SIZE", 0) for e in p2.get(name, [])) / 1024 / max(len(p2.get(name, [])), 1)
        w = sum(e["c"].get("WRITE_SIZE", 0) for e in p3.get(name, [])) / 1024 / max(len(p3.get(name, [])), 1)
        rows.append((n * t, name, n, t, gui / (t * 1e3) if t else 0, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (1024 * gui) if gui else 0,
                     100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc,
                     100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c["SQ_LDS_IDX_ACTIVE"] else 0, f, w))
    rows.sort(reverse=True)
    print("%-64s %6s %8s %6s %6s %7s %6s %8s %10s %9s" % ("kernel", "calls", "us", "GHz", "mfma%", "wInst%", "act%", "ldsConf%", "fetchMB*2", "writeMB"))
    for _, name, n, t, ghz, mfu, wi, act, lc, f, w in rows[:40]:
        print("%-64s %6d %8.1f %6.2f %6.1f %7.1f %6.1f %8.1f %10.1f %9.1f" % (name, n, t, ghz, mfu, wi, act, lc, f, w))


if __name__ == "__main__":
    main(sys.argv[1])
