"""Per-kernel table from the three counter passes of tools/pmc_cmd.sh (last dispatch of each kernel name)."""
import collections
import csv
import re
import sys


def load(f):
    d = collections.OrderedDict()
    try:
        rows = csv.DictReader(open(f))
    except OSError:
        return d
    for r in rows:
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, dict(name=r['Kernel_Name'], grid=int(r['Grid_Size']), wg=int(r['Workgroup_Size']),
                                 t=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c={}))
        e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    return d


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)[:64]


def main(d):
    ps = [load('%s/pmc%d.csv' % (d, i)) for i in (1, 2, 3)]
    last = collections.OrderedDict()
    for k, e in ps[0].items():
        last[short(e['name']) + ' g%d' % (e['grid'] // e['wg'])] = k
    for nm, k in last.items():
        e = ps[0][k]
        c = dict(e['c'])
        for p in ps[1:]:
            if k in p:
                c.update(p[k]['c'])
        waves = c.get('SQ_WAVES', 0) or 1
        gui = c.get('GRBM_GUI_ACTIVE', 0) / 8.0
        wc = c.get('SQ_WAVE_CYCLES', 1) or 1
        if e['t'] < 8:
            continue
        print("%-70s %8.1f us %.2f GHz waves %6d | mfma busy %5.1f%% | wave: waitInst %4.1f%% waitAny %4.1f%% active %4.1f%% | "
              "per wave: VALU %6.0f SALU %5.0f MFMA %5.0f VMEMrd %4.0f wr %4.0f LDS %5.0f SMEM %4.0f | cyc/wave %8.0f" % (
                  nm, e['t'], gui / (e['t'] * 1e3) if e['t'] else 0, waves,
                  100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui) if gui else 0,
                  100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_WAIT_ANY', 0) / wc,
                  100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                  c.get('SQ_INSTS_VALU', 0) / waves, c.get('SQ_INSTS_SALU', 0) / waves, c.get('SQ_INSTS_MFMA', 0) / waves,
                  c.get('SQ_INSTS_VMEM_RD', 0) / waves, c.get('SQ_INSTS_VMEM_WR', 0) / waves,
                  c.get('SQ_INSTS_LDS', 0) / waves, c.get('SQ_INSTS_SMEM', 0) / waves, 4 * wc / waves))


if __name__ == "__main__":
    main(sys.argv[1])
