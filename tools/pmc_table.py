"""Per-dispatch table from rocprofv3 --pmc CSVs (gpurun_out/pmc/pmc{1,2,3}.csv):
duration, MFMA-pipe busy share, wave wait breakdown, FETCH/WRITE MB.
GRBM_GUI_ACTIVE and the SQ counters are summed over the 8 XCDs by rocprofv3."""
import collections
import csv
import re
import sys


def load(f):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, dict(name=r['Kernel_Name'], grid=int(r['Grid_Size']),
                                 wg=int(r['Workgroup_Size']),
                                 t=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c={}))
        e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    return d


def short(n):
    m = re.search(r'conv_(mfma|sp)_kernel<([^>]*)>', n)
    if m:
        return ('sp<' if m.group(1) == 'sp' else 'conv<') + m.group(2).replace(' ', '').replace('(anonymousnamespace)::', '')[:28] + '>'
    m = re.search(r'conv_spq_kernel<([^>]*)>', n)
    if m:
        return 'spq<BN=%s>' % m.group(1)
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0].split('<')[0][-34:]


def product_only(d):
    """the step's own kernels in launch order: the asynchronous range guard (sp_range_collect + a copy) polls at moments
    that depend on event timing, so Dispatch_Ids differ between the counter passes -- rows are matched by their position
    among the step's kernels, counted from the end"""
    skip = ("sp_range_collect", "__amd_rocclr", "at::native", "elementwise_kernel")
    return [e for e in d.values() if not any(k in e['name'] for k in skip)]


def main(d, last):
    p1, p2, p3 = (product_only(load('%s/pmc%d.csv' % (d, i))) for i in (1, 2, 3))
    p1, p2, p3 = p1[-last:], p2[-last:], p3[-last:]
    for a_, b_, c_ in zip(p1, p2, p3):
        assert a_['name'] == b_['name'] == c_['name'], (a_['name'][:60], b_['name'][:60], c_['name'][:60])
    print("%-34s %7s %8s %6s %6s %6s %6s %6s %9s %9s" % (
        "kernel", "blocks", "us", "GHz", "mfma%", "wInst%", "wAny%", "act%", "fetchMB*2", "writeMB"))
    for k, e in enumerate(p1):
        c = e['c']
        gui = c.get('GRBM_GUI_ACTIVE', 0) / 8.0          # per-XCD cycles
        ghz = gui / (e['t'] * 1e3) if e['t'] else 0
        mfu = 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui) if gui else 0
        wc = c.get('SQ_WAVE_CYCLES', 1) or 1
        f = 2 * p2[k]['c'].get('FETCH_SIZE', 0) / 1024   # gfx950: x2 (MI355X guide)
        w = p3[k]['c'].get('WRITE_SIZE', 0) / 1024
        print("%-34s %7d %8.1f %6.2f %6.1f %6.1f %6.1f %6.1f %9.1f %9.1f" % (
            short(e['name']), e['grid'] // e['wg'], e['t'], ghz, mfu,
            100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_WAIT_ANY', 0) / wc,
            100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc, f, w))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc", int(sys.argv[2]) if len(sys.argv) > 2 else 38)
