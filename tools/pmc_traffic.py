"""HBM traffic of the conv kernel per launch from the rocprofv3 counter passes (tools/profile_set.sh): FETCH_SIZE (KiB, x2 on gfx950 per MI355X_MICROARCH.md) and WRITE_SIZE (KiB) of the LAST
step's conv launches (the eager bench launches one step at a time).

    python tools/pmc_traffic.py gpurun_out/r02prof sp conv_sp_kernel > profiles/r02_pmc_traffic_sp.json
"""
import csv
import json
import sys


def per_dispatch(path, counter, kernel):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in kernel.split(",")):
            k = int(r["Dispatch_Id"])
            out[k] = out.get(k, 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def main(d, math, kernel, launches=0, layers=0, rnd=None):
    """launches: conv kernel launches of ONE step (a K-sliced layer counts twice: main + fix-up pass); 0 = the run
    was 3 identical eager steps (rounds 1-3).  layers: LAYER launches of one step (what bench.py's algorithmic bytes per
    launch are divided by); 0 = launches"""
    if rnd is None:      # gpurun_out/r05prof[/seg] -> 5
        import re
        m = re.search(r"r(\d+)prof", d)
        rnd = int(m.group(1)) if m else 0
    fetch = per_dispatch("%s/pmc2.csv" % d, "FETCH_SIZE", kernel)
    write = per_dispatch("%s/pmc3.csv" % d, "WRITE_SIZE", kernel)
    launches = launches or len(fetch) // 3
    assert len(fetch) == len(write) and len(fetch) % launches == 0, (len(fetch), len(write), launches)
    fetch, write = fetch[-launches:], write[-launches:]
    fb, wb = 2 * 1024 * sum(fetch), 1024 * sum(write)
    print(json.dumps({
        "round": rnd, "kernel": kernel, "conv_math": math, "launches_per_step": layers or launches,
        "fetch_bytes_per_step": fb, "write_bytes_per_step": wb,
        "kernel_launches_per_step": launches, "hbm_bytes_per_launch": (fb + wb) / (layers or launches),
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the eager bench "
                "(tools/profile_set.sh); FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; "
                "WRITE_SIZE as reported; last step's launches"}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "conv_mfma_kernel",
         int(sys.argv[4]) if len(sys.argv) > 4 else 0, int(sys.argv[5]) if len(sys.argv) > 5 else 0)
