"""HBM traffic of the conv kernel per launch from the rocprofv3 counter passes
(tools/pmc_run.sh): FETCH_SIZE (KiB, x2 on gfx950 per MI355X_MICROARCH.md) and
WRITE_SIZE (KiB) of the last step's conv_mfma_kernel launches.

    python tools/pmc_traffic.py gpurun_out/pmc f16x3 22 > profiles/r01_pmc_traffic_f16x3.json
"""
import csv
import json
import sys


def per_dispatch(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "conv_mfma_kernel" in r["Kernel_Name"]:
            k = int(r["Dispatch_Id"])
            out[k] = out.get(k, 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def main(d, math, launches):
    fetch = per_dispatch("%s/pmc2.csv" % d, "FETCH_SIZE")[-launches:]
    write = per_dispatch("%s/pmc3.csv" % d, "WRITE_SIZE")[-launches:]
    assert len(fetch) == launches and len(write) == launches, (len(fetch), len(write))
    fb, wb = 2 * 1024 * sum(fetch), 1024 * sum(write)
    print(json.dumps({
        "round": 1, "kernel": "conv_mfma_kernel", "conv_math": math, "launches_per_step": launches,
        "fetch_bytes_per_step": fb, "write_bytes_per_step": wb,
        "hbm_bytes_per_launch": (fb + wb) / launches,
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 2 "
                "--no-graph (tools/pmc_run.sh); FETCH_SIZE doubled per the gfx950 correction in "
                "MI355X_MICROARCH.md; WRITE_SIZE as reported"}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
