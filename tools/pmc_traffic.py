"""HBM traffic of the conv kernel per launch from the rocprofv3 counter passes (tools/r02_profile.sh,
tools/pmc_run.sh): FETCH_SIZE (KiB, x2 on gfx950 per MI355X_MICROARCH.md) and WRITE_SIZE (KiB) of the LAST
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


def main(d, math, kernel, steps=3):
    fetch = per_dispatch("%s/pmc2.csv" % d, "FETCH_SIZE", kernel)
    write = per_dispatch("%s/pmc3.csv" % d, "WRITE_SIZE", kernel)
    assert len(fetch) == len(write) and len(fetch) % steps == 0, (len(fetch), len(write))
    launches = len(fetch) // steps            # warm-up + timed steps, all eager and identical
    fetch, write = fetch[-launches:], write[-launches:]
    fb, wb = 2 * 1024 * sum(fetch), 1024 * sum(write)
    print(json.dumps({
        "round": 3, "kernel": kernel, "conv_math": math, "launches_per_step": launches,
        "fetch_bytes_per_step": fb, "write_bytes_per_step": wb,
        "hbm_bytes_per_launch": (fb + wb) / launches,
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the eager bench "
                "(tools/r03_profile.sh); FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; "
                "WRITE_SIZE as reported; last step's launches"}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "conv_mfma_kernel")
