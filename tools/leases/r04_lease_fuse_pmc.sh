#!/bin/bash
# PMC passes over the eager bench for the fusion block's two launches (what the attention launch waits on): L2 hit rate, L1 traffic,
# instruction mix.  Counter collection only (kernel-trace), one group per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04fuse; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
E="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-alt-math --no-cpu-baseline --no-kernel-events --train-steps 0 --no-voxelize --no-agent-leg"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/fp$i -o p -- $E > $O/p$i.log 2>&1
  f=$(find /tmp/fp$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $O/pmc$i.csv
  tail -2 $O/p$i.log | cut -c1-200
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc*.csv")):
    agg = collections.OrderedDict()
    rows = list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if "disco_fuse_mlp" in r["Kernel_Name"] or "warp_neighbors" in r["Kernel_Name"]})
    last = set(ids[-2:])
    for r in rows:
        if int(r["Dispatch_Id"]) in last:
            k = ("fuse_mlp" if "fuse_mlp" in r["Kernel_Name"] else "warp", r["Counter_Name"])
            agg[k] = agg.get(k, 0) + float(r["Counter_Value"])
    print(f.split("/")[-1], {("%s.%s" % k): v for k, v in agg.items()})
PY
