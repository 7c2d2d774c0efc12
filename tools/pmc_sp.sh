#!/bin/bash
# rocprofv3 counter passes over tools/sp_conv_check.bin for one layer (on the GPU box):
#   tools/pmc_sp.sh LAYER_FILTER [mode]  -> gpurun_out/pmcsp_<filter>/pmc{1,2,3,4}.csv
# Counters only with --kernel-trace (gpurun refuses --pmc with the other trace domains).
F=$1; MODE=${2:-auto}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcsp_$F
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/tools/sp_conv_check.bin 20 $F $MODE"
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pa_$F -o p1 -- $B > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pb_$F -o p2 -- $B > $OUT/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/pc_$F -o p3 -- $B > $OUT/p3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pd_$F -o p4 -- $B > $OUT/p4.log 2>&1
i=1
for d in pa pb pc pd; do
  f=$(find /tmp/${d}_$F -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pmc$i.csv
  i=$((i+1))
done
ls -la $OUT
