#!/bin/bash
# Three rocprofv3 counter passes over any command (on the GPU box):
#   tools/pmc_cmd.sh TAG <command...>   -> gpurun_out/pmc_TAG/pmc{1,2,3}.csv + table.txt
# Counters only with --kernel-trace (gpurun refuses --pmc with the other trace domains).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"
P3="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout 250 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_${TAG}_$i -o p -- "$@" > $OUT/p$i.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pmc$i.csv
  i=$((i+1))
done
python3 $R/tools/pmc_cmd_table.py $OUT > $OUT/table.txt 2>&1
cat $OUT/table.txt
