#!/bin/bash
# Instruction mix of every launch of the step (eager bench, last step): VALU / MFMA / SALU / LDS / VMEM instruction counts per wave,
# issue-cycle shares.  Counter collection only, one group per pass.  -> gpurun_out/r04mix/table.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04mix; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
E="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-alt-math --no-cpu-baseline --no-kernel-events --train-steps 0 --no-voxelize --no-agent-leg"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/mx$i -o p -- $E > $O/p$i.log 2>&1
  f=$(find /tmp/mx$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $O/pmc$i.csv
done
python3 $R/tools/inst_mix_table.py $O > $O/table.txt; cat $O/table.txt
