#!/bin/bash
# PMC rows of the opt-in one-launch warp + attention kernel (DISCONET_FUSE_WARP=1), same three passes as tools/r03_profile.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03fusewarp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DISCONET_FUSE_WARP=1
E="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-alt-math --no-cpu-baseline --no-kernel-events --train-steps 0 --no-voxelize --no-agent-leg"
timeout 250 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/r03_f1 -o p1 -- $E > $OUT/p1.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/r03_f2 -o p2 -- $E > $OUT/p2.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/r03_f3 -o p3 -- $E > $OUT/p3.log 2>&1
for i in 1 2 3; do
  f=$(find /tmp/r03_f$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc$i.csv
done
python3 $R/tools/pmc_table.py $OUT 30 > $OUT/pmc_table.txt 2>&1
grep -E "kernel|fuse|warp" $OUT/pmc_table.txt | head
