#!/bin/bash
# Three rocprofv3 counter passes over a short eager bench run (on the GPU box):
#   tools/pmc_run.sh NAME [extra bench.py args]   -> gpurun_out/pmc_NAME/pmc{1,2,3}.csv
# Counters only with --kernel-trace (gpurun refuses --pmc with the other trace domains).
# The bench's training and voxelizer legs are off and steps run one at a time, so the last
# launches of the trace are exactly one inference step.
NAME=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-alt-math --no-cpu-baseline --no-kernel-events --train-steps 0 --no-voxelize --in-flight 1 $*"
timeout 250 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc1_$NAME -o p1 -- $B > $OUT/p1.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc2_$NAME -o p2 -- $B > $OUT/p2.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc3_$NAME -o p3 -- $B > $OUT/p3.log 2>&1
for i in 1 2 3; do
  f=$(find /tmp/pmc${i}_$NAME -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pmc$i.csv
done
ls -la $OUT
