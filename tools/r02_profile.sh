#!/bin/bash
# Round-2 profile set (on the GPU box):  tools/r02_profile.sh  -> gpurun_out/r02prof/...
#   1. rocprofv3 --kernel-trace --stats over the default bench command (graph replay, one stream)
#   2. three --pmc passes over the eager bench (FETCH_SIZE / WRITE_SIZE / busy counters), kernel-trace only
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-kernel-events"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02_kt -o kt -- $B > $OUT/kt.log 2>&1
for f in kernel_stats kernel_trace; do
  p=$(find /tmp/r02_kt -name "*${f}.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/$f.csv
done
python3 $R/tools/trace_last_step.py $OUT/kernel_trace.csv > $OUT/step_timeline.txt 2>&1
E="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-alt-math --no-cpu-baseline --no-kernel-events --train-steps 0 --no-voxelize"
timeout 250 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/r02_p1 -o p1 -- $E > $OUT/p1.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/r02_p2 -o p2 -- $E > $OUT/p2.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/r02_p3 -o p3 -- $E > $OUT/p3.log 2>&1
for i in 1 2 3; do
  f=$(find /tmp/r02_p$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc$i.csv
done
python3 $R/tools/pmc_table.py $OUT > $OUT/pmc_table.txt 2>&1
python3 $R/tools/pmc_traffic.py $OUT sp conv_sp_kernel > $OUT/pmc_traffic_sp.json 2> $OUT/pmc_traffic.err
python3 $R/tools/rocprof_conv.py $OUT/kernel_trace.csv conv_sp_kernel 20 > $OUT/rocprof_conv_sp.json 2> $OUT/rocprof_conv.err
ls -la $OUT
