#!/bin/bash
# Round-4 profile set (on the GPU box):  tools/profile_set.sh $RD  -> gpurun_out/${RD}prof/...
#   1. rocprofv3 --kernel-trace --stats over the default bench command (graph replay, one stream)
#   2. three --pmc passes over the eager bench (busy counters / FETCH_SIZE / WRITE_SIZE), kernel-trace only
RD=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${RD}prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
V=$(python3 -c "import sys; sys.path.insert(0, '$R'); from disconet_amd import _lib; print(_lib.load().dn_version())")
B="python $R/bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-kernel-events --no-agent-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${RD}_kt -o kt -- $B > $OUT/kt.log 2>&1
for f in kernel_stats kernel_trace; do
  p=$(find /tmp/${RD}_kt -name "*${f}.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/$f.csv
done
python3 $R/tools/trace_last_step.py $OUT/kernel_trace.csv > $OUT/step_timeline.txt 2>&1
python3 $R/tools/layers_from_trace.py $OUT/kernel_trace.csv > $OUT/bench_layers.txt 2>&1
E="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-alt-math --no-cpu-baseline --no-kernel-events --train-steps 0 --no-voxelize --no-agent-leg"
timeout 250 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/${RD}_p1 -o p1 -- $E > $OUT/p1.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${RD}_p2 -o p2 -- $E > $OUT/p2.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/${RD}_p3 -o p3 -- $E > $OUT/p3.log 2>&1
for i in 1 2 3; do
  f=$(find /tmp/${RD}_p$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc$i.csv
done
python3 $R/tools/pmc_table.py $OUT 29 > $OUT/pmc_table.txt 2>&1
python3 $R/tools/pmc_traffic.py $OUT sp conv_sp_kernel,conv_spq_kernel,conv_pre_pair_kernel 20 19 > $OUT/pmc_traffic_sp.json 2> $OUT/pmc_traffic.err
python3 $R/tools/rocprof_conv.py $OUT/kernel_trace.csv conv_sp_kernel,conv_spq_kernel,conv_pre_pair_kernel 19 ${RD#r0} $V > $OUT/rocprof_conv_sp.json 2> $OUT/rocprof_conv.err
# 3. segmentation task (configs[3]): FETCH_SIZE / WRITE_SIZE of its conv launches, same method
S="python $R/bench.py --task seg --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --train-steps 0"
mkdir -p $OUT/seg
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${RD}_s2 -o p2 -- $S > $OUT/seg/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${RD}_s3 -o p3 -- $S > $OUT/seg/p3.log 2>&1
for i in 2 3; do
  f=$(find /tmp/${RD}_s$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/seg/pmc$i.csv
done
python3 $R/tools/pmc_traffic.py $OUT/seg seg conv_sp_kernel,conv_spq_kernel > $OUT/pmc_traffic_seg.json 2> $OUT/pmc_traffic_seg.err
# 4. training step (eager launches), timed alone: per-kernel totals of 10 steps of the default step (tools/train_step_probe.py: the first
#    is the fp32 calibration pass), grouped into DESIGN.md section 8's rows by tools/train_groups.py
T="python $R/tools/train_step_probe.py --dgrad sp --wgrad sp --steps 8"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${RD}_tr -o tr -- $T > $OUT/train.log 2>&1
p=$(find /tmp/${RD}_tr -name "*kernel_stats.csv" | head -1); [ -n "$p" ] && cp "$p" $OUT/train_step_kernel_stats.csv
python3 $R/tools/train_groups.py $OUT/train_step_kernel_stats.csv 10 > $OUT/train_groups.json 2> $OUT/train_groups.err
ls -la $OUT
