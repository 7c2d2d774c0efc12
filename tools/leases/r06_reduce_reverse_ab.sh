#!/bin/bash
# Round 6: do the BatchNorm reductions (bn_stats_v4, bn_bwd_reduce_v4, channel sums) find their input in the memory-side cache when
# they walk it from the END (what the producing conv wrote last) instead of from the start?  DN_REDUCE_REVERSE = 0 / 1 builds
# (AB_FILES=train_ops tools/ab/build.sh DN_REDUCE_REVERSE 0 1), the training step interleaved in one lease
# -> gpurun_out/r06/reduce_reverse_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export DISCONET_ALLOW_STALE_LIB=1
: > $O/reduce_reverse_ab.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "reverse=$v " >> $O/reduce_reverse_ab.txt
    DISCONET_HIP_LIB=$R/tools/ab/DN_REDUCE_REVERSE_$v/libdisconet_hip.so timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], repr(d['loss_last']), d['range_flags'])" >> $O/reduce_reverse_ab.txt
  done
done
cat $O/reduce_reverse_ab.txt
