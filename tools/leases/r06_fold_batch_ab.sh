#!/bin/bash
# Round 6: the fold / finish launches with eight partials in flight per lane (DN_FOLD_BATCH = 8; bn_update_running and bn_param_grad
# fetch eight groups at a time) against the previous build (tools/ab/FOLD_old: train_ops.hip of the commit before, same other
# objects): bit-for-bit parity tests of the shipped build, then the training step interleaved -> gpurun_out/r06/fold_batch_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py -q -m gpu -k "bn or fold or bias or golden or oracle or repeat" 2>&1 | tail -3 > $O/fold_batch_ab.txt
export DISCONET_ALLOW_STALE_LIB=1
for rep in 1 2 3; do
  for v in old new; do
    echo -n "fold=$v " >> $O/fold_batch_ab.txt
    DISCONET_HIP_LIB=$R/tools/ab/FOLD_$v/libdisconet_hip.so timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], repr(d['loss_last']), d['range_flags'])" >> $O/fold_batch_ab.txt
  done
done
cat $O/fold_batch_ab.txt
