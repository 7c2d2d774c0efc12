#!/bin/bash
# Round 6: the folds of a backward pass's bias gradients as ONE launch behind the pass (dn_channel_sum_fold_multi,
# train_ops.DeferredFolds) against a fold behind every sum (DN_TRAIN_DEFER_FOLDS=0): bit-for-bit test, then the training step
# interleaved in one lease -> gpurun_out/r06/defer_folds_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py tests/test_c_abi.py -q -m gpu -k "folds or one_launch or c_abi or abi" 2>&1 | tail -3 > $O/defer_folds_ab.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "defer=$v " >> $O/defer_folds_ab.txt
    DN_TRAIN_DEFER_FOLDS=$v timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], repr(d['loss_last']), d['range_flags'])" >> $O/defer_folds_ab.txt
  done
done
cat $O/defer_folds_ab.txt
