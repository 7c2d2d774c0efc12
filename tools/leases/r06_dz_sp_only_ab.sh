#!/bin/bash
# Round 6: no fp32 copy of dz where nothing reads it (the weight gradient takes dz from its SP copy: dn_conv_wgrad_sp_z) against
# DN_TRAIN_DZ_SP_ONLY=0 DN_TRAIN_WGRAD_ZSP=0 (rounds 5's data flow): bit-for-bit tests, then the training step interleaved in one
# lease (and the middle setting: fp32 dz written, weight gradient on the SP copy) -> gpurun_out/r06/dz_sp_only_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py -q -m gpu -k "wgrad or fp32_copy_of_dz or bn_" 2>&1 | tail -3 > $O/dz_sp_only_ab.txt
for rep in 1 2 3; do
  for v in "0 0" "0 1" "1 1"; do
    set -- $v
    echo -n "dz_sp_only=$1 wgrad_zsp=$2 " >> $O/dz_sp_only_ab.txt
    DN_TRAIN_DZ_SP_ONLY=$1 DN_TRAIN_WGRAD_ZSP=$2 timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], repr(d['loss_last']), d['range_flags'])" >> $O/dz_sp_only_ab.txt
  done
done
cat $O/dz_sp_only_ab.txt
