#!/bin/bash
# Round 6: the CB = 64 stride-2 split-f16 weight gradient with 1 x 16 against 2 x 16 output tiles (DN_WSP_S2_TH64 = 1 / 2, built by
# AB_FILES=conv_wgrad tools/ab/build.sh DN_WSP_S2_TH64 1 2): parity of the shipped build, then the training step interleaved
# -> gpurun_out/r06/wgrad_s2_tile.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -m gpu -k "wgrad" 2>&1 | tail -3 > $O/wgrad_s2_tile.txt
export DISCONET_ALLOW_STALE_LIB=1
for rep in 1 2 3; do
  for v in 1 2; do
    echo -n "th64=$v " >> $O/wgrad_s2_tile.txt
    DISCONET_HIP_LIB=$R/tools/ab/DN_WSP_S2_TH64_$v/libdisconet_hip.so timeout 300 python tools/train_step_probe.py --dgrad sp --wgrad sp --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_first'], d['loss_last'], d['range_flags'])" >> $O/wgrad_s2_tile.txt
  done
done
cat $O/wgrad_s2_tile.txt
