#!/bin/bash
# round 4, lease 14: the bit-grid source of conv_pre_1 (tests + A/B against the hi-only planes) and the heads' 1x1 weight
# fragments read per tile vs held in registers (A/B), all in one lease
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r04 gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_conv.py tests/test_gpu_model.py -q -m gpu -x -k "bit or scatter or split_planar or fused_1x1 or heads or stride1 or default_init" 2>&1 | tail -5
cp disconet_amd/libdisconet_hip.so /tmp/lib_keep.so
run() {  # name lib form
  cp tools/ab/$2/libdisconet_hip.so disconet_amd/libdisconet_hip.so
  DN_BEV_FORM=$3 timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg 2>gpurun_out/ab/$1.err | tail -1 > gpurun_out/ab/$1.json
  python3 -c "
import json; r=json.load(open('gpurun_out/ab/$1.json')); print('%-22s value %.1f  median-of-5 %.1f  conv %.4f ms  others %s' % ('$1', r['value'], r['repeat']['scenes_per_s']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['other_kernels_ms_per_step']))"
}
for r in 1 2 3; do
  run w2tile_bits_$r DN_HEADS_W2_REGS_0 bits
  run w2tile_hi_$r DN_HEADS_W2_REGS_0 hi
  run w2regs_bits_$r DN_HEADS_W2_REGS_1 bits
done
cp /tmp/lib_keep.so disconet_amd/libdisconet_hip.so
