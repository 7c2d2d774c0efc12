#!/bin/bash
# round 4, lease 15: bit grid (uniform tile coordinates in its kernel) vs hi-only planes; every kernel's tile coordinates
# through v_readfirstlane (DN_UNIFORM_TILE=1) vs not -- interleaved in one lease; the bit-grid tests first
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r04 gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_conv.py tests/test_gpu_model.py -q -m gpu -x -k "bit or scatter or split_planar" 2>&1 | tail -3
cp disconet_amd/libdisconet_hip.so /tmp/lib_keep.so
run() {  # name lib form
  cp tools/ab/$2/libdisconet_hip.so disconet_amd/libdisconet_hip.so
  DN_BEV_FORM=$3 timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg 2>gpurun_out/ab/$1.err | tail -1 > gpurun_out/ab/$1.json
  python3 -c "
import json; r=json.load(open('gpurun_out/ab/$1.json')); print('%-22s value %.1f  median-of-5 %.1f  conv %.4f ms  others %s' % ('$1', r['value'], r['repeat']['scenes_per_s']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['other_kernels_ms_per_step']))"
}
for r in 1 2 3; do
  run u0_bits_$r DN_UNIFORM_TILE_0 bits
  run u0_hi_$r DN_UNIFORM_TILE_0 hi
  run u1_bits_$r DN_UNIFORM_TILE_1 bits
done
cp /tmp/lib_keep.so disconet_amd/libdisconet_hip.so
