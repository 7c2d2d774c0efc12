#!/bin/bash
# the fused stem launch (dn_spconv2d_pre_pair): tests, then DN_STEM_PAIR=1 vs 0 interleaved in one lease
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r04 gpurun_out/ab
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -q -m gpu -x -k "stem_pair or split_planar or bit_grid" > gpurun_out/r04/pytest_stem.log 2>&1
grep -E "passed|failed|error|Error" gpurun_out/r04/pytest_stem.log | tail -5
run() {
  DN_STEM_PAIR=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg 2>gpurun_out/ab/$1.err | tail -1 > gpurun_out/ab/$1.json
  python3 -c "
import json; r=json.load(open('gpurun_out/ab/$1.json')); print('%-12s value %.1f  median-of-5 %.1f  conv %.4f ms  launches %s others %s' % ('$1', r['value'], r['repeat']['scenes_per_s']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['kernel'][-22:], r['roofline']['other_kernels_ms_per_step']))"
}
for r in 1 2 3; do run pair1_$r 1; run pair0_$r 0; done
