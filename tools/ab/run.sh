#!/bin/bash
# tools/ab/run.sh ROUNDS VARIANT_DIR...   (on the GPU box): swaps each variant's libdisconet_hip.so into the scratch copy and runs the
# default bench without extras, ROUNDS interleaved rounds; prints value / median of the repeats / conv ms per variant and round.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export DISCONET_ALLOW_STALE_LIB=1   # variant libraries carry the tree's id of ANOTHER flag set: say so (csrc/build.py)
N=$1; shift
mkdir -p gpurun_out/ab; cp disconet_amd/libdisconet_hip.so /tmp/lib_keep.so
for r in $(seq 1 $N); do for v in "$@"; do
  cp tools/ab/$v/libdisconet_hip.so disconet_amd/libdisconet_hip.so
  timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg 2>/dev/null | tail -1 > gpurun_out/ab/${v}_$r.json
  python3 -c "
import json; r=json.load(open('gpurun_out/ab/${v}_$r.json')); print('%-22s run $r: value %.1f  median-of-5 %.1f  conv %.4f ms  others %s' % ('$v', r['value'], r['repeat']['scenes_per_s']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['other_kernels_ms_per_step']))"
done; done
cp /tmp/lib_keep.so disconet_amd/libdisconet_hip.so
