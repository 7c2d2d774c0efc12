#!/bin/bash
# the intra-step side branch (conv4_1 / conv4_2 beside warp + attention as parallel branches of the captured graph) against the
# one-stream step, interleaved in one lease; every line's graph_equals_eager compares a replay with the eager one-stream step
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/ab
run() { DISCONET_OVERLAP=$2 DISCONET_UNSAFE_OVERLAP=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg 2>gpurun_out/ab/$1.err | tail -1 > gpurun_out/ab/$1.json
  python3 -c "
import json; r=json.load(open('gpurun_out/ab/$1.json')); print('%-10s value %.1f  median-of-5 %.1f  ms/step %.4f  graph_equals_eager %s  launch %s' % ('$1', r['value'], r['repeat']['scenes_per_s']['median'], r['ms_per_step'], r.get('graph_equals_eager'), r['config']['launch'][:60]))"; }
for r in 1 2 3; do run ov0_$r 0; run ov1_$r 1; done
