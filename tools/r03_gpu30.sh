# four-wave attention launch with the shared weight ring: bitwise vs the one-wave form, then timings at both sizes
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for w in 1 4; do
  DN_FUSE_MLP_WAVES=$w timeout 200 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>&1 >/dev/null | grep -E "fuse_mlp" | sed "s/^/waves $w: /"
done
timeout 200 python bench.py --mode agent --no-pg --emulate-world 8 --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; r=json.load(sys.stdin); print('share', r['emulated_share']['ms_per_step'], r['emulated_share']['projected_speedup'], r['emulated_share']['phases_us'])"
