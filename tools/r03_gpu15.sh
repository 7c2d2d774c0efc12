# four-wave attention launch: bitwise vs the one-wave form, then the step and the emulated share with both
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_fusion.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for w in 4 1; do
  DN_FUSE_MLP_WAVES=$w timeout 200 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>gpurun_out/r03_bench15_w$w.err | tail -1 > gpurun_out/r03_bench15_w$w.json
  echo "waves $w:"; grep -E "fuse_mlp|warp" gpurun_out/r03_bench15_w$w.err | head -4
  python3 -c "
import json; r=json.load(open('gpurun_out/r03_bench15_w$w.json')); print(r['value'], r['ms_per_step'])"
  DN_FUSE_MLP_WAVES=$w timeout 200 python bench.py --mode agent --no-pg --emulate-world 8 --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; r=json.load(sys.stdin); print('share', r['emulated_share']['ms_per_step'], r['emulated_share']['projected_speedup'], r['emulated_share']['phases_us'])"
done
