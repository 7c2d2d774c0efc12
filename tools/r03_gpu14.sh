# agent-sharded projection at batch 4 / 16 (deep-regime tiles on), test of the share vs the unsharded rows
mkdir -p gpurun_out
for b in 4 16; do
  timeout 250 python bench.py --mode agent --no-pg --emulate-world 8 --agent-batch $b --steps 20 --warmup 3 2>gpurun_out/r03_bench14_b$b.err | tail -1 > gpurun_out/r03_bench14_b$b.json
  python3 - <<PY
import json
r = json.load(open("gpurun_out/r03_bench14_b$b.json"))
print("batch $b: full", r["ms_per_step"], "ms;", json.dumps(r["emulated_share"])[:600])
PY
done
timeout 500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -4
