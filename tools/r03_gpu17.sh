# epilogue with buffer stores / uniform ReLU floor / second output behind one branch: correctness of every tile, then the step
mkdir -p gpurun_out
( timeout 400 tools/sp_conv_check.bin 20 all tiles > gpurun_out/r03_spcheck17.log 2>&1; echo "spcheck rc $?" >> gpurun_out/r03_spcheck17.log )
grep -c FAIL gpurun_out/r03_spcheck17.log; tail -2 gpurun_out/r03_spcheck17.log | cut -c1-200
timeout 500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 200 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>gpurun_out/r03_bench17.err | tail -1 > gpurun_out/r03_bench17.json
grep "^\[sp\]" gpurun_out/r03_bench17.err | cut -c1-80
python3 -c "
import json; r=json.load(open('gpurun_out/r03_bench17.json')); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms_per_step'], r['roofline']['frac'])"
