# quad-merged kernel with wave-private weight blocks (fewer barriers): correctness on every up+cat shape, repeated, + timing
mkdir -p gpurun_out
( for i in 1 2 3; do timeout 200 tools/sp_conv_check.bin 20 "up" tiles; done > gpurun_out/r03_spcheck23.log 2>&1; echo "rc $?" >> gpurun_out/r03_spcheck23.log )
grep -c FAIL gpurun_out/r03_spcheck23.log; grep -E "CHECK|rc " gpurun_out/r03_spcheck23.log
grep -E "^conv|^ragged|^share" gpurun_out/r03_spcheck23.log | tail -9 | sed 's/ref.*\[quad\]/[quad]/' | cut -c1-260
timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
