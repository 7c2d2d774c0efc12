mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_fusion.py tests/test_gpu_sharded.py tests/test_gpu_graph.py -q -m gpu -x > gpurun_out/r03_pytest10a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest10a.log )
( timeout 120 python bench.py --no-cpu-baseline --no-alt-math --train-steps 0 --no-agent-leg --layers > gpurun_out/r03_bench10.json 2> gpurun_out/r03_bench10.err )
tail -4 gpurun_out/r03_pytest10a.log; grep "^\[sp\]" gpurun_out/r03_bench10.err | grep -v conv; tail -c 250 gpurun_out/r03_bench10.json
