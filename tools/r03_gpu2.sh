mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_sharded.py -q -m gpu > gpurun_out/r03_pytest2a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest2a.log )
( timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest2.log )
( timeout 200 python bench.py --layers > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err; echo "bench rc $?" >> gpurun_out/r03_bench2.err )
( DISCONET_FUSE_WARP=0 timeout 120 python bench.py --no-cpu-baseline --no-alt-math --train-steps 0 --no-agent-leg --layers > gpurun_out/r03_bench2_nofw.json 2> gpurun_out/r03_bench2_nofw.err )
( timeout 200 python bench.py --mode agent --emulate-world 8 --agent-check 1000 > gpurun_out/r03_agent2.json 2> gpurun_out/r03_agent2.err; echo "agent rc $?" >> gpurun_out/r03_agent2.err )
tail -5 gpurun_out/r03_pytest2a.log; tail -5 gpurun_out/r03_pytest2.log; tail -c 300 gpurun_out/r03_bench2.json; tail -3 gpurun_out/r03_agent2.err
