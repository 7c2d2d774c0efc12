mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_fusion.py tests/test_c_abi.py tests/test_gpu_model.py -q -m gpu -k "fused or c_host or trained_like or range_guard or split_planar_bevs" > gpurun_out/r03_pytest3a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest3a.log )
( timeout 120 python bench.py --no-cpu-baseline --no-alt-math --train-steps 0 --no-agent-leg --layers > gpurun_out/r03_bench3_fw.json 2> gpurun_out/r03_bench3_fw.err )
( DISCONET_FUSE_WARP=0 timeout 200 python bench.py --mode agent --emulate-world 8 > gpurun_out/r03_agent3_nofw.json 2> gpurun_out/r03_agent3_nofw.err )
( timeout 200 python bench.py --mode agent --emulate-world 8 > gpurun_out/r03_agent3_fw.json 2> gpurun_out/r03_agent3_fw.err )
tail -6 gpurun_out/r03_pytest3a.log; grep "fuse_warp\|^\[sp\] conv_pre" gpurun_out/r03_bench3_fw.err; tail -c 400 gpurun_out/r03_agent3_nofw.json; tail -c 400 gpurun_out/r03_agent3_fw.json
