mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest_gpu.txt )
( timeout 300 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc $?" >> gpurun_out/r03_bench_default.err )
( timeout 200 python bench.py --mode agent --emulate-world 8 --agent-check 1000 > gpurun_out/r03_agent_share.json 2> gpurun_out/r03_agent_share.err )
( timeout 200 python bench.py --task seg --train-steps 3 > gpurun_out/r03_bench_seg.json 2> gpurun_out/r03_bench_seg.err )
bash tools/r03_profile.sh > gpurun_out/r03_profile.log 2>&1
tail -3 gpurun_out/r03_pytest_gpu.txt; tail -c 300 gpurun_out/r03_bench_default.json; tail -5 gpurun_out/r03_profile.log
