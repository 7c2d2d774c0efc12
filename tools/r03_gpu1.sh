mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 240 tools/sp_conv_check.bin 20 all tiles > gpurun_out/r03_spcheck1.log 2>&1; echo "spcheck rc $?" >> gpurun_out/r03_spcheck1.log ) 
( timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest1.log )
( timeout 200 python bench.py > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err; echo "bench rc $?" >> gpurun_out/r03_bench1.err )
tail -3 gpurun_out/r03_spcheck1.log; tail -5 gpurun_out/r03_pytest1.log; tail -c 600 gpurun_out/r03_bench1.json
