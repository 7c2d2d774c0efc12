mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 200 tools/sp_conv_check.bin 20 all auto > gpurun_out/r03_spcheck6.log 2>&1; echo "spcheck rc $?" >> gpurun_out/r03_spcheck6.log )
( timeout 120 python bench.py --no-cpu-baseline --no-alt-math --train-steps 0 --no-agent-leg --layers > gpurun_out/r03_bench6.json 2> gpurun_out/r03_bench6.err )
tail -3 gpurun_out/r03_spcheck6.log; grep "^\[sp\]" gpurun_out/r03_bench6.err; tail -c 200 gpurun_out/r03_bench6.json
