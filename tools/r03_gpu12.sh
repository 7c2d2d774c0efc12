# deep-regime (all nine taps per step) tiles on the 4-image share layers: correctness + us per tile
mkdir -p gpurun_out
( timeout 300 tools/sp_conv_check.bin 4 share tiles > gpurun_out/r03_spcheck12.log 2>&1; echo "spcheck rc $?" >> gpurun_out/r03_spcheck12.log )
cat gpurun_out/r03_spcheck12.log | cut -c1-1500
timeout 200 python bench.py --mode agent --no-pg --emulate-world 8 --steps 20 --warmup 3 2>gpurun_out/r03_bench12_agent.err | tail -1 > gpurun_out/r03_bench12_agent.json
cut -c1-1200 gpurun_out/r03_bench12_agent.json
DN_SP_DEEP=0 timeout 200 python bench.py --mode agent --no-pg --emulate-world 8 --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-600
