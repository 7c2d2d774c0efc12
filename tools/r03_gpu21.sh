mkdir -p gpurun_out
timeout 300 python bench.py --force-process-group --steps 5 --warmup 1 --no-cpu-baseline --no-alt-math --train-steps 0 --no-agent-leg > gpurun_out/r03_out21_pg.txt 2> gpurun_out/r03_out21_pg.err; echo "rc $?"
wc -l gpurun_out/r03_out21_pg.txt; tail -12 gpurun_out/r03_out21_pg.err | cut -c1-300
