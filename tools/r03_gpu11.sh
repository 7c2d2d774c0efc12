mkdir -p gpurun_out/r03share
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/r03_share -o sh -- python $R/bench.py --mode agent --no-pg --emulate-world 8 --steps 10 --warmup 2 > $R/gpurun_out/r03share/run.log 2>&1
p=$(find /tmp/r03_share -name "*kernel_trace.csv" | head -1); [ -n "$p" ] && cp "$p" $R/gpurun_out/r03share/kernel_trace.csv
tail -2 $R/gpurun_out/r03share/run.log | cut -c1-300; ls -la $R/gpurun_out/r03share
