mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r03_pytest27.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest27.log )
grep -E "passed|failed|pytest rc" gpurun_out/r03_pytest27.log | tail -3
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke27.log 2>&1; echo "smoke rc $?" >> gpurun_out/r03_smoke27.log ); tail -1 gpurun_out/r03_smoke27.log
