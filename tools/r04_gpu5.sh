mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04/pytest5.log; tail -6 gpurun_out/r04/pytest5.log
