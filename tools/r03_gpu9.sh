mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py tests/test_gpu_seg_train.py tests/test_gpu_seg.py -q -m gpu -x > gpurun_out/r03_pytest9a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest9a.log )
( timeout 200 python bench.py --no-cpu-baseline --no-alt-math --no-agent-leg --train-steps 6 > gpurun_out/r03_bench9_train.json 2> gpurun_out/r03_bench9_train.err )
( timeout 200 python bench.py --task seg --no-cpu-baseline --train-steps 4 > gpurun_out/r03_bench9_seg.json 2> gpurun_out/r03_bench9_seg.err )
tail -4 gpurun_out/r03_pytest9a.log; grep -o '"train_step": {[^}]*}' gpurun_out/r03_bench9_train.json gpurun_out/r03_bench9_seg.json
