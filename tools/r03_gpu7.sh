mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py tests/test_gpu_seg_train.py -q -m gpu -x > gpurun_out/r03_pytest7a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest7a.log )
( timeout 400 python tools/hazard/train_flake_check.py 500 > gpurun_out/r03_train_flake_check.txt 2>&1 )
tail -6 gpurun_out/r03_pytest7a.log; tail -8 gpurun_out/r03_train_flake_check.txt
