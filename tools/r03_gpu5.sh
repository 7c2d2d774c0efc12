mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_seg_train.py -q -m gpu -x > gpurun_out/r03_pytest5a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest5a.log )
( timeout 150 python bench.py --task seg --train-steps 3 --no-cpu-baseline > gpurun_out/r03_bench5_seg.json 2> gpurun_out/r03_bench5_seg.err )
tail -40 gpurun_out/r03_pytest5a.log; tail -c 500 gpurun_out/r03_bench5_seg.json
