mkdir -p gpurun_out/r04
timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>gpurun_out/r04/bench9.err | tail -1 > gpurun_out/r04/bench9.json
grep "^\[sp\] heads\|^\[sp\] conv8_2\|^\[sp\] conv_pre_[12]\|^\[sp\] conv1_2" gpurun_out/r04/bench9.err | cut -c1-60
python3 -c "
import json; r=json.load(open('gpurun_out/r04/bench9.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('repeat',{}).get('scenes_per_s'))"
timeout 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py -m gpu -x -q 2>&1 | tail -4
for p in 1 0; do DN_DGRAD_PARITY=$p timeout 300 python bench.py --steps 5 --warmup 2 --no-alt-math --no-cpu-baseline --no-voxelize --no-agent-leg --no-kernel-events --train-steps 6 2>/dev/null | tail -1 | python3 -c "
import json,sys; r=json.load(sys.stdin); print('DGRAD_PARITY=$p train_step', r.get('train_step'))" | cut -c1-300; done
