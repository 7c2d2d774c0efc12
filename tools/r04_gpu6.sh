mkdir -p gpurun_out/r04
for fm in 0 1 0 1; do DN_FUSE_FM=$fm timeout 120 python tools/fuse_ab.py 2>&1 | tail -2 | tr '\n' ' '; echo; done
timeout 600 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_model.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -4
for fm in 1 0; do
DN_FUSE_FM=$fm timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg 2>gpurun_out/r04/bench6_fm$fm.err | tail -1 > gpurun_out/r04/bench6_fm$fm.json
python3 -c "
import json; r=json.load(open('gpurun_out/r04/bench6_fm$fm.json')); print('FM=$fm', r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('repeat',{}).get('scenes_per_s'), r['roofline']['other_kernels_ms_per_step'])"
done
