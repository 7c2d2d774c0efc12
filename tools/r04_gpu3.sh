# K slices in the step: default bench and the emulated agent share, DN_SP_KSLICES=1 vs 0; then the conv / model / sharded tests
mkdir -p gpurun_out/r04
for k in 1 0; do
  DN_SP_KSLICES=$k timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>gpurun_out/r04/bench3_k$k.err | tail -1 > gpurun_out/r04/bench3_k$k.json
  python3 -c "
import json; r=json.load(open('gpurun_out/r04/bench3_k$k.json')); print('KSLICES=$k', r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('repeat',{}).get('scenes_per_s'))"
  grep "^\[sp\] conv[3-6]_[12]" gpurun_out/r04/bench3_k$k.err | cut -c1-70
  DN_SP_KSLICES=$k timeout 300 python bench.py --mode agent --no-pg --emulate-world 8 --steps 20 --warmup 3 2>gpurun_out/r04/agent3_k$k.err | tail -1 > gpurun_out/r04/agent3_k$k.json
  python3 -c "
import json; r=json.load(open('gpurun_out/r04/agent3_k$k.json')); e=r.get('emulated_share',{}); print('KSLICES=$k agent', r['ms_per_step'], e.get('ms_per_step'), e.get('phases_us'), e.get('projected_speedup'), e.get('outputs_equal_unsharded_rows'))"
done
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_sharded.py tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -8
