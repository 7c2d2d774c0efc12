mkdir -p gpurun_out/r04
for mode in split nosplit off split nosplit off; do
  case $mode in split) export DN_SP_KSLICES=1 DN_SP_KS_NOSPLIT=0;; nosplit) export DN_SP_KSLICES=1 DN_SP_KS_NOSPLIT=1;; off) export DN_SP_KSLICES=0 DN_SP_KS_NOSPLIT=0;; esac
  timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>gpurun_out/r04/bench8_$mode.err | tail -1 > gpurun_out/r04/bench8_$mode.json
  echo -n "$mode: "; grep "^\[sp\] conv5_1" gpurun_out/r04/bench8_$mode.err | cut -c1-40 | tr '\n' ' '; python3 -c "
import json; r=json.load(open('gpurun_out/r04/bench8_$mode.json')); print(r['value'], r.get('repeat',{}).get('scenes_per_s'))"
done
