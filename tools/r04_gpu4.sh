mkdir -p gpurun_out/r04
( timeout 300 tools/sp_conv_check.bin 20 "ks conv" auto > gpurun_out/r04/ks2.log 2>&1; echo "rc $?" >> gpurun_out/r04/ks2.log )
grep "^\[ks" gpurun_out/r04/ks2.log | cut -c1-60,160-400; tail -2 gpurun_out/r04/ks2.log
for k in 1 0 all; do
  DN_SP_KSLICES=$k timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>gpurun_out/r04/bench4_k$k.err | tail -1 > gpurun_out/r04/bench4_k$k.json
  python3 -c "
import json; r=json.load(open('gpurun_out/r04/bench4_k$k.json')); print('KSLICES=$k', r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('repeat',{}).get('scenes_per_s'))"
  grep "^\[sp\] conv[3-6]_[12]" gpurun_out/r04/bench4_k$k.err | cut -c1-40 | tr '\n' ';'; echo
  DN_SP_KSLICES=$k timeout 300 python bench.py --mode agent --no-pg --emulate-world 8 --steps 20 --warmup 3 2>gpurun_out/r04/agent4_k$k.err | tail -1 > gpurun_out/r04/agent4_k$k.json
  python3 -c "
import json; r=json.load(open('gpurun_out/r04/agent4_k$k.json')); e=r.get('emulated_share',{}); print('KSLICES=$k agent', r['ms_per_step'], e.get('ms_per_step'), e.get('phases_us'), e.get('projected_speedup'), e.get('outputs_equal_unsharded_rows'))"
done
