# NaN-test cost A/B in one lease: DN_NANCHECK 0 (none) / 1 (every quad) / 2 (quad 0 of a tile), two rounds, graph replay value + per-layer eager times
mkdir -p gpurun_out/r04
cp disconet_amd/libdisconet_hip.so /tmp/lib_keep.so
for rnd in a b; do for v in 0 1 2; do
  cp tools/ab/DN_NANCHECK_$v/libdisconet_hip.so disconet_amd/libdisconet_hip.so
  timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg --layers 2>gpurun_out/r04/bench10_$v$rnd.err | tail -1 > gpurun_out/r04/bench10_$v$rnd.json
  echo -n "NANCHECK=$v $rnd: "; python3 -c "
import json,re; r=json.load(open('gpurun_out/r04/bench10_$v$rnd.json'))
L={m.group(1):float(m.group(2)) for m in (re.match(r'\[sp\] (\S+)\s+([0-9.]+)', l) for l in open('gpurun_out/r04/bench10_$v$rnd.err')) if m}
print(r['value'], r['repeat']['scenes_per_s']['median'], 'conv ms', r['roofline']['kernel_ms_per_step'], ' '.join('%s %.1f' % (k, 1e3*L[k]) for k in ('conv_pre_1','conv1_2+3d','conv8_1','heads','conv_pre_2','conv8_2') if k in L))"
done; done
cp /tmp/lib_keep.so disconet_amd/libdisconet_hip.so
