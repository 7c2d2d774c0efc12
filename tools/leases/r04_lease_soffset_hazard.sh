# soffset store anomaly: the four epilogue variants of tools/soff on the WTN = 2 tiles, then a sanity bench line
mkdir -p gpurun_out/r04
export SPCHECK_CFGS=0,6,9,11
L="ragged 3x3 40x72,ragged 1x1,conv1_2 128,conv2_2,conv3_2 32,conv3d_2,mlp 1x1,conv7_2"
for v in 1 2 3 4; do
  ( timeout 150 tools/soff/v$v/sp_conv_check.bin 20 "$L" tiles > gpurun_out/r04/soff_v$v.log 2>&1; echo "rc $?" >> gpurun_out/r04/soff_v$v.log )
  echo "v$v: $(grep -c FAIL gpurun_out/r04/soff_v$v.log) failing cases; $(tail -1 gpurun_out/r04/soff_v$v.log)"
done
unset SPCHECK_CFGS
( timeout 150 tools/sp_conv_check.bin 20 "$L" tiles > gpurun_out/r04/soff_v0.log 2>&1; echo "rc $?" >> gpurun_out/r04/soff_v0.log )
echo "v0 (shipped): $(grep -c FAIL gpurun_out/r04/soff_v0.log) failing cases"
timeout 300 python bench.py --steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg 2>gpurun_out/r04/bench1.err | tail -1 > gpurun_out/r04/bench1.json
python3 -c "
import json; r=json.load(open('gpurun_out/r04/bench1.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'])"
