# projected strong-scaling curve of the agent-sharded step: one rank's share of a 2-, 4- and 8-rank run, batch 4 and 16
mkdir -p gpurun_out/r03curve
for b in 4 16; do for w in 2 4 8; do
  timeout 300 python bench.py --mode agent --no-pg --emulate-world $w --agent-batch $b --steps 10 --warmup 2 2> gpurun_out/r03curve/w${w}_b${b}.err | tail -1 > gpurun_out/r03curve/w${w}_b${b}.json
  python3 -c "
import json; a=json.load(open('gpurun_out/r03curve/w${w}_b${b}.json')); e=a['emulated_share']; print('batch $b world $w: full', a['ms_per_step'], 'share', e['ms_per_step'], 'x', e['projected_speedup'], e['outputs_equal_unsharded_rows'])"
done; done
