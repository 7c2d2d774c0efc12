mkdir -p gpurun_out/r04
timeout 900 python tools/soak_replay.py 20000 > gpurun_out/r04/soak.txt 2>&1; tail -3 gpurun_out/r04/soak.txt
timeout 600 python bench.py --mode agent --emulate-world 8 --agent-check 1000 2> gpurun_out/r04/agent_rccl.err | tail -1 > gpurun_out/r04/agent_rccl.json
python3 -c "
import json; a=json.load(open('gpurun_out/r04/agent_rccl.json')); print(a['ms_per_step'], a.get('replay_check'), {k: a['emulated_share'][k] for k in ('ms_per_step','projected_speedup','collective_in_share','ms_per_step_without_collective','phases_us')})"
