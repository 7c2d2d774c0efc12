# 1000-step replay check of the agent-sharded step WITH the RCCL all-gather between the two graphs (process group of one rank)
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
timeout 500 python bench.py --mode agent --emulate-world 8 --agent-check 1000 2> gpurun_out/r03_agent_rccl.err | tail -1 > gpurun_out/r03_agent_rccl.json
python3 -c "
import json; a=json.load(open('gpurun_out/r03_agent_rccl.json')); print(a['value'], a['phases_us'], a.get('replay_check'), a['emulated_share']['projected_speedup'])"
tail -3 gpurun_out/r03_agent_rccl.err | cut -c1-300
