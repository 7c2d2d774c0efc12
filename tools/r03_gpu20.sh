# stdout of bench.py must be exactly one JSON line in every mode (RCCL's banner goes to stderr)
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
timeout 300 python bench.py --mode agent --emulate-world 8 --steps 5 --warmup 1 > gpurun_out/r03_out20_agent.txt 2>/dev/null
timeout 300 python bench.py --force-process-group --steps 5 --warmup 1 --no-cpu-baseline --no-alt-math --train-steps 0 --no-agent-leg > gpurun_out/r03_out20_pg.txt 2>/dev/null
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-alt-math --train-steps 0 > gpurun_out/r03_out20_tdr.txt 2>/dev/null
timeout 300 python bench.py --task seg --steps 5 --warmup 1 --no-cpu-baseline --train-steps 0 > gpurun_out/r03_out20_seg.txt 2>/dev/null
for f in agent pg tdr seg; do echo "$f: $(wc -l < gpurun_out/r03_out20_$f.txt) line(s), starts $(head -c 12 gpurun_out/r03_out20_$f.txt)"; python3 -c "import json; json.load(open('gpurun_out/r03_out20_$f.txt')); print('  parses')"; done
