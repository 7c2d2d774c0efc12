mkdir -p gpurun_out
T0=$(date +%s); timeout 900 python bench.py 2> gpurun_out/r03_bench16.err | tail -1 > gpurun_out/r03_bench16.json; echo "bench wall $(( $(date +%s) - T0 )) s"

python3 - <<PY
import json
d = json.load(open("gpurun_out/r03_bench16.json"))
print(d["value"], d["ms_per_step"])
for k in ("agent_sharded", "agent_sharded_batch16"):
    a = d[k]; print(k, a.get("value"), a.get("ms_per_step"), a.get("emulated_share", {}).get("projected_speedup"), a.get("error"))
PY
