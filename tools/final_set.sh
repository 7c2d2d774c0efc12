#!/bin/bash
# Validation + profile set of a round on the GPU box:  bash tools/final_set.sh r05  -> gpurun_out/r05final/
#   full -m gpu suite, smoke(), tools/profile_set.sh $RD (its summaries are copied into profiles/ of the box's copy so
#   that the bench lines quote the profile of THIS build), then the bench lines that profiles/ keeps.
RD=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${RD}final
mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log )
tail -22 $O/pytest_gpu.log | cut -c1-200
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log ); tail -2 $O/smoke.log
# guard bands around every buffer of the conv check (SPCHECK_GUARD=1): all shapes at 20 / 7 / 3 images, K-sliced and ragged included
( for n in 20 7 3; do SPCHECK_GUARD=1 timeout 300 tools/sp_conv_check.bin $n all auto 2>&1 | grep -E "guard bands|GUARD BAND|SP CONV CHECK|dn_version"; done ) > $O/guard_bands.txt 2>&1; cat $O/guard_bands.txt
bash tools/profile_set.sh $RD > $O/profile.log 2>&1
P=$R/gpurun_out/${RD}prof
cp $P/rocprof_conv_sp.json profiles/${RD}_rocprof_conv_sp.json
cp $P/pmc_traffic_sp.json profiles/${RD}_pmc_traffic_sp.json
[ -s $P/pmc_traffic_seg.json ] && cp $P/pmc_traffic_seg.json profiles/${RD}_pmc_traffic_seg.json
cd $R
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
timeout 400 python bench.py --mode agent --no-pg --emulate-world 8 --agent-check 1000 2> $O/agent_share.err | tail -1 > $O/agent_share.json
for b in 8 16 32; do
  timeout 300 python bench.py --mode agent --no-pg --emulate-world 8 --agent-batch $b --steps 10 --warmup 2 2> $O/agent_share_b$b.err | tail -1 > $O/agent_share_b$b.json
done
timeout 500 python bench.py --task seg --train-steps 4 2> $O/bench_seg.err | tail -1 > $O/bench_seg.json
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("rocprof"), d.get("train_step", {}).get("ms_per_step"))
print("agent_sharded leg", json.dumps(d.get("agent_sharded"))[:300])
for f in ("agent_share", "agent_share_b8", "agent_share_b16", "agent_share_b32"):
    try:
        a = json.load(open("$O/%s.json" % f)); e = a["emulated_share"]
        print(f, a["ms_per_step"], e["ms_per_step"], e["projected_speedup"], a.get("replay_check"))
    except Exception as ex:
        print(f, "failed", ex)
s = json.load(open("$O/bench_seg.json"))
print("seg", s["value"], s["ms_per_step"], s["roofline"]["frac"], s["roofline"]["traffic"], s.get("train_step", {}).get("ms_per_step"))
PY
ls $P
