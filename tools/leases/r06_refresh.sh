#!/bin/bash
# The round's committed profile set wants a MEDIAN-class lease (two classes of boxes exist: ~2300 and ~2500 scenes/s): run the default
# line first; on a fast box (> 2400) stop after it (the line is kept, labelled); else the rest of tools/final_set.sh without the test suite.
RD=r06; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${RD}final; mkdir -p $O; cd $R
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
V=$(python3 -c "import json; print(int(json.load(open('$O/bench_default.json'))['value']))")
echo "default line: $V scenes/s"
if [ "$V" -gt 2352 ] || [ "$V" -lt 2288 ]; then echo "fast box: stopping (not median-class: $V)"; cp $O/bench_default.json $R/gpurun_out/${RD}_fast_$V.json; exit 0; fi
bash tools/profile_set.sh $RD > $O/profile.log 2>&1
P=$R/gpurun_out/${RD}prof
cp $P/rocprof_conv_sp.json profiles/${RD}_rocprof_conv_sp.json
cp $P/pmc_traffic_sp.json profiles/${RD}_pmc_traffic_sp.json
[ -s $P/pmc_traffic_seg.json ] && cp $P/pmc_traffic_seg.json profiles/${RD}_pmc_traffic_seg.json
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
timeout 400 python bench.py --mode agent --no-pg --emulate-world 8 --agent-check 1000 2> $O/agent_share.err | tail -1 > $O/agent_share.json
for b in 8 16 32; do
  timeout 300 python bench.py --mode agent --no-pg --emulate-world 8 --agent-batch $b --steps 10 --warmup 2 2> $O/agent_share_b$b.err | tail -1 > $O/agent_share_b$b.json
done
timeout 500 python bench.py --task seg --train-steps 4 2> $O/bench_seg.err | tail -1 > $O/bench_seg.json
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print('default (with this lease\'s profile)', d['value'], d['roofline']['frac'], d['roofline'].get('rocprof'), d['train_step']['ms_per_step'])"
