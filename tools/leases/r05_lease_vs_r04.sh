# Round 4's tree (commit 29a3378 as a git worktree under tools/ab/r04_tree, its own library built from ITS sources and its own
# bench.py) against this round's, SAME lease, interleaved three times: the default bench command without the extras, then once
# each with the training step (eager launches: host-dependent).   -> gpurun_out/r05/vs_r04.txt
mkdir -p gpurun_out/r05
R=$PWD
B="--steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg"
for rnd in a b c; do
  ( cd tools/ab/r04_tree && timeout 300 python bench.py $B 2>/dev/null | tail -1 > $R/gpurun_out/r05/ab_r04_$rnd.json )
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/r05/ab_r05_$rnd.json
  python3 - <<PY >> gpurun_out/r05/vs_r04.txt
import json
for t in ("r04", "r05"):
    r = json.load(open("gpurun_out/r05/ab_%s_$rnd.json" % t))
    print("$rnd", t, "value", r["value"], "ms/step", r["ms_per_step"], "conv ms", r["roofline"]["kernel_ms_per_step"], "frac", r["roofline"]["frac"],
          "others", r["roofline"]["other_kernels_ms_per_step"], "repeat", (r.get("repeat") or {}).get("scenes_per_s"))
PY
done
T="--steps 5 --warmup 2 --no-alt-math --no-cpu-baseline --train-steps 6 --no-voxelize --no-agent-leg"
for rnd in a b; do
  ( cd tools/ab/r04_tree && timeout 300 python bench.py $T 2>/dev/null | tail -1 > $R/gpurun_out/r05/abt_r04_$rnd.json )
  timeout 300 python bench.py $T 2>/dev/null | tail -1 > gpurun_out/r05/abt_r05_$rnd.json
  python3 - <<PY >> gpurun_out/r05/vs_r04.txt
import json
for t in ("r04", "r05"):
    r = json.load(open("gpurun_out/r05/abt_%s_$rnd.json" % t))["train_step"]
    print("$rnd", t, "training step ms", r["ms_per_step"], "with KD", r["with_kd"]["ms_per_step"])
PY
done
cat gpurun_out/r05/vs_r04.txt
