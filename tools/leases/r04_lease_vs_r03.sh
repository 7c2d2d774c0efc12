# Round 3's tree (commit 7586278, tools/ab/r03_tree, its own library and bench.py) against this round's, SAME lease, interleaved:
# the default bench command without the extras; r04 additionally with --pre-roll 0 (round 3 had no pre-roll)
mkdir -p gpurun_out/r04
R=$PWD
B="--steps 20 --warmup 3 --no-alt-math --no-cpu-baseline --train-steps 0 --no-voxelize --no-agent-leg"
for rnd in a b c; do
  ( cd tools/ab/r03_tree && timeout 300 python bench.py $B 2>/dev/null | tail -1 > $R/gpurun_out/r04/ab11_r03_$rnd.json )
  timeout 300 python bench.py $B --pre-roll 0 2>/dev/null | tail -1 > gpurun_out/r04/ab11_r04np_$rnd.json
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/r04/ab11_r04_$rnd.json
  python3 - <<PY
import json
for t in ("r03", "r04np", "r04"):
    r = json.load(open("gpurun_out/r04/ab11_%s_$rnd.json" % t))
    print("$rnd", t, "value", r["value"], "ms/step", r["ms_per_step"], "conv ms", r["roofline"]["kernel_ms_per_step"], "frac", r["roofline"]["frac"],
          "others", r["roofline"]["other_kernels_ms_per_step"], "repeat", (r.get("repeat") or {}).get("scenes_per_s"))
PY
done
