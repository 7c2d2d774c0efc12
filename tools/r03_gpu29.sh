# instruction mix / wait counters of the attention launch, one-wave vs four-wave form at the BASELINE size
R=${GRAFT_REPO_ROOT:-/root/repo}
E="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-alt-math --no-cpu-baseline --no-kernel-events --train-steps 0 --no-voxelize --no-agent-leg"
DN_FUSE_MLP_WAVES=1 bash $R/tools/pmc_cmd.sh fm1 $E > /dev/null 2>&1
DN_FUSE_MLP_WAVES=4 bash $R/tools/pmc_cmd.sh fm4 $E > /dev/null 2>&1
grep -h "fuse_mlp\|warp_neighbors" $R/gpurun_out/pmc_fm1/table.txt $R/gpurun_out/pmc_fm4/table.txt
