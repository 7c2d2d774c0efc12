mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_model.py tests/test_gpu_train_step.py -q -m gpu -k "fused or trained_like or train_step_matches or one_launch" > gpurun_out/r03_pytest4a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest4a.log )
( timeout 120 python bench.py --no-cpu-baseline --no-alt-math --train-steps 0 --no-agent-leg --layers > gpurun_out/r03_bench4_fw.json 2> gpurun_out/r03_bench4_fw.err )
for fw in 0 1; do
  ( DISCONET_FUSE_WARP=$fw timeout 150 python bench.py --mode agent --agent-check 300 --steps 5 > gpurun_out/r03_agentchk_fw${fw}_pg.json 2> gpurun_out/r03_agentchk_fw${fw}_pg.err )
  ( DISCONET_FUSE_WARP=$fw timeout 150 python bench.py --mode agent --no-pg --agent-check 300 --steps 5 > gpurun_out/r03_agentchk_fw${fw}_nopg.json 2> gpurun_out/r03_agentchk_fw${fw}_nopg.err )
done
tail -4 gpurun_out/r03_pytest4a.log; grep "fuse_warp" gpurun_out/r03_bench4_fw.err; for f in gpurun_out/r03_agentchk_*.json; do echo $f; grep -o '"replay_check": {[^}]*}[^}]*}' $f; done
