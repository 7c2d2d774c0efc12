mkdir -p gpurun_out
timeout 600 python tools/soak_replay.py 20000 > gpurun_out/r03_soak.txt 2>&1; echo "rc $?"; tail -2 gpurun_out/r03_soak.txt | cut -c1-300
