#!/bin/bash
# PMC counters of the training step's kernels (two steady steps of tools/train_step_probe.py): busy counters / FETCH_SIZE / WRITE_SIZE
# in separate passes (kernel-trace only), summarised per kernel name by tools/train_pmc_table.py -> profiles/r05_train_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05tpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="python $R/tools/train_step_probe.py --dgrad sp --wgrad sp --steps 2"
timeout 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/tp1 -o p1 -- $T > $O/p1.log 2>&1
timeout 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tp2 -o p2 -- $T > $O/p2.log 2>&1
timeout 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tp3 -o p3 -- $T > $O/p3.log 2>&1
for i in 1 2 3; do f=$(find /tmp/tp$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $O/pmc$i.csv; done
python3 $R/tools/train_pmc_table.py $O > $O/train_pmc.txt 2>&1
head -40 $O/train_pmc.txt | cut -c1-170
