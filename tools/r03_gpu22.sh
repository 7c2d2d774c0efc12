# quad-merged kernel: timing-only ablations (abl23 = no weight DMA, abl24 = no patch DMA, abl25 = neither)
mkdir -p gpurun_out
( timeout 300 tools/sp_conv_check.bin 20 "up+cat" abl > gpurun_out/r03_spcheck22.log 2>&1; echo "rc $?" >> gpurun_out/r03_spcheck22.log )
grep -E "^conv|rc " gpurun_out/r03_spcheck22.log | sed 's/.*\[quad\]/[quad]/' | cut -c1-330
