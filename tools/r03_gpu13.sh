# ablations of the deep-regime tile on the share layers (1 = no weight DMA, 2 = no patch DMA, 3 = neither, 4 = no stores, 5 = no LDS reads)
mkdir -p gpurun_out
( timeout 300 tools/sp_conv_check.bin 4 "share conv4_2" abl; timeout 100 tools/sp_conv_check.bin 4 "share conv5_2" abl; timeout 100 tools/sp_conv_check.bin 4 "share conv3_2" abl ) > gpurun_out/r03_spcheck13.log 2>&1
grep -E "share|PASSED|FAILED" gpurun_out/r03_spcheck13.log | sed 's/|/\n   /g' | grep -E "share|abl|auto|64x64"
