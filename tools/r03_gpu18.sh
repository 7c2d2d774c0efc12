mkdir -p gpurun_out
( timeout 300 tools/sp_conv_check.bin 20 "up+cat" tiles > gpurun_out/r03_spcheck18.log 2>&1; echo "rc $?" >> gpurun_out/r03_spcheck18.log )
grep -E "^conv|rc " gpurun_out/r03_spcheck18.log | sed 's/.*\[quad\]/[quad]/' | cut -c1-300
