# K-sliced conv launches: correctness (vs the NHWC engine), batch-invariance (bitwise) and timing of the deep layers
mkdir -p gpurun_out/r04
( timeout 300 tools/sp_conv_check.bin 20 "ks " auto > gpurun_out/r04/ks1.log 2>&1; echo "rc $?" >> gpurun_out/r04/ks1.log )
grep "^\[ks" gpurun_out/r04/ks1.log | cut -c1-400; tail -2 gpurun_out/r04/ks1.log
