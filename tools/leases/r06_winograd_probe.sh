#!/bin/bash
# Round 6: best-case timing of a fused Winograd F(2x2,3x3) kernel for conv3_2 (tools/winograd_probe.hip: the instruction mix of the only
# workgroup shape that fits the part, no epilogue, no output transform) next to the shipped direct kernel on the same layer, one lease
# -> gpurun_out/r06/winograd_probe.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
{
  echo "# tools/leases/r06_winograd_probe.sh, one lease"
  for rep in 1 2; do timeout 120 tools/winograd_probe.bin 320; done
  echo "# the shipped engine on the same layer (tools/sp_conv_check.bin 20 conv3_2 auto):"
  timeout 200 tools/sp_conv_check.bin 20 "conv3_2" auto 2>&1 | grep -E "conv3_2|dn_version" | head -6
} > $O/winograd_probe.txt 2>&1
cat $O/winograd_probe.txt
