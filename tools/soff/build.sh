#!/bin/bash
# Builds libdisconet_hip.so in four variants of the SP conv epilogue's store form (conv_sp.hip, DN_EPI_SOFF) and one
# sp_conv_check binary per variant: the round-3 "plane offset in the store's SCALAR operand" anomaly
# (DESIGN.md 3.6 C).  1 = the failing form, 2 = + s_nop 1 behind every store, 3 = + s_nop 7, 4 = both scalar
# offsets formed before the pair (no SALU write of a store's soffset register behind it).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/disconet_amd/csrc
for v in 1 2 3 4; do
  d=$R/tools/soff/v$v; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -I $R/include -I $C -DDN_EPI_SOFF=$v \
      -c $C/conv_sp.hip -o $d/conv_sp.o
  objs=$(ls $C/build/*.o | grep -v conv_sp.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdisconet_hip.so $objs $d/conv_sp.o
  /opt/rocm/bin/hipcc -O2 -std=c++17 -w -I $R/include $R/tools/sp_conv_check.cpp -L $d -ldisconet_hip \
      -Wl,--disable-new-dtags,-rpath,"\$ORIGIN" -o $d/sp_conv_check.bin
  rm -f $d/conv_sp.o
done
