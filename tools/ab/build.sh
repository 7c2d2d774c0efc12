#!/bin/bash
# A/B builds of libdisconet_hip.so that differ in one macro:  tools/ab/build.sh MACRO v0 v1 ...  ->  tools/ab/<MACRO>_<v>/libdisconet_hip.so
# (conv_sp / conv_spq / fuse_mlp are recompiled with -D<MACRO>=<v>, the other objects come from disconet_amd/csrc/build).
# tools/ab/run.sh swaps a variant in on the GPU box's scratch copy and runs a command.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/disconet_amd/csrc
# tools/ab/build.sh FLAGS name "<extra hipcc flags>"  builds tools/ab/FLAGS_name with those flags instead of a macro
FILES=${AB_FILES:-"conv_sp conv_spq fuse_mlp"}     # AB_FILES="warp" tools/ab/build.sh DN_WARP_SHARED_TAPS 0 1
M=$1; shift
if [ "$M" = FLAGS ]; then set -- "$1:$2"; fi
for v in "$@"; do
  if [ "$M" = FLAGS ]; then d=$R/tools/ab/FLAGS_${v%%:*}; X="${v#*:}"; else d=$R/tools/ab/${M}_$v; X="-D$M=$v"; fi
  mkdir -p $d
  for f in $FILES; do
    # (common.hip refuses to compile without the build id: an A/B variant carries a marker, never a tree hash -- it is loaded
    #  through DISCONET_HIP_LIB / DISCONET_ALLOW_STALE_LIB=1, which skip the id check)
    ID=""; [ "$f" = common ] && ID='-DDN_BUILD_ID="ab-variant------"'
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -I $R/include -I $C $X $ID -c $C/$f.hip -o $d/$f.o &
  done
  wait
  objs=$(ls $C/build/*.o | grep -v -E "/($(echo $FILES | tr ' ' '|'))\.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdisconet_hip.so $objs $(for f in $FILES; do echo $d/$f.o; done)
  rm -f $d/*.o
done
