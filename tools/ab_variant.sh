#!/bin/bash
# Builds dynfu_amd/build/libdynfu_amd_<tag>.so with one source recompiled under extra flags (compile-time A/B):
#   bash tools/ab_variant.sh rb8 tsdf.hip -DDFA_RAY_BATCH=8      then      DFA_LIB_PATH=dynfu_amd/build/libdynfu_amd_rb8.so python ...
set -e
tag=$1; src=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
B=$R/dynfu_amd/build
x=""; case $src in *.cpp) x="-x hip";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -fno-gpu-rdc "$@" $x -c $R/dynfu_amd/csrc/$src -o $B/${src%.*}_$tag.o
objs=""
for o in tsdf warp solve solve6 mc img icp points capi; do
  if [ "$o" == "${src%.*}" ]; then objs="$objs $B/${o}_$tag.o"; else objs="$objs $B/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $B/libdynfu_amd_$tag.so
echo $B/libdynfu_amd_$tag.so
