#!/bin/bash
# Builds dynfu_amd/build/libdynfu_amd_<tag>.so with one source (or `all`) recompiled under extra flags (compile-time A/B):
#   bash tools/ab_variant.sh rb8 tsdf.hip -DDFA_RAY_BATCH=8      then      DFA_LIB_PATH=dynfu_amd/build/libdynfu_amd_rb8.so python ...
#   bash tools/ab_variant.sh prof all -DDFA_PCG_PROFILE -DDFA_DEV_AB
# Sources and flags come from dynfu_amd/build.py (one owner).
set -e
tag=$1; src=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
B=$R/dynfu_amd/build
CC="/opt/rocm/bin/hipcc $(cd $R && python3 -c 'from dynfu_amd import build as B; print(" ".join(B.FLAGS))')"
SRCS=$(cd $R && python3 -c 'from dynfu_amd import build as B; print(" ".join(s.rsplit(".", 1)[0] for s in B.SOURCES))')
objs=""
for o in $SRCS; do
  f=$o.hip; x=""; if [ $o == capi ]; then f=capi.cpp; x="-x hip"; fi
  if [ "$src" == all ] || [ "$f" == "$src" ]; then
    $CC "$@" $x -c $R/dynfu_amd/csrc/$f -o $B/${o}_$tag.o &
    objs="$objs $B/${o}_$tag.o"
  else
    objs="$objs $B/$o.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $B/libdynfu_amd_$tag.so
echo $B/libdynfu_amd_$tag.so
