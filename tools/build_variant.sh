#!/bin/bash
# Builds libfmx with extra compiler flags into findex_amd/lib/variants/libfmx_<tag>.so for A/B runs
# (select one with FMX_LIB=<path>):   tools/build_variant.sh t64 -DFMX_FTHREADS=64
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/findex_amd/lib/variants; mkdir -p $OUT/$TAG
pids=()
for f in fmx_api.cpp fmx_hostpar.cpp fmx_comm.cpp fmx_hostrank.cpp fmx_regex.cpp fmx_build.hip fmx_kernels.hip fmx_search.hip fmx_ktab.hip fmx_jump.hip fmx_select.hip fmx_frontier.hip fmx_refmatch.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$ROOT/findex_amd/csrc "$@" -x hip -c $ROOT/findex_amd/csrc/$f -o $OUT/$TAG/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libfmx_$TAG.so $OUT/$TAG/*.o -ldl && rm -rf $OUT/$TAG && echo $OUT/libfmx_$TAG.so
