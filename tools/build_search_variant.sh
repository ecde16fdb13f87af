#!/bin/bash
# A variant of libfmx.so that differs in fmx_search.hip only (the product's other objects are linked as they are):
#   tools/build_search_variant.sh <tag> [flags, e.g. -DFMX_SEARCH_WAVES=5]   ->  findex_amd/lib/variants/libfmx_<tag>.so
# SRC=<file>: another version of fmx_search.hip (e.g. `git show HEAD:findex_amd/csrc/fmx_search.hip > /tmp/base.hip`); HDR=<dir>: its headers (all of csrc/*.h), when they differ too
# (python -m findex_amd.build first: the product's objects must be current)
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/findex_amd/lib/variants; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I${HDR:-$ROOT/findex_amd/csrc} "$@" -x hip -c ${SRC:-$ROOT/findex_amd/csrc/fmx_search.hip} -o $OUT/search_$TAG.o || exit 1
objs=$(ls $ROOT/findex_amd/lib/*.o | grep -v "fmx_search.hip.o\|faults.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libfmx_$TAG.so $objs $OUT/search_$TAG.o -ldl && rm -f $OUT/search_$TAG.o && echo $OUT/libfmx_$TAG.so
