#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03h}; mkdir -p $O
export FMX_FRONTIER_CHAIN=1
export FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_wl.so
timeout -k 10 200 python tools/wave_timeline.py c4 2>&1 | grep -v amdgpu.ids | cut -c1-3000 > $O/timeline.txt; tail -6 $O/timeline.txt
