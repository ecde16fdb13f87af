#!/bin/bash
# K5: balanced start order A/B + timeline
set -o pipefail
O=gpurun_out/${1:-r03f}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "regex" > $O/pytest_regex.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_regex.log
[ $rc -eq 0 ] || exit $rc
export FMX_FRONTIER_CHAIN=1
run() { timeout -k 10 200 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; }
echo "balance=1" | tee -a $O/ab.txt; run
echo "balance=0" | tee -a $O/ab.txt; FMX_FRONTIER_BALANCE=0 run
echo "balance=1" | tee -a $O/ab.txt; run
export FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_wl.so
timeout -k 10 200 python tools/wave_timeline.py c4 2>&1 | grep -v amdgpu.ids | cut -c1-900 > $O/timeline.txt; tail -7 $O/timeline.txt
