#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03w}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "regex or thompson or dfa or sharded" > $O/pytest_regex.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_regex.log
[ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 200 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; }
unset FMX_LIB; echo "filter on" | tee -a $O/ab.txt; run
export FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_nofilt.so; echo "filter off (row table on)" | tee -a $O/ab.txt; run
unset FMX_LIB; run
