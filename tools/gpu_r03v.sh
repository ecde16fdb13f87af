#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03v}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "regex or thompson or dfa or sharded" > $O/pytest_regex.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_regex.log
[ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 200 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; }
echo "row1=on" | tee -a $O/ab.txt; run
echo "row1=off" | tee -a $O/ab.txt; FMX_ROW1=0 run
echo "row1=on" | tee -a $O/ab.txt; run
