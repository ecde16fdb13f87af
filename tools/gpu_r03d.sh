#!/bin/bash
# K5 A/B: narrow rounds on/off, chain length 1/2, direct export; parity first
set -o pipefail
O=gpurun_out/${1:-r03d}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "regex or thompson or dfa or sharded" > $O/pytest_regex.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_regex.log
[ $rc -eq 0 ] || exit $rc
for v in default nonarrow; do
  for ch in 2 1; do
    if [ $v = default ]; then unset FMX_LIB; else export FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_$v.so; fi
    FMX_FRONTIER_CHAIN=$ch timeout -k 10 200 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  done
done
