#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03k}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail $O/build.log; exit 1; }
timeout -k 10 1100 python -m pytest tests/ -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v amdgpu.ids $O/pytest_gpu.log | tail -15
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
