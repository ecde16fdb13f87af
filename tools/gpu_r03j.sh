#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03j}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "c5_shape or full_size" --durations=10 > $O/pytest_full.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v amdgpu.ids $O/pytest_full.log | tail -25
