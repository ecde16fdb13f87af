#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03n}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "calc_gaps or naive_bwt or reference" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v amdgpu.ids $O/pytest.log | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/calcgaps_chain.py 27 128 2>&1 | grep -v amdgpu.ids | tee $O/calcgaps_chain.txt
timeout -k 10 300 python bench.py --workload c4ref --steps 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4ref ms/step %.3f kernel %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"
