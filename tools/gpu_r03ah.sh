#!/bin/bash
# round 3: three-step lookups inside the fused search kernel
O=gpurun_out/${1:-r03ah}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not regex and not c4" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc = 0 ] || exit 1
for wl in c3 c2; do
  for rows in -1 0; do
    if [ $rows = -1 ]; then unset FMX_ROWS; else export FMX_ROWS=$rows; fi
    timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline 2> $O/${wl}_$rows.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl FMX_ROWS=$rows', 'value %.0f ms/step %.3f kernel_ms %.3f requests %d tables_ms %.0f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['requests_per_launch'], d['config']['tables_build_ms']))"
  done
done
