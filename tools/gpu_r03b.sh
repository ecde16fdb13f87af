#!/bin/bash
# reference-order kernel iteration: parity + c4ref bench
set -o pipefail
O=gpurun_out/${1:-r03b}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail -5 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reference or random_regexes" > $O/pytest_ref.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_ref.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 420 python bench.py --workload c4ref --steps 10 > $O/c4ref.json 2> $O/c4ref.log; echo "c4ref rc=$?"; tail -2 $O/c4ref.log
python - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/c4ref.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("c4ref value %.0f M rq/s  ms/step %.3f kernel_ms %.3f us/pop %.2f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["us_per_pop_critical_path"]))
print(r["algorithmic_bytes"])
PY
