#!/bin/bash
# round 3, first GPU session: reference-order wave kernel parity + c4ref / c4 bench lines
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail -5 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reference or regex" > $O/pytest_regex.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_regex.log
timeout -k 10 300 python bench.py --workload c4reftiny --steps 5 > $O/c4reftiny.json 2> $O/c4reftiny.log; echo "c4reftiny rc=$?"; tail -2 $O/c4reftiny.log
timeout -k 10 420 python bench.py --workload c4ref --steps 10 > $O/c4ref.json 2> $O/c4ref.log; echo "c4ref rc=$?"; tail -4 $O/c4ref.log
FMX_REFMATCH=group timeout -k 10 420 python bench.py --workload c4ref --steps 5 --no-cpu-baseline > $O/c4ref_group.json 2> $O/c4ref_group.log; echo "c4ref(group) rc=$?"
timeout -k 10 420 python bench.py --workload c4 --steps 20 > $O/c4.json 2> $O/c4.log; echo "c4 rc=$?"; tail -3 $O/c4.log
python - <<'PY'
import json
for f in ("c4reftiny","c4ref","c4ref_group","c4"):
    try:
        d=json.loads(open("gpurun_out/r03a/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "value %.0f M rq/s  ms/step %.3f kernel_ms %.3f regexes/s %.3g fresh %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["regexes_per_sec"], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["fresh_batch"].items() if k!="what"}))
    except Exception as e:
        print(f, "no result:", e)
PY
