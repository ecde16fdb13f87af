#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03s}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "jump or kmer or fixture or synthetic or randomized or ragged or words or kats or bytes_layout_fix or bytes_layout_syn or pipelined or concurrent_calls or multi_replicas" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v amdgpu.ids $O/pytest.log | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --workload c3 --steps 20 > $O/c3.json 2> $O/c3.log; echo "c3 rc=$?"
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-20s value %9.0f M rq/s ms/step %.3f kernel_ms %.3f req/launch %d (lines %d ktab %d jump %d) cpu %s" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel_ms"], r["requests_per_launch"], r["rank_line_requests"], r["ktab_lookups"], r.get("jump_lookups",0), d.get("cpu_baseline",{}).get("value")))
    except Exception as e:
        print(os.path.basename(f), "no result:", e)
PY
