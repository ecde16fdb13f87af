#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03q}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "jump or kmer or fixture or synthetic or randomized or ragged or words or kats or bytes_layout_fix or bytes_layout_syn or pipelined or concurrent_calls" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v amdgpu.ids $O/pytest.log | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --workload c3 --steps 20 > $O/c3_decoupled.json 2> $O/c3_decoupled.log; echo "c3 decoupled rc=$?"
FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_lockstep.so timeout -k 10 400 python bench.py --workload c3 --steps 20 --no-cpu-baseline > $O/c3_lockstep.json 2> $O/c3_lockstep.log; echo "c3 lockstep rc=$?"
timeout -k 10 400 python tools/ragged_bench.py > $O/ragged.txt 2>&1; tail -5 $O/ragged.txt
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-20s value %9.0f M rq/s ms/step %.3f kernel_ms %.3f req/launch %d (lines %d ktab %d jump %d)" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel_ms"], r["requests_per_launch"], r["rank_line_requests"], r["ktab_lookups"], r.get("jump_lookups",0)))
    except Exception as e:
        print(os.path.basename(f), "no result:", e)
PY
