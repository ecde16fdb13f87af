#!/bin/bash
# round 3: balance at the end of a frontier launch -- pool sizes, lingering, end-phase dump + second launch
O=gpurun_out/${1:-r03ab}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
for v in "nofan -DFMX_FAN=0" "dump2 -DFMX_ENDDUMP=2" "dump4 -DFMX_ENDDUMP=4" "dump2nofan -DFMX_ENDDUMP=2 -DFMX_FAN=0" "keep96 -DFMX_POOL_KEEP=96" "keep96l8 -DFMX_POOL_KEEP=96 -DFMX_IDLE_LOOKS=8" "keep128l8 -DFMX_POOL_KEEP=128 -DFMX_IDLE_LOOKS=8"; do
  set -- $v; bash tools/build_variant.sh "$@" > $O/build_$1.log 2>&1 || { echo variant $1 build failed; exit 1; }
done
FMX_LIB=findex_amd/lib/variants/libfmx_dump2.so FMX_FRONTIER_CHAIN=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "regex or c4" > $O/pytest_dump2.log 2>&1; echo "pytest(dump2) rc=$?"; tail -2 $O/pytest_dump2.log
run() { # lib chain
  if [ $1 = default ]; then unset FMX_LIB; else export FMX_LIB=findex_amd/lib/variants/libfmx_$1.so; fi
  if [ $2 = - ]; then unset FMX_FRONTIER_CHAIN; else export FMX_FRONTIER_CHAIN=$2; fi
  timeout -k 10 300 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/c4_$1_$2.txt
}
run default -; run nofan -; run dump2 -; run dump2 2; run dump4 2; run dump2nofan 2; run keep96 -; run keep96l8 -; run keep128l8 -; run default -; run nofan -
unset FMX_LIB FMX_FRONTIER_CHAIN
# ---- the three-step row table
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "jump or c5 or bytes_layout or full_size" > $O/pytest_rows3.log 2>&1; echo "pytest(rows3) rc=$?"; tail -3 $O/pytest_rows3.log
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 400 python bench.py --workload c5 --no-cpu-baseline > $O/c5_bench.json 2> $O/c5_bench.log; echo "c5 rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_c5 -- python3 $REPO/bench.py --workload c5 --no-cpu-baseline > $REPO/$O/c5_under_rocprof.json 2> $REPO/$O/c5_under_rocprof.err); echo "rocprof bench c5 exit $?"
python - $O <<'PY'
import csv,glob,sys,json
O=sys.argv[1]
for f in glob.glob("%s/trace_c5/*/*_kernel_stats.csv"%O):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Name"] or "row3" in r["Name"]: print("c5", r["Name"].split("(")[0][:60], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3))
d=json.loads(open(O+"/c5_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("c5 value %.0f ms/step %.3f kernel_ms %.3f tables_build_ms %.0f row_lookups %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], d["config"]["tables_build_ms"], r.get("row_lookups")))
PY
rm -rf $O/trace_c5
