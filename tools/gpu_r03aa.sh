#!/bin/bash
O=gpurun_out/${1:-r03aa}; mkdir -p $O
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
bash tools/build_variant.sh wl -DFMX_WAVELOG > $O/build_wl.log 2>&1 || { echo variant build failed; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "jump or pipelined or host" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
FMX_LIB=findex_amd/lib/variants/libfmx_wl.so timeout -k 10 300 python tools/wave_timeline.py 2>&1 | grep -v amdgpu.ids > $O/wave_trace_fan.txt; head -16 $O/wave_trace_fan.txt | cut -c1-400
for cfg in "c5 -1" "c3 1"; do
  set -- $cfg
  (cd /tmp && export TMPDIR=/tmp && FMX_ROWS=$2 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_$1 -- python3 $REPO/bench.py --workload $1 --no-cpu-baseline > $REPO/$O/$1_rows_under_rocprof.json 2> $REPO/$O/$1_rows_under_rocprof.err); echo "rocprof bench $1 exit $?"
  python - $O $1 <<'PY'
import csv,glob,sys
O,wl=sys.argv[1],sys.argv[2]
for f in glob.glob("%s/trace_%s/*/*_kernel_stats.csv"%(O,wl)):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Name"]: print(wl, r["Name"].split("(")[0][:60], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3))
PY
  rm -rf $O/trace_$1
done
