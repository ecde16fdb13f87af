#!/bin/bash
# round 3 profiles: the bench command under rocprofv3 --kernel-trace --stats (c3, c4, c4ref) + the PMC passes
O=gpurun_out/${1:-r03m}; mkdir -p $O
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
for wl in c3 c4 c4ref; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/bench_trace_$wl -- python3 $REPO/bench.py --workload $wl --no-cpu-baseline > $REPO/$O/${wl}_bench_under_rocprof.json 2> $REPO/$O/${wl}_bench_under_rocprof.err); echo "rocprof bench $wl exit $?"
  echo "pass $wl" 
done
for wl in c4 c4ref c3; do
  timeout -k 10 1000 bash tools/rocprof_passes.sh $O/prof_$wl $wl > $O/passes_$wl.log 2>&1; tail -1 $O/passes_$wl.log
  python tools/summarize_prof.py $O/prof_$wl $O/sum_$wl > /dev/null 2>&1 && echo "summarized $wl"
  # keep the merge small: drop the raw traces, keep the summaries
  rm -rf $O/prof_$wl/trace $O/prof_$wl/pmc*/*/*.db 2>/dev/null
done
for wl in c3 c4 c4ref; do
  python - $O $wl <<'PY'
import csv,glob,sys
O,wl=sys.argv[1],sys.argv[2]
rows=[]
for f in glob.glob("%s/bench_trace_%s/*/*_kernel_stats.csv"%(O,wl)):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Name"]: rows.append(r)
with open("%s/%s_bench_kernel_stats.csv"%(O,wl),"w",newline="") as fo:
    w=csv.writer(fo); w.writerow(["Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","StdDev"])
    for r in rows: w.writerow([r["Name"].split("(")[0].replace("fmx::",""),r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["MinNs"],r["MaxNs"],r["StdDev"]])
print(wl, "bench kernel stats rows", len(rows))
PY
  rm -rf $O/bench_trace_$wl
done
ls $O
