#!/bin/bash
# round 3, final measurements on one box: bench lines of every workload, the bench command under rocprofv3
# --kernel-trace --stats, the PMC passes (c3, c4, c4ref), one-rank RCCL rehearsals.   tools/gpu_r03_final.sh <tag> [part]
O=gpurun_out/${1:-r03final}; mkdir -p $O
PART=${2:-all}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
if [ $PART = all ] || [ $PART = bench ]; then
  for wl in c3 c4 c4ref c2 c5; do
    timeout -k 10 400 python bench.py --workload $wl > $O/${wl}_bench.json 2> $O/${wl}_bench.log; echo "$wl rc=$?"
  done
  FMX_JUMP=0 timeout -k 10 400 python bench.py --workload c3 --no-cpu-baseline > $O/c3_nojump_bench.json 2> $O/c3_nojump_bench.log; echo "c3 (no jump table) rc=$?"
  FMX_ROWS=0 timeout -k 10 400 python bench.py --workload c5 --no-cpu-baseline > $O/c5_norows_bench.json 2> $O/c5_norows_bench.log; echo "c5 (no row table) rc=$?"
  for wl in tiny c4tiny c4reftiny; do
    timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload $wl --steps 10 --warmup 2 > $O/${wl}_rccl1.json 2> $O/${wl}_rccl1.log; echo "$wl (1-rank RCCL) rc=$?"
  done
  timeout -k 10 300 python tools/calcgaps_chain.py 27 128 2>&1 | grep -v amdgpu.ids > $O/calcgaps_chain.txt
  timeout -k 10 300 python tools/measure_host_path.py c3 2>&1 | grep -v amdgpu.ids > $O/host_path.txt; tail -3 $O/host_path.txt
fi
if [ $PART = all ] || [ $PART = prof ]; then
  for wl in c3 c4 c4ref c5; do
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/bench_trace_$wl -- python3 $REPO/bench.py --workload $wl --no-cpu-baseline > $REPO/$O/${wl}_bench_under_rocprof.json 2> $REPO/$O/${wl}_bench_under_rocprof.err); echo "rocprof bench $wl exit $?"
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
PY
    rm -rf $O/bench_trace_$wl
    [ $wl = c5 ] && continue      # kernel times only; the PMC passes are taken on c3, c4, c4ref
    timeout -k 10 1000 bash tools/rocprof_passes.sh $O/prof_$wl $wl > $O/passes_$wl.log 2>&1; tail -1 $O/passes_$wl.log
    python tools/summarize_prof.py $O/prof_$wl $O/sum_$wl > /dev/null 2>&1 && echo "summarized $wl"
    rm -rf $O/prof_$wl
  done
fi
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-30s value %9.0f M rq/s ms/step %.3f kernel_ms %.3f frac %.3f traffic %s cpu %s" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("traffic"), ("%.1f" % d["cpu_baseline"]["value"]) if "cpu_baseline" in d else "-"))
    except Exception as e:
        print(os.path.basename(f), "no result:", e)
PY
