#!/bin/bash
# round 4's GPU-box sessions, by part:  tools/gpu_r04.sh <tag> <part> [...]
#   bound   : what binds k_search4: counters the box lists, the request-mix microbenchmark,
#             the length / batch-size sweeps of the real kernel, PMC passes with the translation / latency counters
O=gpurun_out/${1:-r04}; mkdir -p $O
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail -5 $O/build.log; exit 1; }
for PART in "$@"; do
case $PART in
bound)
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 -L > $REPO/$O/counters_avail.txt 2>&1); echo "list-avail rc=$?"
  grep -o -i -E '\b(TCP_UTCL1|TCP_TCC|TCP_TCP|TCP_PENDING|TCC_EA0_RDREQ|TCC_TAG_STALL|TCC_BUBBLE|TCP_TA|TA_BUSY|TA_ADDR_STALL|TD_)[A-Z0-9_]*' $O/counters_avail.txt | sort -u > $O/counters_of_interest.txt; wc -l $O/counters_of_interest.txt
  timeout -k 10 400 tools/ubench/mix > $O/ubench_mix.txt 2>&1; echo "mix rc=$?"; tail -3 $O/ubench_mix.txt
  timeout -k 10 500 python tools/c3_bound.py 2>&1 | grep -v amdgpu.ids > $O/c3_bound.txt; echo "c3_bound rc=$?"; tail -4 $O/c3_bound.txt
  ;;
pmc_bound)
  # translation / latency counters on the real kernel and on the microbenchmark's D-only and mixed programs
  i=0
  for grp in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
             "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum" \
             "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"; do
    i=$((i+1))
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/$O/pmcb$i -- python3 $REPO/tools/prof_workload.py --workload c3 > $REPO/$O/pmcb$i.log 2>&1) || echo "pass $i failed: $grp"
    (cd /tmp && export TMPDIR=/tmp && MIX_MODE=sep timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/$O/pmcm$i -- $REPO/tools/ubench/mix 77 64 32 4 D KDDRJJJ J > $REPO/$O/pmcm$i.log 2>&1) || echo "mix pass $i failed: $grp"
    echo "pmc_bound pass $i done: $grp"
  done
  python - $O <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
agg=collections.defaultdict(list)
for f in sorted(glob.glob(O+"/pmc[bm]*/*/*_counter_collection.csv")):
    tag="mix" if "/pmcm" in f else "c3"
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("fmx::","")
        if tag=="c3" and not ("k_search4" in k or "k_occ" in k or "k_search_defer" in k): continue
        agg[(tag,k,r["Counter_Name"])].append((int(r["Dispatch_Id"]),float(r["Counter_Value"])))
with open(O+"/bound_counters.csv","w",newline="") as fo:
    w=csv.writer(fo); w.writerow(["Run","Kernel","Counter","Dispatches","Values"])
    for (t,k,c),v in sorted(agg.items()):
        v.sort()
        w.writerow([t,k,c,len(v)," ".join("%.6g"%x for _,x in v[:12])])
print("bound_counters.csv written:",len(agg),"rows")
PY
  rm -rf $O/pmcb* $O/pmcm*.d 2>/dev/null
  ;;
tests)
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -5 $O/pytest_gpu.log
  ;;
c3bound)
  timeout -k 10 500 python tools/c3_bound.py 2>&1 | grep -v amdgpu.ids > $O/c3_bound.txt; echo "c3_bound rc=$?"; grep "^fit\|m  32  miss 0.10" $O/c3_bound.txt
  ;;
bench:*)
  wl=${PART#bench:}
  timeout -k 10 500 python bench.py --workload $wl > $O/${wl}_bench.json 2> $O/${wl}_bench.log; echo "bench $wl rc=$?"
  python - $O/${wl}_bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("value %.0f M rq/s  ms/step %.4f  kernel_ms %.4f  frac %.3f  req/s %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("requests_G_per_s")))
except Exception as e:
    print("no result:", e)
PY
  ;;
ab:*)
  # A/B of library variants on one box:  ab:<workload>:<tag>[,<tag>..]   ("-" = the product build)
  spec=${PART#ab:}; wl=${spec%%:*}; tags=${spec#*:}
  for rep in 1 2; do for tag in ${tags//,/ }; do
    lib=""; [ "$tag" != "-" ] && lib="FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_$tag.so"
    echo -n "$wl $tag: "
    env $lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-host-path --steps 20 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.4f kernel %.4f value %.0f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
  done; done
  ;;
envab:*)
  # the bench under environment settings:  envab:<workload>:<VAR=val>[+<VAR=val>..],..   ("-" = none)
  spec=${PART#envab:}; wl=${spec%%:*}; sets=${spec#*:}
  for rep in 1 2; do for e in ${sets//,/ }; do
    ev=""; [ "$e" != "-" ] && ev=${e//+/ }
    echo -n "$wl [$e]: "
    env $ev timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-host-path --steps 20 2>$O/envab.err | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.4f kernel %.4f value %.0f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
    grep -h "census" $O/envab.err | sort | uniq -c | sed 's/^/    /'
  done; done
  ;;
wgs:*)
  # the search kernel on fewer resident workgroups per CU than the occupancy query allows:  wgs:<workload>:<n>,<n>..
  spec=${PART#wgs:}; wl=${spec%%:*}; vals=${spec#*:}
  for rep in 1 2; do for v in ${vals//,/ }; do
    echo -n "$wl FMX_SEARCH_WGS=$v: "
    FMX_SEARCH_WGS=$v timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-host-path --steps 20 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.4f kernel %.4f value %.0f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
  done; done
  ;;
overlap)
  # does the exchange run beside the next step's search?  one-rank RCCL (its gather is a device kernel like any rank's),
  # the collective's stream at default / high priority, the search kernel on all / fewer workgroups per CU
  for prio in 0 1; do for wgs in 0 4 3 2; do
    FMX_BENCH_NCCL_PRIO=$prio FMX_SEARCH_WGS=$( [ $wgs = 0 ] && echo 8 || echo $wgs ) timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload c3 --steps 20 --warmup 3 --no-cpu-baseline > $O/overlap_p${prio}_w${wgs}.json 2> $O/overlap_p${prio}_w${wgs}.log
    python - $O/overlap_p${prio}_w${wgs}.json $prio $wgs <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); x=d["exchange"]
    print("nccl prio %s search wgs/CU %s: ms/step %.4f search_ms %.4f gather_ms %.4f sum %.4f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], x["search_ms"], x["gather_ms"], x["search_ms"]+x["gather_ms"]))
except Exception as e:
    print("no result:", e)
PY
  done; done
  ;;
rccl1:*)
  wl=${PART#rccl1:}
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/${wl}_one_rank_rccl.json 2> $O/${wl}_one_rank_rccl.log; echo "$wl (1-rank RCCL) rc=$?"
  python - $O/${wl}_one_rank_rccl.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); x=d.get("exchange")
    print("value %.0f ms/step %.4f exchange %s" % (d["value"], d["ms_per_step"], {k:x[k] for k in ("form","delivery","gather_ms","payload_bytes_per_rank","search_ms")} if x else None))
    if x: print("  all forms:", {k:round(v["gather_ms"],4) for k,v in x["all_forms"].items()})
except Exception as e:
    print("no result:", e)
PY
  ;;
trace:*)
  wl=${PART#trace:}
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_$wl -- python3 $REPO/bench.py --workload $wl --no-cpu-baseline --no-host-path > $REPO/$O/${wl}_bench_under_rocprof.json 2> $REPO/$O/${wl}_bench_under_rocprof.err); echo "rocprof bench $wl exit $?"
  python - $O $wl <<'PY'
import csv,glob,sys
O,wl=sys.argv[1],sys.argv[2]
rows=[]
for f in glob.glob("%s/trace_%s/*/*_kernel_stats.csv"%(O,wl)):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Name"]: rows.append(r)
with open("%s/%s_bench_kernel_stats.csv"%(O,wl),"w",newline="") as fo:
    w=csv.writer(fo); w.writerow(["Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","StdDev"])
    for r in rows: w.writerow([r["Name"].split("(")[0].replace("fmx::",""),r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["MinNs"],r["MaxNs"],r["StdDev"]])
for r in rows:
    if int(r["Calls"])>=5: print("%-60s calls %4s avg %9.1f us min %9.1f" % (r["Name"].split("(")[0].replace("fmx::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
  # the last call's launches on the device's clock: begin and end of each, microseconds from the first one's begin
  python - $O $wl <<'PY' > $O/${wl}_last_call_timeline.txt
import csv,glob,sys
O,wl=sys.argv[1],sys.argv[2]
rows=[]
for f in glob.glob("%s/trace_%s/*/*_kernel_trace.csv"%(O,wl)):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("fmx::","").replace("void ","")))
rows.sort()
# a call = from a k_frontier_reset (or the step's first kernel) to the next one
firsts=[i for i,r in enumerate(rows) if r[2].startswith(("k_frontier_reset","k_search4"))]
if len(firsts)>=3:
    a,b=firsts[-3],firsts[-1]
    t0=rows[a][0]
    for s,e,n in rows[a:b]: print("%9.2f .. %9.2f  (%7.2f us)  %s"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,n[:50]))
PY
  rm -rf $O/trace_$wl
  ;;
prof:*)
  wl=${PART#prof:}
  timeout -k 10 1100 bash tools/rocprof_passes.sh $O/prof_$wl $wl > $O/passes_$wl.log 2>&1; tail -1 $O/passes_$wl.log
  python tools/summarize_prof.py $O/prof_$wl $O/sum_$wl > /dev/null 2>&1 && echo "summarized $wl"
  rm -rf $O/prof_$wl
  ;;
*) echo "unknown part $PART";;
esac
done
