#!/bin/bash
# round 5's GPU-box sessions, by part:  tools/gpu_r05.sh <tag> <part> [...]     (other parts: tools/gpu_r04.sh)
#   cold[:wl]     tools/c3_cold.py: per-launch durations of first / replayed / ring / flushed launches (tables on and off)
#   coldpmc[:wl]  the same under rocprofv3 --pmc, one counter group per pass (never with other traces), labelled per phase
#   coldtrace     the bench under rocprofv3 --kernel-trace: every k_search4 launch of the session, in order
#   alloc         tools/ubench/alloc: what device allocation costs by API, chunk size and host threads
O=gpurun_out/${1:-r05}; mkdir -p $O
TAG=$1
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail -5 $O/build.log; exit 1; }
for PART in "$@"; do
case $PART in
cold|cold:*)
  wl=c3; [ "$PART" != cold ] && wl=${PART#cold:}
  timeout -k 10 400 python tools/c3_cold.py --workload $wl 2>&1 | grep -v amdgpu.ids > $O/${wl}_cold.txt; echo "cold $wl rc=$?"; grep "^SUMMARY" $O/${wl}_cold.txt
  timeout -k 10 400 python tools/c3_cold.py --workload $wl --tables off 2>&1 | grep -v amdgpu.ids > $O/${wl}_cold_tables_off.txt; echo "cold (tables off) rc=$?"; grep "^SUMMARY" $O/${wl}_cold_tables_off.txt
  ;;
coldpmc|coldpmc:*)
  wl=c3; [ "$PART" != coldpmc ] && wl=${PART#coldpmc:}
  mkdir -p $O/coldpmc_$wl
  i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
             "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
             "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/$O/coldpmc_$wl/pass$i -- python3 $REPO/tools/c3_cold.py --workload $wl > $REPO/$O/coldpmc_$wl/pass$i.log 2>&1) || echo "pass $i failed: $grp"
    echo "coldpmc pass $i done: $grp"
  done
  python tools/c3_cold_pmc.py $O/coldpmc_$wl $O/${wl}_cold_counters.csv > $O/${wl}_cold_counters.txt 2>&1; cat $O/${wl}_cold_counters.txt
  rm -rf $O/coldpmc_$wl/pass*/
  ;;
coldtrace)
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_bench -- python3 $REPO/bench.py --steps 20 --warmup 5 > $REPO/$O/c3_bench_under_rocprof.json 2> $REPO/$O/c3_bench_under_rocprof.err); echo "rocprof bench exit $?"
  python - $O <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
rows=[]
for f in glob.glob("%s/trace_bench/*/*_kernel_trace.csv"%O):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("fmx::","").replace("void ","")))
rows.sort()
with open(O+"/c3_bench_launches_in_order.txt","w") as fo:
    fo.write("# every fmx kernel of one `python bench.py --steps 20 --warmup 5` session under rocprofv3 --kernel-trace, in order: start (ms since the first), duration (us), kernel\n")
    t0=rows[0][0] if rows else 0
    for s,e,n in rows:
        if n.startswith(("k_search4","k_lf_walk","k_jump","k_row3","k_ktab","k_occ")): fo.write("%10.3f %9.2f  %s\n"%((s-t0)/1e6,(e-s)/1e3,n[:70]))
srch=[(e-s)/1e3 for s,e,n in rows if n.startswith("k_search4")]
print("k_search4 launches:",len(srch)); print(" ".join("%.1f"%x for x in srch))
# rocprofv3's own per-kernel statistics, and beside them the search kernel's WITHOUT fmx_prepare's calibration launches (the leading
# launches of an instantiation that last about as long as the 60-us spin they consist of: < 0.6 of the instantiation's median)
import statistics
stats=[]
for f in glob.glob("%s/trace_bench/*/*_kernel_stats.csv"%O):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Name"]: stats.append(r)
by=collections.OrderedDict() if False else {}
for s_,e,n in rows:
    if n.startswith("k_search4"): by.setdefault(n,[]).append((e-s_))
with open(O+"/c3_bench_kernel_stats.csv","w",newline="") as fo:
    w=csv.writer(fo); w.writerow(["Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","StdDev"])
    for r in stats: w.writerow([r["Name"].split("(")[0].replace("fmx::",""),r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["MinNs"],r["MaxNs"],r["StdDev"]])
    for n,v in by.items():
        med=statistics.median(v); lead=0
        while lead < len(v) and v[lead] < 0.6*med: lead+=1
        u=v[lead:]
        w.writerow([n+" [searches only: without the %d calibration launches of fmx_prepare]"%lead,len(u),sum(u),"%.1f"%(sum(u)/len(u)),min(u),max(u),"%.1f"%(statistics.pstdev(u) if len(u)>1 else 0.0)])
        print("%s: %d searches, avg %.1f us (min %.1f, max %.1f); %d calibration launches left out"%(n[:50],len(u),sum(u)/len(u)/1e3,min(u)/1e3,max(u)/1e3,lead))
PY
  rm -rf $O/trace_bench
  ;;
alloc)
  hipcc -O2 --offload-arch=gfx950 -o tools/ubench/alloc tools/ubench/alloc.hip -lpthread > $O/alloc_build.log 2>&1 || { echo "alloc build failed"; tail -3 $O/alloc_build.log; }
  timeout -k 10 500 tools/ubench/alloc > $O/alloc.txt 2>&1; echo "alloc rc=$?"; cat $O/alloc.txt
  ;;
sqpmc:*)
  # the SQ counters of tools/c3_cold.py's launches under an environment setting:  sqpmc:<VAR=val|->   (one rocprofv3 --pmc pass)
  setting=${PART#sqpmc:}; tag=${setting//[^A-Za-z0-9]/_}
  mkdir -p $O/sq_$tag
  ( [ "$setting" != "-" ] && export $setting; cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $REPO/$O/sq_$tag/pass1 -- python3 $REPO/tools/c3_cold.py > $REPO/$O/sq_$tag/pass1.log 2>&1 ) || echo "sq pass failed"
  ( [ "$setting" != "-" ] && export $setting; cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $REPO/$O/sq_$tag/pass2 -- python3 $REPO/tools/c3_cold.py > $REPO/$O/sq_$tag/pass2.log 2>&1 ) || echo "sq pass 2 failed"
  python tools/c3_cold_pmc.py $O/sq_$tag $O/sq_${tag}_counters.csv > $O/sq_${tag}_counters.txt 2>&1; grep "^ring\|^s_ring\|^replay " $O/sq_${tag}_counters.txt
  rm -rf $O/sq_$tag/pass*/
  ;;
writevalue)
  hipcc -O2 --offload-arch=gfx950 -o tools/ubench/writevalue tools/ubench/writevalue.hip > $O/wv_build.log 2>&1
  timeout -k 10 120 tools/ubench/writevalue > $O/writevalue.txt 2>&1; echo "writevalue rc=$?"; cat $O/writevalue.txt
  ;;
abenv:*)
  # A/B of library variants and environment settings on one box:  abenv:<workload>:<tag>[+VAR=val..],..   ("-" = the product build)
  spec=${PART#abenv:}; wl=${spec%%:*}; items=${spec#*:}
  for rep in 1 2; do for it in ${items//,/ }; do
    tag=${it%%+*}; envs=""; [ "$it" != "$tag" ] && envs=${it#*+} && envs=${envs//+/ }
    lib=""; [ "$tag" != "-" ] && lib="FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_$tag.so"
    echo -n "$wl [$it]: "
    env $lib $envs timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-host-path --no-rank-only --steps 40 --warmup 5 2>$O/abenv.err | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.4f kernel %.4f replayed %.4f req/s %s miss_none %s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['replayed_batch_ms'], d.get('requests_G_per_s', d['roofline'].get('device_rank_queries_G_per_s')), (d.get('miss_none') or {}).get('ms_per_step')))" || tail -3 $O/abenv.err
  done; done
  ;;
*) bash tools/gpu_r04.sh $TAG $PART ;;
esac
done
