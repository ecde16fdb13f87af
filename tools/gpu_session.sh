#!/bin/bash
# One GPU-box session, in stages chosen on the command line (outputs under gpurun_out/<tag>/):
#   tools/gpu_session.sh <tag> [tests] [smoke] [bench:<workload>]... [prof:<workload>]... [trace:<workload>]...
# Stages run in the order given and the session stops at the first stage that fails or times out.
TAG=$1; shift
OUT=gpurun_out/$TAG
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
for stage in "$@"; do
  case $stage in
    tests)
      timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
      echo "pytest exit $rc: $(tail -1 $OUT/pytest_gpu.log)"; [ $rc -eq 0 ] || { tail -30 $OUT/pytest_gpu.log; exit 1; } ;;
    smoke)
      timeout -k 10 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; rc=$?
      echo "smoke exit $rc"; [ $rc -eq 0 ] || { tail -20 $OUT/smoke.log; exit 1; } ;;
    bench:*)
      wl=${stage#bench:}
      timeout -k 10 600 python bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; rc=$?
      echo "bench $wl exit $rc"; tail -4 $OUT/bench_$wl.err; [ $rc -eq 0 ] || exit 1
      python - <<PY
import json
d = json.load(open("$OUT/bench_$wl.json"))
r = d["roofline"]
print("  value %.0f %s, %.3f ms/step, kernel %.3f ms, roofline %.0f GB/s frac %.3f, requests %.1f G/s" % (
    d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r.get("requests_G_per_s", 0)))
if "cpu_baseline" in d:
    print("  cpu_baseline %.1f %s on %d cores (n=%d)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["n"]))
PY
      ;;
    trace:*)
      wl=${stage#trace:}
      (cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/bench_trace_$wl -- python3 $REPO/bench.py --workload $wl --no-cpu-baseline > $REPO/$OUT/bench_under_rocprof_$wl.json 2> $REPO/$OUT/bench_under_rocprof_$wl.err); rc=$?
      echo "rocprof bench $wl exit $rc"; [ $rc -eq 0 ] || { tail -5 $OUT/bench_under_rocprof_$wl.err; exit 1; } ;;
    prof:*)
      wl=${stage#prof:}
      tools/rocprof_passes.sh $OUT/prof_$wl $wl > $OUT/passes_$wl.log 2>&1; rc=$?
      tail -3 $OUT/passes_$wl.log; [ $rc -eq 0 ] || exit 1
      python tools/summarize_prof.py $OUT/prof_$wl $OUT/sum_$wl > /dev/null && cat $OUT/sum_${wl}_summary.md | head -40 ;;
    *) echo "unknown stage $stage"; exit 2 ;;
  esac
done
