#!/bin/bash
# per-kernel durations of the round-3 tree against this tree on ONE box (rocprofv3 --kernel-trace --stats):  tools/ab_r03_trace.sh <workload>
WL=${1:-c5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/ab_r03_trace; mkdir -p $O
for rep in 1 2; do for tree in .r03tree .; do
  tag=$( [ $tree = . ] && echo now || echo r03 )
  extra=$( [ $tree = . ] && echo "--no-host-path" )
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $REPO/$tree/bench.py --workload $WL --no-cpu-baseline $extra > $O/$tag.json 2> $O/$tag.err) || echo "$tag failed"
  python - $O/$tag $tag <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Name"] and int(r["Calls"]) >= 5:
            print("%s %-55s calls %3s avg %8.1f us min %8.1f" % (sys.argv[2], r["Name"].split("(")[0].replace("fmx::","")[:55], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
  rm -rf $O/$tag
done; done
