#!/bin/bash
O=gpurun_out/${1:-r03t}; mkdir -p $O
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/tr -- python3 $REPO/bench.py --workload c3 --no-cpu-baseline > $REPO/$O/c3_under_rocprof.json 2> $REPO/$O/err.log); echo "rc=$?"
python - $O <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/tr/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Name"]: print(r["Name"].split("(")[0][-40:], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3), "min %.1f" % (float(r["MinNs"])/1e3))
PY
rm -rf $O/tr
