#!/bin/bash
# rocprofv3 kernel trace of tools/regex_c4.py: per-level durations of the frontier kernel.
OUT=${1:-gpurun_out/c4prof}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/trace" -- python3 "$REPO/tools/regex_c4.py" > "$REPO/$OUT/trace.log" 2>&1 || echo "trace failed"
cd "$REPO"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_frontier" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-24:]))
rows.sort()
# the last match call = the last run of 64+1 launches
last = rows[-66:]
t0 = last[0][0]
print("launch  start_us  dur_us  kernel")
for i, (a, b, k) in enumerate(last):
    print("%3d %9.1f %8.1f  %s" % (i, (a - t0) / 1e3, (b - a) / 1e3, k))
PY
