#!/bin/bash
# One PMC pass over tools/ab_layout.py for the search kernel:
#   tools/prof_counters.sh <out-dir> "<counters>" [workload] [layoutA] [layoutB]
OUT=${1:-gpurun_out/pc}
GRP=${2:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQC_ICACHE_MISSES"}
WL=${3:-c3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc $GRP --kernel-trace --output-format csv -d "$REPO/$OUT/pmc" -- python3 "$REPO/tools/ab_layout.py" "$WL" ${4:-onehot} ${5:-onehot} > "$REPO/$OUT/pmc.log" 2>&1 || echo "pass failed"
cd "$REPO"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_search4" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
for c, (n, v) in sorted(acc.items()):
    print("%-40s n=%3d mean %.5g" % (c, n, v / n))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_search4" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-30:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    print("%-40s n=%3d mean %.1f us (under PMC)" % (k, len(v), sum(v) / len(v) / 1e3))
PY
