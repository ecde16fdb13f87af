#!/bin/bash
# rocprofv3 kernel trace + SQ counters of tools/ab_layout.py (both layouts in one process).
OUT=${1:-gpurun_out/prof_ab}
WL=${2:-c3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/trace" -- python3 "$REPO/tools/ab_layout.py" "$WL" > "$REPO/$OUT/trace.log" 2>&1 && echo "trace done" &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d "$REPO/$OUT/pmc1" -- python3 "$REPO/tools/ab_layout.py" "$WL" > "$REPO/$OUT/pmc1.log" 2>&1 && echo "pmc done"
cd "$REPO"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Name"]:
            print("%-60s calls %5s avg %10.1f ns" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])))
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/pmc1/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_search4" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            acc[key][0] += 1; acc[key][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print("%-42s %-22s n=%3d mean %.4g" % (k, c, n, v / n))
PY
