#!/bin/bash
# Runs the rocprofv3 passes the bench's roofline/traffic figures come from (on the GPU box):
#   pass 0: --kernel-trace --stats            (per-kernel durations)
#   pass 1..: --pmc <counters> --kernel-trace (one counter group per run; never with other traces)
# Usage: tools/rocprof_passes.sh <out-dir> [workload]
OUT=${1:-gpurun_out/prof}
WL=${2:-c3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/trace" -- python3 "$REPO/tools/prof_workload.py" --workload "$WL" > "$REPO/$OUT/trace.log" 2>&1
echo "trace done"
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$REPO/$OUT/pmc$i" -- python3 "$REPO/tools/prof_workload.py" --workload "$WL" > "$REPO/$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
  echo "pmc pass $i done: $grp"
  i=$((i+1))
done
