#!/bin/bash
# A/B of the round-3 tree (a git worktree at .r03tree, built in place) against this tree on ONE box:  tools/ab_r03.sh <workload>
WL=${1:-c5}
fmt='import json,sys; d=json.load(sys.stdin); print("%s ms/step %.4f kernel %.4f value %.0f" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"]))'
for i in 1 2; do
  (cd .r03tree && timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline --steps 20 2>/dev/null | python -c "$fmt" "r03 $WL")
  timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline --no-host-path --steps 20 2>/dev/null | python -c "$fmt" "now $WL"
done
