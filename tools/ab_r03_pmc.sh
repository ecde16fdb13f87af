#!/bin/bash
# dynamic instruction counts of the round-3 tree's kernels against this tree's, one rocprofv3 --pmc pass each:  tools/ab_r03_pmc.sh <workload>
WL=${1:-c5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/ab_r03_pmc; mkdir -p $O
for tree in .r03tree .; do
  tag=$( [ $tree = . ] && echo now || echo r03 )
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/$tag -- python3 $REPO/$tree/tools/prof_workload.py --workload $WL > $O/$tag.log 2>&1) || echo "$tag failed"
  python - $O/$tag $tag <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("fmx::","")
        if "k_search" in k: agg[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k,c),v in sorted(agg.items()):
    print("%s %-55s %-20s n=%d mean %.4g" % (sys.argv[2], k[:55], c, len(v), sum(v)/len(v)))
PY
  rm -rf $O/$tag
done
