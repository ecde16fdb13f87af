#!/bin/bash
O=gpurun_out/${1:-r03r}; mkdir -p $O
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
export FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_lockstep.so
timeout -k 10 1000 bash tools/rocprof_passes.sh $O/prof_c3 c3 > $O/passes_c3.log 2>&1; tail -1 $O/passes_c3.log
python tools/summarize_prof.py $O/prof_c3 $O/sum_c3 > /dev/null 2>&1 && echo summarized
rm -rf $O/prof_c3/trace $O/prof_c3/pmc*/*/*.db 2>/dev/null
grep -E "k_search4|k_jump" $O/sum_c3_summary.md
