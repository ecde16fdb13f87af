#!/bin/bash
# round 3: one lane per pattern for the one-row part of a search (k_search_rows), the frontier's row loop variants,
# the three-stream host pipeline
O=gpurun_out/${1:-r03z}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
for v in "nofan -DFMX_FAN=0" "fanstrict -DFMX_FAN_STRICT=1" "wl -DFMX_WAVELOG"; do
  set -- $v; bash tools/build_variant.sh "$@" > $O/build_$1.log 2>&1 || { echo variant $1 build failed; exit 1; }
done
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc = 0 ] || exit 1
for wl in c3 c2 c5; do
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline > $O/${wl}_bench.json 2> $O/${wl}_bench.log; echo "$wl rc=$?"
  FMX_ROWS=0 timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline > $O/${wl}_norows_bench.json 2> $O/${wl}_norows_bench.log; echo "$wl (FMX_ROWS=0) rc=$?"
done
for lib in default nofan fanstrict; do
  if [ $lib = default ]; then unset FMX_LIB; else export FMX_LIB=findex_amd/lib/variants/libfmx_$lib.so; fi
  timeout -k 10 300 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/c4_$lib.txt
done
FMX_LIB=findex_amd/lib/variants/libfmx_wl.so timeout -k 10 300 python tools/wave_timeline.py 2>&1 | grep -v amdgpu.ids > $O/wave_trace_fan.txt
unset FMX_LIB
for ch in 4 2 8; do
  FMX_PIPE_CHUNKS=$ch timeout -k 10 300 python tools/measure_host_path.py c3 2>&1 | grep "pinned\|pageable" > $O/host_chunks_$ch.txt; echo "chunks=$ch"; cat $O/host_chunks_$ch.txt
done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-30s value %9.0f M rq/s ms/step %.3f kernel_ms %.3f frac %.3f" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"]))
    except Exception as e:
        print(os.path.basename(f), "no result:", e)
PY
