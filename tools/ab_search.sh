#!/bin/bash
# A/B of the search-kernel variants on the GPU box: parity subset + bench line per variant.
# Usage: tools/ab_search.sh "1 4" [workload]     (1 = generic kernel, anything else = k_search4)
WL=${2:-c3}
mkdir -p gpurun_out/ab
for v in $1; do
  export FMX_SEARCH_VARIANT=$v
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fixture or synthetic or ragged or full_size or c1" > gpurun_out/ab/pytest_v$v.log 2>&1
  echo "variant $v pytest exit $? : $(tail -1 gpurun_out/ab/pytest_v$v.log)"
  python bench.py --steps 30 --warmup 5 --workload $WL --no-cpu-baseline > gpurun_out/ab/bench_v$v.json 2> gpurun_out/ab/bench_v$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab/bench_v$v.json"))
print("variant $v: %.0f M ranks/s, kernel %.4f ms, frac %.3f, ms/step %.4f" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["ms_per_step"]))
PY
done
