#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (plain and under rocprofv3 --kernel-trace --stats),
# PMC passes, host-path timing.  Usage: tools/gpu_round.sh <tag>   (outputs under gpurun_out/<tag>/)
TAG=${1:-r01}
OUT=gpurun_out/$TAG
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? $(tail -1 $OUT/pytest_gpu.log)"
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -3 $OUT/bench.err
python bench.py --workload c2 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench c2 exit $?"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/bench_trace -- python3 $REPO/bench.py --no-cpu-baseline > $REPO/$OUT/bench_under_rocprof.json 2> $REPO/$OUT/bench_under_rocprof.err); echo "rocprof bench exit $?"
tools/rocprof_passes.sh $OUT/prof c3 > $OUT/passes.log 2>&1; tail -2 $OUT/passes.log
python tools/measure_host_path.py c3 > $OUT/host_path.log 2>&1; tail -1 $OUT/host_path.log
