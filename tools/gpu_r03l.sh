#!/bin/bash
# round 3 bench lines on one box + the one-rank RCCL rehearsal of the driver's launch
set -o pipefail
O=gpurun_out/${1:-r03l}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail $O/build.log; exit 1; }
for wl in c3 c4 c4ref c2 c5; do
  timeout -k 10 400 python bench.py --workload $wl > $O/${wl}_bench.json 2> $O/${wl}_bench.log; echo "$wl rc=$?"
done
# the driver's N > 1 launch, one rank: RCCL initialised, the gather inside the timed region
for wl in tiny c4tiny c4reftiny; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload $wl --steps 10 --warmup 2 > $O/${wl}_rccl1.json 2> $O/${wl}_rccl1.log; echo "$wl (1-rank RCCL) rc=$?"
done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-22s value %9.0f M rq/s ms/step %.3f kernel_ms %.3f frac %.3f ranks_in_group %s cpu %s" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], d["config"].get("ranks_in_group"), ("%.1f" % d["cpu_baseline"]["value"]) if "cpu_baseline" in d else "-"))
    except Exception as e:
        print(os.path.basename(f), "no result:", e)
PY
