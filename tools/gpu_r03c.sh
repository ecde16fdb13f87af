#!/bin/bash
# reference-order kernel: occupancy / priority sweep on c4ref
set -o pipefail
O=gpurun_out/${1:-r03c}; mkdir -p $O
cat > $O/run.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch, findex_amd, bench
dev = torch.device("cuda", 0)
n = 1 << 30
bwt, eof = bench.make_bwt(torch, n, bench.C4_ALPHABET, 4, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=torch.cuda.current_stream().cuda_stream)
res, trees = bench.make_regexes(100000, 4000)
batch = findex_amd.ReTree.prepare_batch(hip, trees)
for wgs in sys.argv[1:]:
    os.environ["FMX_REF_WGS"] = wgs
    ks = []
    for i in range(6):
        t0 = time.perf_counter(); out, per = batch.match_raw(mode="reference", maxBranching=1024, maxIterations=1000, copy=False); dt = time.perf_counter() - t0
        ks.append((hip.last_kernel_ms(), dt * 1e3))
    print("lib=%s wgs=%s kernel ms %s call ms %.3f results %d" % (os.path.basename(os.environ.get("FMX_LIB", "default")), wgs, " ".join("%.3f" % k[0] for k in ks[1:]), min(k[1] for k in ks[1:]), out.size), flush=True)
PY
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 300 python $O/run.py 7 6 5 4 3 2 1 2>&1 | grep -v amdgpu.ids | tee $O/sweep_default.txt
for v in prio64 prio200; do
  FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_$v.so timeout -k 10 300 python $O/run.py 7 5 3 2>&1 | grep -v amdgpu.ids | tee $O/sweep_$v.txt
done
