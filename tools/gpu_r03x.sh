#!/bin/bash
O=gpurun_out/${1:-r03x}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
timeout -k 10 300 python - 2>&1 <<'PY' | grep -v amdgpu.ids | tail -60 | tee $O/trace.txt
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch, bench, findex_amd
from findex_amd.searcher import PinnedArray
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
log2n, sigma, k, m, seed = bench.LITERAL["c3"]; n = 1 << log2n
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream); del bwt
pats, off = bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000, dev, stream)
h_p = pats.cpu().numpy(); h_o = off.cpu().numpy().astype(np.uint64)
pp, po = PinnedArray(h_p.shape, np.uint8), PinnedArray(h_o.shape, np.uint64)
psp, pep = PinnedArray((k,), np.uint64), PinnedArray((k,), np.uint64)
pp.array[:] = h_p; po.array[:] = h_o
for _ in range(8): hip.search_batch(pp.array, po.array, out=(psp.array, pep.array))
os.environ["FMX_TRACE"] = "1"
t0 = time.perf_counter(); hip.search_batch(pp.array, po.array, out=(psp.array, pep.array)); print("call %.3f ms" % ((time.perf_counter() - t0) * 1e3))
for jump in ("0",):
    pass
PY
