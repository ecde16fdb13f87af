import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench, findex_amd
from findex_amd.searcher import PinnedArray
log2n, sigma, k, m, seed = bench.LITERAL["tiny"]
k = 1_000_000
n = 1 << 26
dev = torch.device("cuda", 0)
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
pats, off = bench.make_patterns(torch, hip, n, sigma, k, m, 5, dev, 0)
h_p = pats.cpu().numpy(); h_o = off.cpu().numpy().astype(np.uint64)
pp, po = PinnedArray(h_p.shape, np.uint8), PinnedArray(h_o.shape, np.uint64)
psp, pep = PinnedArray((k,), np.uint64), PinnedArray((k,), np.uint64)
pp.array[:] = h_p; po.array[:] = h_o
for i in range(8):
    print("--- pinned call", i, file=sys.stderr)
    t0=time.perf_counter(); hip.search_batch(pp.array, po.array, out=(psp.array, pep.array)); print("call %.3f ms"%((time.perf_counter()-t0)*1e3), file=sys.stderr)
sp = np.zeros(k, dtype=np.uint64); ep = np.zeros(k, dtype=np.uint64)
for i in range(2):
    print("--- pageable call (prefaulted out)", i, file=sys.stderr)
    t0=time.perf_counter(); hip.search_batch(h_p, h_o, out=(sp, ep)); print("call %.3f ms"%((time.perf_counter()-t0)*1e3), file=sys.stderr)
