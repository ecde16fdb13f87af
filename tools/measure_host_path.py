#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (fmx_search_batch: H2D of patterns and
offsets, k_search4, D2H of the intervals; chunked over two streams) on the C3 workload -- the figure DESIGN.md
quotes beside the HBM-resident `value`; never the bench value."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import findex_amd  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
log2n, sigma, k, m, seed = bench.LITERAL[wl]
n = 1 << log2n
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
del bwt
pats, off = bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000, dev, stream)
h_p = pats.cpu().numpy()
h_o = off.cpu().numpy().astype(np.uint64)
def steady(fn, warm=6, reps=20):
    """median of `reps` calls after `warm` untimed ones (the first calls on fresh buffers pay page faults and the
    driver's first mapping of the pages: 8-17 ms each)"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


sp = np.zeros(k, dtype=np.uint64)
ep = np.zeros(k, dtype=np.uint64)
hip.stats_reset()
hip.search_batch(h_p, h_o, out=(sp, ep))
ranks = hip.stats()["rank_queries"]
dt = steady(lambda: hip.search_batch(h_p, h_o, out=(sp, ep)))
st = hip.stats()
print("host-pointer path %s, pageable buffers: %.3f ms per call (kernel part %.3f ms), %.0f M rank-queries/s, %.1f M patterns/s "
      "PCIe-inclusive" % (wl, dt * 1e3, st["last_kernel_ms"], ranks / dt / 1e6, k / dt / 1e6))

# the same batch in page-locked buffers from fmx_host_alloc (what a JNI adapter would wrap in direct ByteBuffers)
from findex_amd.searcher import PinnedArray  # noqa: E402
pp, po = PinnedArray(h_p.shape, np.uint8), PinnedArray(h_o.shape, np.uint64)
psp, pep = PinnedArray((k,), np.uint64), PinnedArray((k,), np.uint64)
pp.array[:] = h_p
po.array[:] = h_o
for pipeline in ("off", "on"):
    findex_amd.set_pipeline(pipeline)
    dtp = steady(lambda: hip.search_batch(pp.array, po.array, out=(psp.array, pep.array)))
    assert np.array_equal(psp.array, sp) and np.array_equal(pep.array, ep)
    print("host-pointer path %s, pinned buffers, pipeline %s: %.3f ms per call, %.0f M rank-queries/s, %.1f M patterns/s PCIe-inclusive"
          % (wl, pipeline, dtp * 1e3, ranks / dtp / 1e6, k / dtp / 1e6))
findex_amd.set_pipeline("off")

# per-call latency of the single-query forms the Scala adapter's search()/getPrevRange() map to
one_p = h_p.reshape(k, m)[0].copy()
one_o = np.array([0, m], dtype=np.uint64)
for name, fn in (("search_batch k=1", lambda: hip.search_batch(one_p, one_o)),
                 ("prev_range_batch k=1", lambda: hip.prev_range_batch(np.array([0], dtype=np.uint64), np.array([n], dtype=np.uint64), np.array([65], dtype=np.uint8))),
                 ("occ_batch k=1", lambda: hip.occ_batch(np.array([65], dtype=np.uint8), np.array([12345], dtype=np.int64))),
                 ("search_batch k=1000", lambda: hip.search_batch(h_p[:1000 * m], h_o[:1001]))):
    best = 1e9
    for _ in range(3):                      # best of three rounds: the first rounds of a process see clock ramps
        fn()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        best = min(best, (time.perf_counter() - t0) / 200)
    print("%-22s %.1f us per call" % (name, best * 1e6))
