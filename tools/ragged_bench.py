#!/usr/bin/env python3
"""Uniform vs ragged pattern batches through the search path (lockstep batches of 16: groups whose pattern
ends early idle until their batch ends).  FMX_SEARCH_VARIANT=1 selects the generic one-pattern-per-group kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, findex_amd
log2n, sigma, k, m, seed = bench.LITERAL["c3"]
n = 1 << log2n
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
del bwt
pats, off = bench.make_patterns(torch, hip, n, sigma, k, 64, 7, dev, stream)      # 64-char hits
P = pats.reshape(k, 64)
def run(lens, tag):
    lens_t = torch.as_tensor(lens, device=dev, dtype=torch.int64)
    offs = torch.zeros(k + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(lens_t, 0)
    idx = torch.arange(64, device=dev)[None, :]
    keep = idx >= (64 - lens_t)[:, None]               # the last `len` chars of each 64-char hit: still a hit
    buf = P[keep].contiguous()
    sp = torch.empty(k, dtype=torch.int64, device=dev); ep = torch.empty_like(sp)
    hip.stats_reset()
    hip.search_batch_dev(buf.data_ptr(), offs.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    torch.cuda.synchronize()
    ranks = hip.stats()["rank_queries"]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        hip.search_batch_dev(buf.data_ptr(), offs.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-28s FMX_SEARCH_VARIANT=%s: %.3f ms, %.1f G rank-queries/s, hits %d" % (tag, os.environ.get("FMX_SEARCH_VARIANT", "default"), ms, ranks / ms / 1e6, int((sp < ep).sum())))
rng = np.random.default_rng(1)
run(np.full(k, 32), "uniform 32")
run(rng.integers(1, 65, k), "ragged 1..64")
run(np.where(rng.random(k) < 0.1, 8, 32), "90% 32 / 10% 8")
