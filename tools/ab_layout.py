#!/usr/bin/env python3
"""A/B of two index layouts on one workload: same BWT, same patterns, results compared bit for bit.
Usage: python tools/ab_layout.py [workload] [layoutA] [layoutB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, findex_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
la = sys.argv[2] if len(sys.argv) > 2 else "onehot"
lb = sys.argv[3] if len(sys.argv) > 3 else "bytes"
log2n, sigma, k, m, seed = bench.LITERAL[wl]
n = 1 << log2n
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
findex_amd.set_layout(la)
A = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
pats, off = bench.make_patterns(torch, A, n, sigma, k, m, seed * 1000, dev, stream)

def run(h, tag):
    sp = torch.empty(k, dtype=torch.int64, device=dev); ep = torch.empty_like(sp)
    h.stats_reset()
    h.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    torch.cuda.synchronize()
    st = h.stats()
    ranks = st["rank_queries"]
    for _ in range(3):
        h.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        h.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("%-10s index %.1f GiB, built %.1f ms: %.4f ms/step incl. prep, %.1f G rank-queries/s, hits %d"
          % (tag, st["index_bytes"] / 2**30, st["build_ms"], ms, ranks / ms / 1e6, int((sp < ep).sum())), flush=True)
    return sp, ep

spa, epa = run(A, la)
findex_amd.set_layout(lb)
B = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
spb, epb = run(B, lb)
print("results identical:", bool(torch.equal(spa, spb) and torch.equal(epa, epb)))
