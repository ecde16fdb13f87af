#!/usr/bin/env python3
"""What binds the literal search kernel at C3: one open of the C3 index, then the kernel's time
against (a) the pattern length (every 8 more characters are one more row-jump lookup per pattern and one more dependent
trip per batch: the slope is the marginal cost of a request, the intercept what a pattern costs before its first), (b)
the batch size (fixed cost per launch) and (c) the share of patterns that miss.  Prints one line per point and the least
squares fit  ms = a + b * requests.

    python tools/c3_bound.py [--workload c3] [--reps 7]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import findex_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--lens", default="8,12,16,20,24,32,40,48,64")
ap.add_argument("--ks", default="125000,250000,500000,1000000,2000000,4000000")
a = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
log2n, sigma, k0, m0, seed = bench.LITERAL[a.workload]
n = 1 << log2n
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
hip.prepare(ktab=True, jump=True)
del bwt
torch.cuda.empty_cache()


def patterns(k, m, miss, sd):
    g = torch.Generator(device=dev)
    g.manual_seed(sd)
    rows = torch.randint(0, n, (k,), generator=g, device=dev, dtype=torch.int64)
    walk = torch.empty((k, m), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hip.lf_walk_batch_dev(rows.data_ptr(), k, m, walk.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    pats = torch.flip(walk, dims=[1]).contiguous()
    if miss > 0:
        mut = torch.rand(k, generator=g, device=dev) < miss
        pos = torch.randint(0, m, (k,), generator=g, device=dev)
        sym = torch.randint(1, sigma + 1, (k,), generator=g, device=dev, dtype=torch.uint8)
        idx = torch.nonzero(mut).squeeze(1)
        pats[idx, pos[idx]] = sym[idx]
    off = torch.arange(0, (k + 1) * m, m, dtype=torch.int64, device=dev)
    return pats.reshape(-1), off


RING = 4      # distinct batches per point, rotated through the timed launches (round 5: no launch sees the batch the device has just searched)


def point(k, m, miss, sd=1):
    ring = [patterns(k, m, miss, sd + 7919 * j) for j in range(RING)]
    sp = torch.empty(k, dtype=torch.int64, device=dev)
    ep = torch.empty(k, dtype=torch.int64, device=dev)
    for pats, off in ring:
        hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    torch.cuda.synchronize()
    pats, off = ring[0]
    hip.stats_reset()
    hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    torch.cuda.synchronize()
    s = hip.stats()
    hits = float((sp < ep).sum().item()) / k
    ts = []
    for r in range(max(a.reps, RING)):
        bp, bo = ring[(r + 1) % RING]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    req = int(s["search_requests"]) + int(s["ktab_lookups"]) + int(s["jump_lookups"]) + int(s["row_lookups"])
    print("k %8d  m %3d  miss %.2f : %7.4f ms (min %7.4f)  requests %9d = %8d rank + %8d ktab + %8d jump + %8d row  -> %6.2f G req/s, "
          "%7.1f G rank-q/s, hits %.3f" % (k, m, miss, ms, min(ts), req, s["search_requests"], s["ktab_lookups"], s["jump_lookups"],
                                          s["row_lookups"], req / ms / 1e6, s["rank_queries"] / ms / 1e6, hits), flush=True)
    return ms, req


def fit(pts, what):
    x = np.array([p[1] for p in pts], dtype=np.float64) / 1e6
    y = np.array([p[0] for p in pts], dtype=np.float64)
    b, c = np.polyfit(x, y, 1)
    print("fit over %s: ms = %.4f + %.5f * M requests  (marginal rate %.1f G req/s, fixed %.1f us)" % (what, c, b, 1.0 / b, c * 1e3), flush=True)


point(k0, m0, 0.10)
st = hip.stats()
print("index %.1f GiB: jump %.1f GiB, rows %.1f GiB, ktab K=%d, tables built in %.0f ms"
      % (st["index_bytes"] / 2**30, st["jump_bytes"] / 2**30, st["row_bytes"] / 2**30, st["ktab_k"], st["tables_build_ms"]), flush=True)
print("---- pattern length, 1M patterns, all hits")
pts = [point(k0, int(m), 0.0) for m in a.lens.split(",")]
fit(pts, "lengths (hits only)")
fit([p for p, m in zip(pts, a.lens.split(",")) if int(m) >= 16], "lengths >= 16 (hits only)")
print("---- pattern length, 1M patterns, 10 % with one byte replaced")
pts = [point(k0, int(m), 0.10) for m in a.lens.split(",")]
fit(pts, "lengths (10 % misses)")
print("---- batch size, m = %d, 10 %% misses" % m0)
pts = [point(int(k), m0, 0.10) for k in a.ks.split(",")]
fit(pts, "batch sizes")
print("---- share of misses, 1M x %d" % m0)
for miss in (0.0, 0.05, 0.10, 0.25, 0.5, 1.0):
    point(k0, m0, miss)
