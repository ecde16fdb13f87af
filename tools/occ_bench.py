#!/usr/bin/env python3
"""Throughput of the batched primitives beside the search kernel on one workload's index:
occ_batch (K2), prev_range_batch (K4), lf_walk_batch, psi_batch, next_substr_batch.  Usage: python tools/occ_bench.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, findex_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
log2n, sigma, k, m, seed = bench.LITERAL[wl]
n = 1 << log2n
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
del bwt
st0 = hip.stats()["index_bytes"]
g = torch.Generator(device=dev); g.manual_seed(42)
kc = 1 << 24
qc = torch.randint(1, sigma + 1, (kc,), generator=g, device=dev, dtype=torch.uint8)
qi = torch.randint(0, n, (kc,), generator=g, device=dev, dtype=torch.int64)
qj = torch.minimum(qi + torch.randint(1, 1 << 20, (kc,), generator=g, device=dev, dtype=torch.int64), torch.tensor(n, device=dev))
o1 = torch.empty(kc, dtype=torch.int64, device=dev); o2 = torch.empty_like(o1)

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

ms = timed(lambda: hip.occ_batch_dev(qc.data_ptr(), qi.data_ptr(), o1.data_ptr(), kc, stream))
print("occ_batch        %d queries: %.3f ms, %.1f G rank-queries/s" % (kc, ms, kc / ms / 1e6))
ms = timed(lambda: hip.prev_range_batch_dev(qi.data_ptr(), qj.data_ptr(), qc.data_ptr(), o1.data_ptr(), o2.data_ptr(), kc, stream))
print("prev_range_batch %d steps:   %.3f ms, %.1f G rank-queries/s" % (kc, ms, 2 * kc / ms / 1e6))
kw, lw = 1 << 20, 32
rows = qi[:kw].contiguous(); wb = torch.empty((kw, lw), dtype=torch.uint8, device=dev)
ms = timed(lambda: hip.lf_walk_batch_dev(rows.data_ptr(), kw, lw, wb.data_ptr(), 0, stream))
print("lf_walk_batch    %d x %d:    %.3f ms, %.1f G LF steps/s" % (kw, lw, ms, kw * lw / ms / 1e6))
ks = 1 << 22
srows = qi[:ks].contiguous(); so = torch.empty(ks, dtype=torch.int64, device=dev)
hip.psi_batch_dev(srows.data_ptr(), so.data_ptr(), ks, stream); torch.cuda.synchronize()      # builds the select directory
ms = timed(lambda: hip.psi_batch_dev(srows.data_ptr(), so.data_ptr(), ks, stream))
print("psi_batch        %d rows:    %.3f ms, %.2f G Psi steps/s (select directory %.2f GiB)"
      % (ks, ms, ks / ms / 1e6, (hip.stats()["index_bytes"] - st0) / 2**30))
kn, ln = 1 << 19, 16
nb = torch.empty((kn, ln), dtype=torch.uint8, device=dev); nl = torch.empty(kn, dtype=torch.int32, device=dev)
ms = timed(lambda: hip.next_substr_batch_dev(srows.data_ptr(), kn, ln, nb.data_ptr(), nl.data_ptr(), stream))
print("next_substr      %d x %d:    %.3f ms, %.2f G Psi steps/s" % (kn, ln, ms, kn * ln / ms / 1e6))
