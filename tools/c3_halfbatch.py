#!/usr/bin/env python3
"""Is k_search4 at C3 bound by the round trips of a wave's lockstep batch (time ~ trips, whatever the requests per trip) or by
requests?  The same 1M pattern SLOTS searched (a) full: every slot a 32-character pattern; (b) half: every other slot an EMPTY
pattern (offsets repeat), so that each wave's batches hold 8 real patterns instead of 16 -- half the requests, the same batches
and trips; (c) quarter: one slot in four.  If (b) takes as long as (a), a wave's time is its trips and twice the requests per
trip would be free.   python tools/c3_halfbatch.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, findex_amd
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
log2n, sigma, k, m, seed = bench.LITERAL["c3"]
n = 1 << log2n
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
hip.prepare(ktab=True, jump=True)
del bwt
ring = [bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000 + 7919 * j, dev, stream) for j in range(4)]
sp = torch.empty(k, dtype=torch.int64, device=dev); ep = torch.empty(k, dtype=torch.int64, device=dev)
def run(label, keep_every, miss=True):
    offs = []
    for pats, off in ring:
        lens = torch.zeros(k, dtype=torch.int64, device=dev)
        lens[::keep_every] = m
        # slot q's pattern = the bytes of original pattern q (kept slots) -- offsets into the SAME buffer: kept patterns stay where they are,
        # dropped slots are empty ranges at their own start
        start = torch.arange(0, k * m, m, dtype=torch.int64, device=dev)
        # a valid non-decreasing offsets array needs contiguous ranges: build a compacted buffer instead
        keep = torch.arange(0, k, keep_every, device=dev)
        buf = pats.view(k, m)[keep].contiguous().view(-1)
        o = torch.zeros(k + 1, dtype=torch.int64, device=dev)
        o[1:] = torch.cumsum(lens, 0)
        offs.append((buf, o))
    for b, o in offs:
        hip.search_batch_dev(b.data_ptr(), o.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    torch.cuda.synchronize()
    hip.stats_reset()
    hip.search_batch_dev(offs[0][0].data_ptr(), offs[0][1].data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    torch.cuda.synchronize()
    s = hip.stats()
    req = int(s["search_requests"] + s["ktab_lookups"] + s["jump_lookups"] + s["row_lookups"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 24
    e0.record()
    for i in range(reps):
        b, o = offs[i % len(offs)]
        hip.search_batch_dev(b.data_ptr(), o.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-8s %7d real patterns in %d slots: %.4f ms per launch, %8d requests -> %.2f G req/s" % (label, k // keep_every + (1 if k % keep_every else 0), k, ms, req, req / ms / 1e6), flush=True)
run("full", 1)
run("half", 2)
run("quarter", 4)
run("eighth", 8)
