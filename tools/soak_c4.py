#!/usr/bin/env python3
"""Soak of the frontier kernel's in-launch hand-over: the C4 batch matched N times on one resident batch, every
call's (sorted) result list compared with the first call's through a checksum and the count.  With a build that hands
over eagerly (tools/build_variant.sh k64 -DFMX_POOL_KEEP=64; FMX_LIB=...) most elements pass through the queue.
    python tools/soak_c4.py [calls] [workload]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, findex_amd
from findex_amd.regex import RegexBatch
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
wl = sys.argv[2] if len(sys.argv) > 2 else "c4"
log2n, k, seed, max_len = bench.REGEX[wl][:4]
n = 1 << log2n
dev = torch.device("cuda", 0)
bwt, eof = bench.make_bwt(torch, n, bench.C4_ALPHABET, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
del bwt
res, trees = bench.make_regexes(k, seed * 1000)
rb = RegexBatch(hip, trees)
out, per = rb.match_raw(max_steps=max_len, copy=False)
want_n, want_crc, want_per = out.size, zlib.crc32(out.tobytes()), zlib.crc32(per.tobytes())
hip.stats_reset()
bad = 0
t0 = time.time()
for i in range(calls):
    out, per = rb.match_raw(max_steps=max_len, copy=False)
    if out.size != want_n or zlib.crc32(out.tobytes()) != want_crc or zlib.crc32(per.tobytes()) != want_per:
        bad += 1
        print("call %d differs: %d results (want %d)" % (i, out.size, want_n), flush=True)
    if (i + 1) % 1000 == 0:
        print("%d calls, %d differ, %.1fs" % (i + 1, bad, time.time() - t0), flush=True)
st = hip.stats()
print("soak %s: %d calls, %d results each, %d differ; queue entries per call: %d written, %d read" % (
    wl, calls, want_n, bad, st["frontier_queue_writes"] // calls, st["frontier_queue_reads"] // calls))
sys.exit(1 if bad else 0)
