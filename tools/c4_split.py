#!/usr/bin/env python3
"""C4 matched as 1 / 2 / 4 concurrent slices on ONE device (fmx_regex_batch_*_multi with the same handle repeated):
do the launches of one slice fill the wave slots another slice's launch leaves idle?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, findex_amd
from findex_amd.regex import RegexBatchMulti
log2n, k, seed, max_len = bench.REGEX["c4"][:4]
n = 1 << log2n
dev = torch.device("cuda", 0)
bwt, eof = bench.make_bwt(torch, n, bench.C4_ALPHABET, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
del bwt
res, trees = bench.make_regexes(k, seed * 1000)
ref = None
for parts in (1, 2, 3, 4):
    mb = RegexBatchMulti([hip] * parts, trees)
    for _ in range(3):
        out, per = mb.match_raw(max_steps=max_len, cap=1 << 22, copy=False)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); out, per = mb.match_raw(max_steps=max_len, cap=1 << 22, copy=False); ts.append(time.perf_counter() - t0)
    if ref is None: ref = out.copy()
    assert out.size == ref.size and all(np.array_equal(out[f], ref[f]) for f in ("regex", "len", "sp", "ep"))
    print("%d slice(s): median %.3f ms per call, best %.3f" % (parts, sorted(ts)[len(ts)//2] * 1e3, min(ts) * 1e3))
