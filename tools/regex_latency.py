#!/usr/bin/env python3
"""Latency of single-regex matches through a resident batch (the interactive shape of ReTree.matchSA):
a literal-heavy regex is a frontier of one or a few elements for many levels.  FMX_FRONTIER_TAIL=0 shows
the grid-kernel-only path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import findex_amd
from helpers import synth_bwt

bwt, eof, counts = synth_bwt(2_000_000, 97, 100, 12)
sa = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
for re in ("abcdabcdabcdabcdabcdabcd", "ab(c|d)abcdab[abc]dabcdabcd", "a[ab]*c"):
    batch = findex_amd.ReTree.prepare_batch(sa, [findex_amd.ReTree(findex_amd.REParser.re2post(re))])
    out, per = batch.match_raw(max_steps=64, cap=1 << 16)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            out, per = batch.match_raw(max_steps=64, cap=1 << 16)
        best = min(best, (time.perf_counter() - t0) / 100)
    print("%-32s %4d results: %.1f us per call (kernels %.1f us)  [FMX_FRONTIER_TAIL=%s]"
          % (re, out.size, best * 1e6, sa.stats()["last_kernel_ms"] * 1e3, os.environ.get("FMX_FRONTIER_TAIL", "1")))
