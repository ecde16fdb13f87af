#!/usr/bin/env python3
"""The measurement behind not building BWTMerger2.calcGaps (SURVEY 8f-4, F/bwtmerger.scala:981-1023) on the GPU.
Its inner loop is ONE dependent chain: curRank = cf(c) + occ(c, curRank - 1) for every byte of the text merged so
far, with host-side corrections (KMP buffer, longSuffixCmp) between steps -- nothing to batch.  This times such a
chain (an LF walk: same dependency shape, one more dependent read per step) on the GPU, one lane group, against the
oracle's C loop on one host core, on the same index.

    python tools/calcgaps_chain.py [log2n=27] [sigma=128]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import findex_amd  # noqa: E402
import oracle  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = 1 << log2n
dev = torch.device("cuda", 0)
bwt, eof = bench.make_bwt(torch, n, sigma, 77, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
h_bwt = bwt.cpu().numpy()
orc = oracle.NaiveFMSearcher.from_mem(h_bwt, eof, oracle.histogram(h_bwt, eof, threads=8), threads=8)
steps = 200_000
row = n // 7
t0 = time.perf_counter()
end_cpu = orc.lf_chain(row, steps)
t_cpu = time.perf_counter() - t0
hip.lf_walk_batch(np.array([row], dtype=np.uint64), 16, want_bytes=False)       # warm up
t0 = time.perf_counter()
_, end = hip.lf_walk_batch(np.array([row], dtype=np.uint64), steps, want_bytes=False)
t_gpu = time.perf_counter() - t0
assert int(end[0]) == end_cpu, "the two chains must end on the same row"
print("single dependent rank chain, n=2^%d sigma=%d, %d steps, same end row on both:" % (log2n, sigma, steps))
print("  GPU (one lane group, k_lf_walk):        %.3f us/step  (call %.1f ms)" % (t_gpu * 1e6 / steps, t_gpu * 1e3))
print("  CPU (oracle, inverted lists, one core): %.3f us/step" % (t_cpu * 1e6 / steps))
print("  -> the chain is %.1fx %s on the GPU; calcGaps stays on the host"
      % ((t_gpu * 1e6 / steps) / (t_cpu * 1e6 / steps), "slower"))

# ---- the same kind of chain on the host over the product's own dictionary (fmx_calc_gaps_chain: BWT' + symbol counts
# every 256 positions in front of those bytes, one count + a scan per step) against the reference's structure (inverted lists,
# binary-searched) on one core: curRank = cf(c) + occ(c, curRank - 1) over a random "older text"
rng = np.random.default_rng(3)
text = rng.integers(1, sigma + 1, steps).astype(np.uint8)
hip.occ_host(1, 0)                      # builds the host-side dictionary (not timed)
t0 = time.perf_counter()
ranks, done = hip.calc_gaps_chain(text, rank0=row)
t_host = time.perf_counter() - t0
assert done == steps
cur = row
t0 = time.perf_counter()
want = orc.occ_chain(text, row) if hasattr(orc, "occ_chain") else None
t_ref = time.perf_counter() - t0
if want is None:                        # the oracle has no C loop for this chain: time its occ on a sample, in Python
    t0 = time.perf_counter()
    cur = row
    for j in range(20000):
        cur = orc.cf(int(text[j])) + (0 if cur == 0 else orc.occ(int(text[j]), cur - 1))
        assert cur == int(ranks[j])
    t_ref = (time.perf_counter() - t0) * steps / 20000
    note = "oracle occ through ctypes, 20k steps scaled (includes ~1 us of Python per step)"
else:
    assert np.array_equal(want, ranks)
    note = "oracle C loop"
print("calcGaps rank chain on the host, same index, %d steps, same ranks:" % steps)
print("  product dictionary (fmx_calc_gaps_chain, one core): %.3f us/step" % (t_host * 1e6 / steps))
print("  reference structure (%s): %.3f us/step" % (note, t_ref * 1e6 / steps))
