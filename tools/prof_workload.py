#!/usr/bin/env python3
"""Workload for rocprofv3 passes: a bench step of the chosen workload (k_search4, or the regex frontier kernels
for c4) plus a calibration launch of
k_occ whose HBM byte count is known -- 2^24 uniformly random (c, i) rank queries over the
77 GiB rank dictionary touch 2^24 distinct 64-byte blocks (collisions < 0.2 %), i.e. 1 GiB, plus
the 9 bytes of (c, i) streamed in per query.  FETCH_SIZE read for k_occ calibrates the counter
for this access pattern (MI355X_MICROARCH.md, HBM: FETCH_SIZE halves wide reads on gfx950; other
widths are uncalibrated -- calibrate on a known byte count in your own access pattern).

    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_workload.py [--workload c3] [--steps 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import findex_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--calib-queries", type=int, default=1 << 24)
ap.add_argument("--ring", type=int, default=8)
a = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
regex = a.workload in bench.REGEX
print("SRC_SHA16=%s" % bench.source_hash())
ref_mode = False
if regex:
    log2n, k, seed, max_len, ref_mode = bench.REGEX[a.workload]
    n, sigma = 1 << log2n, None
    bwt, eof = bench.make_bwt(torch, n, bench.C4_ALPHABET, seed, dev)
else:
    log2n, sigma, k, m, seed = bench.LITERAL[a.workload]
    n = 1 << log2n
    bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
hip.prepare(ktab=True, jump=not regex, frontier=regex)
del bwt
torch.cuda.empty_cache()
if regex:
    res, trees = bench.make_regexes(k, seed * 1000)
    batch = findex_amd.ReTree.prepare_batch(hip, trees)
    cap = 1 << 22
    d_out = torch.empty(3 * cap, dtype=torch.int64, device=dev)
    d_per = torch.empty(k, dtype=torch.int32, device=dev)

    def regex_step():       # the bench's step
        if ref_mode:
            batch.match_raw(mode="reference", maxBranching=bench.REF_LIMITS[0], maxIterations=bench.REF_LIMITS[1], cap=cap, copy=False)
        else:
            batch.match_dev(d_out.data_ptr(), cap, d_per.data_ptr(), max_steps=max_len)
    regex_step()          # first call: allocations (the level chain is captured from the 2nd)
    regex_step()
else:
    ring = [bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000 + 7919 * j, dev, stream) for j in range(a.ring)]      # rotated through the steps (round 5)
    pats, off = ring[0]
    sp = torch.empty(k, dtype=torch.int64, device=dev)
    ep = torch.empty(k, dtype=torch.int64, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(42)
kc = a.calib_queries
if regex:
    alpha = torch.tensor([ord(c) for c in bench.C4_ALPHABET], dtype=torch.uint8, device=dev)
    qc = alpha[torch.randint(0, alpha.numel(), (kc,), generator=g, device=dev)]
else:
    qc = torch.randint(1, sigma + 1, (kc,), generator=g, device=dev, dtype=torch.uint8)
qi = torch.randint(0, n, (kc,), generator=g, device=dev, dtype=torch.int64)
qo = torch.empty(kc, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.stats_reset()
if not regex:
    for bp, bo in ring:           # every batch once before the counted steps: none of them is the device's first sight of it
        hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    torch.cuda.synchronize()
    hip.stats_reset()
for i in range(a.steps):
    if regex:
        regex_step()
    else:
        bp, bo = ring[i % a.ring]
        hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
torch.cuda.synchronize()
st = hip.stats()
if regex:
    print("frontier: %d calls; per call %d elements stepped, %d rank-line requests, %d queue reads, %d queue writes, %d results"
          % (a.steps, st["frontier_elements"] // a.steps, st["frontier_requests"] // a.steps,
             st["frontier_queue_reads"] // a.steps, st["frontier_queue_writes"] // a.steps, st["frontier_results"] // a.steps))
else:
    print("k_search4: %d launches, %d rank queries and %d block requests per launch"
          % (a.steps, st["rank_queries"] // a.steps, st["search_requests"] // a.steps))
for _ in range(3):
    hip.occ_batch_dev(qc.data_ptr(), qi.data_ptr(), qo.data_ptr(), kc, stream)
torch.cuda.synchronize()
line = int(st["block_bytes"])
print("CALIB kernel=k_occ queries=%d line_bytes=%d read_bytes=%d" % (kc, line, kc * line + kc * 9))
