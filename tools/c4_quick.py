#!/usr/bin/env python3
"""C4 in a few seconds for A/B runs (FMX_LIB=<variant>, FMX_FRONTIER_* knobs): the bench's index and regex batch,
device-resident matches timed, result list compared with the first call's.   python tools/c4_quick.py [calls] [log2n] [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, findex_amd, bench
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
dev = torch.device("cuda", 0)
bwt, eof = bench.make_bwt(torch, 1 << log2n, bench.C4_ALPHABET, 4, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), 1 << log2n, eof, None, device=0, stream=torch.cuda.current_stream().cuda_stream)
res, trees = bench.make_regexes(k, 4000)
batch = findex_amd.ReTree.prepare_batch(hip, trees)
cap = 1 << 22
d_out = torch.empty(3 * cap, dtype=torch.int64, device=dev)
d_per = torch.empty(k, dtype=torch.int32, device=dev)
hip.stats_reset()
n0 = batch.match_dev(d_out.data_ptr(), cap, d_per.data_ptr(), max_steps=64)
st = hip.stats()
first = d_out[: 3 * n0].clone()
per0 = d_per.clone()
for _ in range(3):
    batch.match_dev(d_out.data_ptr(), cap, d_per.data_ptr(), max_steps=64)
ks, ts = [], []
for _ in range(calls):
    t0 = time.perf_counter()
    n1 = batch.match_dev(d_out.data_ptr(), cap, d_per.data_ptr(), max_steps=64)
    ts.append(time.perf_counter() - t0)
    ks.append(hip.last_kernel_ms())
    assert n1 == n0 and torch.equal(d_out[: 3 * n0], first) and torch.equal(d_per, per0), "results differ between calls"
if os.environ.get("C4_GROUPS"):      # sizes of the per-regex result groups (what the grouping kernels order)
    c = per0.cpu().numpy().astype(np.int64)
    for lo, hi in ((0, 0), (1, 1), (2, 4), (5, 12), (13, 32), (33, 64), (65, 128), (129, 256), (257, 1024), (1025, 1 << 40)):
        sel = c[(c >= lo) & (c <= hi)]
        print("groups of %4d..%-6d: %6d groups, %7d results" % (lo, min(hi, int(c.max())), sel.size, int(sel.sum())))
    blk = np.add.reduceat(c, np.arange(0, k, 256))
    print("results per 256 regexes: mean %.0f max %d; blocks over 1024: %d" % (blk.mean(), blk.max(), int((blk > 1024).sum())))
ks.sort(); ts.sort()
steps = st["backward_steps"]
print("lib=%s chain=%s | results %d steps %d | device ms median %.4f min %.4f | call ms median %.4f min %.4f | %.1f G rq/s device (median)"
      % (os.path.basename(os.environ.get("FMX_LIB", "default")), os.environ.get("FMX_FRONTIER_CHAIN", "-"), n0, steps,
         ks[len(ks) // 2], ks[0], ts[len(ts) // 2] * 1e3, ts[0] * 1e3, 2 * steps / ks[len(ks) // 2] / 1e6), flush=True)
