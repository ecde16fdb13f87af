#!/usr/bin/env python3
"""When do the waves of k_search4 begin and end?  Diagnostic build -DFMX_SEARCHLOG (works in the round-3 worktree too,
patched with the same log):
    tools/build_variant.sh slog -DFMX_SEARCHLOG && FMX_LIB=findex_amd/lib/variants/libfmx_slog.so python tools/search_wave_timeline.py c5
Prints the launch's span on the device's 100 MHz clock, when the waves began / entered their batch loop / left it /
ended (percentiles), how long a batch took, and how many waves were alive at tenths of the span."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, findex_amd
from findex_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
log2n, sigma, k, m, seed = bench.LITERAL[wl]
n = 1 << log2n
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
if hasattr(hip, "prepare"):
    hip.prepare(ktab=True, jump=True)
del bwt
pats, off = bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000, dev, stream)
sp = torch.empty(k, dtype=torch.int64, device=dev)
ep = torch.empty(k, dtype=torch.int64, device=dev)
for _ in range(5):
    hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
torch.cuda.synchronize()
L = _lib.load()
L.fmx_debug_searchlog.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
log = np.zeros((1 << 15, 4), dtype=np.uint64)
assert L.fmx_debug_searchlog(log.ctypes.data_as(ctypes.c_void_p), log.nbytes) == 0
print("device ms of the logged call (all its kernels): %.4f" % hip.last_kernel_ms())
t3 = log[:, 3] & np.uint64((1 << 48) - 1)
nb = (log[:, 3] >> np.uint64(48)).astype(np.int64)
live = log[:, 0] != 0
live &= log[:, 0] + np.uint64(100000) > log[live, 0].max()      # the log is never cleared: entries of earlier, larger launches (within 1 ms of the newest: this launch)
t0 = log[live, 0].astype(np.int64)
t1 = t0 + (log[live, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)      # (entry 1: the batch loop's begin and the last walk's, both from t0)
tw = t0 + (log[live, 1] >> np.uint64(32)).astype(np.int64)
t2, t3, nb = log[live, 2].astype(np.int64), t3[live].astype(np.int64), nb[live]
T = 0.01      # us per tick
z = t0.min()
span = (t3.max() - z) * T
print("%d waves; span of the launch %.1f us" % (live.sum(), span))
def pct(name, v):
    q = np.percentile(v, [0, 10, 50, 90, 99, 100])
    print("%-34s min %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f  p99 %7.1f  max %7.1f us" % ((name,) + tuple(q)))
pct("wave begins at", (t0 - z) * T)
pct("enters its batch loop at", (t1 - z) * T)
pct("begins its last walk at", (tw - z) * T)
pct("last walk takes", (t2 - tw) * T)
pct("has walked at", (t2 - z) * T)
pct("ends at", (t3 - z) * T)
pct("alive for", (t3 - t0) * T)
pct("set-up (tables into LDS)", (t1 - t0) * T)
pct("per batch", (tw - t1) * T / np.maximum(nb, 1))
for b in sorted(set(nb.tolist())):
    sel = nb == b
    print("waves with %d batches: %5d; batch loop ends p50 %6.1f p99 %6.1f, wave ends p50 %6.1f p99 %6.1f max %6.1f us" % (b, sel.sum(), np.percentile((tw[sel] - z) * T, 50), np.percentile((tw[sel] - z) * T, 99), np.percentile((t3[sel] - z) * T, 50), np.percentile((t3[sel] - z) * T, 99), ((t3[sel] - z) * T).max()))
print("batches per wave: min %d max %d" % (nb.min(), nb.max()))
for f in range(0, 11):
    at = z + int(f / 10 * (t3.max() - z))
    print("  at %3d %% of the span: %5d waves alive, %5d not begun, %5d ended" % (10 * f, int(((t0 <= at) & (t3 > at)).sum()), int((t0 > at).sum()), int((t3 <= at).sum())))
# by XCD (workgroup id mod 8) and by the order of the waves
wid = np.nonzero(live)[0]
for x in range(8):
    sel = ((wid // 4) % 8) == x
    print("  XCD %d: waves begin p50 %6.1f, end p50 %6.1f max %6.1f us" % (x, np.percentile((t0[sel] - z) * T, 50), np.percentile((t3[sel] - z) * T, 50), ((t3[sel] - z) * T).max()))
# What handing the partial last round to the FAST waves would give: every wave keeps its own pace (us per batch);
# statically the first (batches mod waves) waves take the extra batch, dynamically the ones with the lowest pace do.
pace = (t2 - t1) * T / np.maximum(nb, 1)
base_n = int(nb.min())
n_extra = int((nb > base_n).sum())
start = (t1 - z) * T
static_end = start + nb * pace
fast = np.argsort(pace)[:n_extra]
dyn_n = np.full(nb.shape, base_n)
dyn_n[fast] += 1
dyn_end = start + dyn_n * pace
print("last round to the fastest waves (same paces): static ends p50 %.1f p99 %.1f max %.1f us; by pace p50 %.1f p99 %.1f max %.1f us"
      % (np.percentile(static_end, 50), np.percentile(static_end, 99), static_end.max(),
         np.percentile(dyn_end, 50), np.percentile(dyn_end, 99), dyn_end.max()))
# and with perfect balance (work stealing at batch granularity): total work / waves
print("perfect balance: %.1f us" % (start.mean() + (nb * pace).sum() / nb.size))
