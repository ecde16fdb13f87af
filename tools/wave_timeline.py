#!/usr/bin/env python3
"""Occupancy of the wave slots over the launches of one C4 match (diagnostic build -DFMX_WAVELOG):
    tools/build_variant.sh wl -DFMX_WAVELOG && FMX_LIB=findex_amd/lib/variants/libfmx_wl.so python tools/wave_timeline.py
Prints, per launch: waves that had a share, their rounds (mean / max), start-up and run times, and how many waves were
resident at tenths of the launch's span."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, findex_amd
from findex_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
log2n, k, seed, max_len = bench.REGEX[wl][:4]
n = 1 << log2n
dev = torch.device("cuda", 0)
bwt, eof = bench.make_bwt(torch, n, bench.C4_ALPHABET, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
del bwt
res, trees = bench.make_regexes(k, seed * 1000)
from findex_amd.regex import RegexBatch
rb = RegexBatch(hip, trees)
for _ in range(4):
    rb.match_raw(max_steps=max_len)
L = _lib.load()
L.fmx_debug_wavelog.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
L.fmx_debug_wavelog(None, 0, 1)
hip.stats_reset()
rb.match_raw(max_steps=max_len)
st = hip.stats()
print({k: v for k, v in st.items() if k.startswith("frontier") or k in ("launches", "backward_steps", "ktab_lookups")})
log = np.zeros((16, 1 << 15, 8), dtype=np.uint64)
assert L.fmx_debug_wavelog(log.ctypes.data_as(ctypes.c_void_p), log.nbytes, 0) == 0
print("kernel ms of the logged call: %.3f" % hip.last_kernel_ms())
T = 0.01   # us per tick (100 MHz)
SLOTS = int(os.environ.get("SLOTS", "4096"))
g0 = None
for p in range(16):
    e = log[p]
    m = e[:, 2] != 0
    if not m.any():
        continue
    t0, t1, t2 = e[m, 0].astype(np.int64), e[m, 1].astype(np.int64), e[m, 2].astype(np.int64)
    rounds = (e[m, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    steps = (e[m, 3] >> np.uint64(32)).astype(np.int64)
    lo, hi = t0.min(), t2.max()
    if g0 is None:
        g0 = lo
    span = (hi - lo) * T
    busy = ((t2 - t0) * T).sum()
    print("pass %2d: start %7.1f us span %6.1f us | %5d waves, rounds mean %.1f max %d, steps %d | start-up %.1f us, in rounds %.1f us mean (%.2f us/round), resident wave-time %.0f us = %.0f%% of %d slots"
          % (p, (lo - g0) * T, span, m.sum(), rounds.mean(), rounds.max(), steps.sum(), ((t1 - t0) * T).mean(), ((t2 - t1) * T).mean(),
             ((t2 - t1) * T).sum() / max(1, rounds.sum()), busy, 100 * busy / (SLOTS * span), SLOTS))
    occ = []
    for q in range(10):
        t = lo + (hi - lo) * (q + 0.5) / 10
        occ.append(int(((t0 <= t) & (t2 > t)).sum()))
    print("         resident waves at tenths of the span:", occ)
    late = np.argsort(t2)[-6:]
    print("         last waves to end (start us, first round us, end us, rounds, steps):",
          [(round((t0[i] - lo) * T, 1), round((t1[i] - lo) * T, 1), round((t2[i] - lo) * T, 1), int(rounds[i]), int(steps[i])) for i in late])
    # who are the stragglers: waves by the time they end, in five groups
    wr = (e[m, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    rd = (e[m, 4] >> np.uint64(32)).astype(np.int64)
    ap = (e[m, 5] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    gr = (e[m, 5] >> np.uint64(32)).astype(np.int64)
    order = np.argsort(t2)
    for name, sel in (("first half", order[: len(order) // 2]), ("50-80 %", order[len(order) // 2: len(order) * 8 // 10]),
                      ("80-95 %", order[len(order) * 8 // 10: len(order) * 95 // 100]), ("95-99 %", order[len(order) * 95 // 100: len(order) * 99 // 100]),
                      ("last 1 %", order[len(order) * 99 // 100:])):
        if len(sel):
            print("         waves ending %-10s: end %6.1f us mean | rounds %5.1f  steps %6.0f  us/round %5.2f | queue appends %6.0f reads %6.0f  reservations %5.1f  looks %4.1f"
                  % (name, ((t2[sel] - lo) * T).mean(), rounds[sel].mean(), steps[sel].mean(), (((t2 - t1)[sel] * T).sum() / max(1, rounds[sel].sum())),
                     wr[sel].mean(), rd[sel].mean(), ap[sel].mean(), gr[sel].mean()))
    hist = np.bincount(np.minimum(rounds, 20), minlength=21)
    print("         waves by rounds (0..20+):", hist.tolist())

# ---- with a -DFMX_PHASELOG=<round> build: where the rounds from that round on spend their time
if hasattr(L, "fmx_debug_phaselog"):
    ph = np.zeros((1 << 15, 8), dtype=np.uint64)
    L.fmx_debug_phaselog.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    if L.fmx_debug_phaselog(ph.ctypes.data_as(ctypes.c_void_p), ph.nbytes) == 0:
        m = ph[:, 4] > 0
        if m.any():
            tot = ph[m, :4].sum(axis=0).astype(np.float64)
            n = float(ph[m, 4].sum())
            print("late rounds (%d waves, %d rounds): cycles per round  take %.0f | stage %.0f | wait + ranks %.0f | bookkeeping %.0f  (total %.0f)"
                  % (m.sum(), n, tot[0] / n, tot[1] / n, tot[2] / n, tot[3] / n, tot.sum() / n))

# ---- per-round trace of the waves that ended last (launch 0): held elements / pool / deepest length / narrow / express
if hasattr(L, "fmx_debug_wavetrace"):
    tr = np.zeros((1 << 15, 128), dtype=np.uint32)
    L.fmx_debug_wavetrace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    if L.fmx_debug_wavetrace(tr.ctypes.data_as(ctypes.c_void_p), tr.nbytes) == 0:
        e = log[0]
        order = np.argsort(e[:, 2])[::-1][:4]
        for wv in order:
            row = tr[wv]
            txt = " ".join("%d/%d/L%d%s%s" % (v & 0xFF, (v >> 8) & 0xFFF, (v >> 20) & 0xFF, "n" if (v >> 28) & 1 else "", ("x%d" % (v >> 29)) if v >> 29 else "")
                           for v in row.tolist() if v)
            print("wave %d (held/pool/deepest by round): %s" % (wv, txt))
