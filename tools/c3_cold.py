#!/usr/bin/env python3
"""What a literal-search launch costs on a batch the device has NOT just seen.

One open of the workload's index (default C3: n = 2^32, sigma = 128, all tables), R distinct 1M-pattern batches, and the
search kernel launched in phases, every launch timed by its own pair of HIP events:

  first    batch 0, once: the first touch after the table build and the pattern generation
  replay   batch 0, six more times: what rounds 1-4 timed (the batch's ~335 MB of sectors straddle the 256 MiB Infinity Cache)
  ring     batches 1, 2, .., R-1, 0, 1, ..: 3 R launches, R x 335 MB between two uses of any line
  replay2  batch 0 again, six times: the first of them follows the ring (cold), the others follow themselves
  fl32M .. flushed  a device fill of 32 MiB / 128 MiB / 512 MiB / 1 GiB, then batch 0 -- three times each: what does a launch
           lose when other traffic has gone through the L2s (32 MiB together) or the Infinity Cache (256 MiB)?
  rd512M, rd1G  the same buffer READ (a sum) instead of written, then batch 0: clean lines of something else in the caches
  idle20ms 20 ms without any device work, then batch 0
  stream   the ring and the replay as 3 R back-to-back launches each, one event pair around all of them (no gaps, no syncs)

Prints one line per launch and a PHASES line (JSON) that tools/c3_cold_pmc.py uses to label the dispatches of a
`rocprofv3 --pmc ... --kernel-trace` run of this very script.

    python tools/c3_cold.py [--workload c3] [--ring 10] [--tables on|off]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import findex_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--ring", type=int, default=10)
ap.add_argument("--tables", default="on", choices=["on", "off"])
a = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
log2n, sigma, k, m, seed = bench.LITERAL[a.workload]
n = 1 << log2n
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
del bwt
torch.cuda.empty_cache()
if a.tables == "on":
    hip.prepare(ktab=True, jump=True)
else:
    hip.config_set("ktab", "off")
    hip.config_set("jump", "off")
    hip.prepare(ktab=False, search=True)
R = a.ring
batches = [bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000 + 7919 * j, dev, stream) for j in range(R)]
sp = torch.empty(k, dtype=torch.int64, device=dev)
ep = torch.empty(k, dtype=torch.int64, device=dev)
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
st = hip.stats()
print("index %.1f GiB, tables %s, jump_bytes %.0f GiB, residency 0x%x" % (st["index_bytes"] / 2**30, a.tables, st["jump_bytes"] / 2**30, st["search_residency"]))


def launch(b):
    bp, bo = batches[b]
    hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)


def one(b):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch(b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


phases = []          # (phase, batch) per k_search4 launch, in order
rows = []


def run(phase, b, pre=None):
    if pre:
        pre()
    ms = one(b)
    phases.append((phase, b))
    rows.append((phase, b, ms))
    print("%-8s batch %2d  %.4f ms" % (phase, b, ms), flush=True)


hip.stats_reset()
run("first", 0)
s1 = hip.stats()
reqs = int(s1["search_requests"] + s1["ktab_lookups"] + s1["jump_lookups"] + s1["row_lookups"])
print("requests per launch (batch 0): %d = %d rank lines + %d k-mer + %d jump + %d row words" %
      (reqs, s1["search_requests"], s1["ktab_lookups"], s1["jump_lookups"], s1["row_lookups"]))
for _ in range(6):
    run("replay", 0)
for i in range(3 * R):
    run("ring", (1 + i) % R)
for _ in range(6):
    run("replay2", 0)
# what does a launch lose when something else has run?  A device fill of 32 MiB (the eight L2s together), 128 MiB, 512 MiB and
# 1 GiB (past the 256 MiB Infinity Cache) before batch 0, three times each; and 20 ms of idling (clocks) without any fill
for label, mib in (("fl32M", 32), ("fl128M", 128), ("fl512M", 512), ("flushed", 1024)):
    for _ in range(3):
        run(label, 0, pre=lambda mib=mib: (flush[: mib << 20].fill_(1), torch.cuda.synchronize()))
# ... and the same bytes READ instead of written (a sum over the buffer): the caches then hold clean lines of something else
flush64 = flush.view(torch.int64)
for label, mib in (("rd512M", 512), ("rd1G", 1024)):
    for _ in range(3):
        run(label, 0, pre=lambda mib=mib: (flush64[: (mib << 20) // 8].sum(), torch.cuda.synchronize()))
import time  # noqa: E402
for _ in range(3):
    run("idle20ms", 0, pre=lambda: time.sleep(0.02))


def streamed(label, seq):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for b in seq:
        launch(b)
        phases.append((label, b))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / len(seq)
    print("%-8s %d back-to-back launches: %.4f ms each" % (label, len(seq), ms), flush=True)
    return ms


ms_ring = streamed("s_ring", [(1 + i) % R for i in range(3 * R)])
ms_rep = streamed("s_replay", [0] * (3 * R))


def mean(ph, skip=0):
    v = [ms for p, _, ms in rows if p == ph][skip:]
    return sum(v) / len(v)


summary = {"workload": a.workload, "tables": a.tables, "ring": R, "requests_per_launch": reqs,
           "first_ms": mean("first"), "replay_ms": mean("replay", 1), "ring_ms": mean("ring"), "replay2_first_ms": [ms for p, _, ms in rows if p == "replay2"][0],
           "replay2_rest_ms": mean("replay2", 1), "flushed_ms": mean("flushed"), "fl32M_ms": mean("fl32M"), "fl128M_ms": mean("fl128M"), "fl512M_ms": mean("fl512M"), "rd512M_ms": mean("rd512M"), "rd1G_ms": mean("rd1G"), "idle20ms_ms": mean("idle20ms"), "stream_ring_ms": ms_ring, "stream_replay_ms": ms_rep,
           "ring_G_requests_per_s": reqs / ms_ring / 1e6, "replay_G_requests_per_s": reqs / ms_rep / 1e6}
print("SUMMARY " + json.dumps(summary))
print("PHASES " + json.dumps(phases))
