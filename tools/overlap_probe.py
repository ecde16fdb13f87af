#!/usr/bin/env python3
"""Does work on a second stream run BESIDE the search kernel on this platform?  bench.py's multi-rank step wants the
gather of step i to overlap the search of step i + 1; the one-rank RCCL rehearsal showed step = search + gather
exactly.  This probe replaces the gather by other side-stream work of the same size to tell RCCL's behaviour from the
device's: (a) nothing, (b) an elementwise kernel over 8 MB, (c) a device-to-device copy of 8 MB, (d) the one-rank RCCL
gather (run under torch.distributed.run --nproc-per-node 1).

    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 tools/overlap_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import findex_amd  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
use_dist = "RANK" in os.environ
if use_dist:
    dist.init_process_group("nccl", device_id=dev)
stream = torch.cuda.current_stream().cuda_stream
log2n, sigma, k, m, seed = bench.LITERAL[os.environ.get("PROBE_WL", "c3")]
n = 1 << log2n
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev)
torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
hip.prepare(ktab=True, jump=True)
del bwt
pats, off = bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000, dev, stream)
sp = [torch.empty(k, dtype=torch.int64, device=dev) for _ in range(2)]
ep = [torch.empty(k, dtype=torch.int64, device=dev) for _ in range(2)]
a = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
b = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
outs = [torch.empty(1 << 20, dtype=torch.int64, device=dev)]
side = torch.cuda.Stream(device=dev)


def run(kind, steps=30):
    works = [None, None]
    evs = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        j = i & 1
        if works[j] is not None:
            works[j].wait()
            works[j] = None
        torch.cuda.current_stream().wait_event(evs[j])
        hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp[j].data_ptr(), ep[j].data_ptr(), k, stream)
        done = torch.cuda.Event()
        done.record()
        if kind == "kernel":
            with torch.cuda.stream(side):
                side.wait_event(done)
                torch.add(a, 1, out=b)
                evs[j].record(side)
        elif kind == "copy":
            with torch.cuda.stream(side):
                side.wait_event(done)
                b.copy_(a, non_blocking=True)
                evs[j].record(side)
        elif kind == "rccl":
            works[j] = dist.gather(a, outs, dst=0, async_op=True)
    for w in works:
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for kind in ("none", "kernel", "copy") + (("rccl",) if use_dist else ()):
    run(kind, 5)
    print("side work %-7s: %.4f ms per step" % (kind, run(kind)), flush=True)
if use_dist:
    dist.destroy_process_group()
