#!/usr/bin/env python3
"""bench.py -- the reference's headline measurement on MI355X (BASELINE.json):
M rank-queries/s and patterns/s of batched FM-index backward search over a 4 GiB BWT resident in
HBM, with the achieved fraction of the HBM roofline and the CPU path timed beside it.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (fmx_search_batch_dev: one k_search launch) over one batch
of synthetic patterns that already sit in HBM.  Workload at every N (weak scaling): per GPU,
config C3 -- 1M 32-char literal patterns over a 4 GiB sigma=128 synthetic BWT whose rank
dictionary is replicated on each GPU; with N > 1 each rank searches its own 1M-pattern shard
and the step ends with the RCCL all-gather of the (sp, ep) intervals.

torch is plumbing here (device buffers, the stream, torch.distributed); the product path is
libfmx.so through its C ABI.  Only the cpu_baseline leg touches oracle/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
# Algorithmic bytes per rank query, the figure roofline.achieved is priced with: 128 B at
# sigma = 4, 132 B (one 128-B block + a 4-B checkpoint) at sigma = 128.  What this build's layouts really
# fetch per rank query (64 B one-hot block / 132 B bytes layout) is reported next to it.
def survey_bytes_per_rank(sigma):
    return 128 if sigma <= 4 else 132


# Ceiling of the memory system for this access pattern, measured with tools/ubench/chain.hip on MI355X:
# dependent random 64-byte requests at 16 chains per wave and full occupancy.
REQUEST_CEILING_G_PER_S = 53.0

WORKLOADS = {
    # name: (log2 n, sigma, patterns per GPU, pattern length, seed#)
    "c3": (32, 128, 1_000_000, 32, 3),
    "c2": (28, 4, 1_000_000, 16, 2),
    "c5": (34, 128, 1_000_000, 24, 5),      # 16 GiB BWT: opens in the bytes+checkpoints layout (80 GiB)
    "tiny": (22, 128, 100_000, 32, 9),
}


def pmc_traffic(workload, kernel="k_search4"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC profile of this workload
    (profiles/r*_<workload>_counters.csv, written by tools/summarize_prof.py from separate rocprofv3 --pmc
    passes): FETCH_SIZE KiB x the bytes one KiB stands for in this access pattern (calibrated in the same
    profile on a k_occ launch of known byte count, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024.
    None when no profile of this workload is in the tree."""
    import csv
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_counters.csv" % workload))):
        vals, per_kib = {}, None
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel"]:
                vals[r["Counter"]] = float(r["Mean"])
            if r["Counter"] == "FETCH_BYTES_PER_KIB":
                per_kib = float(r["Mean"])
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and per_kib:
            best = (int(vals["FETCH_SIZE"] * per_kib + vals["WRITE_SIZE"] * 1024), os.path.basename(f))
    return best


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def make_bwt(torch, n, sigma, seed, device):
    """i.i.d. uniform symbols 1..sigma on the device (any byte string is a valid
    LF permutation); eof = n/3."""
    g = torch.Generator(device=device)
    g.manual_seed(0xF1DE0000 + seed)
    bwt = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        bwt[a:b] = torch.randint(1, sigma + 1, (b - a,), generator=g, device=device, dtype=torch.uint8)
    return bwt, n // 3


def make_patterns(torch, hip, n, sigma, k, m, seed, device, stream):
    """90 % hit patterns by LF walk (every backward step keeps a non-empty interval), 10 % with
    one byte replaced (early-exit path), SURVEY 8d.  Generated on the device with the library's
    own LF-walk kernel; hit-ness is then verified from the search results."""
    g = torch.Generator(device=device)
    g.manual_seed(0x5EED0000 + seed)
    rows = torch.randint(0, n, (k,), generator=g, device=device, dtype=torch.int64)
    walk = torch.empty((k, m), dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    hip.lf_walk_batch_dev(rows.data_ptr(), k, m, walk.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    pats = torch.flip(walk, dims=[1]).contiguous()
    mut = torch.rand(k, generator=g, device=device) < 0.10
    pos = torch.randint(0, m, (k,), generator=g, device=device)
    sym = torch.randint(1, sigma + 1, (k,), generator=g, device=device, dtype=torch.uint8)
    idx = torch.nonzero(mut).squeeze(1)
    pats[idx, pos[idx]] = sym[idx]
    off = torch.arange(0, (k + 1) * m, m, dtype=torch.int64, device=device)
    return pats.reshape(-1), off


def effective_cores():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota
    (the GPU boxes expose 256 CPUs but grant a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, -(-q // p)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(torch, hip_full, sigma, m, device, stream, rank):
    """The CPU path beside the GPU number: the oracle's restatement of the reference algorithm
    (inverted position lists + binary-search occ, bwtmerger.scala:354-375) on this host's cores,
    on a bounded sample of the same workload: the same generator and alphabet at n = 2^27 (so the
    32 GiB position list of the full index need not be built) and 200k patterns of the same
    length and hit mix.  The GPU answers for the sample are checked against it bit for bit."""
    import oracle
    import findex_amd
    n_s, k_s = 1 << 27, 200_000
    bwt_s, eof_s = make_bwt(torch, n_s, sigma, 77, device)
    hip_s = findex_amd.HipFMSearcher.from_device(bwt_s.data_ptr(), n_s, eof_s, None, device=device.index, stream=stream)
    pats, off = make_patterns(torch, hip_s, n_s, sigma, k_s, m, 78, device, stream)
    sp = torch.empty(k_s, dtype=torch.int64, device=device)
    ep = torch.empty(k_s, dtype=torch.int64, device=device)
    hip_s.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k_s, stream)
    torch.cuda.synchronize()
    h_bwt = bwt_s.cpu().numpy()
    counts = np.bincount(h_bwt, minlength=256).astype(np.int64)
    counts[h_bwt[eof_s]] -= 1
    t0 = time.time()
    orc = oracle.NaiveFMSearcher.from_mem(h_bwt, eof_s, counts)
    t_build = time.time() - t0
    h_pats = pats.cpu().numpy()
    h_off = off.cpu().numpy().astype(np.uint64)
    cores = effective_cores()
    t0 = time.time()
    wsp, wep, steps = orc.search_batch(h_pats, h_off, threads=cores)
    dt = time.time() - t0
    ok = bool(np.array_equal(wsp, sp.cpu().numpy().astype(np.uint64)) and
              np.array_equal(wep, ep.cpu().numpy().astype(np.uint64)))
    if not ok:
        raise SystemExit("bench: GPU results differ from the CPU oracle on the baseline sample")
    ranks = 2 * int(steps.sum())
    log(rank, "cpu_baseline: %d cores, %.2fs for %d patterns (%d rank queries), list build %.1fs, parity ok"
        % (cores, dt, k_s, ranks, t_build))
    hip_s.close()
    return {"value": ranks / dt / 1e6, "unit": "M rank-queries/s", "cores": cores, "kind": "port",
            "patterns_per_s": k_s / dt,
            "sample": "same generator/alphabet at n=2^27 (not 2^32), 200k x %d-char patterns, 90%% LF-walk hits; "
                      "inverted lists + binary-search occ in C with OpenMP; GPU results bit-equal" % m}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # this image exports NCCL_DEBUG=VERSION, which makes RCCL print a banner on stdout; stdout carries
    # the one JSON line, so drop that setting (any other value the user chose is kept)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]
    import torch
    import torch.distributed as dist
    import findex_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(rank, "note: WORLD_SIZE=%d but --gpus=%d; using WORLD_SIZE" % (world, args.gpus))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # under torch.distributed.run (RANK/MASTER_* set) the RCCL path runs even for one rank
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=device)
    stream = torch.cuda.current_stream().cuda_stream

    log2n, sigma, k, m, seed = WORKLOADS[args.workload]
    n = 1 << log2n
    t0 = time.time()
    bwt, eof = make_bwt(torch, n, sigma, seed, device)          # same seed on every rank: replicas
    torch.cuda.synchronize()
    hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=local, stream=stream)
    del bwt
    torch.cuda.empty_cache()
    st = hip.stats()
    log(rank, "index: n=2^%d sigma=%d, %.1f GiB in HBM (%d symbols x %d blocks x %d B), built in %.1f ms (+%.1fs setup)"
        % (log2n, sigma, st["index_bytes"] / 2**30, st["n_symbols"], st["n_blocks"], st["block_bytes"],
           st["build_ms"], time.time() - t0))
    pats, off = make_patterns(torch, hip, n, sigma, k, m, seed * 1000 + rank, device, stream)
    # the intervals land in the slots of a pipelined gather: with N > 1 the all-gather of step i (RCCL over
    # xGMI, 16 B per pattern) runs on the collective's stream while step i+1 is being searched
    from findex_amd.distributed import IntervalGather
    gather = IntervalGather(k, device)
    step_no = [0]

    def step():
        sp, ep = gather.slot(step_no[0])
        hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
        if use_dist:        # the path's one exchange: gather the hit intervals
            gather.launch(step_no[0])
        step_no[0] += 1
        return sp, ep

    # rank queries one step executes (device counter; identical every step)
    hip.stats_reset()
    sp, ep = step()
    torch.cuda.synchronize()
    s1 = hip.stats()
    ranks_per_step = int(s1["rank_queries"])
    requests_per_step = int(s1["search_requests"])
    hits = int((sp < ep).sum().item())
    for _ in range(max(0, args.warmup - 1)):
        step()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        sp_i, ep_i = gather.slot(step_no[0])
        a.record()          # torch's current stream == the stream the kernel is launched on
        hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp_i.data_ptr(), ep_i.data_ptr(), k, stream)
        b.record()
        if use_dist:
            gather.launch(step_no[0])
        step_no[0] += 1
    gather.finish()         # every step's gather completes inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    tot = torch.tensor([dt, float(ranks_per_step), float(hits), kernel_ms], dtype=torch.float64, device=device)
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0].item())
        ranks_all = float(sm[1].item())
        hits_all = float(sm[2].item())
        kernel_ms_max = float(mx[3].item())
    else:
        ranks_all, hits_all, kernel_ms_max = float(ranks_per_step), float(hits), kernel_ms

    if rank == 0:
        bytes_per_rank = survey_bytes_per_rank(sigma)
        kernel_name = "k_search4"
        achieved = ranks_per_step * bytes_per_rank / (kernel_ms * 1e-3) / 1e9
        req_rate = requests_per_step / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "rank_queries_per_sec",
            "value": ranks_all * args.steps / dt / 1e6,
            "unit": "M rank-queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "patterns_per_sec": world * k * args.steps / dt,
            "config": {
                "workload": "%s: %d x %d-char literal patterns per GPU, 2^%d-byte sigma=%d synthetic BWT resident "
                            "in HBM (rank dictionary replicated per GPU)" % (args.workload.upper(), k, m, log2n, sigma),
                "n": n, "sigma": sigma, "patterns_per_gpu": k, "pattern_len": m,
                "hit_patterns_fraction": hits_all / (world * k),
                "rank_queries_per_step": ranks_all,
                "parallelism": "patterns sharded over %d GPU(s), index replicated%s"
                               % (world, ", all_gather of (sp,ep) per step overlapped with the next step's search" if use_dist else ""),
                "index_gib": st["index_bytes"] / 2**30, "index_build_ms": st["build_ms"],
                "index_layout": "one-hot bit-vectors, 64-B blocks" if st["layout"] == 0 else "BWT bytes + checkpoints",
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": (pmc_traffic(args.workload) or (None, None))[0],
                # the same in bandwidth terms: measured HBM bytes per launch / this run's kernel time, and its
                # share of the 8 TB/s peak (what the hardware really moved; `frac` above prices SURVEY's 132 B)
                "traffic_GBps": ((pmc_traffic(args.workload) or (0, None))[0] or 0) / (kernel_ms * 1e-3) / 1e9 or None,
                "traffic_frac": ((pmc_traffic(args.workload) or (0, None))[0] or 0) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS or None,
                "traffic_source": (pmc_traffic(args.workload) or (None, "no PMC profile of this workload committed"))[1],
                "bytes_per_rank_query": bytes_per_rank, "rank_queries_per_launch": ranks_per_step,
                "kernel_ms": kernel_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
                # what the layout really moves and the limit that binds it: distinct memory requests per second
                "layout_bytes_per_request": 64 if st["layout"] == 0 else 66,
                "requests_per_launch": requests_per_step, "requests_G_per_s": req_rate,
                "request_ceiling_G_per_s": REQUEST_CEILING_G_PER_S, "request_frac": req_rate / REQUEST_CEILING_G_PER_S,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, hip, sigma, m, device, stream, rank)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
