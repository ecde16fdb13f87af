#!/usr/bin/env python3
"""bench.py -- the reference's headline measurement on MI355X (BASELINE.json): M rank-queries/s and
patterns/s (regexes/s) of batched FM-index backward search over an index resident in HBM, with the
achieved fraction of the HBM roofline and the CPU path timed beside it.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c4|c5|tiny|c4tiny]

`--gpus N` with N > 1 starts the N ranks itself (a child `python -m torch.distributed.run
--nproc-per-node N bench.py ...`, before this process touches any GPU) and relays rank 0's JSON line; when
the driver has already started the ranks (RANK/WORLD_SIZE in the environment) it just takes its place.

A "step" is one pass of the hot path over one batch that already sits in HBM:
  literal workloads (c3 default, c2, c5, tiny): fmx_search_batch_dev = ONE launch of k_search4 over 1M
    patterns per GPU; with N > 1 each rank searches its own shard and the step ends with the RCCL all-gather
    of the (sp, ep) intervals (overlapped with the next step's search);
  regex workloads (c4, c4tiny): fmx_regex_batch_match_dev on a resident batch of compiled regexes (the Glushkov
    SA-interval frontier, ReTree._matchSA with limits that do not bind), results and per-regex counts LEFT IN HBM
    like the literal step's intervals; the rate with the results delivered into page-locked host memory is reported
    beside it (`host_delivered`), and so is the one-shot rate of a batch that arrives as strings (`fresh_batch`:
    fmx_regex_compile_batch + fmx_regex_batch_create + the first match).  With N > 1 every rank matches its own
    batch of the same size (weak scaling) and the per-rank result lists are all-gathered.
  c4ref, c4reftiny: the same batch through FMX_MATCH_REFERENCE with the reference's default limits
    (ReTree.matchSA: maxBranching 1024, maxIterations 1000, re2/retree.scala:570) -- its own pop order, its own
    answer, results delivered to the host in its list order.

torch is plumbing here (device buffers, the stream, torch.distributed); the product path is libfmx.so
through its C ABI.  Only the cpu_baseline leg touches oracle/.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming ceiling)


def survey_bytes_per_rank(sigma):
    """SURVEY.md 8d priced a rank query with the structure it assumed (128 B at sigma = 4, a 128-B block + a 4-B
    checkpoint at sigma = 128).  Kept as a secondary figure (`survey_equiv_GBps`); roofline.achieved prices what
    this build's layout really asks of memory."""
    return 128 if sigma <= 4 else 132


# Ceiling of the memory system for dependent random requests of the search kernel's own MIX (64-byte dictionary blocks,
# 16-byte row-jump and k-mer entries, 8-byte row words, from four tables of 177 GiB), measured with tools/ubench/mix.hip on
# MI355X: 46.5 G requests/s for every kind and every mix (profiles/r04_ubench_mix.txt; chain.hip, whose chains carry
# less arithmetic, reaches 51-54 on one table).  What the memory system counts is cache-line requests that miss the
# CU's L1 / translation cache, whatever they ask for (profiles/r04_c3_bound.md).  Quoted for HBM-resident workloads only.
REQUEST_CEILING_G_PER_S = 46.5
INFINITY_CACHE_BYTES = 256 << 20

LITERAL = {
    # name: (log2 n, sigma, patterns per GPU, pattern length, seed#)
    "c3": (32, 128, 1_000_000, 32, 3),
    "c2": (28, 4, 1_000_000, 16, 2),
    "c5": (34, 128, 1_000_000, 24, 5),      # 16 GiB BWT: opens in the bytes+checkpoints layout
    "tiny": (22, 128, 100_000, 32, 9),
}
REGEX = {
    # name: (log2 n, regexes per GPU, seed#, max match length explored, reference-order mode)
    "c4": (30, 100_000, 4, 64, False),
    "c4tiny": (22, 5_000, 4, 64, False),
    "c4ref": (30, 100_000, 4, 0, True),
    "c4reftiny": (22, 5_000, 4, 0, True),
}
# The same two shapes over the BWT of a TEXT with natural repeats (tools/text_bwt.py: words.txt's words drawn with
# replacement, index over the reversed text like findex's) -- beside the i.i.d. inputs SURVEY 8d prescribes, which are the
# best case for the derived tables and have LF cycles no text has.  Never the headline.
TEXT = {
    # name: (log2 n, patterns or regexes per GPU, pattern length / max match length (0: none), seed#)
    "c3text": (30, 1_000_000, 32, 13),
    "c4text": (30, 100_000, 0, 14),
    "c3texttiny": (20, 50_000, 32, 13),
    "c4texttiny": (20, 3_000, 0, 14),
}
REF_LIMITS = (1024, 1000)      # ReTree.matchSA's defaults, re2/retree.scala:570
C4_ALPHABET = "abcdefghijklmnopqrstuvwxyz \n"


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------- launching N ranks
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """--gpus N > 1 without a launcher above us: start the N ranks as a child process (nothing in this process
    has touched a GPU yet), pass rank 0's JSON line through, fail if any rank fails."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in p.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = p.wait()
    if rc != 0:
        print("[bench] a rank failed (exit %d)" % rc, file=sys.stderr, flush=True)
        return rc
    if line is None:
        print("[bench] the ranks printed no result line", file=sys.stderr, flush=True)
        return 1
    print(line, flush=True)
    return 0


def check_launch(args, rank, world):
    """--check-launch: everything of a multi-rank run except the GPU work -- rendezvous (gloo), the sharding of
    the workload, the pipelined interval gather and the result line -- so that the launcher and the distributed
    plumbing can be tested on a machine without GPUs.  No search happens: the intervals are made-up values each
    rank can predict for every other rank."""
    import torch
    import torch.distributed as dist
    from findex_amd.distributed import IntervalGather
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if use_dist:
        dist.init_process_group("gloo")
    if os.environ.get("FMX_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit("bench: rank %d told to fail (FMX_BENCH_FAIL_RANK)" % rank)
    t0 = time.perf_counter()
    if args.workload in REGEX:
        # the regex workloads' exchange: result lists of different lengths per rank -- rank 1 has none at all in odd
        # steps -- as int64 words (three per result), ids made global, sizes gathered, then the padded payload
        from findex_amd.distributed import exchange_result_words, RESULT_DTYPE
        k = REGEX[args.workload][1]

        def fake(r, i):       # the list rank r would send in step i: len depends on the rank, values predictable
            cnt = 0 if (r == 1 and i % 2) else 1000 + 37 * r + i
            a = np.zeros(cnt, dtype=RESULT_DTYPE)
            a["regex"] = np.arange(cnt) % k
            a["len"] = 4 + (np.arange(cnt) + r) % 60
            a["sp"] = np.arange(cnt, dtype=np.uint64) * np.uint64(r + 1) + np.uint64(i)
            a["ep"] = a["sp"] + np.uint64(r + 1)
            return a
        for i in range(args.steps):
            mine = fake(rank, i)
            got = exchange_result_words(torch.from_numpy(mine.view(np.int64).reshape(-1).copy()), rank * k)
            want = []
            for r in range(world):
                w = fake(r, i)
                w["regex"] += np.uint32(r * k)
                want.append(w)
            want = np.concatenate(want)
            assert got.size == want.size and got.tobytes() == want.tobytes(), "regex exchange content"
    else:
        k = (LITERAL.get(args.workload) or (0, 0, 1000))[2]
        gather = IntervalGather(k, torch.device("cpu"), form=args.exchange, delivery=args.delivery)
        for i in range(args.steps):
            sp, ep = gather.slot(i)
            sp.copy_(torch.arange(k, dtype=torch.int64) * (rank + 1) + i)
            ep.copy_(sp + rank + 1)
            if i == 1:
                ep[5] = sp[5] + (1 << 30)       # a wide interval: travels through the escape list of the packed form
            out = gather.launch(i)
            gather.finish()
            if out is None:
                assert args.delivery == "root" and rank != 0, "only the root receives"
                continue
            for r in range(world if use_dist else 1):
                gsp, gep = gather.intervals(out, r)
                assert int(gsp[7]) == 7 * (r + 1) + i and int(gep[7]) == 7 * (r + 1) + i + r + 1, "gather content"
                assert int(gep[5] - gsp[5]) == ((1 << 30) if i == 1 else r + 1), "wide interval"
    dt = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launch_check", "value": None, "unit": "", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                          "config": {"workload": "launch check (%s shapes), gloo, no GPU work" % args.workload,
                                     "exchange": "%s / %s" % (args.exchange, args.delivery),
                                     "ranks_in_group": dist.get_world_size() if use_dist else 1}}), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------- synthetic inputs
def make_bwt(torch, n, symbols, seed, device):
    """i.i.d. uniform symbols on the device (any byte string is a valid LF permutation); eof = n/3.
    `symbols` = sigma (bytes 1..sigma) or a string (its characters)."""
    g = torch.Generator(device=device)
    g.manual_seed(0xF1DE0000 + seed)
    bwt = torch.empty(n, dtype=torch.uint8, device=device)
    alpha = None if isinstance(symbols, int) else torch.tensor([ord(c) for c in symbols], dtype=torch.uint8, device=device)
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        if alpha is None:
            bwt[a:b] = torch.randint(1, symbols + 1, (b - a,), generator=g, device=device, dtype=torch.uint8)
        else:
            bwt[a:b] = alpha[torch.randint(0, alpha.numel(), (b - a,), generator=g, device=device)]
    return bwt, n // 3


def make_text_bwt(torch, n, seed, device, rank=0):
    """BWT of a seeded natural-language-like text of n - 1 bytes (tools/text_bwt.py), suffix-sorted on the device."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import text_bwt
    t0 = time.time()
    text = text_bwt.make_text(torch, n - 1, seed, device)
    bwt, eof = text_bwt.bwt_of_reversed_text(torch, text, lambda m: log(rank, m))
    sample = text[: min(n - 1, 1 << 24)].cpu().numpy()          # a stretch of the text: where the regex workload takes its literals
    del text
    torch.cuda.empty_cache()
    log(rank, "text: %d bytes, BWT by prefix doubling in %.1fs; begins %r" % (n - 1, time.time() - t0, sample[:60].tobytes()))
    return bwt, eof, sample


def make_patterns(torch, hip, n, sigma, k, m, seed, device, stream):
    """90 % hit patterns by LF walk (every backward step keeps a non-empty interval), 10 % with one byte
    replaced (early-exit path), SURVEY 8d.  Generated on the device with the library's own LF-walk kernel;
    hit-ness is then verified from the search results."""
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
    if isinstance(sigma, int):
        sym = torch.randint(1, sigma + 1, (k,), generator=g, device=device, dtype=torch.uint8)
    else:       # a string: the alphabet's characters
        alpha = torch.tensor([ord(c) for c in sigma], dtype=torch.uint8, device=device)
        sym = alpha[torch.randint(0, alpha.numel(), (k,), generator=g, device=device)]
    idx = torch.nonzero(mut).squeeze(1)
    pats[idx, pos[idx]] = sym[idx]
    off = torch.arange(0, (k + 1) * m, m, dtype=torch.int64, device=device)
    return pats.reshape(-1), off


def make_regexes(k, seed, text_sample=None):
    """The seeded C4 grammar (tools/regex_workload.py); only shapes the reference's ReTree.apply accepts are kept:
    the candidate stream is compiled in chunks with fmx_regex_compile_batch and the first k that compile are the
    batch (the same list tools/regex_workload.generate draws one by one).  text_sample (the text workloads): a
    regex's literal characters are a stretch of that text instead of random letters, so that it has matches.
    Returns (regex strings, CompiledRegexes holding their handles)."""
    import random
    import findex_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import regex_workload
    rng = random.Random(seed)
    res, sets = [], []
    literal = None
    if text_sample is not None:
        def literal(r, nlit):
            while True:
                a = r.randrange(0, text_sample.size - nlit)
                w = text_sample[a:a + nlit].tobytes().decode("latin-1")
                if w[0] not in " \n":              # (the grammar's extras are inserted behind the first character)
                    return w
    while len(res) < k:
        cand = [regex_workload.gen_one(rng, literal=literal) for _ in range(k - len(res) + 64)]
        cs = findex_amd.ReTree.compile_batch(cand)
        ok = np.nonzero(cs.ok())[0][: k - len(res)]
        res += [cand[i] for i in ok]
        sets.append(cs.select(ok))
    if len(sets) == 1:
        return res, sets[0]
    return res, findex_amd.ReTree.compile_batch(res)


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


def host_mem_available():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
                break
        else:
            return 0
        for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                txt = open(path).read().strip()
                if txt != "max":
                    used = 0
                    for up in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                        try:
                            used = int(open(up).read())
                            break
                        except Exception:
                            pass
                    avail = min(avail, int(txt) - used)
                break
            except Exception:
                continue
        return avail
    except Exception:
        return 0


def source_hash():
    """sha256 (first 16 hex digits) over the library's sources: what a committed PMC profile is stamped with, so that
    a profile of another build is not quoted as this one's traffic (there is no .git on the GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "findex_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "fmx.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel from the newest COMMITTED PMC profile of this workload
    (profiles/r*_<workload>_counters.csv, written by tools/summarize_prof.py from separate rocprofv3 --pmc
    passes): FETCH_SIZE KiB x the bytes one KiB stands for in this access pattern (calibrated in the same
    profile on a k_occ launch of known byte count, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024.
    Not measured in this run: returns (bytes, file name) or None when no such profile is in the tree -- or (None,
    reason) when the newest one was taken from other sources than the ones running now (its SRC_SHA16 stamp)."""
    import csv
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_counters.csv" % workload))):
        vals, per_kib, calls, disp, stamp = {}, None, None, None, None
        for r in csv.DictReader(open(f)):
            if r["Counter"] == "SRC_SHA16":
                stamp = r["Mean"]
            if kernel in r["Kernel"]:
                vals[r["Counter"]] = float(r["Mean"])
                disp = float(r["Dispatches"])
            if r["Counter"] == "FETCH_BYTES_PER_KIB":
                per_kib = float(r["Mean"])
            if r["Counter"] == "CALLS":
                calls = float(r["Mean"])
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and per_kib:
            per_launch = calls is None and 1.0 or disp / calls      # a regex call = several dispatches of the kernel
            best = (int((vals["FETCH_SIZE"] * per_kib + vals["WRITE_SIZE"] * 1024) * per_launch), os.path.basename(f))
            if stamp != source_hash():
                best = (None, "%s is stale: taken at source hash %s, this build is %s" % (os.path.basename(f), stamp, source_hash()))
    return best


# ---------------------------------------------------------------- CPU baselines (the only users of oracle/)
def oracle_index(torch, bwt_dev, eof, cores, rank):
    """The oracle's index over the SAME BWT the GPU run uses: bytes copied to the host, then
      - the reference's own structure, inverted position lists (4 bytes per row) sorted on all cores, where it can exist:
        n <= 2^32 (32-bit entries) and 6 n bytes of host memory;
      - else BASELINE.md's fallback for large n, symbol checkpoints every 256 positions + a scan of the BWT bytes
        (oracle.SampledFMSearcher: n + 4 sigma n / 256 bytes -- 48 GiB at C5), which computes the same occ / search
        (tests/test_oracle_kat.py holds it to the inverted lists) -- `orc.kind` says which.
    None when the host can hold neither."""
    import oracle
    n = bwt_dev.numel()
    avail = host_mem_available()

    def to_host():
        h = np.empty(n, dtype=np.uint8)
        chunk = 1 << 28
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            h[a:b] = bwt_dev[a:b].cpu().numpy()
        return h
    t0 = time.time()
    if n <= (1 << 32) and not (avail and avail < 6 * n + (2 << 30)):
        h_bwt = to_host()
        counts = oracle.histogram(h_bwt, eof, threads=cores)
        orc = oracle.NaiveFMSearcher.from_mem(h_bwt, eof, counts, threads=cores)
        del h_bwt
        orc.kind = "inverted lists + binary-search occ (the reference's structure, bwtmerger.scala:354-375)"
        return orc, time.time() - t0
    need = 3 * n + (2 << 30)          # the bytes + checkpoints of up to 128 symbols (4 * 128 / 256 = 2 bytes per row)
    if avail and avail < need:
        log(rank, "cpu_baseline: n=%d needs %.0f GiB of host memory even as checkpoints (%.0f GiB available)" % (n, need / 2**30, avail / 2**30))
        return None, 0.0
    h_bwt = to_host()
    orc = oracle.SampledFMSearcher(h_bwt, eof, threads=cores)
    orc.kind = ("symbol checkpoints every 256 positions + a scan of the BWT bytes (BASELINE.md's structure for n >= 2^31: the "
                "reference's 32-bit inverted lists cannot describe 2^%d rows); same occ / search as the lists, %.0f GiB"
                % (n.bit_length() - 1, (orc.bytes() + n) / 2**30))
    log(rank, "cpu_baseline: n=%d: checkpoints + BWT bytes on the host (%.0f GiB) in %.1fs" % (n, (orc.bytes() + n) / 2**30, time.time() - t0))
    return orc, time.time() - t0


def cpu_baseline_literal(torch, orc, t_build, n, pats, off, sp, ep, sample, m, cores, rank, note):
    """The CPU path beside the GPU number: the oracle's restatement of the reference algorithm (inverted
    position lists + binary-search occ, bwtmerger.scala:354-375) on this host's cores over the first `sample`
    patterns of the timed batch.  The GPU answers for them are checked against it bit for bit."""
    h_pats = pats[: sample * m].cpu().numpy()
    h_off = off[: sample + 1].cpu().numpy().astype(np.uint64)
    t0 = time.time()
    wsp, wep, steps = orc.search_batch(h_pats, h_off, threads=cores)
    dt = time.time() - t0
    ok = bool(np.array_equal(wsp, sp[:sample].cpu().numpy().astype(np.uint64)) and
              np.array_equal(wep, ep[:sample].cpu().numpy().astype(np.uint64)))
    if not ok:
        raise SystemExit("bench: GPU results differ from the CPU oracle on the baseline sample")
    ranks = 2 * int(steps.sum())
    # the same on ONE core, over the first sample / cores patterns
    s1 = max(1, sample // max(cores, 1))
    t0 = time.time()
    _, _, steps1 = orc.search_batch(h_pats[: s1 * m], h_off[: s1 + 1], threads=1)
    dt1 = time.time() - t0
    log(rank, "cpu_baseline: %d cores, %.2fs for %d patterns (%d rank queries), index build %.1fs, parity ok; one core: %.2fs for %d patterns"
        % (cores, dt, sample, ranks, t_build, dt1, s1))
    return {"value": ranks / dt / 1e6, "unit": "M rank-queries/s", "cores": cores, "kind": "port",
            "one_core_value": 2 * int(steps1.sum()) / dt1 / 1e6,
            "patterns_per_s": sample / dt, "index_build_s": t_build, "n": n,
            "structure": getattr(orc, "kind", "inverted lists + binary-search occ"),
            "sample": "%s; first %d of the timed batch's %d-char patterns; %s, in C with OpenMP; the GPU's (sp, ep) for them "
                      "are bit-equal" % (note, sample, m, getattr(orc, "kind", "inverted lists + binary-search occ"))}


def cpu_baseline_regex(orc, t_build, n, res, trees, gpu_out, sample, max_len, ref_mode, cores, rank, note):
    """ReTree._matchSA (re2/retree.scala:618-653) in C on this host's cores over the first `sample` regexes of
    the timed batch.  Frontier workloads: limits not binding, match length capped like the GPU run, result
    multisets compared.  Reference-order workloads: the reference's default limits, result LISTS compared in the
    reference's own order."""
    import findex_amd
    t0 = time.time()
    findex_amd.ReTree.compile_batch(res[:sample])           # the front-end's share of a one-shot batch (same host code)
    t_compile = time.time() - t0
    tables = [trees[i].tables() for i in range(sample)]
    t0 = time.time()
    if ref_mode:
        want, pops, trunc = orc.match_tables_batch(tables, maxBranching=REF_LIMITS[0], maxIterations=REF_LIMITS[1],
                                                   threads=cores, ordered=True)
    else:
        want, pops, trunc = orc.match_tables_batch(tables, max_len=max_len, threads=cores)
    dt = time.time() - t0
    got = gpu_out[gpu_out["regex"] < sample]
    ok = got.size == want.size and all(np.array_equal(got[f], want[f]) for f in ("regex", "len", "sp", "ep"))
    if not ok:
        raise SystemExit("bench: GPU regex results differ from the CPU oracle on the baseline sample "
                         "(%d vs %d results)" % (got.size, want.size))
    log(rank, "cpu_baseline: %d cores, %.2fs for %d regexes (%d getPrevRange steps, %d results), index build %.1fs, "
        "front-end %.2fs, parity ok (%s)" % (cores, dt, sample, pops, want.size, t_build, t_compile,
                                             "lists in the reference's order" if ref_mode else "multisets"))
    how = ("the reference's default limits (maxBranching %d, maxIterations %d); result lists equal in the reference's "
           "own order" % REF_LIMITS) if ref_mode else ("limits not binding, match length capped at %d like the GPU run; "
                                                       "result multisets bit-equal" % max_len)
    return {"value": 2 * pops / dt / 1e6, "unit": "M rank-queries/s", "cores": cores, "kind": "port",
            "regexes_per_s": sample / dt, "index_build_s": t_build, "n": n,
            "fresh_batch_regexes_per_s": sample / (dt + t_compile), "front_end_s": t_compile,
            "sample": "%s; first %d regexes of the timed batch; ReTree._matchSA with its priority queue in C, one "
                      "regex per OpenMP task, %s; fresh_batch adds the regex front-end (the library's "
                      "fmx_regex_compile_batch on the same cores) for the sample" % (note, sample, how)}


def measure_exchange(args, torch, dist, hip, gather, k, device, stream, use_dist, reps=10):
    """The exchange on its own (nothing else on the device), so that a scaling curve says which side of
    max(search, gather) binds a step: per form (packed 8 B / pairs 16 B per pattern) and delivery (root only /
    all-gather) the time from the pack kernel's launch to the collective's end, HIP events on the current stream around
    `launch` + `finish` (the collective's own stream is joined by finish()).  None without a process group."""
    if not use_dist:
        return None
    from findex_amd.distributed import IntervalGather
    res = {}
    for form in ("packed", "pairs"):
        for delivery in ("root", "all"):
            g = gather if (form == args.exchange and delivery == args.delivery) else \
                IntervalGather(k, device, form=form, delivery=delivery, searcher=hip, depth=1)
            sp, ep = g.slot(0)
            if g is not gather:
                sp.copy_(gather.mine[0][0])
                ep.copy_(gather.mine[0][1])
            ms = []
            for i in range(reps + 2):
                dist.barrier()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.launch(0, stream, packed_already=(form == "packed" and g is gather))      # the configured form arrives packed by the search
                g.finish()
                e1.record()
                torch.cuda.synchronize()
                if i >= 2:
                    ms.append(e0.elapsed_time(e1))
            t = torch.tensor([sum(ms) / len(ms)], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res["%s/%s" % (form, delivery)] = {"gather_ms": float(t.item()), "payload_bytes_per_rank": g.payload_bytes}
            if g is not gather:
                del g
    mine = res["%s/%s" % (args.exchange, args.delivery)]
    return {"form": args.exchange, "delivery": args.delivery, "gather_ms": mine["gather_ms"],
            "payload_bytes_per_rank": mine["payload_bytes_per_rank"], "all_forms": res,
            "gather_ms_is": "pack kernel + collective alone on the device (max over ranks, mean of %d); inside a step it runs "
                            "beside the next step's search: a step costs max(search_ms, gather_ms)" % reps}


def measure_host_path(torch, hip, pats, off, sp_dev, ep_dev, k, m, reps=5):
    """The same batch through the HOST-pointer entry points (what findex's API hands over: JVM arrays,
    findex.scala:15-31) -- PCIe-inclusive, never `value`: fmx_search_batch from pageable and page-locked buffers, and the
    lean form (fmx_search_batch_ex: no offsets for equal-length patterns, intervals back in the 8-byte form)."""
    import findex_amd
    from findex_amd.searcher import PinnedArray
    h_pat = pats.cpu().numpy()
    h_off = off.cpu().numpy().astype(np.uint64)
    want_sp = sp_dev.cpu().numpy().astype(np.uint64)
    want_ep = ep_dev.cpu().numpy().astype(np.uint64)

    def timed(fn):
        fn()
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return r, sorted(ts)[len(ts) // 2]
    out = {}
    (gsp, gep), t = timed(lambda: hip.search_batch(h_pat, h_off))
    assert np.array_equal(gsp, want_sp) and np.array_equal(gep, want_ep)
    out["pageable"] = {"ms_per_call": t * 1e3, "patterns_per_s": k / t, "bytes_per_pattern": m + 8 + 16}
    p_pat, p_off = PinnedArray(h_pat.shape, np.uint8), PinnedArray(h_off.shape, np.uint64)
    p_sp, p_ep = PinnedArray((k,), np.uint64), PinnedArray((k,), np.uint64)
    p_pat.array[:] = h_pat
    p_off.array[:] = h_off
    _, t = timed(lambda: hip.search_batch(p_pat.array, p_off.array, out=(p_sp.array, p_ep.array)))
    assert np.array_equal(p_sp.array, want_sp) and np.array_equal(p_ep.array, want_ep)
    out["page_locked"] = {"ms_per_call": t * 1e3, "patterns_per_s": k / t, "bytes_per_pattern": m + 8 + 16}
    cap = max(16, k // 256)
    pk, t = timed(lambda: hip.search_batch_ex(h_pat, fixed_len=m, packed=True, escape_cap=cap))
    usp, uep = hip.unpack_intervals(pk, k, cap)
    assert np.array_equal(usp, want_sp) and np.array_equal(uep, want_ep)
    out["lean_pageable"] = {"ms_per_call": t * 1e3, "patterns_per_s": k / t, "bytes_per_pattern": m + 8,
                            "what": "fmx_search_batch_ex: fixed_len = %d (no offsets travel), intervals back in the 8-byte form" % m}
    out["what"] = ("fmx_search_batch on host arrays, median of %d calls after two warm-ups, results checked against the device-"
                   "resident step's; the link moves ~53 GB/s each way on these boxes" % reps)
    return out


def measure_rank_only(args, torch, hip, batches, k, stream, device, onehot, rank):
    """The RANK kernel's own roofline (north_star: "each rank(c,i) is one coalesced load plus an in-lane popcount ... achieved
    HBM GB/s against the ~8 TB/s peak"; bwtmerger.scala:354-375 is what it replaces): the same ring of batches with every
    derived table OFF -- k-mer table, row jump table, row tables dropped and disabled on this handle -- so that every backward
    step of every pattern is executed as rank-dictionary line requests (one per step once the interval is a row, two while sp
    and ep lie in different blocks).  What is reported is EXECUTED work: line requests per second (device counter), their
    bytes at the line's size against the HBM peak.  Changes the handle's tables: call it last."""
    sp = torch.empty(k, dtype=torch.int64, device=device)
    ep = torch.empty(k, dtype=torch.int64, device=device)
    hip.drop_tables(jump=True, frontier=True, ktab=True)
    hip.config_set("ktab", "off")
    hip.config_set("jump", "off")
    hip.prepare(ktab=False, search=True)          # nothing to build: calibrates the table-less kernel's grid
    ring = len(batches)
    per = []
    for bp, bo in batches:
        hip.stats_reset()
        hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
        torch.cuda.synchronize()
        sj = hip.stats()
        assert sj["ktab_lookups"] == 0 and sj["jump_lookups"] == 0 and sj["row_lookups"] == 0 and sj["tables_held_bytes"] == 0
        per.append((int(sj["search_requests"]), int(sj["rank_queries"])))
    steps = max(args.steps, 5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        bp, bo = batches[i % ring]
        hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    reqs = sum(per[i % ring][0] for i in range(steps)) / float(steps)
    occs = sum(per[i % ring][1] for i in range(steps)) / float(steps)
    line = 64.0 if onehot else 66.0
    achieved = reqs * line / (ms * 1e-3) / 1e9
    log(rank, "rank_only (all derived tables off): %.4f ms per step, %.2f G executed rank-line requests/s = %.0f GB/s (%.3f of peak)"
        % (ms, reqs / ms / 1e6, achieved, achieved / HBM_PEAK_GBS))
    return {"ms": ms, "steps": steps, "executed_rank_queries_per_s": reqs / (ms * 1e-3),
            "executed_rank_queries_is": "rank-dictionary line requests the kernel issued per second (device counter): each is one "
                                        "%d-byte block fetched and popcounted in the lane group -- one per backward step on a one-row "
                                        "interval, one or two on a wider one" % int(line),
            "rank_line_requests_per_launch": reqs, "reference_occ_evaluations_per_s": occs / (ms * 1e-3),
            "bytes_per_request": line, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "requests_G_per_s": reqs / ms / 1e6, "request_frac": reqs / ms / 1e6 / REQUEST_CEILING_G_PER_S,
            "kernel": "k_search4 with KT = 0, no row tables: every step on the rank dictionary",
            "what": "the same ring of %d batches, fmx_drop_tables + this handle's ktab = off / jump = off; HIP events around %d "
                    "back-to-back steps" % (ring, steps)}


# ---------------------------------------------------------------- the two kinds of workload
def run_literal(args, torch, dist, findex_amd, rank, world, local, device, use_dist, stream):
    is_text = args.workload in TEXT
    if is_text:
        log2n, k, m, seed = TEXT[args.workload]
        sigma = C4_ALPHABET
    else:
        log2n, sigma, k, m, seed = LITERAL[args.workload]
    n = 1 << log2n
    t0 = time.time()
    if is_text:
        bwt, eof, _ = make_text_bwt(torch, n, seed, device, rank)
    else:
        bwt, eof = make_bwt(torch, n, sigma, seed, device)          # same seed on every rank: replicas
    torch.cuda.synchronize()
    hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=local, stream=stream)
    hip.prepare(ktab=True, jump=True)      # a serving handle: its derived tables up front (tables_build_ms), not at the threshold
    st = hip.stats()
    n_sigma = sigma if isinstance(sigma, int) else len(sigma)
    log(rank, "index: n=2^%d sigma=%d, %.1f GiB in HBM (%d symbols x %d blocks x %d B), built in %.1f ms (+%.1fs setup)"
        % (log2n, n_sigma, st["index_bytes"] / 2**30, st["n_symbols"], st["n_blocks"], st["block_bytes"],
           st["build_ms"], time.time() - t0))
    want_cpu = world == 1 and rank == 0 and not args.no_cpu_baseline
    if not want_cpu:
        del bwt
        torch.cuda.empty_cache()
    # A RING of distinct pattern batches, rotated through the steps (round 5).  The reference's search(in) never sees the
    # same query twice (findex.scala:15-31), but until round 4 every timed figure replayed ONE batch: a C3 step touches
    # ~335 MB of distinct 64-byte sectors, which straddles the 256 MiB Infinity Cache, so part of a replayed step was served
    # on die (profiles/r05_c3_cold.md).  With R >= 8 batches, R x that lies between two uses of any line: every step runs
    # against HBM.  The replayed-batch figure is kept beside the headline as `replayed_batch_ms`.
    ring = max(1, args.ring)
    batches = [make_patterns(torch, hip, n, sigma, k, m, seed * 1000 + rank + 7919 * j, device, stream) for j in range(ring)]
    pats, off = batches[0]
    # the intervals land in the slots of a pipelined gather: with N > 1 the gather of step i (RCCL over
    # xGMI, 8 B per pattern) runs on the collective's stream while step i+1 is being searched
    from findex_amd.distributed import IntervalGather
    gather = IntervalGather(k, device, form=args.exchange, delivery=args.delivery, searcher=hip)
    step_no = [0]

    def step():
        i = step_no[0]
        step_no[0] += 1
        bp, bo = batches[i % ring]
        if use_dist:        # the intervals are written in the exchange's form by the search itself, then gathered: the path's one exchange
            gather.search_into(i, hip, bp.data_ptr(), bo.data_ptr(), stream)
            gather.launch(i, stream, packed_already=True)
            return None, None
        sp, ep = gather.slot(i)
        hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, stream)
        return sp, ep

    # what one step of each batch executes (device counters; the batches differ by a fraction of a per cent), and batch 0's
    # intervals for the hit count, the host path and the CPU baseline
    sp0 = torch.empty(k, dtype=torch.int64, device=device)
    ep0 = torch.empty(k, dtype=torch.int64, device=device)
    per_batch = []
    hits_b = []
    for j in range(ring):
        bp, bo = batches[j]
        hip.stats_reset()
        hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp0.data_ptr(), ep0.data_ptr(), k, stream)
        torch.cuda.synchronize()
        sj = hip.stats()
        per_batch.append({f: int(sj[f]) for f in ("rank_queries", "search_requests", "ktab_lookups", "jump_lookups", "row_lookups")})
        hits_b.append(int((sp0 < ep0).sum().item()))
    hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp0.data_ptr(), ep0.data_ptr(), k, stream)
    torch.cuda.synchronize()
    s1 = hip.stats()
    for _ in range(max(0, args.warmup)):
        step()
    gather.finish()
    torch.cuda.synchronize()

    # The kernel's own duration is measured live, with HIP events on its stream, on every EVENT_EVERY-th step of the timed
    # region.  Not on every step: an event record is a marker packet the command processor drains the stream for, and
    # two of them between consecutive launches were 8-11 us of a 150 us step (device clock: launches inside one C5 step
    # follow each other without a gap, steps were 11 us apart) -- measurement overhead, not the path's.  (3: coprime with the
    # ring's 8, so the timed launches rotate through the batches too.)
    EVENT_EVERY = 3
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if j % EVENT_EVERY == 0 else None
          for j in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    first_timed = step_no[0]
    t0 = time.perf_counter()
    for pair in ev:
        i = step_no[0]
        step_no[0] += 1
        bp, bo = batches[i % ring]
        if use_dist:
            gather.slot(i)      # (waits for the collective that last used the slot, outside the search's events)
            if pair:
                pair[0].record()          # torch's current stream == the stream the kernel is launched on
            gather.search_into(i, hip, bp.data_ptr(), bo.data_ptr(), stream)
            if pair:
                pair[1].record()
            gather.launch(i, stream, packed_already=True)
        else:
            sp_i, ep_i = gather.slot(i)
            if pair:
                pair[0].record()
            hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), sp_i.data_ptr(), ep_i.data_ptr(), k, stream)
            if pair:
                pair[1].record()
    gather.finish()         # every step's gather completes inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist and gather.overflow(step_no[0] - 1):       # (a collective; outside the timed region) more wide intervals than the packed form's escape list holds
        raise SystemExit("bench: the last step's escape list overflowed -- the exchange must run in the 16-byte form (--exchange pairs)")
    timed = [p for p in ev if p]
    kernel_ms = sum(a.elapsed_time(b) for a, b in timed) / len(timed)
    # the work of exactly the timed steps
    timed_batches = [(first_timed + j) % ring for j in range(args.steps)]

    def mean_of(f):
        return sum(per_batch[b][f] for b in timed_batches) / float(args.steps)
    ranks_per_step = mean_of("rank_queries")
    requests_per_step = mean_of("search_requests")
    lookups_per_step = mean_of("ktab_lookups")
    jumps_per_step = mean_of("jump_lookups")
    rows_per_step = mean_of("row_lookups")
    hits = sum(hits_b[b] for b in timed_batches) / float(args.steps)

    # ---- the same step on ONE batch replayed (what rounds 1-4 reported): part of it is served by the Infinity Cache
    def replay(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp0.data_ptr(), ep0.data_ptr(), k, stream)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp0.data_ptr(), ep0.data_ptr(), k, stream)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    replayed_ms = replay(max(args.steps, 5))

    # ---- the same ring asked with FMX_SEARCH_MISS_NONE (fmx.h): a pattern that does not occur may come back as (0, 0) -- None in
    # the reference either way (findex.scala:30) -- which spares the kernel the walk to the loop's values at the failing step.
    # What the JVM adapter's search() runs; never the headline (the headline step returns those values).
    def miss_none(reps):
        a, b = torch.empty(k, dtype=torch.int64, device=device), torch.empty(k, dtype=torch.int64, device=device)
        hip.stats_reset()
        hip.search_batch_ex_dev(pats.data_ptr(), off.data_ptr(), a.data_ptr(), b.data_ptr(), k, stream, miss_none=True)
        torch.cuda.synchronize()
        sj = hip.stats()
        hit = sp0 < ep0                  # (sp0 / ep0: batch 0's intervals from the plain call)
        assert torch.equal(a[hit], sp0[hit]) and torch.equal(b[hit], ep0[hit]) and bool((a[~hit] >= b[~hit]).all()), "MISS_NONE changed a result"
        assert int(sj["rank_queries"]) == per_batch[0]["rank_queries"], "MISS_NONE changed the count of the reference's steps"
        reqs = int(sj["search_requests"] + sj["ktab_lookups"] + sj["jump_lookups"] + sj["row_lookups"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(3):
            bp, bo = batches[i % ring]
            hip.search_batch_ex_dev(bp.data_ptr(), bo.data_ptr(), a.data_ptr(), b.data_ptr(), k, stream, miss_none=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            bp, bo = batches[i % ring]
            hip.search_batch_ex_dev(bp.data_ptr(), bo.data_ptr(), a.data_ptr(), b.data_ptr(), k, stream, miss_none=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        return {"ms_per_step": ms, "patterns_per_sec": k / (ms * 1e-3), "requests_per_launch": reqs, "requests_G_per_s": reqs / ms / 1e6,
                "what": "the ring's steps through fmx_search_batch_ex_dev with FMX_SEARCH_MISS_NONE: hits bit-equal to the plain call's, "
                        "every miss sp >= ep, the reference-equivalent step count unchanged; %d back-to-back steps" % reps}
    miss_none_rec = miss_none(max(args.steps, 10)) if not (use_dist or os.environ.get("FMX_BENCH_NO_MISS_NONE")) else None      # (the variable: A/B runs against older libraries)

    exchange = measure_exchange(args, torch, dist, hip, gather, k, device, stream, use_dist)
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
    if rank != 0:
        return None
    if exchange is not None:
        exchange["search_ms"] = kernel_ms_max
        exchange["expected_step_ms"] = max(kernel_ms_max, exchange["gather_ms"])
        exchange["step_is_bound_by"] = "gather" if exchange["gather_ms"] > kernel_ms_max else "search"

    # ---- roofline of the one kernel a step launches (k_search4).  Algorithmic bytes of THIS layout per launch:
    # every memory request for a rank-dictionary line the kernel issued (device counter), at the line's size, plus
    # the operands it streams: pattern bytes, (k+1) offsets, 16 B of (sp, ep) per pattern.
    onehot = st["layout"] == 0
    line_bytes = 64.0 if onehot else 66.0        # bytes layout: a 128-B block and its 4-B checkpoint, one request each
    operand_bytes = k * m + 8 * (k + 1) + 16 * k
    jump_entry = 32.0 if s1["jump_bytes"] >= 32 * n else 16.0      # a pair of entries per lookup, or one
    alg_bytes = requests_per_step * line_bytes + 16.0 * lookups_per_step + jump_entry * jumps_per_step + 8.0 * rows_per_step + operand_bytes
    all_requests = requests_per_step + lookups_per_step + jumps_per_step + rows_per_step      # every one a dependent random request
    ksec = kernel_ms * 1e-3
    achieved = alg_bytes / ksec / 1e9
    resident = "hbm" if st["index_bytes"] > INFINITY_CACHE_BYTES else "infinity-cache"
    traffic = pmc_traffic(args.workload, "k_search4")
    roof = {
        "bound": "hbm", "kernel": "k_search4",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic[0] if traffic else None,
        "traffic_source": (traffic[1] if traffic and traffic[0] is None else
                           ("committed profile profiles/%s (separate rocprofv3 --pmc passes of this very source; not "
                            "measured in this run)" % traffic[1]) if traffic else "no PMC profile of this workload committed"),
        "algorithmic_over_traffic": round(alg_bytes / traffic[0], 3) if traffic and traffic[0] else None,
        "traffic_note": "HBM is read in 64-byte sectors: the %d 16- and 32-byte table lookups of a launch cost %d MB there against %d MB "
                        "algorithmic -- the difference between traffic and algorithmic bytes, not re-reads (the lookups are "
                        "random rows: no second entry of a sector is ever wanted); the bound is requests per second, not bytes"
                        % (lookups_per_step + jumps_per_step, (lookups_per_step + jumps_per_step) * 64 // 10**6,
                           int(lookups_per_step * 16 + jumps_per_step * jump_entry) // 10**6),
        "algorithmic_bytes_per_launch": alg_bytes,
        "algorithmic_bytes": "%d rank-line requests x %g B + %d k-mer table entries x 16 B + %d row jump table lookups x %g B + %d row "
                             "table words x 8 B + %d operand bytes (patterns, offsets, intervals)"
                             % (requests_per_step, line_bytes, lookups_per_step, jumps_per_step, jump_entry, rows_per_step, operand_bytes),
        "kernel_ms": kernel_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
        "kernel_ms_is": "HIP events around the launch on %d of the %d timed steps (every %dth: the markers themselves cost a "
                        "step 8-11 us)" % (len(timed), args.steps, EVENT_EVERY),
        "requests_per_launch": all_requests, "rank_line_requests": requests_per_step, "ktab_lookups": lookups_per_step,
        "jump_lookups": jumps_per_step, "jump_table_gib": s1["jump_bytes"] / 2**30,
        "row_lookups": rows_per_step, "row_table_gib": s1["row_bytes"] / 2**30,
        "ktab_k": int(s1["ktab_k"]), "rank_queries_per_launch": ranks_per_step,
        "rank_queries_per_request": ranks_per_step / max(all_requests, 1),
        # SURVEY 8d's own pricing (its structure fetches 128/132 B per rank query; this layout does not): reported
        # for comparison only, it is not a fraction of anything
        "survey_equiv_GBps": ranks_per_step * survey_bytes_per_rank(n_sigma) / ksec / 1e9,
        "index_resident_in": resident,
        "index_resident_in_is": ("the steps rotate through %d distinct batches: ~%.1f GB of distinct 64-byte sectors are touched between two "
                                 "uses of any line, %.0fx the 256 MiB Infinity Cache -- nothing of a step is served on die "
                                 "(profiles/r05_c3_cold.md); replayed_batch_ms is the same step on ONE batch replayed"
                                 % (ring, ring * all_requests * 64 / 1e9, ring * all_requests * 64 / INFINITY_CACHE_BYTES)) if resident == "hbm" else None,
    }
    if resident == "hbm":
        # the limit that binds this access pattern: distinct dependent memory requests per second
        roof["requests_G_per_s"] = all_requests / ksec / 1e9
        roof["request_ceiling_G_per_s"] = REQUEST_CEILING_G_PER_S
        roof["request_ceiling_source"] = ("tools/ubench/mix.hip (dependent chains with this kernel's mix of 64 / 16 / 8-byte requests over "
                                          "four tables of 177 GiB, 16 chains per wave): profiles/r04_ubench_mix.txt, profiles/r04_c3_bound.md")
        roof["request_frac"] = roof["requests_G_per_s"] / REQUEST_CEILING_G_PER_S
    else:
        roof["note"] = ("the rank dictionary (%.0f MB) stays in the 256 MiB Infinity Cache: bytes are served on die, the "
                        "HBM peak is quoted because the contract asks for it, not because it binds"
                        % (st["index_bytes"] / 1e6))
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
        "data": "synthetic" if not is_text else "synthetic text with natural repeats (words.txt's words drawn with replacement), its true BWT",
        "patterns_per_sec": world * k * args.steps / dt,
        "ring_batches": ring,
        "ms_per_step_is": "steps rotate through %d distinct pattern batches (seeds differ): no step sees a batch the device has just "
                          "searched; the same step on ONE replayed batch: replayed_batch_ms" % ring,
        "replayed_batch_ms": replayed_ms,
        "miss_none": miss_none_rec,
        # co-headline: what the memory system is asked per second; `value` counts the reference's occ evaluations,
        # which the tables of rounds 2-4 serve with fewer and fewer requests -- compare rounds by patterns_per_sec / this
        "requests_G_per_s": world * all_requests * args.steps / dt / 1e9,
        "value_is": "reference-equivalent occ evaluations per second (config.rank_queries_are); executed memory requests: requests_G_per_s",
        "exchange": exchange,
        "config": {
            "workload": "%s: %d x %d-char literal patterns per GPU, 2^%d-byte sigma=%d %s resident "
                        "in HBM (rank dictionary replicated per GPU)"
                        % (args.workload.upper(), k, m, log2n, n_sigma, "BWT of a text with natural repeats" if is_text else "synthetic BWT"),
            "n": n, "sigma": n_sigma, "patterns_per_gpu": k, "pattern_len": m,
            "hit_patterns_fraction": hits_all / (world * k),
            "rank_queries_per_step": ranks_all,
            "rank_queries_are": "occ evaluations of the reference's loop on these inputs (2 per backward step, early "
                                "exits counted) -- a reference-equivalent count, not executed popcounts: the kernel serves them "
                                "with rank_queries_per_request per memory request (k-mer table for the first K steps, one "
                                "row-jump-table %s per %d steps once the interval is a row, shared blocks)"
                                % ("request -- a pair of entries, 32 bytes --" if s1["jump_bytes"] >= 32 * n else "entry",
                                   max(int(s1["jump_chars"]), 1) * (2 if s1["jump_bytes"] >= 32 * n else 1)),
            "parallelism": "patterns sharded over %d GPU(s), index replicated%s"
                           % (world, (", one gather of the hit intervals per step (%s form, delivered to %s), overlapped with the "
                                      "next step's search" % (args.exchange, "rank 0" if args.delivery == "root" else "every rank"))
                              if use_dist else ""),
            "ranks_in_group": dist.get_world_size() if use_dist else 1,
            "index_gib": s1["index_bytes"] / 2**30, "index_build_ms": st["build_ms"],
            "tables_build_ms": s1["tables_build_ms"],
            "tables_alloc_ms": s1["tables_alloc_ms"],
            "tables_alloc_ms_is": "of tables_build_ms: the time inside hipMalloc -- ~0 when the device's memory has not been held since the box "
                                  "came up, seconds when a process released it just before (the driver wipes released memory before it "
                                  "hands it out again: profiles/r05_alloc.md); the rest is the build kernels",
            "tables_build_ms_is": "the k-mer jump table (K = %d), the row jump table (%.1f GiB) and the row table (%.1f GiB), built "
                                  "here by fmx_prepare (by default: when a handle has searched n / 64 patterns): paid once per open on "
                                  "top of index_build_ms; the most a build held at once: %.1f GiB"
                                  % (int(s1["ktab_k"]), s1["jump_bytes"] / 2**30, s1["row_bytes"] / 2**30,
                                     s1["peak_table_build_bytes"] / 2**30),
            "index_layout": "one-hot bit-vectors, 64-B blocks" if onehot else "BWT bytes + checkpoints",
            # workgroups per CU the search kernel's grid is sized for (bit 8: confirmed by the kernel's residency census,
            # DESIGN.md 3 "Residency" -- the occupancy query can answer one too many)
            "search_residency": "0x%x" % int(hip.stats().get("search_residency", 0)),
        },
        "roofline": roof,
    }
    if world == 1 and not args.no_host_path:
        # (batch 0's intervals again: sp0 / ep0 held the replayed batch's, which they are)
        out["host_path"] = measure_host_path(torch, hip, pats, off, sp0, ep0, k, m)
        log(rank, "host path: pageable %.3f ms, page-locked %.3f ms, lean (no offsets, 8-byte intervals) %.3f ms per call"
            % (out["host_path"]["pageable"]["ms_per_call"], out["host_path"]["page_locked"]["ms_per_call"],
               out["host_path"]["lean_pageable"]["ms_per_call"]))
    if world == 1 and rank == 0 and not args.no_rank_only:      # (like the CPU leg: at N = 1 only -- the other ranks would have left)
        # (after everything else that uses this handle's tables, before the CPU leg, which does not use the handle)
        out["rank_only"] = measure_rank_only(args, torch, hip, batches, k, stream, device, onehot, rank)
    if want_cpu:
        cores = effective_cores()
        sample = min(k, 200_000)
        orc, t_build = oracle_index(torch, bwt, eof, cores, rank)
        note = "the timed run's own index (n=2^%d)" % log2n
        if orc is None:
            # fallback: the same generator and alphabet at n = 2^27, its own pattern batch of the same shape
            n_s = 1 << 27
            bwt_s, eof_s = make_bwt(torch, n_s, sigma, 77, device)
            hip_s = findex_amd.HipFMSearcher.from_device(bwt_s.data_ptr(), n_s, eof_s, None, device=local, stream=stream)
            hip_s.prepare(ktab=True, jump=True)
            p_s, o_s = make_patterns(torch, hip_s, n_s, sigma, sample, m, 78, device, stream)
            sp_s = torch.empty(sample, dtype=torch.int64, device=device)
            ep_s = torch.empty(sample, dtype=torch.int64, device=device)
            hip_s.search_batch_dev(p_s.data_ptr(), o_s.data_ptr(), sp_s.data_ptr(), ep_s.data_ptr(), sample, stream)
            torch.cuda.synchronize()
            orc, t_build = oracle_index(torch, bwt_s, eof_s, cores, rank)
            out["cpu_baseline"] = cpu_baseline_literal(torch, orc, t_build, n_s, p_s, o_s, sp_s, ep_s, sample, m, cores, rank,
                                                       "FALLBACK: the host cannot hold the 2^%d-row position list, so the same "
                                                       "generator and alphabet at n=2^27 with its own pattern batch" % log2n)
            hip_s.close()
        else:
            out["cpu_baseline"] = cpu_baseline_literal(torch, orc, t_build, n, pats, off, sp0, ep0, sample, m, cores, rank, note)
        orc.close()
    return out


def run_regex(args, torch, dist, findex_amd, rank, world, local, device, use_dist, stream):
    is_text = args.workload in TEXT
    text_sample = None
    if is_text:
        log2n, k, max_len, seed = TEXT[args.workload]
        ref_mode = False
    else:
        log2n, k, seed, max_len, ref_mode = REGEX[args.workload]
    n = 1 << log2n
    t0 = time.time()
    if is_text:
        bwt, eof, text_sample = make_text_bwt(torch, n, seed, device, rank)
    else:
        bwt, eof = make_bwt(torch, n, C4_ALPHABET, seed, device)
    torch.cuda.synchronize()
    hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=local, stream=stream)
    st = hip.stats()
    log(rank, "index: n=2^%d sigma=%d, %.1f GiB in HBM, built in %.1f ms (+%.1fs setup)"
        % (log2n, len(C4_ALPHABET), st["index_bytes"] / 2**30, st["build_ms"], time.time() - t0))
    want_cpu = world == 1 and rank == 0 and not args.no_cpu_baseline
    if not want_cpu:
        del bwt
        torch.cuda.empty_cache()
    t0 = time.time()
    res, trees = make_regexes(k, seed * 1000 + rank, text_sample)           # weak scaling: every rank its own k regexes
    t_gen = time.time() - t0
    cap = 1 << 22
    lim = dict(mode="reference", maxBranching=REF_LIMITS[0], maxIterations=REF_LIMITS[1]) if ref_mode else dict(max_steps=max_len)

    # ---- a batch that arrives as strings and is matched once: front-end + resident + first match (fresh_batch)
    hip.stats()                  # tables of the index built (k-mer table): not part of a batch's cost
    findex_amd.ReTree.prepare_batch(hip, trees[:16]).match_raw(cap=cap, **lim)
    t0 = time.perf_counter()
    fresh_trees = findex_amd.ReTree.compile_batch(res)
    t1 = time.perf_counter()
    fresh = findex_amd.ReTree.prepare_batch(hip, fresh_trees)
    t2 = time.perf_counter()
    fresh.match_raw(cap=cap, copy=False, **lim)
    t3 = time.perf_counter()
    fresh_batch = {"regexes_per_s": k / (t3 - t0), "seconds_per_batch": t3 - t0, "compile_s": t1 - t0,
                   "resident_s": t2 - t1, "first_match_s": t3 - t2, "host_threads": effective_cores(),
                   "what": "fmx_regex_compile_batch (all host cores) + fmx_regex_batch_create + the first "
                           "fmx_regex_batch_match of %d regexes given as strings, results in page-locked host memory" % k}
    del fresh, fresh_trees
    log(rank, "fresh batch of %d regexes: compile %.3fs + resident %.3fs + first match %.3fs (workload generation %.1fs)"
        % (k, t1 - t0, t2 - t1, t3 - t2, t_gen))
    batch = findex_amd.ReTree.prepare_batch(hip, trees)
    # a ring of resident batches the steps rotate through (round 5, like the literal workloads' pattern batches): no call
    # matches the batch the device has just matched.  Batch 0 is the one the CPU baseline checks.
    ring = max(1, args.regex_ring)
    batches = [batch]
    for j in range(1, ring):
        _, trees_j = make_regexes(k, seed * 1000 + rank + 7919 * j, text_sample)
        batches.append(findex_amd.ReTree.prepare_batch(hip, trees_j))

    from findex_amd.distributed import all_gather_varlen
    from findex_amd.regex import RESULT_DTYPE

    # A frontier step = fmx_regex_batch_match_dev: the regex path's device-pointer form, results (grouped by regex,
    # ordered) and per-regex counts left in HBM -- as the literal workloads' step leaves its intervals there.  The rate
    # with the results delivered into page-locked host memory is reported beside it (`host_delivered`).
    # A reference-order step = fmx_regex_batch_match(FMX_MATCH_REFERENCE): the list order is the answer, it is
    # delivered to the host.
    d_out = torch.empty(3 * cap, dtype=torch.int64, device=device)          # 24-byte records as three words
    d_per = torch.empty(max(k, 1), dtype=torch.int32, device=device)

    step_no = [0]

    def step(b=None):
        if b is None:
            b = batches[step_no[0] % ring]
            step_no[0] += 1
        if ref_mode:
            out, _ = b.match_raw(cap=cap, copy=False, **lim)
            n_res = out.size
            if use_dist:
                d_out[: 3 * n_res].copy_(torch.from_numpy(out.view(np.int64).reshape(-1)))
        else:
            n_res = b.match_dev(d_out.data_ptr(), cap, d_per.data_ptr(), max_steps=max_len)
        if use_dist:        # the path's one exchange: every rank receives every rank's result list (sizes, then payload)
            all_gather_varlen(d_out[: 3 * n_res])
        return n_res

    # what a call executes, averaged over the ring's batches (device counters; they differ by a per cent or two)
    hip.stats_reset()
    n_results_b = [int(step(b)) for b in reversed(batches)]          # (batch 0 last: its results are the ones compared below)
    torch.cuda.synchronize()
    s_all = hip.stats()
    s1 = dict(s_all)
    for f in ("rank_queries", "backward_steps", "frontier_requests", "frontier_elements", "frontier_queue_reads", "frontier_queue_writes",
              "frontier_results", "frontier_records", "ktab_lookups", "row_lookups"):
        s1[f] = s_all[f] / float(ring)
    n_results = n_results_b[-1]
    if ref_mode:
        out_res = batch.match_raw(cap=cap, **lim)[0]
    else:
        out_res = d_out[: 3 * n_results].cpu().numpy().view(RESULT_DTYPE)
    steps_per_call = s1["backward_steps"]
    ranks_per_step = 2 * steps_per_call
    n_results = sum(n_results_b) / float(ring)
    for _ in range(max(0, args.warmup)):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    kms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kms.append(hip.last_kernel_ms())
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = sum(kms) / len(kms)
    # ... and the same call on ONE batch replayed (what rounds 1-4 reported)
    for _ in range(2):
        step(batch)
    torch.cuda.synchronize()
    tr = time.perf_counter()
    for _ in range(max(args.steps, 5)):
        step(batch)
    torch.cuda.synchronize()
    replayed_ms = (time.perf_counter() - tr) / max(args.steps, 5) * 1e3
    host_delivered = None
    if not ref_mode:        # the same call with the results written into page-locked host memory (PCIe-inclusive)
        for _ in range(3):
            batch.match_raw(max_steps=max_len, cap=cap, copy=False)
        th = time.perf_counter()
        for _ in range(args.steps):
            batch.match_raw(max_steps=max_len, cap=cap, copy=False)
        th = (time.perf_counter() - th) / args.steps
        host_delivered = {"ms_per_step": th * 1e3, "value": ranks_per_step / th / 1e6, "unit": "M rank-queries/s",
                          "what": "fmx_regex_batch_match into page-locked host buffers (k_res_export over the link); this rank"}
    tot = torch.tensor([dt, float(ranks_per_step), float(n_results), kernel_ms], dtype=torch.float64, device=device)
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0].item())
        ranks_all, results_all, kernel_ms_max = float(sm[1].item()), float(sm[2].item()), float(mx[3].item())
    else:
        ranks_all, results_all, kernel_ms_max = float(ranks_per_step), float(n_results), kernel_ms
    if rank != 0:
        return None
    ksec = kernel_ms * 1e-3
    line_bytes = 64.0 if st["layout"] == 0 else 66.0
    if ref_mode:
        # ---- k_match_ref_wave: one regex per wave, the heap in LDS; what it asks of memory (device counters):
        # rank-dictionary lines of the steps made at push time, 16 B push record per pushed element, a 32-B slot
        # written per element pushed with a non-empty interval and read when it is popped, 32 B per result.
        # Priced with what reaches HBM (round 3 priced every access and came out ABOVE the PMC traffic): the push records are
        # the batch's own tables, read again and again from L2 -- counted once each (16 B x the batch's follow entries);
        # an element slot is written once (32 B) and read back from L2 when it is popped.
        info = batch.info()
        rec_once = 16.0 * min(s1["frontier_records"], info["follows"] + info["firsts"])
        alg_bytes = (s1["frontier_requests"] * line_bytes + rec_once + 32.0 * s1["frontier_queue_writes"] + 32.0 * s1["frontier_results"])
        achieved = alg_bytes / ksec / 1e9
        roof = {
            "bound": "hbm", "kernel": "k_match_ref_wave (ReTree._matchSA replayed, one regex per wave, heap keys in LDS, steps "
                                       "evaluated at push time)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_source": "see profiles/ (rocprofv3 --pmc of this workload)",
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes": "%d rank-line requests x %g B + %d distinct push records x 16 B (%d loads, L2-resident) + %d element "
                                 "slots written x 32 B (their %d reads hit L2) + %d results x 32 B"
                                 % (s1["frontier_requests"], line_bytes, int(rec_once / 16), s1["frontier_records"],
                                    s1["frontier_queue_writes"], s1["frontier_queue_reads"], s1["frontier_results"]),
            "kernel_ms": kernel_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
            "pops_per_launch": steps_per_call, "elements_stepped_at_push": int(s1["frontier_elements"]),
            "rank_queries_per_launch": ranks_per_step, "device_rank_queries_G_per_s": ranks_per_step / ksec / 1e9,
            "note": "latency-bound by construction: a regex's pops are serial in the reference's own order (the answer "
                    "depends on it); the launch lasts as long as the regexes that use all %d iterations "
                    "(kernel_ms / %d = time per pop on the critical path)" % (REF_LIMITS[1] - 1, REF_LIMITS[1] - 1),
            "us_per_pop_critical_path": kernel_ms * 1e3 / (REF_LIMITS[1] - 1),
        }
        traffic = pmc_traffic(args.workload.replace("tiny", ""), "k_match_ref_wave")
    else:
        # ---- roofline of the frontier kernels (k_frontier*), priced with what reaches HBM, from the device's own
        # counters: 64 B per rank-line request, 16 B per k-mer table entry, 24 B per work-queue entry read or appended,
        # 24 B per result written, and the state records ONCE each (32 B x the batch's states: the 6.9 M record loads
        # of a call hit L2 -- round 2 priced every load and came out above the PMC figure).
        n_states_batch = batch.info()["states"]
        rec_bytes = 32.0 * min(s1["frontier_records"], n_states_batch)
        alg_bytes = (s1["frontier_requests"] * line_bytes + 16.0 * s1["ktab_lookups"] + 8.0 * s1["row_lookups"] + rec_bytes +
                     24.0 * (s1["frontier_queue_reads"] + s1["frontier_queue_writes"]) + 24.0 * s1["frontier_results"])
        achieved = alg_bytes / ksec / 1e9
        traffic = pmc_traffic(args.workload.replace("tiny", ""), "k_frontier")
        all_req = s1["frontier_requests"] + s1["ktab_lookups"] + s1["row_lookups"] + s1["frontier_records"]
        roof = {
            "bound": "hbm", "kernel": "k_frontier* (all device work of one fmx_regex_batch_match: HIP events around reset, start "
                                       "elements, the launch chain, result grouping and export)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes": "%d rank-line requests x %g B + %d k-mer table entries x 16 B + %d row-table words x 8 B (one-row "
                                 "elements) + %d distinct state records x 32 B "
                                 "(%d loads, L2-resident) + (%d + %d) queue entries x 24 B + %d results x 24 B"
                                 % (s1["frontier_requests"], line_bytes, s1["ktab_lookups"], s1["row_lookups"], int(rec_bytes / 32),
                                    s1["frontier_records"], s1["frontier_queue_reads"], s1["frontier_queue_writes"],
                                    s1["frontier_results"]),
            "kernel_ms": kernel_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
            "requests_per_launch": int(all_req),
            "rank_line_requests": int(s1["frontier_requests"]), "ktab_lookups": int(s1["ktab_lookups"]),
            "state_records": int(s1["frontier_records"]), "ktab_k": int(s1["ktab_k"]), "row_table_lookups": int(s1["row_lookups"]),
            "rank_queries_per_launch": ranks_per_step,
            "requests_G_per_s": all_req / ksec / 1e9,
            "request_ceiling_G_per_s": REQUEST_CEILING_G_PER_S,
            "request_ceiling_source": "tools/ubench/mix.hip, profiles/r04_ubench_mix.txt",
            "request_frac": all_req / ksec / 1e9 / REQUEST_CEILING_G_PER_S,
            "device_rank_queries_G_per_s": ranks_per_step / ksec / 1e9,
            "note": ("the index is the BWT of a text: every frontier dies by itself, no match length limit is set"
                     if is_text else
                     "the synthetic BWT is an i.i.d. string, not the BWT of a text: a handful of starred classes sit on LF "
                     "cycles and never die, so the launch's critical path is %d dependent rounds cut by max_match_len "
                     "(truncated_at_max_len); on a real text a frontier dies by itself (workload c4text)" % max_len),
        }
    roof["traffic"] = traffic[0] if traffic else None
    if traffic and traffic[0]:
        # algorithmic bytes over the PMC traffic of the committed profile: below 1 where the kernel re-reads lines that left
        # the L2 (the frontier's state records), a few per cent above 1 where requests that are priced as HBM reads hit the L2
        # (the reference-order kernel: 3 % of its rank lines, and element slots overwritten before they are written back)
        roof["algorithmic_over_traffic"] = round(roof["algorithmic_bytes_per_launch"] / traffic[0], 3)
    roof["traffic_source"] = (traffic[1] if traffic and traffic[0] is None else
                              ("committed profile profiles/%s (separate rocprofv3 --pmc passes of this very source; not "
                               "measured in this run)" % traffic[1]) if traffic else "no PMC profile of this workload committed")
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
        "data": "synthetic" if not is_text else "synthetic text with natural repeats (words.txt's words drawn with replacement), its true BWT; regex literals are stretches of the text",
        "regexes_per_sec": world * k * args.steps / dt,
        "ring_batches": ring,
        "ms_per_step_is": "calls rotate through %d distinct resident regex batches (seeds differ): no call matches the batch the device "
                          "has just matched; the same call on ONE batch replayed: replayed_batch_ms" % ring,
        "replayed_batch_ms": replayed_ms,
        "regexes_per_sec_is": "a RESIDENT batch (compiled and on the device) matched again and again; a batch given as "
                              "strings and matched once runs at fresh_batch.regexes_per_s",
        "fresh_batch": fresh_batch,
        "host_delivered": host_delivered,
        "config": {
            "workload": ("%s: %d seeded regexes (<= 32 Glushkov positions) per GPU, 2^%d-byte sigma=%d %s "
                         "resident in HBM, " % (args.workload.upper(), k, log2n, len(C4_ALPHABET),
                                                "BWT of a text with natural repeats" if is_text else "synthetic BWT")) +
                        ("ReTree.matchSA in the reference's own pop order under its default limits (maxBranching %d, "
                         "maxIterations %d), result lists delivered to the host" % REF_LIMITS if ref_mode else
                         "SA-interval frontier expansion, results (grouped by regex, ordered) left in HBM"),
            "n": n, "sigma": len(C4_ALPHABET), "regexes_per_gpu": k,
            "results_per_call": results_all, "backward_steps_per_call": ranks_all / 2,
            "parallelism": "regexes sharded over %d GPU(s), index replicated" % world,
            "ranks_in_group": dist.get_world_size() if use_dist else 1,
            "index_gib": s1["index_bytes"] / 2**30, "index_build_ms": st["build_ms"], "tables_build_ms": s1["tables_build_ms"],
        },
        "roofline": roof,
    }
    if ref_mode:
        out["config"]["max_branching"], out["config"]["max_iterations"] = REF_LIMITS
    else:
        out["config"]["max_match_len"] = max_len
        out["config"]["truncated_at_max_len"] = bool(batch.truncated)
    if want_cpu:
        cores = effective_cores()
        sample = min(k, 20_000)
        orc, t_build = oracle_index(torch, bwt, eof, cores, rank)
        if orc is not None:
            out["cpu_baseline"] = cpu_baseline_regex(orc, t_build, n, res, trees, out_res, sample, max_len, ref_mode, cores,
                                                     rank, "the timed run's own index (n=2^%d)" % log2n)
            orc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(LITERAL) + sorted(REGEX) + sorted(TEXT))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive sub-record of the literal workloads")
    ap.add_argument("--no-rank-only", action="store_true", help="skip the tables-off sub-record of the literal workloads (executed rank work)")
    ap.add_argument("--ring", type=int, default=8,
                    help="literal workloads: distinct pattern batches the steps rotate through (1 = one batch replayed, rounds 1-4)")
    ap.add_argument("--regex-ring", type=int, default=4,
                    help="regex workloads: distinct resident batches the calls rotate through (1 = one batch replayed, rounds 1-4)")
    ap.add_argument("--exchange", default="packed", choices=["packed", "pairs"],
                    help="N > 1, literal workloads: the intervals travel as 8 bytes (default) or 16 bytes per pattern")
    ap.add_argument("--delivery", default="root", choices=["root", "all"],
                    help="N > 1, literal workloads: the gather is delivered to rank 0 only (default) or to every rank")
    ap.add_argument("--check-launch", action="store_true",
                    help="rendezvous + sharding + gather over gloo without any GPU work (tests the launcher)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.check_launch:
        sys.exit(check_launch(args, rank, world))

    # this image exports NCCL_DEBUG=VERSION, which makes RCCL print a banner on stdout; stdout carries
    # the one JSON line, so drop that setting (any other value the user chose is kept)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]
    import torch
    import torch.distributed as dist
    import findex_amd

    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # under torch.distributed.run (RANK/MASTER_* set) the RCCL path runs even for one rank
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        # the collective's stream at high priority: its kernels are few workgroups that must find room beside a search
        # kernel that fills the device (FMX_BENCH_NCCL_PRIO=0: default priority)
        opts = None
        if os.environ.get("FMX_BENCH_NCCL_PRIO", "1") != "0":
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                opts = None
        dist.init_process_group("nccl", device_id=device, pg_options=opts)
    stream = torch.cuda.current_stream().cuda_stream
    run = run_regex if (args.workload in REGEX or args.workload.startswith("c4text")) else run_literal
    out = run(args, torch, dist, findex_amd, rank, world, local, device, use_dist, stream)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
