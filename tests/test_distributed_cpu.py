"""The N > 1 host path (pattern sharding + the one gather of result intervals) on CPU: two gloo
ranks, the oracle standing in for the per-rank searcher so that no GPU is needed."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from findex_amd import distributed as D  # noqa: E402
from helpers import lf_walk_patterns, pack_patterns, synth_bwt  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleSearcher:
    """search_batch / regex matching through the CPU oracle (test stand-in for HipFMSearcher)."""

    def __init__(self, bwt, eof, counts):
        import oracle
        self.o = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
        self.n = self.o.n

    def search_batch(self, pat, off):
        off = np.asarray(off, dtype=np.uint64)
        base = int(off[0]) if off.size else 0
        sp, ep, _ = self.o.search_batch(np.asarray(pat, dtype=np.uint8)[base:], off - np.uint64(base))
        return sp, ep


def workload():
    bwt, eof, counts = synth_bwt(60_000, 97, 100, 11)
    rng = np.random.default_rng(4)
    s = OracleSearcher(bwt, eof, counts)
    pats = []
    for m in (0, 1, 3, 9, 17):
        pats += lf_walk_patterns(s.o, rng, 60, m, 0.2, alphabet=[97, 98, 99, 100])
    order = rng.permutation(len(pats))
    pats = [pats[i] for i in order]
    buf, off = pack_patterns(pats)
    return (bwt, eof, counts), buf, off


REGEXES = ["ab", "a[bc]d", "b(a|c)+d", "dd?a", "c[ab]*d"]     # no star over the whole alphabet: that never runs dry


def oracle_match(sa, res):
    from oracle import retree as R
    out = []
    for re in res:
        r, left, _ = sa.o.match_tables(R.ReTree(R.re2post(re)).tables(), 1 << 40, 0)
        assert left == 0
        out.append(sorted(r))
    return out


class FakeSearcher:
    """Answers made of the pattern's bytes: first byte 0 -> a miss (sp == ep), 1 -> an interval of 2^24 - 1 rows (the
    narrowest that must use the escape list), 2 -> 2^24 - 2 rows (the widest that fits the word), 3 -> 2^37 rows,
    else a few rows."""

    def search_batch(self, pat, off):
        k = off.size - 1
        sp = np.zeros(k, dtype=np.uint64)
        ep = np.zeros(k, dtype=np.uint64)
        for q in range(k):
            b = pat[int(off[q]):int(off[q + 1])]
            sp[q] = int(b.sum()) * 1000003 % (1 << 38) if b.size else 0
            c = int(b[0]) if b.size else 0
            ep[q] = int(sp[q]) + {0: 0, 1: (1 << 24) - 1, 2: (1 << 24) - 2, 3: 1 << 37}.get(c, c)
        return sp, ep


def fake_workload():
    rng = np.random.default_rng(12)
    pats = [bytes(rng.integers(0, 9, int(rng.integers(1, 6)), dtype=np.uint8)) for _ in range(40)]
    pats.insert(20, bytes(rng.integers(0, 9, 400, dtype=np.uint8)))      # one pattern holds most of the bytes
    off = np.zeros(len(pats) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in pats])
    return np.frombuffer(b"".join(pats), dtype=np.uint8).copy(), off


def _rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx, buf, off = workload()
        s = OracleSearcher(*idx)
        sp, ep = D.search_batch_sharded(s, buf, off)
        res = D.group_results(D.match_batch_sharded(s, REGEXES, match_fn=oracle_match, weights=[1, 3, 9, 2, 8]), len(REGEXES))
        cuts = D.shard_bounds(off, world)
        # the exchange the RCCL branch runs on device tensors, here on host tensors: a non-zero first regex id on
        # rank 1, lists of different lengths, and an empty list on rank 0 in the second round
        mine = np.array([(j, 5 + j + rank, 1000 * rank + j, 1000 * rank + j + 3) for j in range(2 + 3 * rank)], dtype=D.RESULT_DTYPE)
        ex1 = D.exchange_result_words(torch.from_numpy(mine.view(np.int64).reshape(-1).copy()), 7 * rank)
        none = mine[:0] if rank == 0 else mine
        ex2 = D.exchange_result_words(torch.from_numpy(none.view(np.int64).reshape(-1).copy()), 7 * rank)
        try:
            D.match_batch_sharded(s, REGEXES, match_fn=oracle_match, mode="reference")
            kw_refused = False
        except TypeError:
            kw_refused = True
        t = torch.arange(3 + 2 * rank, dtype=torch.int64) + 100 * rank
        parts = D.all_gather_varlen(t)
        # the pipelined gather bench.py uses: 5 batches through 2 slots, gather i overlapping batch i+1 -- in both
        # forms (packed 8 B / pairs 16 B per pattern; batch 3 holds a wide interval and a miss) and both deliveries
        seen = {}
        for form in ("packed", "pairs"):
            for delivery in ("all", "root"):
                g = D.IntervalGather(4, torch.device("cpu"), form=form, delivery=delivery, escape_cap=2)
                outs, got = [], []
                for i in range(5):
                    a, b = g.slot(i)
                    a.copy_(torch.arange(4) + 10 * i + 1000 * rank)
                    b.copy_(torch.arange(4) + 10 * i + 1000 * rank + 5)
                    if i == 3:
                        b[1] = a[1] + (1 << 33)          # wide: through the escape list
                        b[2] = a[2]                      # a miss: sp == ep
                    outs.append(g.launch(i))
                    if i >= 1:                      # batch i-1's gather has had a batch to finish; wait and read it
                        g.work[(i - 1) % g.depth].wait()
                        got.append(None if outs[i - 1] is None else [[x.tolist() for x in g.intervals(outs[i - 1], r)] for r in range(world)])
                g.finish()
                got.append(None if outs[4] is None else [[x.tolist() for x in g.intervals(outs[4], r)] for r in range(world)])
                seen[form + "/" + delivery] = (got, g.payload_bytes)
        # an overfull escape list: rank 1's batch alone holds three wide intervals where the list takes two --
        # EVERY rank learns of it from overflow(i) (until round 4 only the root did, by an exception while decoding), and
        # the slot is repeated in the 16-byte form, which has no such limit
        g = D.IntervalGather(4, torch.device("cpu"), form="packed", delivery="root", escape_cap=2)
        a, b = g.slot(0)
        a.copy_(torch.arange(4) + 1000 * rank)
        b.copy_(a + 5)
        if rank == 1:
            b[:3] = a[:3] + (1 << 30)
        out0 = g.launch(0)
        over0 = g.overflow(0)
        a1, b1 = g.slot(1)
        a1.copy_(a)
        b1.copy_(a + 7)                              # nothing wide: no overflow on any rank
        g.launch(1)
        over1 = g.overflow(1)
        g.finish()
        root_raises = None
        if out0 is not None:
            try:
                g.intervals(out0, 1)
                root_raises = False
            except OverflowError:
                root_raises = True
        g2 = D.IntervalGather(4, torch.device("cpu"), form="pairs", delivery="root")
        a2, b2 = g2.slot(0)
        a2.copy_(a)
        b2.copy_(b)
        out2 = g2.launch(0)
        g2.finish()
        redo = None if out2 is None else [[x.tolist() for x in g2.intervals(out2, r)] for r in range(world)]
        assert g2.overflow(0) is False
        seen["overflow"] = (over0, over1, root_raises, redo)
        # the one-call form: every (form, delivery), a searcher whose answers include wide intervals and misses, and a
        # byte distribution that leaves the middle rank of three without a pattern
        fake = FakeSearcher()
        fbuf, foff = fake_workload()
        sharded = {}
        for form in ("packed", "pairs"):
            for delivery in ("all", "root"):
                r_ = D.search_batch_sharded(fake, fbuf, foff, form=form, delivery=delivery, root=world - 1)
                sharded[form + "/" + delivery] = None if r_ is None else (r_[0].tolist(), r_[1].tolist())
        fcuts = D.shard_bounds(foff, world)
        q.put((rank, sp, ep, res, cuts, [p.tolist() for p in parts], seen, ex1.tolist(), ex2.tolist(), kw_refused, sharded, fcuts))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_match_single_process(world):
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    idx, buf, off = workload()
    s = OracleSearcher(*idx)
    wsp, wep = s.search_batch(buf, off)
    wres = oracle_match(s, REGEXES)
    want1 = [(j + 7 * r, 5 + j + r, 1000 * r + j, 1000 * r + j + 3) for r in range(world) for j in range(2 + 3 * r)]
    want2 = [w for w in want1 if w[2] >= 1000]                 # rank 0 contributed nothing in the second round
    fbuf, foff = fake_workload()
    fsp, fep = FakeSearcher().search_batch(fbuf, foff)
    assert int(((fep - fsp) >= np.uint64(0xFFFFFF)).sum()) >= 3 and int((fep == fsp).sum()) >= 3
    for rank, sp, ep, res, cuts, parts, seen, ex1, ex2, kw_refused, sharded, fcuts in got:
        assert [tuple(x) for x in ex1] == want1 and [tuple(x) for x in ex2] == want2
        assert kw_refused
        over0, over1, root_raises, redo = seen.pop("overflow")
        assert over0 is True and over1 is False, "every rank must learn of rank 1's overfull escape list (rank %d: %s %s)" % (rank, over0, over1)
        assert root_raises == (True if rank == 0 else None)
        if rank == 0:                                # the repeat in the 16-byte form carries all three wide intervals
            for r in range(world):
                a = [j + 1000 * r for j in range(4)]
                b = [x + 5 for x in a]
                if r == 1:
                    b[:3] = [x + (1 << 30) for x in a[:3]]
                assert redo[r] == [a, b], r
        else:
            assert redo is None
        for key, (batches, payload) in seen.items():
            form, delivery = key.split("/")
            assert payload == (8 * (4 + 1 + 2 * 2) if form == "packed" else 16 * 4)
            for i, batch in enumerate(batches):    # every receiving rank sees every rank's rows of batch i
                if delivery == "root" and rank != 0:
                    assert batch is None
                    continue
                for r in range(world):
                    a = [j + 10 * i + 1000 * r for j in range(4)]
                    b = [x + 5 for x in a]
                    if i == 3:
                        b[1], b[2] = a[1] + (1 << 33), a[2]
                    assert batch[r] == [a, b], (key, i, r)
        for key, val in sharded.items():
            if key.endswith("/root") and rank != world - 1:
                assert val is None
            else:
                assert val == (fsp.tolist(), fep.tolist()), key
        if world == 3:
            assert fcuts[1] == fcuts[2] or fcuts[2] - fcuts[1] <= 1      # the heavy pattern leaves a rank (nearly) empty
        assert np.array_equal(sp, wsp) and np.array_equal(ep, wep)
        assert [sorted(r) for r in res] == wres
        assert cuts[0] == 0 and cuts[-1] == off.size - 1 and cuts == sorted(cuts)
        assert parts == [[100 * r + j for j in range(3 + 2 * r)] for r in range(world)]
    if world == 2:      # byte balance of the shards
        cuts = got[0][4]
        b0 = int(off[cuts[1]] - off[cuts[0]])
        b1 = int(off[cuts[2]] - off[cuts[1]])
        assert abs(b0 - b1) <= 20


def test_packed_interval_form_round_trips():
    """pack_intervals_np / unpack_intervals_np (the host twin of fmx_pack_intervals_dev): misses, the two widths either
    side of the escape threshold, rows near 2^38, an escape list that is exactly full, one entry too short, and empty."""
    rng = np.random.default_rng(1)
    sp = rng.integers(0, 1 << 38, 1000, dtype=np.int64).astype(np.uint64)
    w = rng.integers(0, 50, 1000, dtype=np.int64).astype(np.uint64)
    w[::97] = np.uint64(0xFFFFFF)
    w[5::97] = np.uint64(0xFFFFFE)
    w[9::97] = np.uint64(1 << 37)
    w[3::50] = 0
    ep = sp + w
    n_wide = int((w >= np.uint64(0xFFFFFF)).sum())
    pk = D.pack_intervals_np(sp, ep, n_wide)
    assert pk.size == D.packed_words(1000, n_wide) and int(pk[1000]) == n_wide
    a, b = D.unpack_intervals_np(pk, 1000, n_wide)
    assert np.array_equal(a, sp) and np.array_equal(b, ep)
    short = D.pack_intervals_np(sp, ep, n_wide - 1)
    assert int(short[1000]) == n_wide
    with pytest.raises(OverflowError):
        D.unpack_intervals_np(short, 1000, n_wide - 1)
    e = D.pack_intervals_np(sp[:0], ep[:0], 4)
    assert e.size == 9 and D.unpack_intervals_np(e, 0, 4)[0].size == 0


def test_work_bounds_balance_by_weight():
    assert D.work_bounds([], 3) == [0, 0, 0, 0]
    assert D.work_bounds([1, 1, 1, 1], 2) == [0, 2, 4]
    c = D.work_bounds([100, 1, 1, 1, 1, 96], 2)                  # one heavy regex at each end
    assert c == [0, 1, 6]
    c = D.work_bounds(np.ones(10), 4)
    assert c[0] == 0 and c[-1] == 10 and c == sorted(c)
    r = np.array([(2, 3, 5, 9), (0, 1, 2, 3), (2, 1, 4, 6)], dtype=D.RESULT_DTYPE)
    assert D.group_results(r, 3) == [[(1, 2, 3)], [], [(3, 5, 9), (1, 4, 6)]]
    assert np.array_equal(D.gather_results(r), r)                # no process group: identity


def test_shard_bounds_edge_cases():
    assert D.shard_bounds(np.array([0], dtype=np.uint64), 4) == [0, 0, 0, 0, 0]
    assert D.shard_bounds(np.zeros(9, dtype=np.uint64), 4) == [0, 2, 4, 6, 8]          # all empty patterns
    off = np.array([0, 100, 101, 102, 103], dtype=np.uint64)                             # one huge pattern
    c = D.shard_bounds(off, 2)
    assert c[0] == 0 and c[-1] == 4 and c == sorted(c)
    off = np.arange(0, 8 * 33, 8, dtype=np.uint64)
    assert D.shard_bounds(off, 8) == [0, 4, 8, 12, 16, 20, 24, 28, 32]


def test_single_process_path_needs_no_process_group():
    idx, buf, off = workload()
    s = OracleSearcher(*idx)
    sp, ep = D.search_batch_sharded(s, buf, off)
    wsp, wep = s.search_batch(buf, off)
    assert np.array_equal(sp, wsp) and np.array_equal(ep, wep)
