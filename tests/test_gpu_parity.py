"""Parity tests proper: the HIP path, called through the C ABI (ctypes -> libfmx.so), against
the CPU oracle on the same inputs, bit for bit.  Needs a real MI355X:  pytest -m gpu
"""
import os

import numpy as np
import pytest

import findex_amd
import oracle
from oracle import retree as R
from helpers import bwt_of_text, lf_walk_patterns, pack_patterns, synth_bwt

pytestmark = pytest.mark.gpu

FIXTURES = [("test1024.cmp", False), ("test2048.cmp", False), ("test2048-2.cmp", False), ("test3072.cmp", False),
            ("test.cmp", False), ("test-part.cmp", False), ("words", True)]


def pair_from_files(testdata, name, be):
    p = os.path.join(testdata, name + ".bwt")
    return findex_amd.HipFMSearcher(p, bigEndian=be), oracle.NaiveFMSearcher(p, bigEndian=be)


def pair_from_mem(bwt, eof, counts):
    return findex_amd.HipFMSearcher.from_mem(bwt, eof, counts), oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)


def jump_row_bytes(layout="onehot"):
    """Bytes per row of the row jump table these small indexes get: 16, or 32 when pairs of entries are forced on
    (FMX_JUMP_PAIRS=1; by default only indexes of 2^30 rows and more get pairs; never the bytes layout)."""
    return 32 if (os.environ.get("FMX_JUMP_PAIRS") == "1" and layout != "bytes") else 16


def check_occ(hip, orc, rng, k, symbols):
    c = rng.choice(np.asarray(symbols, dtype=np.uint8), size=k)
    i = rng.integers(-1, orc.n + 2, size=k, dtype=np.int64)
    i[:4] = [-1, 0, orc.n - 1, orc.n + 5]
    want = orc.occ_batch(c, i)
    got = hip.occ_batch(c, i)
    assert np.array_equal(got.astype(np.int64), want)


def check_search(hip, orc, pats):
    buf, off = pack_patterns(pats)
    wsp, wep, wsteps = orc.search_batch(buf, off)
    hip.stats_reset()
    gsp, gep = hip.search_batch(buf, off)
    assert np.array_equal(gsp, wsp) and np.array_equal(gep, wep)
    st = hip.stats()
    assert st["backward_steps"] == int(wsteps.sum()) and st["rank_queries"] == 2 * int(wsteps.sum())
    return int((wsp < wep).sum())


def reference_loop_by_prev_range(hip, pats2d):
    """SuffixAlgo.search (findex.scala:15-31) restated over fmx_prev_range_batch -- K4, one plain getPrevRange per
    pattern and step, no derived table -- for k equal-length patterns: (sp, ep, steps) of the reference's loop.  K4 is
    verified against the oracle up to 2^32 rows and against brute-force prefix counts above, so this is the at-size
    check for what the table-served search kernels return where the oracle's 32-bit lists cannot follow."""
    k, m = pats2d.shape
    sp = np.zeros(k, dtype=np.uint64)
    ep = np.full(k, hip.n, dtype=np.uint64)
    steps = np.zeros(k, dtype=np.int64)
    for i in range(m - 1, -1, -1):
        act = np.nonzero(sp < ep)[0]
        if act.size == 0:
            break
        a, b = hip.prev_range_batch(sp[act], ep[act], pats2d[act, i])
        sp[act], ep[act] = a, b
        steps[act] += 1
    return sp, ep, steps


# ---------------------------------------------------------------- reference fixtures
@pytest.mark.parametrize("name,be", FIXTURES)
def test_fixture_files_occ_search_step(testdata, name, be):
    hip, orc = pair_from_files(testdata, name, be)
    assert hip.n == orc.n and hip.eof == orc.eof
    assert [hip.cf(c) for c in range(256)] == [orc.cf(c) for c in range(256)]
    rng = np.random.default_rng(abs(hash(name)) % 2**32)
    present = [c for c in range(256) if orc.occ(c, orc.n - 1) > 0]
    check_occ(hip, orc, rng, 4000, present + [0, 1, 255, 200])
    pats = lf_walk_patterns(orc, rng, 300, 8, 0.1, alphabet=present)
    pats += lf_walk_patterns(orc, rng, 100, 40, 0.3, alphabet=present)
    pats += [b"", b"a", bytes([present[0]]), b"\x00", b"zz\x00", b"\x80\xff", bytes(rng.integers(1, 256, 17, dtype=np.uint8))]
    hits = check_search(hip, orc, pats)
    assert hits > 200
    # single steps on random valid intervals
    sp = rng.integers(0, orc.n + 1, size=3000).astype(np.uint64)
    ep = rng.integers(0, orc.n + 1, size=3000).astype(np.uint64)
    sp, ep = np.minimum(sp, ep), np.maximum(sp, ep)
    c = rng.choice(np.asarray(present + [0, 255], dtype=np.uint8), size=3000)
    w1, w2 = orc.prev_range_batch(sp, ep, c)
    g1, g2 = hip.prev_range_batch(sp, ep, c)
    assert np.array_equal(g1, w1) and np.array_equal(g2, w2)
    for a, b in ((0, orc.n), (5, 700), (int(sp[0]), int(ep[0]))):
        assert hip.getIntervalPrevRange(a, b, 0, 255) == orc.getIntervalPrevRange(a, b, 0, 255)
        assert hip.getIntervalPrevRange(a, b, 97, 122) == orc.getIntervalPrevRange(a, b, 97, 122)


def test_words_c1_thousand_8char_literals(testdata):
    """BASELINE config C1: 1k 8-char literals cut from words lines, reversed (the index is over
    the reversed text); counts cross-checked against the text itself."""
    hip, orc = pair_from_files(testdata, "words", True)
    txt = open(os.path.join(testdata, "words.txt"), "rb").read()
    rng = np.random.default_rng(1)
    lines = [w for w in txt.split(b"\r\n") if len(w) >= 8]
    pats = [bytes(lines[i][:8][::-1]) for i in rng.integers(0, len(lines), 1000)]
    buf, off = pack_patterns(pats)
    sp, ep = hip.search_batch(buf, off)
    wsp, wep, _ = orc.search_batch(buf, off)
    assert np.array_equal(sp, wsp) and np.array_equal(ep, wep) and (sp < ep).all()
    for j in range(0, 1000, 50):
        p = pats[j][::-1]
        want, k = 0, txt.find(p)
        while k >= 0:
            want += 1
            k = txt.find(p, k + 1)
        assert int(ep[j] - sp[j]) == want
    assert hip.search(b"aardvark"[::-1]) == (1044943, 1044945) and hip.search(b"aardvark") is None


def test_reference_kats_through_the_product(testdata):
    """The reference's own known answers (T/Indexer.scala:247-351,1076-1124, T/REParser.scala:236-291),
    asked of the HIP path."""
    hip = findex_amd.HipFMSearcher.from_mem(*bwt_of_text(b"abracadabra"))
    assert hip.cf(0) == 0 and hip.cf(ord("a")) == 1 and hip.cf(ord("b")) == 6
    rows = {0: [0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1], ord("a"): [1, 1, 1, 1, 1, 1, 2, 3, 4, 5, 5, 5],
            ord("b"): [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2], ord("c"): [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1],
            ord("d"): [0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], ord("r"): [0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2],
            ord("x"): [0] * 12}
    for c, want in rows.items():
        assert [hip.occ(c, i) for i in range(12)] == want
    assert hip.search(b"bra") == (6, 8)
    assert hip.getPrevI(6) == 2 and hip.getNextI(6) == 10 and hip.getNextI(10) == 1
    assert hip.psi_batch(np.arange(12, dtype=np.uint64)).tolist() == [3, 0, 6, 7, 8, 9, 10, 11, 5, 2, 1, 4]
    hip = findex_amd.HipFMSearcher.from_mem(*bwt_of_text(b"mmabcacadabbbca"[::-1]))
    assert hip.occ(ord("b"), 6) == 3
    assert hip.getPrevRange(0, 16, ord("a")) == (1, 6) and hip.getPrevRange(1, 6, ord("b")) == (6, 8)
    # file backed, little endian golden
    hip = findex_amd.HipFMSearcher(os.path.join(testdata, "test1024.cmp.bwt"), bigEndian=False)
    eof = hip.eof
    assert eof == 462 and hip.bwt_read(0) == ord("u") and hip.bwt_read(eof) == 0
    assert hip.getPrevI(eof) == 0 and hip.getNextI(eof) == 517 and hip.getPrevI(1) == 48 and hip.getPrevI(48) == 649
    assert hip.nextSubstr(1, 3) == b"haa" and hip.prevSubstr(1, 5) == b"bqxxa"
    assert hip.prevSubstr(eof, 5) == b"\0uexm" and hip.prevSubstr(hip.getPrevI(eof), 4) == b"uexm"
    assert hip.nextSubstr(eof, 100) == (
        b"ajrtzbeqwbxdfpwjflmmsseewuudgfbtzqenjqafwzcnfanycigwsflfvxojxpqhhzekjdkhgsptqveavquuoqujbezdkarayoml")
    # small2.txt
    hip = findex_amd.HipFMSearcher.from_mem(*bwt_of_text(b"ippisissim"[::-1]))
    assert [hip.getNextI(i) for i in (0, 5, 4, 10, 9)] == [5, 4, 10, 9, 3]
    assert [hip.getPrevI(i) for i in (3, 9, 10, 4, 5, 0)] == [9, 10, 4, 5, 0, 1]


def test_naive_bwt_searcher_through_the_product(golden):
    """NaiveBWTSearcher (findex.scala:459-506) through the C ABI -- the 22 known answers of
    T/Indexer.scala:703-725,737-742 (the 0xff symbol and the EOF hole among them), then every (c, key) of random
    blocks against the oracle's restatement, the reference's last-slot rule (:500-502) included."""
    import json
    from oracle.naive_bwt import NaiveBWTSearcher
    kat = json.load(open(os.path.join(golden, "naive_bwt_searcher_kat.json")))
    n_kat = 0
    for name in ("case1", "case2"):
        k = kat[name]
        bwt = np.array(k["bwt"], dtype=np.int64).astype(np.uint8)
        hip = findex_amd.HipFMSearcher.from_block(bwt, k["bs"], k["rk0"])
        assert hip.n == bwt.size and hip.eof == k["rk0"]
        for c, key, want in k["occ"]:
            assert hip.occ(c & 0xFF, key) == want, (name, c, key)
            n_kat += 1
        for c in (97, 100, 106, 122, 255):
            assert hip.cf(c) == k["bs"][c]
    assert n_kat == 22
    rng = np.random.default_rng(5)
    for trial in range(12):
        n = int(rng.integers(1, 700))
        bwt = rng.integers(1, 7 if trial % 2 else 256, n).astype(np.uint8)
        rk0 = int(rng.integers(0, n))
        if trial == 3 and n > 3:          # the last-slot rule: the first byte occurs nowhere else
            bwt[0] = 250
            bwt[1:][bwt[1:] == 250] = 1
            rk0 = n - 1
        cnt = np.bincount(bwt, minlength=256)
        cnt[bwt[rk0]] -= 1
        cnt[bwt[rk0]] += 1                # the bucket starts come from the block's text: the skipped row's byte is in it
        bs = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
        hip = findex_amd.HipFMSearcher.from_block(bwt, bs, rk0)
        ref = NaiveBWTSearcher(bwt, bs, rk0)
        cs = np.repeat(np.arange(256, dtype=np.uint8), 6)
        keys = rng.integers(-1, n, cs.size).astype(np.int64)
        got = hip.occ_batch(cs, keys)
        want = [ref.occ(int(c), int(key)) if key >= 0 else 0 for c, key in zip(cs, keys)]
        assert got.tolist() == want, trial
        for c in {int(bwt[0]), int(bwt[rk0]), 1, 255}:     # full columns for the interesting symbols
            ks = np.arange(n, dtype=np.int64)
            assert hip.occ_batch(np.full(n, c, dtype=np.uint8), ks).tolist() == [ref.occ(c, int(key)) for key in ks], (trial, c)


def test_calc_gaps_rank_chain_on_the_host(golden, testdata):
    """SURVEY 8f-4, BWTMerger2.calcGaps' rank loop (bwtmerger.scala:981-1023) over the product's dictionary on the
    host: fmx_occ_host gives NaiveBWTSearcher.occ's answers (the 22 known ones, random blocks with the last-slot
    rule, a file-backed index), and fmx_calc_gaps_chain reproduces the loop -- curRank = bucketStarts(c) +
    occ(c, curRank - 1), the bump past rklst for c == lastChar, a stop where the reference consults its KMP buffer --
    against a plain restatement of the loop over the oracle's NaiveBWTSearcher."""
    import json
    from oracle.naive_bwt import NaiveBWTSearcher
    kat = json.load(open(os.path.join(golden, "naive_bwt_searcher_kat.json")))
    for name in ("case1", "case2"):
        k = kat[name]
        bwt = np.array(k["bwt"], dtype=np.int64).astype(np.uint8)
        hip = findex_amd.HipFMSearcher.from_block(bwt, k["bs"], k["rk0"])
        for c, key, want in k["occ"]:
            assert hip.occ_host(c & 0xFF, key) == want, (name, c, key)
    rng = np.random.default_rng(15)
    for trial in range(8):
        n = int(rng.integers(2, 5000))
        sigma_hi = 7 if trial % 2 else 256
        bwt = rng.integers(1, sigma_hi, n).astype(np.uint8)
        rk0 = int(rng.integers(0, n))
        if trial == 3:
            bwt[0] = 250
            bwt[1:][bwt[1:] == 250] = 1
            rk0 = n - 1
        cnt = np.bincount(bwt, minlength=256)
        bs = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
        hip = findex_amd.HipFMSearcher.from_block(bwt, bs, rk0)
        ref = NaiveBWTSearcher(bwt, bs, rk0)
        for c in {int(bwt[0]), int(bwt[rk0]), 1, 2, 255}:
            ks = np.concatenate([[-1], rng.integers(0, n, 300), [n - 1, n + 3]])
            assert [hip.occ_host(c, int(key)) for key in ks] == [ref.occ(c, int(min(key, n - 1))) if key >= 0 else 0 for key in ks], (trial, c)
        # the loop itself over an "older text" of 3000 bytes
        text = rng.integers(1, sigma_hi, 3000).astype(np.uint8)
        last_char, rklst = int(text[7]), int(rng.integers(0, n))
        c0 = int(text[0])
        cur = int(bs[c0])                                # :985-987
        want, stops = [cur], []
        for j in range(1, text.size):
            ch = int(text[j])
            cur = int(bs[ch]) if cur == 0 else int(bs[ch]) + ref.occ(ch, cur - 1)      # :999-1001
            if ch == last_char:                              # :1003-1013
                if cur == rklst:
                    stops.append(j)
                    cur += j & 1                             # the "KMP buffer's" verdict, made up: the caller's business
                elif cur > rklst:
                    cur += 1
            want.append(cur)
        # drive the chain the way a caller does: byte 0 by hand, then stretches between the stops
        got = [int(bs[c0])]
        at, cur = 1, got[0]
        while at < text.size:
            ranks, done = hip.calc_gaps_chain(text[at:], rank0=cur, last_char=last_char, rklst=rklst)
            got += [int(x) for x in ranks[:done]]
            at += done
            if done:
                cur = got[-1]
            if at < text.size:                               # stopped at a rank == rklst: decide, go on
                assert int(ranks[done]) == rklst and at in stops
                cur = rklst + (at & 1)
                got.append(cur)
                at += 1
        assert got == want, trial
    # any handle serves it: a file-backed index gives the same occ on the host as on the device
    hip, orc = pair_from_files(testdata, "test1024.cmp", False)
    cs = rng.integers(0, 256, 2000).astype(np.uint8)
    ks = rng.integers(-1, hip.n + 2, 2000).astype(np.int64)
    dev = hip.occ_batch(cs, ks)
    assert [hip.occ_host(int(c), int(k)) for c, k in zip(cs, ks)] == dev.tolist()


def test_extract_is_next_and_prev_substr(testdata):
    hip, orc = pair_from_files(testdata, "test1024.cmp", False)
    for row in (0, 1, 48, 462, 517, hip.n - 1):
        for ln in (0, 1, 7, 40):
            assert hip.extract(row, ln, +1) == hip.nextSubstr(row, ln)
            assert hip.extract(row, ln, -1) == hip.prevSubstr(row, ln)
    with pytest.raises(findex_amd.FmxError):
        hip.extract(0, 3, 0)
    rows = np.arange(hip.n, dtype=np.uint64)
    for ln in (0, 1, 9):
        assert hip.nextSubstr_batch(rows, ln) == [orc.nextSubstr(int(r), ln) for r in rows]


@pytest.mark.parametrize("name,be", [("test1024.cmp", False), ("test.cmp", False), ("words", True)])
def test_walks_all_rows(testdata, name, be):
    """Psi == the reference's .fm payload and LF == getPrevI, for every row (a sample on words)."""
    hip, orc = pair_from_files(testdata, name, be)
    rows = np.arange(orc.n, dtype=np.uint64)
    if orc.n > 20000:
        rows = np.random.default_rng(3).integers(0, orc.n, 20000).astype(np.uint64)
        rows[:3] = [0, orc.eof, orc.n - 1]
    fm = orc.fm()
    assert np.array_equal(hip.psi_batch(rows), fm[rows.astype(np.int64)].astype(np.uint64))
    b, end = hip.lf_walk_batch(rows, 6)
    for j in range(min(300, rows.size)):
        assert bytes(b[j]) == orc.prevSubstr(int(rows[j]), 6)
    want_end = np.array([orc.getPrevI(int(r)) for r in rows[:2000]], dtype=np.uint64)
    _, e1 = hip.lf_walk_batch(rows[:2000], 1, want_bytes=False)
    assert np.array_equal(e1, want_end)
    for sp, ln in ((1, 3), (int(orc.eof), 50), (int(orc.n - 1), 7), (0, 4)):
        assert hip.nextSubstr(sp, ln) == orc.nextSubstr(sp, ln)
        assert hip.prevSubstr(sp, ln) == orc.prevSubstr(sp, ln)


@pytest.mark.parametrize("name,be", [("test1024.cmp", False), ("test3072.cmp", False), ("words", True)])
def test_fm_file_is_byte_identical(testdata, name, be, tmp_path):
    """T/Indexer.scala:841-900 BWTCreatorTest through the product: the .fm written from the device
    structure equals FMCreator's (oracle restatement) byte for byte: header 0x04 + int64 BE size,
    then 4-byte BE entries == bwt2occ(bwt with eof := 0)."""
    hip, orc = pair_from_files(testdata, name, be)
    a, b = str(tmp_path / "hip.fm"), str(tmp_path / "orc.fm")
    hip.write_fm(a)
    orc.write_fm(b)
    assert open(a, "rb").read() == open(b, "rb").read()
    raw = np.fromfile(a, dtype=np.uint8)
    assert raw[0] == 4 and raw.size == 9 + 4 * hip.n


def test_open_streams_large_files(tmp_path):
    """fmx_open streams the .bwt payload to the device in 32 MiB chunks through two pinned buffers; a 75 MB file
    (3 chunks, the last one partial) must give the same index as the same bytes handed over in memory."""
    n = 75_000_003
    bwt, eof, counts = synth_bwt(n, 1, 4, 21)
    base = tmp_path / "big"
    with open(str(base) + ".bwt", "wb") as f:
        f.write(np.array([n, eof], dtype="<i8").tobytes())
        f.write(bwt.tobytes())
    with open(str(base) + ".aux", "wb") as f:
        f.write(counts.astype("<i8").tobytes())
    a = findex_amd.HipFMSearcher(str(base) + ".bwt", bigEndian=False)
    b = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    assert a.n == b.n == n and a.eof == b.eof == eof
    assert [a.cf(c) for c in range(256)] == [b.cf(c) for c in range(256)]
    rng = np.random.default_rng(3)
    c = rng.integers(0, 6, size=200_000).astype(np.uint8)
    i = rng.integers(-1, n + 1, size=200_000, dtype=np.int64)
    i[:6] = [0, (32 << 20) - 1, 32 << 20, (64 << 20) - 1, 64 << 20, n - 1]        # chunk seams
    assert np.array_equal(a.occ_batch(c, i), b.occ_batch(c, i))
    # brute force at the seams
    for q in range(6):
        assert int(a.occ_batch(c[q:q + 1], i[q:q + 1])[0]) == int(
            (np.where(np.arange(int(i[q]) + 1) == eof, 0, bwt[: int(i[q]) + 1]) == c[q]).sum())


# ---------------------------------------------------------------- synthetic indexes
@pytest.mark.parametrize("n,lo,hi,seed", [
    (1, 1, 1, 1), (2, 1, 2, 2), (447, 1, 4, 3), (448, 1, 4, 4), (449, 1, 4, 5), (896, 1, 4, 6), (897, 65, 68, 7),
    (959, 1, 4, 3), (960, 1, 4, 4), (961, 1, 4, 5), (1344, 1, 3, 12),     # block edges of the one-hot layout (448)
    (100_003, 1, 4, 8),            # DNA-like sigma=4 (C2 shape, small)
    (300_007, 1, 128, 9),          # sigma=128 incl. byte 0x80 (C3 shape, small)
    (200_000, 1, 255, 10),         # every byte value
    (50_000, 200, 255, 11),        # only bytes >= 0x80: the reference would throw, the product must not
])
def test_synthetic_parity(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    for eof in sorted({0, n // 3, n - 1}):
        bwt, eof, counts = synth_bwt(n, lo, hi, seed, eof=eof)
        hip, orc = pair_from_mem(bwt, eof, counts)
        syms = list(range(lo, hi + 1))
        check_occ(hip, orc, rng, 3000, syms + [0, 255])
        m = 1 if n < 10 else (16 if hi - lo < 8 else 6)
        pats = lf_walk_patterns(orc, rng, 400, m, 0.1, alphabet=syms) + [b"", bytes([lo]), b"\x00"]
        check_search(hip, orc, pats)
        if n > 1000:
            pats = lf_walk_patterns(orc, rng, 200, 33, 0.2, alphabet=syms)
            check_search(hip, orc, pats)


@pytest.mark.parametrize("layout", ["onehot", "bytes"])
def test_randomized_small_indexes(layout):
    """Many small random indexes (n around the 448 / 128 block edges and beyond, random alphabet ranges, random
    eof) and random operands -- hits, misses, empty patterns, absent symbols, the EOF symbol 0, keys outside
    [0, n) -- through search, occ and single steps, against the oracle.  FMX_TEST_SEED picks other draws."""
    seed = int(os.environ.get("FMX_TEST_SEED", "5"))
    rng = np.random.default_rng(seed)
    findex_amd.set_layout(layout)
    try:
        for trial in range(24):
            n = int(rng.choice([1, 2, 3, 63, 127, 128, 129, 447, 448, 449, 895, 897, 5000, 40_000]) + rng.integers(0, 3))
            lo = int(rng.integers(1, 250))
            hi = int(min(255, lo + rng.integers(0, 40)))
            eof = int(rng.integers(0, n))
            bwt, eof, counts = synth_bwt(n, lo, hi, seed * 1000 + trial, eof=eof)
            hip, orc = pair_from_mem(bwt, eof, counts)
            symbols = list(range(max(0, lo - 2), min(255, hi + 2) + 1)) + [0, 255]
            check_occ(hip, orc, rng, 300, symbols)
            pats = []
            for m in (0, 1, 2, 5, 9, 33):
                pats += lf_walk_patterns(orc, rng, 12, m, 0.3, alphabet=symbols)
            pats += [bytes(rng.choice(np.asarray(symbols, dtype=np.uint8), size=int(rng.integers(1, 12))).tolist())
                     for _ in range(40)]
            check_search(hip, orc, pats)
            a = rng.integers(0, n + 1, size=200)
            b = rng.integers(0, n + 1, size=200)
            sp, ep = np.minimum(a, b).astype(np.uint64), np.maximum(a, b).astype(np.uint64)
            c = rng.choice(np.asarray(symbols, dtype=np.uint8), size=200)
            gsp, gep = hip.prev_range_batch(sp, ep, c)
            for q in range(200):
                w = orc.getPrevRange(int(sp[q]), int(ep[q]), int(c[q]))
                g = (int(gsp[q]), int(gep[q])) if gsp[q] < gep[q] else None
                assert g == w, (trial, n, q)
    finally:
        findex_amd.set_layout("auto")


def test_many_patterns_ragged_lengths():
    """More patterns than resident octets, lengths 0..70, so octets chain through several patterns."""
    bwt, eof, counts = synth_bwt(400_000, 1, 4, 77)
    hip, orc = pair_from_mem(bwt, eof, counts)
    rng = np.random.default_rng(5)
    pats = []
    for m in (0, 1, 2, 7, 19, 32, 70):
        pats += lf_walk_patterns(orc, rng, 3000 if m else 50, m, 0.15, alphabet=[1, 2, 3, 4])
    order = rng.permutation(len(pats))
    pats = [pats[i] for i in order]
    # ~130k patterns: replicate to exceed the 65536 resident octets
    pats = pats * 8
    hits = check_search(hip, orc, pats)
    assert hits > 50_000


def test_kmer_table_on_and_off_agree():
    """The k-mer jump table (fmx_ktab.hip) answers a search's first K steps with one lookup; with it and without it
    every (sp, ep) -- misses with the reference's values at the failing step included -- and the executed-step
    counter must equal the oracle's.  Patterns shorter than K, patterns with bytes outside the alphabet and a
    pattern inside the buffer's first 16 bytes take the stepwise start."""
    bwt, eof, counts = synth_bwt(300_000, 1, 5, 19)
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(4)
    pats = [b"\x01\x02", b"\x03"]                                   # first in the buffer: end < 16
    for m in (1, 2, 5, 6, 7, 9, 13, 30):
        pats += lf_walk_patterns(orc, rng, 400, m, 0.3, alphabet=[1, 2, 3, 4, 5])
    pats += [bytes([1, 2, 9, 1, 2, 3, 4, 5, 1]), bytes([0, 1, 2, 3, 4, 5, 1, 2]), b""] * 20      # foreign bytes inside / outside the k-mer
    pats = pats[:2] + [pats[2 + i] for i in rng.permutation(len(pats) - 2)]
    long_only = []
    for m in (7, 8, 9, 13, 30):                                      # hits and misses at every depth, all >= K characters
        long_only += lf_walk_patterns(orc, rng, 1500, m, 0.5, alphabet=[1, 2, 3, 4, 5])
    long_only = [long_only[i] for i in rng.permutation(len(long_only))]
    ks = []
    for mode in ("off", "auto"):
        findex_amd.set_ktab(mode)
        try:
            hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
            check_search(hip, orc, pats)                         # mixed: most waves hold an ineligible pattern
            check_search(hip, orc, long_only)                    # every wave takes the table
            ks.append(hip.stats()["ktab_k"])
            if mode == "auto":
                assert hip.stats()["ktab_lookups"] >= len(long_only) - 64
        finally:
            findex_amd.set_ktab("auto")
    assert ks[0] == 0 and ks[1] >= 5                                 # 5^6 = 15625 <= n/8


@pytest.mark.parametrize("layout", ["onehot", "bytes"])
def test_row_jump_table_on_and_off_agree(layout):
    """The row jump table (fmx_jump.hip) takes eight backward steps with one lookup once a search holds one row; with
    it and without it every (sp, ep) -- misses with the reference loop's values at the failing step -- and the
    executed-step counter must equal the oracle's.  Patterns of every length around the eight-step groups, misses at
    every depth, foreign bytes and byte 0 inside a group, patterns whose walk crosses the EOF row, ragged batches in
    which only some groups of a wave can jump; the k-mer table on and off beside it (the jump starts where it ends)."""
    text = bytes(np.random.default_rng(77).integers(97, 101, 40_000).astype(np.uint8))      # real text: LF walks cross the EOF row
    cases = [synth_bwt(300_000, 1, 5, 19), bwt_of_text(text[:3000])]
    rng = np.random.default_rng(8)
    findex_amd.set_layout(layout)
    try:
        for bwt, eof, counts in cases:
            orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
            syms = [int(c) for c in np.nonzero(counts)[0]]
            pats = []
            for m in range(1, 45):
                pats += lf_walk_patterns(orc, rng, 120, m, 0.35, alphabet=syms)
            # walks that start at and around the EOF row, and ones that pass through it
            for r0 in (orc.eof, max(orc.eof - 1, 0), min(orc.eof + 1, orc.n - 1), 0, orc.n - 1):
                r, cs = r0, []
                for _ in range(40):
                    cs.append(orc.bwt_read(r))
                    r = orc.getPrevI(r)
                full = bytes(reversed(cs))
                pats += [full[-m:] for m in (5, 9, 17, 25, 33, 40)]
            pats += [bytes([syms[0]] * 12 + [0] + [syms[0]] * 12), bytes([syms[0]] * 20 + [250] + [syms[0]] * 3), b""] * 5
            pats = [pats[i] for i in rng.permutation(len(pats))]
            uniform = lf_walk_patterns(orc, rng, 4000, 36, 0.1, alphabet=syms)      # every wave jumps together
            seen = []
            for ktab in ("auto", "off"):
                for jump in ("off", "auto", "jumps", "rows", "rows3"):
                    findex_amd.set_ktab(ktab)
                    findex_amd.set_jump(jump)
                    try:
                        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
                        check_search(hip, orc, pats)
                        check_search(hip, orc, uniform)
                        st = hip.stats()
                        seen.append((jump, st["jump_bytes"], st["jump_lookups"]))
                        # "auto": the lane groups of k_search4 walk the one-row part, eight steps per lookup in the row jump
                        # table at chunk boundaries, three per lookup in the three-step table between them; "jumps": with
                        # the jump table alone; "rows3" / "rows": one lane per pattern walks it (k_search_rows) with the
                        # three-step / one-step row table, as on an index too large for a jump table
                        assert st["row_bytes"] == (8 * orc.n if jump in ("auto", "rows", "rows3") else 0)
                        assert (st["row_lookups"] > 0) == (jump in ("auto", "rows", "rows3"))
                        if jump in ("auto", "jumps"):
                            row_b = jump_row_bytes(layout) if jump == "auto" else 16      # (pairs are built from the three-step table: not with "jumps" alone)
                            assert st["jump_bytes"] == row_b * orc.n and st["jump_lookups"] > (1 if row_b == 32 else 2) * len(uniform) and st["jump_chars"] == 9
                            # the table itself: (BWT' along an 8-step LF walk, the row it ends on) -- spot-check through a
                            # search of the walked characters from a one-row start is what check_search just did; the
                            # executed steps equal the oracle's although fewer lines were requested
                            assert st["search_requests"] < st["backward_steps"]
                        else:
                            assert st["jump_bytes"] == 0 and st["jump_lookups"] == 0
                    finally:
                        findex_amd.set_ktab("auto")
                        findex_amd.set_jump("auto")
            # every entry width: 8 .. 11 characters per lookup (9 above), with and without the three-step table beside it
            for jc in (8, 10, 11):
                for jump in ("auto", "jumps"):
                    findex_amd.config_set("jump_chars", jc)
                    findex_amd.set_jump(jump)
                    try:
                        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
                        check_search(hip, orc, pats)
                        check_search(hip, orc, uniform)
                        st = hip.stats()
                        assert st["jump_chars"] == jc and st["jump_lookups"] > len(uniform)
                    finally:
                        findex_amd.config_set("jump_chars", 9)
                        findex_amd.set_jump("auto")
    finally:
        findex_amd.set_layout("auto")
    # a block-mode handle (NaiveBWTSearcher) never has one
    blk = findex_amd.HipFMSearcher.from_block(np.array([1, 2, 3, 1, 2], dtype=np.uint8), np.arange(256, dtype=np.int64) * 0, 2)
    blk.search_batch(np.array([1, 2], dtype=np.uint8), np.array([0, 2], dtype=np.uint64))
    assert blk.stats()["jump_bytes"] == 0


@pytest.mark.parametrize("layout", ["onehot", "bytes"])
def test_row_tables_on_repetitive_texts(layout):
    """The three-step lookups take intervals of up to G rows (lane t looks up row sp + t): texts made of repeats, where
    a pattern's interval stays a few rows wide for many steps and only SOME of its rows agree with the next three
    characters -- every mode of the row tables against the oracle, intervals, misses' values and step counts."""
    rng = np.random.default_rng(2024)
    texts = []
    unit = bytes(rng.integers(97, 100, 37).astype(np.uint8))
    rep = bytearray(unit * 60)
    for j in rng.integers(0, len(rep), 70):          # a few point mutations: the repeats split into families of 2..8 rows
        rep[int(j)] = int(rng.integers(97, 101))
    texts.append(bytes(rep))
    texts.append(b"abracadabra" * 150 + b"abracadabrx" * 3 + b"cadabraabra" * 40)
    texts.append(bytes(rng.integers(97, 99, 2500).astype(np.uint8)))      # two letters: wide intervals for a dozen steps
    findex_amd.set_layout(layout)
    try:
        for text in texts:
            bwt, eof, counts = bwt_of_text(text)
            orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
            syms = [int(c) for c in np.nonzero(counts)[0]]
            pats = []
            for m in (3, 5, 6, 7, 8, 9, 11, 12, 16, 17, 19, 24, 31, 40):
                pats += lf_walk_patterns(orc, rng, 150, m, 0.3, alphabet=syms)
                for _ in range(40):                  # substrings of the text itself: hits with several occurrences
                    a = int(rng.integers(0, len(text) - m))
                    pats.append(text[a:a + m])
            pats = [pats[i] for i in rng.permutation(len(pats))]
            for ktab in ("auto", "off"):
                for jump in ("off", "auto", "jumps", "rows", "rows3"):
                    findex_amd.set_ktab(ktab)
                    findex_amd.set_jump(jump)
                    try:
                        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
                        hits = check_search(hip, orc, pats)
                        assert hits > len(pats) // 3
                        if jump == "auto":
                            assert hip.stats()["row_lookups"] > 0
                    finally:
                        findex_amd.set_ktab("auto")
                        findex_amd.set_jump("auto")
            for jc in (8, 11):                           # intervals of a few rows through the row jump table, lane t row sp + t
                findex_amd.config_set("jump_chars", jc)
                try:
                    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
                    check_search(hip, orc, pats)        # (whether the table is reached at all depends on the text's repeats)
                finally:
                    findex_amd.config_set("jump_chars", 9)
    finally:
        findex_amd.set_layout("auto")


def test_prepare_builds_every_table_up_front():
    """fmx_prepare(KTAB | SELECT | JUMP): the k-mer table, the select directory and the three row tables are built by
    the call, not by the first search / Psi / regex match after it (tables_build_ms does not move again), and the
    answers are the oracle's."""
    bwt, eof, counts = synth_bwt(300_000, 1, 20, 5)
    hip, orc = pair_from_mem(bwt, eof, counts)
    st0 = hip.stats()
    assert st0["jump_bytes"] == 0 and st0["row_bytes"] == 0 and st0["ktab_k"] == 0
    hip.prepare(ktab=True, select=True, jump=True, frontier=True)
    st1 = hip.stats()
    assert st1["ktab_k"] > 0 and st1["jump_bytes"] == jump_row_bytes() * orc.n and st1["row_bytes"] == 16 * orc.n      # R3 + R1: 8 n each
    assert st1["tables_build_ms"] > 0
    rng = np.random.default_rng(6)
    pats = lf_walk_patterns(orc, rng, 3000, 30, 0.2, alphabet=list(range(1, 21)))
    check_search(hip, orc, pats)
    rows = rng.integers(0, orc.n, 500).astype(np.uint64)
    assert hip.nextSubstr_batch(rows, 9) == [orc.nextSubstr(int(r), 9) for r in rows]
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(re, lineOnly=True)) for re in ("ab(c|d)*e", "a[b-d]+f")]
    findex_amd.ReTree.matchSA_batch(hip, trees, mode="frontier", maxBranching=1 << 20, maxIterations=0)
    st2 = hip.stats()
    assert st2["tables_build_ms"] == st1["tables_build_ms"], "a table was built after fmx_prepare"
    assert st2["jump_lookups"] > 0 and st2["row_lookups"] > 0


def test_tables_are_built_lazily():
    """The default policy ("tables_after" = auto): a handle's derived tables are built by fmx_prepare or by the search
    that brings its patterns to the threshold -- 1024 for the k-mer table, max(65536, n / 64) for the row tables -- never
    by a per-call adapter's single queries; fmx_drop_tables frees the row tables and resets the count; the row jump
    table is built in ONE allocation (peak_table_build_bytes = its own size).  Answers are the oracle's throughout."""
    import time
    bwt, eof, counts = synth_bwt(2_000_000, 1, 20, 6)
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(8)
    findex_amd.config_set("tables_after", "auto")
    try:
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
        t0 = time.perf_counter()
        assert hip.search(b"\x01\x02\x03") == orc.search(b"\x01\x02\x03")
        dt = time.perf_counter() - t0
        st = hip.stats()
        assert st["jump_bytes"] == 0 and st["row_bytes"] == 0 and st["ktab_k"] == 0 and st["patterns_seen"] == 1
        assert st["tables_build_ms"] == 0 and dt < 0.1
        pats = lf_walk_patterns(orc, rng, 600, 30, 0.2, alphabet=list(range(1, 21)))
        check_search(hip, orc, pats)                         # 601 patterns so far: still nothing
        assert hip.stats()["ktab_k"] == 0
        check_search(hip, orc, pats)                         # 1201: the k-mer table, not the row tables
        st = hip.stats()
        assert st["ktab_k"] > 0 and st["jump_bytes"] == 0 and st["row_bytes"] == 0
        big = pats * 54                                       # 32400 patterns per call
        check_search(hip, orc, big)
        assert hip.stats()["jump_bytes"] == 0
        check_search(hip, orc, big)                          # 66001 >= 65536: this search builds and uses them
        st = hip.stats()
        assert st["jump_bytes"] == jump_row_bytes() * orc.n and st["row_bytes"] == 8 * orc.n and st["jump_lookups"] > 0
        assert st["peak_table_build_bytes"] == jump_row_bytes() * orc.n, "the row jump table is built in one allocation"
        assert st["patterns_seen"] == 1 + 2 * 600 + 2 * 32400
        hip.drop_tables()
        st = hip.stats()
        assert st["jump_bytes"] == 0 and st["row_bytes"] == 0 and st["patterns_seen"] == 0 and st["ktab_k"] > 0
        check_search(hip, orc, pats)
        assert hip.stats()["jump_bytes"] == 0
        hip.prepare(ktab=False, jump=True)
        assert hip.stats()["jump_bytes"] == jump_row_bytes() * orc.n
        hip.stats_reset()
        check_search(hip, orc, pats)
        assert hip.stats()["jump_lookups"] > 0
        hip.close()
        # one flag per table: prepare(KTAB) alone builds the k-mer table and leaves the row tables to their threshold
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
        hip.prepare(ktab=True)
        check_search(hip, orc, pats)
        st = hip.stats()
        assert st["ktab_k"] > 0 and st["jump_bytes"] == 0 and st["row_bytes"] == 0, "prepare(KTAB) made a search build the row tables"
        hip.close()
        # ... and the reverse: prepare(JUMP) alone leaves the k-mer table to its threshold of 1024 patterns
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
        hip.prepare(ktab=False, jump=True)
        check_search(hip, orc, pats[:300])
        st = hip.stats()
        assert st["ktab_k"] == 0 and st["jump_bytes"] == jump_row_bytes() * orc.n
        hip.close()
        # "jumps" alone: no three-step table to build from -- eight rank queries per row, same table
        findex_amd.config_set("jump", "jumps")
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
        hip.prepare(ktab=True, jump=True)
        st = hip.stats()
        assert st["jump_bytes"] == 16 * orc.n and st["row_bytes"] == 0          # (pairs are built from the three-step table: none here)
        check_search(hip, orc, pats)
        hip.close()
    finally:
        findex_amd.config_set("tables_after", "0")
        findex_amd.config_set("jump", "auto")


def test_two_handles_with_their_own_table_policies():
    """The table policy lives on the handle (round 5; until round 4: process-global atomics, so two handles in one JVM could
    not differ): fmx_config_set writes the defaults a handle copies at open, fmx_index_config_set changes one handle's own.
    Two handles on one index, one process: the first with every table off, the second with 11-character entries in pairs;
    a third, opened afterwards, has the untouched defaults.  Answers are the oracle's on all three."""
    bwt, eof, counts = synth_bwt(600_000, 1, 20, 15)
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(3)
    pats = lf_walk_patterns(orc, rng, 2000, 40, 0.2, alphabet=list(range(1, 21)))
    a = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    b = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    a.config_set("jump", "off")
    a.config_set("ktab", "off")
    b.config_set("jump_chars", 11)
    b.config_set("jump_pairs", "on")
    for h in (a, b):
        h.prepare(ktab=True, jump=True)
    sa, sb = a.stats(), b.stats()
    assert sa["jump_bytes"] == 0 and sa["row_bytes"] == 0 and sa["ktab_k"] == 0 and sa["tables_held_bytes"] == 0
    assert sb["jump_bytes"] == 32 * orc.n and sb["jump_chars"] == 11 and sb["row_bytes"] == 8 * orc.n and sb["ktab_k"] > 0
    assert sb["tables_held_bytes"] >= 40 * orc.n and sb["table_budget_bytes"] == 2**64 - 1
    c = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)          # the defaults were not touched
    c.prepare(ktab=True, jump=True)
    sc = c.stats()
    assert sc["jump_bytes"] == jump_row_bytes() * orc.n and sc["jump_chars"] == 9 and sc["ktab_k"] > 0
    for h in (a, b, c):
        h.stats_reset()
        check_search(h, orc, pats)
    assert a.stats()["jump_lookups"] == 0 and a.stats()["ktab_lookups"] == 0
    assert b.stats()["jump_lookups"] > 0 and c.stats()["jump_lookups"] > 0
    # a handle's policy can change under it: drop + prepare rebuild by the new one
    b.drop_tables(jump=True, frontier=False, ktab=True)
    assert b.stats()["tables_held_bytes"] == 0
    b.config_set("jump_pairs", "off")
    b.config_set("jump_chars", 8)
    b.prepare(ktab=True, jump=True)
    sb = b.stats()
    assert sb["jump_bytes"] == 16 * orc.n and sb["jump_chars"] == 8
    check_search(b, orc, pats)
    with pytest.raises(findex_amd.FmxError):
        a.config_set("layout", "bytes")                                # not a per-handle key
    with pytest.raises(findex_amd.FmxError):
        a.config_set("jump_chars", 12)
    for h in (a, b, c):
        h.close()


def test_table_budget_decides_what_is_built():
    """"table_budget" (round 5): the device bytes ALL derived tables of a handle may hold.  n = 2 000 000, sigma = 20: the
    k-mer table is 2.7 MB (K = 4), the three-step table 16 MB, the row jump table 32 MB as single entries and 64 MB as
    pairs.  Budgets that admit everything / single entries only / the three-step table only / the k-mer table only / next to
    nothing; fmx_prepare_ex and the per-handle key; a fraction.  The handle never holds more than its budget, what is
    left out is what the header says, and every configuration answers like the oracle."""
    bwt, eof, counts = synth_bwt(2_000_000, 1, 20, 6)
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(12)
    pats = lf_walk_patterns(orc, rng, 1500, 36, 0.2, alphabet=list(range(1, 21)))
    n = orc.n
    MB = 1_000_000
    for budget, want_jump, want_rows, want_k in ((200 * MB, 32 * n, 8 * n, True), (60 * MB, 16 * n, 8 * n, True),
                                                 (30 * MB, 0, 8 * n, True), (10 * MB, 0, 0, True), (256, 0, 0, False)):      # (256 bytes: not even the 20 one-character entries)
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
        base = hip.stats()["index_bytes"]                                     # the rank dictionary, the BWT bytes, C[], counters
        hip.config_set("jump_pairs", "on")
        hip.prepare(ktab=True, jump=True, budget_bytes=budget)              # fmx_prepare_ex
        st = hip.stats()
        assert st["table_budget_bytes"] == budget and st["tables_held_bytes"] <= budget, (budget, st["tables_held_bytes"])
        assert st["jump_bytes"] == want_jump and st["row_bytes"] == want_rows and (st["ktab_k"] > 0) == want_k, (budget, st)
        assert st["tables_held_bytes"] == st["index_bytes"] - base
        if st["tables_held_bytes"]:
            assert st["hbm_free_after_tables"] > 0
        check_search(hip, orc, pats)
        hip.nextSubstr_batch(np.arange(5, dtype=np.uint64), 6)                # the select directory is never refused, but it is counted
        assert hip.stats()["tables_held_bytes"] > st["tables_held_bytes"]
        hip.close()
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    hip.config_set("table_budget", "0.0000001")                               # a fraction of the free HBM: ~25 KB here
    hip.prepare(ktab=True, jump=True)
    st = hip.stats()
    assert st["jump_bytes"] == 0 and st["row_bytes"] == 0 and 0 < st["table_budget_bytes"] < MB
    check_search(hip, orc, pats)
    hip.config_set("table_budget", "auto")
    hip.drop_tables(jump=True, frontier=False, ktab=True)
    hip.prepare(ktab=True, jump=True)
    assert hip.stats()["jump_bytes"] == jump_row_bytes() * n
    hip.close()


def test_pipelined_host_batch_pageable_and_pinned():
    """Host-pointer batches of 128k patterns or more travel as whole arrays or, with the "pipeline" setting on and
    page-locked buffers, as chunks over three streams (fmx_api.cpp): ragged
    lengths with empty patterns at chunk borders, offsets that do not start at 0, pageable and page-locked
    (fmx_host_alloc) buffers -- every interval must equal the oracle's."""
    from findex_amd.searcher import PinnedArray
    bwt, eof, counts = synth_bwt(400_000, 1, 6, 77)
    hip, orc = pair_from_mem(bwt, eof, counts)
    rng = np.random.default_rng(21)
    k = 300_001
    lens = rng.integers(0, 14, k)
    lens[[0, k // 8, k // 8 + 1, k // 2, k - 1]] = 0
    off = np.concatenate([[5], 5 + np.cumsum(lens)]).astype(np.uint64)       # the first 5 bytes belong to no pattern
    buf = rng.integers(1, 7, int(off[-1])).astype(np.uint8)
    # make a good share of them hits: overwrite with LF-walk text
    walk, _ = hip.lf_walk_batch(rng.integers(0, hip.n, 20000).astype(np.uint64), 13)
    for j in range(0, k, 15):
        L = int(lens[j])
        buf[int(off[j]):int(off[j]) + L] = walk[(j // 15) % 20000, :L][::-1]
    wsp, wep, _ = orc.search_batch(buf, off)
    sp, ep = hip.search_batch(buf, off)
    assert np.array_equal(sp, wsp) and np.array_equal(ep, wep)
    assert int((wsp < wep).sum()) > k // 20
    pb, po = PinnedArray(buf.shape, np.uint8), PinnedArray(off.shape, np.uint64)
    psp, pep = PinnedArray((k,), np.uint64), PinnedArray((k,), np.uint64)
    pb.array[:] = buf
    for pipeline in ("off", "on"):               # whole arrays and one search / chunks over three streams
        findex_amd.set_pipeline(pipeline)
        try:
            po.array[:] = off
            psp.array[:] = 0
            pep.array[:] = 0
            hip.search_batch(pb.array, po.array, out=(psp.array, pep.array))
            assert np.array_equal(psp.array, wsp) and np.array_equal(pep.array, wep)
            # a decreasing offset is refused on every path (large batches are checked beside / behind their uploads, not
            # by a walk over the offsets before anything starts), and the handle stays usable
            for q in (k // 3, k // 8):                   # inside a chunk; at a chunk's border
                bad = off.copy()
                bad[q] = bad[q + 1] + 1
                with pytest.raises(findex_amd.FmxError) as e:
                    hip.search_batch(buf, bad)
                assert e.value.code == 3
                po.array[:] = bad
                with pytest.raises(findex_amd.FmxError) as e:
                    hip.search_batch(pb.array, po.array, out=(psp.array, pep.array))
                assert e.value.code == 3
            po.array[:] = off
            psp.array[:] = 0
            hip.search_batch(pb.array, po.array, out=(psp.array, pep.array))
            assert np.array_equal(psp.array, wsp) and np.array_equal(pep.array, wep)
        finally:
            findex_amd.set_pipeline("off")


def test_concurrent_calls_on_one_handle():
    """An index handle is immutable after open: host-pointer calls from several threads (each borrows its own
    call context from the handle's pool) must give the single-threaded answers -- small and large batches,
    searches and single steps mixed."""
    import threading
    bwt, eof, counts = synth_bwt(300_000, 1, 20, 31)
    hip, orc = pair_from_mem(bwt, eof, counts)
    rng = np.random.default_rng(9)
    jobs = []
    for j in range(8):
        pats = lf_walk_patterns(orc, rng, 4000 if j % 2 else 3, 12, 0.2, alphabet=list(range(1, 21)))
        buf, off = pack_patterns(pats)
        wsp, wep, _ = orc.search_batch(buf, off)
        jobs.append((buf, off, (wsp, wep)))
    errors = []

    def work(j):
        try:
            buf, off, want = jobs[j]
            for _ in range(5):
                sp, ep = hip.search_batch(buf, off)
                assert np.array_equal(sp, want[0]) and np.array_equal(ep, want[1]), j
                o = hip.occ_batch(np.array([j + 1], dtype=np.uint8), np.array([1000 * j], dtype=np.int64))
                assert int(o[0]) == orc.occ(j + 1, 1000 * j)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(j,)) for j in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_concurrent_regex_batches_on_one_handle():
    """Several resident regex batches matched at the same time from several threads on ONE index handle (each batch
    has its own queue, counters and captured launch graph; the launches of different batches run side by side on
    the device): every call must give the answer the batch gives alone."""
    import threading
    from findex_amd.regex import RegexBatch
    bwt, eof, counts = synth_bwt(600_000, 97, 100, 5)            # 4 letters: a..d
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    sets = [["ab[a-c]*d", "a[ab]*c", "b(a|cd)[ad]*b"] + ["abcd"[(i + j) % 4] + "abcd"[(i // 4) % 4] + "c[ab]?d" for i in range(40)]
            for j in range(4)]
    batches = [RegexBatch(hip, [findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in rs]) for rs in sets]
    want = []
    for b in batches:
        out, per = b.match_raw(max_steps=40, cap=1 << 20)
        b.match_raw(max_steps=40, cap=1 << 20)               # the second match captures the batch's graph
        want.append((out, per))
    errors = []

    def work(j):
        try:
            for _ in range(25):
                out, per = batches[j].match_raw(max_steps=40, cap=1 << 20)
                assert out.size == want[j][0].size and out.tobytes() == want[j][0].tobytes(), j
                assert np.array_equal(per, want[j][1]), j
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(j,)) for j in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]
    assert want[0][0].size > 1000


def test_search_batch_multi_replicas():
    """fmx_search_batch_multi over three replica handles (all on device 0 here; one per GPU in production): the
    slices are searched from three host threads and land in their ranges of the output arrays."""
    bwt, eof, counts = synth_bwt(200_000, 1, 12, 41)
    hips = [findex_amd.HipFMSearcher.from_mem(bwt, eof, counts) for _ in range(3)]
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(2)
    pats = []
    for m in (0, 3, 11, 40):
        pats += lf_walk_patterns(orc, rng, 500, m, 0.2, alphabet=list(range(1, 13)))
    pats = [pats[i] for i in rng.permutation(len(pats))]
    buf, off = pack_patterns(pats)
    wsp, wep, _ = orc.search_batch(buf, off)
    sp, ep = findex_amd.HipFMSearcher.search_batch_multi(hips, buf, off)
    assert np.array_equal(sp, wsp) and np.array_equal(ep, wep)
    # fewer patterns than handles, and all-empty patterns
    sp, ep = findex_amd.HipFMSearcher.search_batch_multi(hips, buf[: int(off[1])], off[:2])
    assert (int(sp[0]), int(ep[0])) == (int(wsp[0]), int(wep[0]))
    sp, ep = findex_amd.HipFMSearcher.search_batch_multi(hips, np.zeros(0, dtype=np.uint8), np.zeros(6, dtype=np.uint64))
    assert sp.tolist() == [0] * 5 and ep.tolist() == [hips[0].n] * 5
    other = findex_amd.HipFMSearcher.from_mem(bwt[:1000], 5, np.bincount(np.delete(bwt[:1000], 5), minlength=256).astype(np.int64))
    with pytest.raises(findex_amd.FmxError):
        findex_amd.HipFMSearcher.search_batch_multi([hips[0], other], buf, off)


def test_regex_batch_multi_replicas_and_gather(testdata):
    """fmx_regex_batch_match_multi over three replica handles (all on device 0 here; one per GPU in production): the
    batch is cut by estimated frontier work, slices are matched from three host threads, the concatenated result
    list equals the single-handle one.  fmx_gather then collects device-resident slices of three handles into one
    host array."""
    import ctypes
    from findex_amd import _lib
    from findex_amd.regex import RegexBatchMulti
    hips = [pair_from_files(testdata, "words", True)[0] for _ in range(3)]
    res = REGEXES + ["co(m|n)+e", "s[aeiou]+t", "abc", "zz?y"]          # light and heavy ones mixed
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(re)) for re in res]
    want, wper = findex_amd.ReTree.prepare_batch(hips[0], trees).match_raw(cap=1 << 20)
    multi = RegexBatchMulti(hips, trees)
    for _ in range(2):
        got, gper = multi.match_raw(cap=1 << 20)
        assert got.size == want.size and all(np.array_equal(got[f], want[f]) for f in ("regex", "len", "sp", "ep"))
        assert np.array_equal(gper, wper)
    one = RegexBatchMulti(hips, trees[:1])                              # fewer regexes than handles
    got, _ = one.match_raw()
    assert np.array_equal(got["sp"], want["sp"][want["regex"] == 0])
    with pytest.raises(findex_amd.FmxError) as e:                       # too small a result buffer is reported
        multi.match_raw(cap=8)
    assert e.value.code == 9
    # fmx_gather: (sp, ep) slices left on the device by three handles' searches
    torch = _torch()
    orc = oracle.NaiveFMSearcher(os.path.join(testdata, "words.bwt"), bigEndian=True)
    rng = np.random.default_rng(8)
    buf, off = pack_patterns(lf_walk_patterns(orc, rng, 900, 6, 0.2, alphabet=list(range(97, 123))))
    wsp, _, _ = orc.search_batch(buf, off)
    cuts = [0, 100, 100, 900]                                           # an empty slice in the middle
    d_pat = torch.from_numpy(buf).cuda()
    outs, ptrs = [], []
    for r in range(3):
        a, b = cuts[r], cuts[r + 1]
        d_off = torch.from_numpy(off[a:b + 1].astype(np.int64)).cuda()
        sp = torch.empty(max(b - a, 1), dtype=torch.int64, device="cuda")
        ep = torch.empty_like(sp)
        hips[r].search_batch_dev(d_pat.data_ptr(), d_off.data_ptr(), sp.data_ptr(), ep.data_ptr(), b - a)
        outs.append((sp, ep, d_off))
        ptrs.append(sp.data_ptr())
    torch.cuda.synchronize()
    dst = np.zeros(900, dtype=np.uint64)
    idxs = (ctypes.c_void_p * 3)(*[h.handle for h in hips])
    srcs = (ctypes.c_void_p * 3)(*ptrs)
    cnt = (ctypes.c_size_t * 3)(100, 0, 800)
    _lib.check(_lib.load().fmx_gather(idxs, 3, srcs, cnt, 8, dst.ctypes.data_as(ctypes.c_void_p)))
    assert np.array_equal(dst, wsp)


def test_device_offsets_validation_knob():
    """fmx_config_set("validate", "1"): the device-pointer search checks d_off on the device."""
    import ctypes
    from findex_amd import _lib
    torch = _torch()
    bwt, eof, counts = synth_bwt(50_000, 1, 6, 3)
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    pat = torch.randint(1, 7, (4000,), dtype=torch.uint8, device="cuda")
    good = torch.arange(0, 4001, 4, dtype=torch.int64, device="cuda")
    bad = good.clone()
    bad[500] = 1
    sp = torch.empty(1000, dtype=torch.int64, device="cuda")
    ep = torch.empty_like(sp)
    L = _lib.load()
    _lib.check(L.fmx_config_set(b"validate", b"1"))
    try:
        hip.search_batch_dev(pat.data_ptr(), good.data_ptr(), sp.data_ptr(), ep.data_ptr(), 1000)
        with pytest.raises(findex_amd.FmxError) as e:
            hip.search_batch_dev(pat.data_ptr(), bad.data_ptr(), sp.data_ptr(), ep.data_ptr(), 1000)
        assert e.value.code == 3
    finally:
        _lib.check(L.fmx_config_set(b"validate", b"0"))
    torch.cuda.synchronize()


def test_counts_must_describe_bwt():
    bwt, eof, counts = synth_bwt(5000, 1, 4, 1)
    bad = counts.copy()
    bad[1] += 1
    bad[2] -= 1
    with pytest.raises(findex_amd.FmxError) as e:
        findex_amd.HipFMSearcher.from_mem(bwt, eof, bad)
    assert e.value.code == 2
    zero = bwt.copy()
    zero[17] = 0
    c2 = np.bincount(zero, minlength=256).astype(np.int64)
    c2[zero[eof]] -= 1
    with pytest.raises(findex_amd.FmxError) as e:
        findex_amd.HipFMSearcher.from_mem(zero, eof, c2)
    assert e.value.code in (2, 6)
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    with pytest.raises(findex_amd.FmxError):
        hip.prev_range_batch([5], [4], [1])            # sp > ep
    with pytest.raises(findex_amd.FmxError):
        hip.lf_walk_batch([5000], 1)                   # row out of range


# ---------------------------------------------------------------- regex frontier (K5)
REGEXES = ["ab", "abc", "a[bcd]e", "th(e|a)", "q[a-z]*k", "co(m|n)+e", "a[b-d]*e", "x?yz", "(ab)*c", "ing\r\n",
           "ab(cd|ef)+gh", "a(cd|ef)*j", "s[aeiou]+t", "[a-c]", "(a|b|d|c)", "z(a|e)*b", "un[a-z][a-z]ed"]


def oracle_results(orc, re, lineOnly=False):
    t = R.ReTree(R.re2post(re, lineOnly))
    res, left, pops = orc.match_tables(t.tables(), 1 << 40, 0, cap=1 << 22)
    assert left == 0
    return sorted(res), pops


@pytest.mark.parametrize("name,be", [("test.cmp", False), ("words", True)])
def test_regex_frontier_parity(testdata, name, be):
    """ReTree.matchSA with limits that do not bind: same result multiset, same number of
    getPrevRange calls."""
    hip, orc = pair_from_files(testdata, name, be)
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(re)) for re in REGEXES]
    hip.stats_reset()
    got = findex_amd.ReTree.matchSA_batch(hip, trees, max_steps=1 << 20, cap=1 << 21)   # test.txt is one 10 KB word
    assert not findex_amd.ReTree.last_truncated
    total_pops = 0
    for re, g in zip(REGEXES, got):
        want, pops = oracle_results(orc, re)
        total_pops += pops
        assert [r.key() for r in g] == want, re
    assert hip.stats()["backward_steps"] == total_pops
    # single-regex entry point and SAResult rendering (re2.scala:9-19)
    one = trees[1].matchAll(hip, max_steps=1 << 20)
    assert [r.key() for r in one] == oracle_results(orc, REGEXES[1])[0]
    for r in one[:5]:
        sub = orc.nextSubstr(r.sp, r.len).decode("latin-1")
        assert str(r) == (sub if r.cnt == 1 else "[%d Results] %s" % (r.cnt, sub))


def test_regex_reference_vectors():
    """T/REParser.scala:591-605: '.*(a|b)ca' over reversed 'mmabcacamabbbca' -> 2 results."""
    hip, orc = pair_from_mem(*bwt_of_text(b"mmabcacamabbbca"[::-1]))
    ref = findex_amd.ReTree(findex_amd.REParser.re2post(".*(a|b)ca")).matchSA(hip, debugLevel=0)
    assert len(ref) == 2        # ret.length == 2, T/REParser.scala:602-603
    got = findex_amd.ReTree(findex_amd.REParser.re2post(".*(a|b)ca")).matchAll(hip)
    assert len(got) == 2
    assert [r.key() for r in got] == oracle_results(orc, ".*(a|b)ca")[0]


def test_regex_duplicate_follows_give_duplicate_results():
    """The reference does not dedup frontier states: x(a|a) emits each hit twice, and follows
    lists may name one CharNode twice ((a*b*)*c)."""
    hip, orc = pair_from_mem(*bwt_of_text(b"xxabxxabyab"[::-1]))
    for re in ("x(a|a)", "(a*b*)*c", "a(b|b|b)", "b(a*b*)*x"):
        want, _ = oracle_results(orc, re)
        got = findex_amd.ReTree(findex_amd.REParser.re2post(re)).matchAll(hip)
        assert [r.key() for r in got] == want, re
    keys = [r.key() for r in findex_amd.ReTree(findex_amd.REParser.re2post("x(a|a)")).matchAll(hip)]
    assert keys and all(keys.count(k) == 2 for k in keys)


REF_REGEXES = ["th(e|a)", "q[a-z]*k", "co(m|n)+e", "a.*(b|c)d.*f", "s[aeiou]+t", "a[b-d]*e", "e.*", "(a|b|d|c)", "ab?j",
               "in.*g", "z(a|e)*b", "x?yz",
               "a.*(b|c)da.*f"]        # the regex the reference's WordsDB suite runs on words (T/REParser.scala:630)


@pytest.mark.parametrize("kernel", ["wave", "group"])
@pytest.mark.parametrize("name,be", [("test.cmp", False), ("words", True)])
def test_regex_reference_order_and_limits(testdata, name, be, kernel, monkeypatch):
    """ReTree.matchSA as the reference runs it: default limits 1024 / 1000 (which bind for '.*'
    regexes) and tighter ones; results must equal the oracle's replay of _matchSA -- same
    elements, same (newest-first) order, same number of getPrevRange calls.  Both kernels: one regex per wave
    with the heap in LDS and the steps made at push time (limits that fit LDS: all here but 2^14), and one
    regex per lane group with the heap in device memory (FMX_REFMATCH=group forces it)."""
    monkeypatch.setenv("FMX_REFMATCH", kernel)
    hip, orc = pair_from_files(testdata, name, be)
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(re, lineOnly=True)) for re in REF_REGEXES]
    for mb, mi in ((1024, 1000), (16, 50), (4, 0), (1 << 14, 3000), (1024, 2), (1, 0), (300, 1 << 20)):
        hip.stats_reset()
        got = findex_amd.ReTree.matchSA_batch(hip, trees, mode="reference", maxBranching=mb, maxIterations=mi)
        total_pops = 0
        for re, g in zip(REF_REGEXES, got):
            want, left, pops = orc.match_tables(R.ReTree(R.re2post(re, True)).tables(), mb, mi)
            total_pops += pops
            assert [r.key() for r in g] == want, (re, mb, mi)
        st = hip.stats()
        assert st["backward_steps"] == total_pops and st["rank_queries"] == 2 * total_pops, (mb, mi)
    # the same through a resident batch (fmx_regex_batch_match in reference mode): one flat, per-regex-ordered list
    batch = findex_amd.ReTree.prepare_batch(hip, trees)
    flat, per = batch.match_raw(mode="reference", maxBranching=1024, maxIterations=1000)
    got = findex_amd.ReTree.matchSA_batch(hip, trees, mode="reference", maxBranching=1024, maxIterations=1000)
    assert per.tolist() == [len(g) for g in got]
    assert [(int(r["regex"]), int(r["len"]), int(r["sp"]), int(r["ep"])) for r in flat] == [
        (j, *r.key()) for j, g in enumerate(got) for r in g]
    # the Scala-signature entry point
    one = trees[3].matchSA(hip)
    want, _, _ = orc.match_tables(R.ReTree(R.re2post(REF_REGEXES[3], True)).tables(), 1024, 1000)
    assert [r.key() for r in one] == want


@pytest.mark.parametrize("layout", ["onehot", "bytes"])
def test_regex_reference_mode_c4_shape(layout):
    """BASELINE C4's regex grammar under the reference's default limits (1024 / 1000) on a C4-like synthetic index
    (sigma = 28, i.i.d.): 3000 regexes, every regex's result LIST equal to the oracle's replay of _matchSA
    (re2/retree.scala:618-653) in the reference's own order, pops counted; a handful of them use all 999
    iterations.  What `bench.py --workload c4ref` checks on its sample, here in both layouts."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import regex_workload
    n = (1 << 22) + 77
    rng = np.random.default_rng(44)
    alpha = np.frombuffer(regex_workload.ALPHABET.encode(), dtype=np.uint8)
    bwt = alpha[rng.integers(0, alpha.size, n)]
    eof = n // 3
    counts = oracle.histogram(bwt, eof, threads=4)
    findex_amd.set_layout(layout)
    try:
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    finally:
        findex_amd.set_layout("auto")
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts, threads=4)
    import random
    prng = random.Random(4004)
    cand = [regex_workload.gen_one(prng) for _ in range(3300)]
    cs = findex_amd.ReTree.compile_batch(cand)
    trees = cs.select(np.nonzero(cs.ok())[0][:3000])
    batch = findex_amd.ReTree.prepare_batch(hip, trees)
    hip.stats_reset()
    got, per = batch.match_raw(mode="reference", maxBranching=1024, maxIterations=1000)
    tables = [trees[i].tables() for i in range(len(trees))]
    want, pops, _ = orc.match_tables_batch(tables, maxBranching=1024, maxIterations=1000, threads=4, ordered=True)
    assert got.size == want.size and all(np.array_equal(got[f], want[f]) for f in ("regex", "len", "sp", "ep"))
    assert np.array_equal(per, np.bincount(want["regex"], minlength=len(trees)).astype(np.uint32))
    assert hip.stats()["backward_steps"] == pops
    # tighter limits bind for many more of them
    got2, _ = batch.match_raw(mode="reference", maxBranching=12, maxIterations=60)
    want2, _, _ = orc.match_tables_batch(tables, maxBranching=12, maxIterations=60, threads=4, ordered=True)
    assert got2.size == want2.size and all(np.array_equal(got2[f], want2[f]) for f in ("regex", "len", "sp", "ep"))
    assert got2.size != got.size or not np.array_equal(got2["sp"], got["sp"])


def test_random_regexes_all_engines_vs_oracle(testdata):
    """Differential test: ~300 random regexes over a small alphabet (test_regex_compile.random_regex),
    every one the reference can parse, through the three engines' tables on the GPU against the
    oracle: ReTree frontier mode (all matches), ReTree reference-order mode under tight limits, and
    the Thompson engine when createNFA accepts the regex."""
    import random
    from oracle import engines as E
    from test_regex_compile import random_regex
    hip, orc = pair_from_mem(*bwt_of_text((b"abcde" * 3 + b"badcebadce" + b"eeddccbbaa" + b"abacadaeab") * 4))
    rng = random.Random(int(os.environ.get("FMX_TEST_SEED", "99")))          # stress runs: other seeds / counts
    regs = []
    while len(regs) < int(os.environ.get("FMX_TEST_NREGEX", "300")):
        re = random_regex(rng)
        if "." in re or "\\w" in re or "\\d" in re:
            continue
        try:
            tb = R.ReTree(R.re2post(re)).tables()
        except (R.MatchError, R.Re2PostSyntax):
            continue
        # follows lists that repeat a node (nested stars) make the undeduplicated frontier grow
        # exponentially, in the reference too: keep the regexes whose full search is small
        if orc.match_tables(tb, 1 << 40, 5000)[1] != 0:
            continue
        regs.append(re)
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(re)) for re in regs]
    got = findex_amd.ReTree.matchSA_batch(hip, trees, max_steps=1 << 16, cap=1 << 22)
    assert not findex_amd.ReTree.last_truncated
    ref = findex_amd.ReTree.matchSA_batch(hip, trees, mode="reference", maxBranching=24, maxIterations=40, cap=1 << 22)
    n_thompson = 0
    for re, g, rf in zip(regs, got, ref):
        t = R.ReTree(R.re2post(re)).tables()
        want, left, _ = orc.match_tables(t, 1 << 40, 0, cap=1 << 22)
        assert left == 0 and [r.key() for r in g] == sorted(want), re
        assert [r.key() for r in rf] == orc.match_tables(t, 24, 40)[0], re
        try:
            onfa = E.createNFA(R.re2post(re))
            if any(s is E.MatchState for s in onfa.outStates()):
                raise R.MatchError("nullable")
        except R.MatchError:
            with pytest.raises(findex_amd.MatchError):
                findex_amd.REParser.createNFA(findex_amd.REParser.re2post(re))
            continue
        nfa = findex_amd.REParser.createNFA(findex_amd.REParser.re2post(re))
        assert [r.key() for r in findex_amd.REParser.matchSA(nfa, hip, maxLength=64)] == sorted(
            E.nfa_matchSA(onfa, _OIdx(orc), maxLength=64)), re
        n_thompson += 1
    assert n_thompson > 50


def oracle_results_capped(bwt, eof, counts, re, max_len):
    """All matches of length <= max_len: breadth-first over the oracle's getPrevRange."""
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    t = R.ReTree(R.re2post(re)).tables()
    front = [(0, 0, orc.n, s) for s in t["firsts"]]
    out = []
    while front:
        nxt = []
        for ln, sp, ep, s in front:
            r = orc.getPrevRange(sp, ep, t["c"][s])
            if r is None:
                continue
            if t["isLast"][s]:
                out.append((ln + 1, r[0], r[1]))
            elif ln + 1 < max_len:
                nxt += [(ln + 1, r[0], r[1], f) for f in t["follows"][s]]
        front = nxt
    return out


def test_regex_overflow_is_reported():
    bwt, eof, counts = synth_bwt(200_000, 97, 100, 3)
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    t = findex_amd.ReTree(findex_amd.REParser.re2post("a[a-d]*b"))
    with pytest.raises(findex_amd.FmxError) as e:
        t.matchAll(hip, max_frontier=64)
    assert e.value.code == 9
    # a level cap is not an error: every match of length <= 3 comes back and the call says so
    part = t.matchAll(hip, max_steps=3, max_frontier=1 << 22)
    assert findex_amd.ReTree.last_truncated and part and all(r.len <= 3 for r in part)
    assert {r.key() for r in part} == {k for k in oracle_results_capped(bwt, eof, counts, "a[a-d]*b", 3)}


def test_regex_resident_batch_shrinking_cap_and_foreign_index(testdata):
    """a resident batch keeps its scratch, and from its second match on a captured launch
    chain, when a later call passes a smaller result capacity -- every size baked into that chain must then still
    be the one the kernels outside it use.  Three matches with cap 2^20, 2^20, 2^12 must all equal the oracle;
    and a batch made for one index must refuse another (its pointers are inside the chain)."""
    hip, orc = pair_from_files(testdata, "words", True)
    res = ["th(e|a)", "co(m|n)+e", "s[aeiou]+t", "un[a-z][a-z]ed", "q[a-z]*k", "ab(cd|ef)+gh", "ing\r\n"]
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(re)) for re in res]
    want = []
    for j, re in enumerate(res):
        want += [(j,) + key for key in oracle_results(orc, re)[0]]
    assert 64 < len(want) < 4000          # several result slices in use, and everything fits the small cap
    batch = findex_amd.ReTree.prepare_batch(hip, trees)
    for cap in (1 << 20, 1 << 20, 1 << 12, 1 << 20, 1 << 12):
        out, per = batch.match_raw(cap=cap)
        got = [(int(r["regex"]), int(r["len"]), int(r["sp"]), int(r["ep"])) for r in out]
        assert got == want, cap
        assert per.sum() == len(want)
    other, _ = pair_from_files(testdata, "words", True)          # same file, same n, same device: still another index
    with pytest.raises(findex_amd.FmxError) as e:
        import ctypes
        from findex_amd import _lib
        lim = _lib.fmx_limits(0, 0, 0, 1024, 1000)
        buf = np.empty(16, dtype=findex_amd.regex.RESULT_DTYPE)
        n_out = ctypes.c_size_t()
        _lib.check(_lib.load().fmx_regex_batch_match(other.handle, batch._h, ctypes.byref(lim),
                                                     buf.ctypes.data_as(ctypes.c_void_p), 16, ctypes.byref(n_out), None))
    assert e.value.code == 3


def test_regex_long_tail_levels():
    """Frontiers that stay small for many levels, grow, and die slowly (n = 2M over 4 letters: "ab[a-c]*d"
    holds min(3^k, 125k * 0.75^k) elements at level k + 2, i.e. ~9.4k at level 11, ~900 at level 19, dead near
    level 45): a search that starts in one wave, spills its pool into the work queue, is dealt out again by the
    next launches on the small grid and moves to the full grid and back.  Every match must equal the oracle's."""
    bwt, eof, counts = synth_bwt(2_000_000, 97, 100, 12)            # 4 letters: a..d
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    for re in ("ab[a-c]*d", "a[ab]*c", "dcba[ab]*d"):
        want = sorted(oracle_results_capped(bwt, eof, counts, re, 60))
        got = findex_amd.ReTree(findex_amd.REParser.re2post(re)).matchAll(hip, max_steps=60, max_frontier=1 << 22)
        assert sorted(r.key() for r in got) == want, re
        assert len(want) > 100, re


_QUEUE_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import findex_amd
from findex_amd.regex import RegexBatch
from helpers import synth_bwt
from test_gpu_parity import oracle_results_capped
bwt, eof, counts = synth_bwt(600_000, 97, 100, 5)            # 4 letters: a..d
hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
res = ["ab[a-c]*d", "a[ab]*c", "dcba[ab]*d", "b(a|cd)[ad]*b"] + ["abcd"[i % 4] + "abcd"[(i // 4) % 4] + "c[ab]?d" for i in range(60)]
trees = [findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in res]
batch = RegexBatch(hip, trees)
want = sorted((i, ) + k for i, r in enumerate(res) for k in oracle_results_capped(bwt, eof, counts, r, 40))
for call in range(40):
    out, per = batch.match_raw(max_steps=40, cap=1 << 20)
    got = sorted(zip(out["regex"].tolist(), out["len"].tolist(), out["sp"].tolist(), out["ep"].tolist()))
    assert got == want, call
    assert per.tolist() == np.bincount(out["regex"], minlength=len(res)).tolist()
st = hip.stats()
assert st["frontier_queue_writes"] > 40 * 1000, st         # entries did go through the queue
print("ok", len(want), st["frontier_queue_writes"] // 40)
"""


def test_regex_queue_tags_wrap_and_launches_hand_over():
    """The work queue's entries carry 16-bit generation tags and the queue is zeroed before a tag can come round
    (fmx_frontier.hip): with the limit at 5 instead of 60000, 4-round launches and chains of 2, forty calls on one
    resident batch rewind and swap the buffers hundreds of times and cross the zeroing every other call -- each
    call's results must equal the oracle's.  (A child process: the knobs are read once per process.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FMX_FRONTIER_TAG_LIMIT="5", FMX_FRONTIER_ROUNDS="4", FMX_FRONTIER_ROUNDS_SMALL="4",
               FMX_FRONTIER_CHAIN="2", FMX_FRONTIER_CHAIN_SMALL="2", FMX_FRONTIER_GRAPH="1")     # (and the captured-graph form of a call)
    r = subprocess.run([sys.executable, "-c", _QUEUE_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_regex_device_resident_results():
    """fmx_regex_batch_match_dev leaves the grouped, ordered results and the per-regex counts in HBM: they must be the
    bytes the host form returns -- for a plain batch, for one with a group of more than 1024 results (ordered on the
    host: staged through host memory and written back) and for a DFA batch whose start state is final (a result the
    host adds)."""
    torch = _torch()
    from findex_amd.regex import RegexBatch, RESULT_DTYPE
    bwt, eof, counts = synth_bwt(600_000, 97, 100, 5)            # 4 letters: a..d
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    plain = ["ab[a-c]*d", "abca", "b(a|cd)[ad]*b"] + ["abcd"[i % 4] + "abcd"[(i // 4) % 4] + "c[ab]?d" for i in range(30)]
    big = ["a[ab]*c", "[ab][a-d][a-d][a-d][a-d]"]                # the second: ~2 * 4^4 * ... distinct intervals, one group > 1024
    sets = [[findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in plain],
            [findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in plain + big]]
    d1, d2 = findex_amd.DFA(2), findex_amd.DFA(4)
    d1.addLink(0, 1, ord("a"))
    d1.finishStates = {0, 1}                                     # final start state: the host adds (0, 0, n)
    for f, t_, ch in [(0, 1, ord("a")), (1, 2, ord("b")), (2, 2, ord("b")), (2, 3, ord("c"))]:
        d2.addLink(f, t_, ch)
    d2.finishStates = {3}
    d1.compileBuckets()
    d2.compileBuckets()
    sets.append([d1, d2])
    seen_big = False
    for trees in sets:
        rb = RegexBatch(hip, trees)
        out, per = rb.match_raw(max_steps=12, cap=1 << 20)
        seen_big = seen_big or (per.size and int(per.max()) > 1024)
        cap = 1 << 20
        d_out = torch.zeros(cap * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        d_per = torch.full((len(trees),), 7, dtype=torch.int32, device="cuda")
        n = rb.match_dev(d_out.data_ptr(), cap, d_per.data_ptr(), max_steps=12)
        torch.cuda.synchronize()
        got = np.frombuffer(d_out[: n * RESULT_DTYPE.itemsize].cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
        assert n == out.size and got.tobytes() == out.tobytes()
        assert np.array_equal(d_per.cpu().numpy().astype(np.uint32), per)
    assert seen_big


def test_match_batch_sharded_device_exchange():
    """distributed.match_batch_sharded with the nccl backend keeps the result lists in HBM until after the exchange
    (fmx_regex_batch_match_dev + all_gather on device tensors): a one-rank RCCL group on this GPU must give what the
    plain resident batch gives."""
    torch = _torch()
    import torch.distributed as dist
    from findex_amd import distributed as D
    from findex_amd.regex import RegexBatch
    bwt, eof, counts = synth_bwt(300_000, 97, 100, 8)
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    res = ["ab[a-c]*d", "a[ab]*c", "abca"] + ["abcd"[i % 4] + "abcd"[(i // 4) % 4] + "c[ab]?d" for i in range(20)]
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in res]
    want, _ = RegexBatch(hip, trees).match_raw(max_steps=20)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        got = D.match_batch_sharded(hip, trees, max_steps=20)
    finally:
        dist.destroy_process_group()
    assert got.tobytes() == want.tobytes() and want.size > 100


def test_allgather_dev_rccl_in_the_library():
    """fmx_comm_* / fmx_allgather_dev / fmx_gather_dev: the exchange as an RCCL call inside libfmx.so (what a JVM caller
    uses).  One GPU here, so the communicators have one rank -- created both ways (ncclCommInitAll over the handles of one
    process; unique id + ncclCommInitRank as one process per GPU does) -- and the gather of the intervals of a real
    search must reproduce them; with two or more GPUs a second handle joins and every rank receives both slices.  The
    search is NOT synchronised before the exchange: the collective is ordered behind the producer streams by the
    library.  Both payloads: (sp, ep) as 16 bytes per pattern and the packed 8-byte form; both deliveries: to every
    rank and to the root only.  A failed RCCL initialisation is a status, not a crash."""
    import ctypes
    torch = _torch()
    from findex_amd import _lib
    from findex_amd.distributed import pack_intervals_np
    L = _lib.load()
    ndev = min(torch.cuda.device_count(), 2)
    bwt, eof, counts = synth_bwt(200_000, 97, 100, 5)
    hips = [findex_amd.HipFMSearcher.from_mem(bwt, eof, counts, device=d) for d in range(ndev)]
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(6)
    buf, off = pack_patterns(lf_walk_patterns(orc, rng, 1000 * ndev, 6, 0.2, alphabet=[97, 98, 99, 100]))
    wsp, wep, _ = orc.search_batch(buf, off)
    k = 1000
    for hip in hips:
        hip.prepare(ktab=True, jump=True)               # no table is built (and no stream synchronised) by the searches below
    comm = ctypes.c_void_p()
    idxs = (ctypes.c_void_p * ndev)(*[h.handle for h in hips])
    _lib.check(L.fmx_comm_create_all(idxs, ndev, ctypes.byref(comm)))
    nr, nl = ctypes.c_int(), ctypes.c_int()
    _lib.check(L.fmx_comm_info(comm, ctypes.byref(nr), ctypes.byref(nl)))
    assert nr.value == ndev and nl.value == ndev
    sends, recvs, packs, precvs, streams, keep = [], [], [], [], [], []
    pw = hips[0].packed_words(k, 8)
    for d, hip in enumerate(hips):                      # rank d searches patterns [d*k, (d+1)*k) on its own GPU
        dev = torch.device("cuda", d)
        with torch.cuda.device(dev):
            lo = int(off[d * k])
            d_pat = torch.from_numpy(buf[lo:int(off[(d + 1) * k])].copy()).to(dev)
            d_off = torch.from_numpy((off[d * k:(d + 1) * k + 1] - np.uint64(lo)).astype(np.int64)).to(dev)
            both = torch.empty(2 * k, dtype=torch.int64, device=dev)
            pk = torch.empty(pw, dtype=torch.int64, device=dev)
            recvs.append(torch.zeros(2 * k * ndev, dtype=torch.int64, device=dev))
            precvs.append(torch.zeros(pw * ndev, dtype=torch.int64, device=dev))
            torch.cuda.synchronize(dev)
            st = torch.cuda.Stream(device=dev)
            hip.search_batch_dev(d_pat.data_ptr(), d_off.data_ptr(), both.data_ptr(), both.data_ptr() + 8 * k, k, st.cuda_stream)
            hip.pack_intervals_dev(both.data_ptr(), both.data_ptr() + 8 * k, k, pk.data_ptr(), escape_cap=8, stream=st.cuda_stream)
            sends.append(both)
            packs.append(pk)
            streams.append(st)
            keep += [d_pat, d_off]
    sp_ = (ctypes.c_void_p * ndev)(*[t.data_ptr() for t in sends])
    rp_ = (ctypes.c_void_p * ndev)(*[t.data_ptr() for t in recvs])
    pp_ = (ctypes.c_void_p * ndev)(*[t.data_ptr() for t in packs])
    pr_ = (ctypes.c_void_p * ndev)(*[t.data_ptr() for t in precvs])
    st_ = (ctypes.c_void_p * ndev)(*[s_.cuda_stream for s_ in streams])
    _lib.check(L.fmx_allgather_dev(comm, sp_, rp_, 16 * k, st_))          # ordered behind the searches by the library
    for d in range(ndev):
        got = recvs[d].cpu().numpy().astype(np.uint64).reshape(ndev, 2, k)
        for r in range(ndev):
            assert np.array_equal(got[r, 0], wsp[r * k:(r + 1) * k]) and np.array_equal(got[r, 1], wep[r * k:(r + 1) * k])
    root = ndev - 1
    _lib.check(L.fmx_gather_dev(comm, pp_, pr_, 8 * pw, root, st_))       # the packed form, delivered to the root only
    got = precvs[root].cpu().numpy().astype(np.uint64).reshape(ndev, pw)
    for r in range(ndev):
        assert np.array_equal(got[r][:k + 1], pack_intervals_np(wsp[r * k:(r + 1) * k], wep[r * k:(r + 1) * k], 8)[:k + 1])      # (no escapes: the list's slots are not written)
        usp, uep = hips[0].unpack_intervals(got[r], k, 8)
        assert np.array_equal(usp, wsp[r * k:(r + 1) * k]) and np.array_equal(uep, wep[r * k:(r + 1) * k])
    for d in range(ndev):
        if d != root:
            assert int(precvs[d].abs().sum().item()) == 0                 # nothing lands on the other ranks
    assert L.fmx_gather_dev(comm, pp_, pr_, 8 * pw, ndev, st_) == 3       # no such root
    _lib.check(L.fmx_comm_free(comm))
    # the one-process-per-GPU bootstrap with a single rank: unique id -> ncclCommInitRank
    uid = ctypes.create_string_buffer(128)
    _lib.check(L.fmx_comm_unique_id(uid))
    _lib.check(L.fmx_comm_create_rank(hips[0].handle, 1, 0, uid, ctypes.byref(comm)))
    recvs[0].zero_()
    torch.cuda.synchronize()
    one_s = (ctypes.c_void_p * 1)(sends[0].data_ptr())
    one_r = (ctypes.c_void_p * 1)(recvs[0].data_ptr())
    _lib.check(L.fmx_allgather_dev(comm, one_s, one_r, 16 * k, None))     # idle buffers: no producer streams
    assert torch.equal(recvs[0][: 2 * k], sends[0])
    _lib.check(L.fmx_comm_free(comm))
    # refused: two handles on one device (RCCL has one rank per GPU)
    twice = (ctypes.c_void_p * 2)(hips[0].handle, hips[0].handle)
    assert L.fmx_comm_create_all(twice, 2, ctypes.byref(comm)) == 3
    # a failed initialisation returns FMX_ERR_HIP with a message and leaves nothing behind: injected, in a process of its own,
    # through the tests' twin of the library -- the product library has no such switch (findex_amd/build.py, fmx_comm.cpp)
    from findex_amd import build as fbuild
    assert b"FMX_COMM_FAIL_INIT" not in open(_lib.LIB_PATH, "rb").read()
    code = (
        "import ctypes, os, torch\n"
        "import findex_amd\n"
        "from findex_amd import _lib\n"
        "L = _lib.load()\n"
        "hip = findex_amd.HipFMSearcher(%r)\n"
        "uid = ctypes.create_string_buffer(128)\n"
        "_lib.check(L.fmx_comm_unique_id(uid))\n"
        "comm = ctypes.c_void_p()\n"
        "os.environ['FMX_COMM_FAIL_INIT'] = '1'\n"
        "assert L.fmx_comm_create_rank(hip.handle, 1, 0, uid, ctypes.byref(comm)) == 5 and not comm.value\n"
        "assert b'ncclCommInitRank' in L.fmx_last_error()\n"
        "idxs = (ctypes.c_void_p * 1)(hip.handle)\n"
        "assert L.fmx_comm_create_all(idxs, 1, ctypes.byref(comm)) == 5 and not comm.value\n"
        "del os.environ['FMX_COMM_FAIL_INIT']\n"
        "_lib.check(L.fmx_comm_create_all(idxs, 1, ctypes.byref(comm)))     # and the library still makes communicators\n"
        "_lib.check(L.fmx_comm_free(comm))\n"
        "print('injected ok')\n" % os.path.join(fbuild.ROOT, "tests", "golden", "testdata", "words.bwt"))
    import subprocess
    import sys
    env = dict(os.environ, FMX_LIB=fbuild.OUT_FAULTS, PYTHONPATH=fbuild.ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "injected ok" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


def _nccl_rank(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from findex_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        bwt, eof, counts = synth_bwt(300_000, 97, 100, 8)
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts, device=rank)
        res = ["ab[a-c]*d", "a[ab]*c", "abca"] + ["abcd"[i % 4] + "abcd"[(i // 4) % 4] + "c[ab]?d" for i in range(20)]
        trees = [findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in res]
        got = D.match_batch_sharded(hip, trees, weights=[len(r) for r in res], max_steps=20)
        rng = np.random.default_rng(3)
        orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
        buf, off = pack_patterns(lf_walk_patterns(orc, rng, 500, 7, 0.2, alphabet=[97, 98, 99, 100]))
        sp, ep = D.search_batch_sharded(hip, buf, off)
        q.put((rank, got.tobytes(), sp.tobytes(), ep.tobytes()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_search_and_match_two_ranks_rccl():
    """The N > 1 path on real RCCL: two processes, one GPU each, a replica of the index per rank, the regex batch
    cut by weight (rank 1's regex ids start past 0), result lists of different lengths exchanged from HBM
    (exchange_result_words), pattern intervals all-gathered.  Every rank must hold what one process computes.
    Skipped on a box with one GPU (the same code path runs on CPU tensors over gloo in tests/test_distributed_cpu.py)."""
    torch = _torch()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import socket
    import torch.multiprocessing as mp
    from findex_amd.regex import RegexBatch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_rank, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=240) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    bwt, eof, counts = synth_bwt(300_000, 97, 100, 8)
    hip, orc = pair_from_mem(bwt, eof, counts)
    res = ["ab[a-c]*d", "a[ab]*c", "abca"] + ["abcd"[i % 4] + "abcd"[(i // 4) % 4] + "c[ab]?d" for i in range(20)]
    want, _ = RegexBatch(hip, [findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in res]).match_raw(max_steps=20)
    rng = np.random.default_rng(3)
    buf, off = pack_patterns(lf_walk_patterns(orc, rng, 500, 7, 0.2, alphabet=[97, 98, 99, 100]))
    wsp, wep, _ = orc.search_batch(buf, off)
    for rank, res_b, sp_b, ep_b in got:
        assert res_b == want.tobytes() and sp_b == wsp.tobytes() and ep_b == wep.tobytes()


# ---------------------------------------------------------------- the reference's other two engines
class _OIdx:
    def __init__(self, sa):
        self.sa, self.n = sa, sa.n

    def getPrevRange(self, sp, ep, c):
        return self.sa.getPrevRange(sp, ep, c)


def test_thompson_engine(testdata):
    """REParser.createNFA + REParser.matchSA (re2.scala:264-334,568-693) on the frontier kernel:
    the reference's vectors (T/REParser.scala:219-234,292-307) and the oracle's restatement."""
    from oracle import engines as E
    P = findex_amd.REParser
    hip, orc = pair_from_mem(*bwt_of_text(b"mmabcacamabbbca"[::-1]))
    for post in ("ma.b.", "ba|c.", "ab|*c.", "a?b.c."):
        got = [r.key() for r in P.matchSA(P.createNFA(P.post2re(post)), hip)]
        assert got == sorted(E.nfa_matchSA(E.createNFA(R.post2re(post)), _OIdx(orc))), post
    got = P.matchSA(P.createNFA(P.post2re("ma.b.")), hip)
    assert len(got) == 1 and got[0].cnt == 2 and got[0].len == 3          # "[2 Results] bam"
    hip, orc = pair_from_files(testdata, "test1024.cmp", False)
    res = P.matchSA(P.createNFA(P.post2re("ba|d|e|c.")), hip)
    assert {str(r) for r in res} == {"ec", "dc", "[2 Results] ac", "bc"}   # T/REParser.scala:303-305
    hip, orc = pair_from_files(testdata, "words", True)
    for re in ("th(e|a)", "co(m|n)+e", "x?yz", "q\\wk", "ab(cd|ef)+gh", "z(a|e)*b", "e\\d", "un..ed"):
        got = [r.key() for r in P.matchSA(P.createNFA(P.re2post(re, lineOnly=True)), hip)]
        want = sorted(E.nfa_matchSA(E.createNFA(R.re2post(re, True)), _OIdx(orc)))
        assert got == want, re
    # maxLength: elements of length >= maxLength are not expanded (re2.scala:645)
    nfa = P.createNFA(P.re2post("s(a|e|i|o|u)+t"))
    got = [r.key() for r in P.matchSA(nfa, hip, maxLength=4)]
    assert got == sorted(E.nfa_matchSA(E.createNFA(R.re2post("s(a|e|i|o|u)+t")), _OIdx(orc), maxLength=4))
    for bad in ("[ab]c", "a*", "(a|b)*"):                                   # scala.MatchError there
        with pytest.raises(findex_amd.MatchError):
            P.createNFA(P.re2post(bad))


def test_dfa_engine(testdata):
    """DFA.compileBuckets + DFA.matchSA (dfa.scala:190-213,231-289): T/dfa.scala:110-122 and the
    oracle's restatement, incl. the rule that multi-character buckets do not expand."""
    from oracle import engines as E

    def both(nstates, links, finish):
        a, b = findex_amd.DFA(nstates), E.DFA(nstates)
        for f, t_, ch in links:
            a.addLink(f, t_, ch)
            b.addLink(f, t_, ch)
        a.finishStates = set(finish)
        b.finishStates = set(finish)
        a.compileBuckets()
        b.compileBuckets()
        return a, b

    hip, orc = pair_from_mem(*bwt_of_text(b"mmabcacadabbbca"[::-1]))
    links = [(0, 1, ord("a")), (1, 2, ord("b")), (2, 2, ord("b")), (2, 3, ord("c"))]
    a, b = both(4, links, {3})
    got = a.matchSA(hip)
    assert [r.key() for r in got] == sorted(b.matchSA(_OIdx(orc))) and len(got) == 2
    hip, orc = pair_from_files(testdata, "words", True)
    cases = [
        (4, links, {3}),
        (4, [(0, 1, ord(ch)) for ch in "cdfmkl"] + links[1:], {3}),            # c-d and k-m are buckets: only 'f' starts
        (3, [(0, 1, ord("q")), (1, 2, ord("u")), (2, 2, ord("e")), (2, 1, ord("i"))], {1, 2}),
        (2, [(0, 1, ord("z"))], {0, 1}),                                       # final start state: (0, 0, n)
    ]
    for nst, lk, fin in cases:
        a, b = both(nst, lk, fin)
        assert [r.key() for r in a.matchSA(hip)] == sorted(b.matchSA(_OIdx(orc))), (nst, lk)


# ---------------------------------------------------------------- second layout: BWT bytes + checkpoints
@pytest.fixture
def bytes_layout(request):
    """The compact layout, in both of its checkpoint forms: absolute 32-bit counts (whenever every symbol occurs
    fewer than 2^32 times) and counts relative to 64-bit superblock totals (forced here; larger counts need it)."""
    findex_amd.set_layout("bytes")
    findex_amd.set_checkpoints(getattr(request, "param", "auto"))
    yield
    findex_amd.set_layout("auto")
    findex_amd.set_checkpoints("auto")


def test_bytes_layout_fixtures(testdata, bytes_layout, tmp_path):
    """The compact layout (two lines per rank query, for sigma*n/8 > HBM) must answer exactly like
    the one-hot layout: every primitive against the oracle on the reference's fixtures."""
    for name, be in (("test1024.cmp", False), ("test.cmp", False), ("words", True)):
        hip, orc = pair_from_files(testdata, name, be)
        assert hip.stats()["layout"] == 1
        rng = np.random.default_rng(5)
        present = [c for c in range(256) if orc.occ(c, orc.n - 1) > 0]
        check_occ(hip, orc, rng, 4000, present + [0, 1, 255])
        pats = lf_walk_patterns(orc, rng, 300, 9, 0.15, alphabet=present) + [b"", b"\x00", b"\xff\x80", bytes([present[0]])]
        assert check_search(hip, orc, pats) > 150
        rows = rng.integers(0, orc.n, 3000).astype(np.uint64)
        rows[:3] = [0, orc.eof, orc.n - 1]
        assert np.array_equal(hip.psi_batch(rows), orc.fm()[rows.astype(np.int64)].astype(np.uint64))
        _, e1 = hip.lf_walk_batch(rows[:500], 1, want_bytes=False)
        assert np.array_equal(e1, np.array([orc.getPrevI(int(r)) for r in rows[:500]], dtype=np.uint64))
        assert hip.nextSubstr(int(orc.eof), 40) == orc.nextSubstr(int(orc.eof), 40)
        assert hip.getIntervalPrevRange(0, orc.n, 0, 255) == orc.getIntervalPrevRange(0, orc.n, 0, 255)
        a, b = str(tmp_path / "h.fm"), str(tmp_path / "o.fm")
        hip.write_fm(a)
        orc.write_fm(b)
        assert open(a, "rb").read() == open(b, "rb").read()
        trees = [findex_amd.ReTree(findex_amd.REParser.re2post(re)) for re in REGEXES[:8]]
        got = findex_amd.ReTree.matchSA_batch(hip, trees, max_steps=1 << 20)
        for re, g in zip(REGEXES[:8], got):
            assert [r.key() for r in g] == oracle_results(orc, re)[0], re
        ref = findex_amd.ReTree.matchSA_batch(hip, trees, mode="reference", maxBranching=16, maxIterations=50)
        for re, g in zip(REGEXES[:8], ref):
            assert [r.key() for r in g] == orc.match_tables(R.ReTree(R.re2post(re)).tables(), 16, 50)[0], re


@pytest.mark.parametrize("bytes_layout", ["auto", "superblock"], indirect=True)
@pytest.mark.parametrize("n,lo,hi", [(1, 1, 1), (127, 1, 4), (128, 1, 4), (129, 1, 4), (4_194_304 + 5, 1, 128),
                                     (300_007, 1, 255)])
def test_bytes_layout_synthetic(n, lo, hi, bytes_layout):
    """Block (128) and superblock (2^22 positions) edges of the compact layout."""
    rng = np.random.default_rng(n)
    for eof in sorted({0, n // 3, n - 1}):
        bwt, eof, counts = synth_bwt(n, lo, hi, n + 1, eof=eof)
        hip, orc = pair_from_mem(bwt, eof, counts)
        assert hip.stats()["layout"] == 1
        syms = list(range(lo, hi + 1))
        check_occ(hip, orc, rng, 3000, syms + [0, 255])
        m = 1 if n < 10 else (12 if hi - lo < 8 else 5)
        check_search(hip, orc, lf_walk_patterns(orc, rng, 300, m, 0.1, alphabet=syms) + [b"", b"\x00"])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("checkpoints", ["auto", "superblock"])
def test_bytes_layout_c5_shape(checkpoints):
    """BASELINE config C5's index at its own size: n = 2^34 rows (a 16 GiB BWT), sigma = 128, the bytes + checkpoints
    layout the library picks by itself (the one-hot vectors would need 314 GB), 80 GiB resident, and C5's per-GPU
    batch of 1M x 24-character patterns.  Run in both checkpoint forms (absolute 32-bit counts; relative counts +
    64-bit superblocks).  The reference cannot represent such an index (Int positions) and the oracle's 32-bit lists
    stop at 2^32, so the checks are size-independent properties:
      * occ(c, i) equals a brute-force count of the device BWT prefix (torch), also past row 2^32 and 2^33,
      * the symbol totals add up to n - 1 and the column sums to i + 1,
      * 1M LF-walk patterns hit and each hit interval contains the row its walk ended on; misses made by replacing
        one byte report sp == ep; ragged prefixes of the patterns hit intervals that contain the full patterns',
      * getPrevRange over the alphabet partitions an interval (getIntervalPrevRange)."""
    import torch
    assert torch.cuda.is_available()
    n = 1 << 34
    g = torch.Generator(device="cuda")
    g.manual_seed(55)
    bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
    for a in range(0, n, 1 << 28):
        b = min(n, a + (1 << 28))
        bwt[a:b] = torch.randint(1, 129, (b - a,), generator=g, device="cuda", dtype=torch.uint8)
    eof = n // 3
    torch.cuda.synchronize()
    findex_amd.set_checkpoints(checkpoints)        # the layout itself is left to the library ("auto")
    try:
        hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
    finally:
        findex_amd.set_checkpoints("auto")
    st = hip.stats()
    assert st["layout"] == 1 and st["block_bytes"] == 132 and st["index_bytes"] > 75 << 30
    rng = np.random.default_rng(2)
    qs_i = np.concatenate([rng.integers(0, n, 4), [0, eof, (1 << 32) - 1, 1 << 32, (1 << 33) + 12345, n - 1]]).astype(np.int64)
    qs_c = rng.integers(1, 129, qs_i.size).astype(np.uint8)
    got = hip.occ_batch(qs_c, qs_i)
    for c, i, gv in zip(qs_c, qs_i, got):
        want = 0
        for a in range(0, int(i) + 1, 1 << 28):
            b = min(int(i) + 1, a + (1 << 28))
            want += int((bwt[a:b] == int(c)).sum().item())
        if eof <= i and int(bwt[eof].item()) == int(c):
            want -= 1
        assert int(gv) == want, (c, i)
    col = hip.occ_batch(np.arange(0, 129, dtype=np.uint8), np.full(129, n - 1, dtype=np.int64))
    assert int(col.sum()) == n
    for i in (0, eof, (1 << 33) + 7):
        assert int(hip.occ_batch(np.arange(0, 129, dtype=np.uint8), np.full(129, i, dtype=np.int64)).sum()) == i + 1
    # the CPU oracle at this size: the sampled-checkpoint structure over this very BWT (the inverted lists stop at 2^32
    # rows; oracle.SampledFMSearcher computes the same occ / search: tests/test_oracle_kat.py) -- 48 GiB of host memory, once
    orc = None
    if checkpoints == "auto":
        import bench
        orc, t_orc = bench.oracle_index(torch, bwt, eof, bench.effective_cores(), 0)
        print("[c5_shape] oracle on the run's own index: %s, %.0fs" % (getattr(orc, "kind", None), t_orc))
    del bwt
    k, m = 1_000_000, 24
    rows = rng.integers(0, n, k).astype(np.uint64)
    b, end = hip.lf_walk_batch(rows, m)
    pats = np.ascontiguousarray(b[:, ::-1])
    off = np.arange(k + 1, dtype=np.uint64) * m
    hip.stats_reset()
    sp, ep = hip.search_batch(pats.reshape(-1), off)
    assert (sp < ep).all() and (sp <= end).all() and (end < ep).all()
    assert hip.stats()["backward_steps"] == k * m                 # every step of a hit pattern is executed
    # misses: one byte replaced by a symbol outside the alphabet -> the loop's final values are an empty interval
    bad = pats[:50_000].copy()
    bad[np.arange(50_000), rng.integers(0, m, 50_000)] = 200
    bsp, bep = hip.search_batch(bad.reshape(-1), off[:50_001])
    assert (bsp == bep).all()
    # misses INSIDE the alphabet: the failing step is a rank query on rows past 2^32 (k_search_rows<2> parks the pattern,
    # k_search_defer<true, bytes> finds the reference loop's (sp, ep) there) -- values and step counts against the loop
    # restated over getPrevRange
    inalpha = pats[:100_000].copy()
    inalpha[np.arange(100_000), rng.integers(0, m, 100_000)] = rng.integers(1, 129, 100_000).astype(np.uint8)
    hip.stats_reset()
    isp, iep = hip.search_batch(inalpha.reshape(-1), off[:100_001])
    isteps = hip.stats()["backward_steps"]
    wsp, wep, wsteps = reference_loop_by_prev_range(hip, inalpha)
    assert np.array_equal(isp, wsp) and np.array_equal(iep, wep) and isteps == int(wsteps.sum())
    assert int((wsp == wep).sum()) > 90_000 and int((wsp[wsp == wep] > np.uint64(1 << 32)).sum()) > 50_000
    if orc is not None:      # ... and bit for bit against the oracle: 100k hits, the 100k in-alphabet misses, random occ operands
        import bench
        cores = bench.effective_cores()
        osp, oep, ost = orc.search_batch(inalpha.reshape(-1), off[:100_001], threads=cores)
        assert np.array_equal(isp, osp) and np.array_equal(iep, oep) and isteps == int(ost.sum())
        osp, oep, ost = orc.search_batch(pats[:100_000].reshape(-1), off[:100_001], threads=cores)
        assert np.array_equal(sp[:100_000], osp) and np.array_equal(ep[:100_000], oep) and int(ost.sum()) == 100_000 * m
        qc = rng.integers(0, 130, 2000).astype(np.uint8)
        qi = np.concatenate([rng.integers(-1, n, 1990), [-1, 0, eof, (1 << 32) - 1, 1 << 32, (1 << 33) + 5, n - 2, n - 1, n, n + 9]]).astype(np.int64)
        assert hip.occ_batch(qc, qi).astype(np.int64).tolist() == [orc.occ(int(c), int(i)) for c, i in zip(qc, qi)]
        orc.close()
    # the same patterns cut into ragged pieces: a prefix of a hit pattern hits an interval that contains the full hit
    cut = rng.integers(1, m + 1, 100_000)
    pieces = [pats[j, :cut[j]].tobytes() for j in range(100_000)]
    pbuf, poff = pack_patterns(pieces)
    psp, pep = hip.search_batch(pbuf, poff)
    assert (psp <= sp[:100_000]).all() and (ep[:100_000] <= pep).all() and (psp < pep).all()
    full = cut == m
    assert np.array_equal(psp[full], sp[:100_000][full]) and np.array_equal(pep[full], ep[:100_000][full])
    for a, bnd in ((0, n), (int(sp[0]), int(ep[0])), ((1 << 33) - 5, (1 << 33) + 100000)):
        parts = hip.getIntervalPrevRange(a, bnd, 0, 255)
        assert sum(e - s for s, e in parts) == bnd - a
        srt = sorted(parts)
        assert all(srt[j][1] <= srt[j + 1][0] for j in range(len(srt) - 1))


# ---------------------------------------------------------------- full-size properties (on device)
def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def oracle_at_size(torch, bwt_dev, eof, what):
    """The CPU oracle over the SAME device BWT a full-size test uses (bytes copied to the host, inverted position
    lists sorted on all the box's cores: 6 n bytes of host memory, 32-bit entries so n <= 2^32) -- or None, with the
    reason printed, when the host cannot hold it: the size-independent properties are then all that is checked."""
    import bench
    cores = bench.effective_cores()
    orc, t_build = bench.oracle_index(torch, bwt_dev, eof, cores, 0)
    if orc is None:
        print("[%s] FALLBACK: properties only (the host cannot hold the oracle's index at n=%d)" % (what, bwt_dev.numel()))
        return None, cores
    print("[%s] oracle built on the run's own index (n=%d) in %.1fs on %d cores: comparing bit for bit" % (what, bwt_dev.numel(), t_build, cores))
    return orc, cores


@pytest.mark.timeout(900)
@pytest.mark.parametrize("log2n,extra,sigma", [(28, 0, 4), (32, 12345, 16), (32, 0, 128)])
def test_full_size_properties(log2n, extra, sigma):
    """BASELINE-size indexes -- C2 (256 MB, sigma=4), n > 2^32 to cross the 32-bit line, and C3's own
    instantiation (n = 2^32 exactly, sigma = 128: the one-hot layout with 32-bit counts, 77 GiB, a
    1M x 32 pattern batch).  No CPU oracle is built at these sizes inside pytest, so check size-independent
    properties:
      * occ(c, i) equals a brute-force count of the device BWT prefix (torch),
      * occ(c, n-1) equals the symbol total, sum_c occ(c, i) == i + 1 - [i >= eof],
      * LF-walk patterns hit, and the hit interval contains the row the walk ended on,
      * the step partition: intervals of getIntervalPrevRange are disjoint and sum to ep - sp."""
    torch = _torch()
    n = (1 << log2n) + extra
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + log2n)
    bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        bwt[a:b] = torch.randint(1, sigma + 1, (b - a,), generator=g, device="cuda", dtype=torch.uint8)
    eof = n // 3
    torch.cuda.synchronize()
    hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
    assert hip.n == n
    rng = np.random.default_rng(9)
    # brute-force rank on a few (c, i)
    qs_i = np.concatenate([rng.integers(0, n, 6), [0, eof - 1, eof, eof + 1, n - 1]]).astype(np.int64)
    qs_c = rng.integers(1, sigma + 1, qs_i.size).astype(np.uint8)
    got = hip.occ_batch(qs_c, qs_i)
    for c, i, gv in zip(qs_c, qs_i, got):
        want = 0
        for a in range(0, int(i) + 1, step):
            b = min(int(i) + 1, a + step)
            want += int((bwt[a:b] == int(c)).sum().item())
        if eof <= i and int(bwt[eof].item()) == int(c):
            want -= 1
        assert int(gv) == want, (c, i)
    # totals and the column sum
    totals = hip.occ_batch(np.arange(1, sigma + 1, dtype=np.uint8), np.full(sigma, n - 1, dtype=np.int64))
    assert int(totals.sum()) == n - 1
    for i in (0, 12345, eof, n - 2):
        col = hip.occ_batch(np.arange(0, sigma + 1, dtype=np.uint8), np.full(sigma + 1, i, dtype=np.int64))
        assert int(col.sum()) == i + 1
    # LF-walk patterns must hit and contain their end row
    k, m = (1_000_000, 32) if sigma == 128 else (20000, 16 if sigma == 4 else 12)
    if sigma == 128:
        st = hip.stats()
        assert st["layout"] == 0 and st["n_symbols"] == 128 and st["index_bytes"] > 70 << 30
    rows = rng.integers(0, n, k).astype(np.uint64)
    b, end = hip.lf_walk_batch(rows, m)
    pats = np.ascontiguousarray(b[:, ::-1])
    off = (np.arange(k + 1, dtype=np.uint64) * m)
    sp, ep = hip.search_batch(pats.reshape(-1), off)
    assert (sp < ep).all() and (sp <= end).all() and (end < ep).all()
    # the oracle at the config's own size: the reference algorithm (inverted lists + binary-search occ,
    # bwtmerger.scala:354-375, findex.scala:15-31) over this very BWT, 100k+ patterns (hits, and misses made by
    # replacing one byte) and random occ / getPrevRange operands, bit for bit, executed steps included
    orc, cores = oracle_at_size(torch, bwt, eof, "full_size[%d-%d-%d]" % (log2n, extra, sigma)) if n <= (1 << 32) else (None, 0)
    if orc is not None:
        ks = min(k, 120_000)
        sample = pats[:ks].copy()
        mut = rng.random(ks) < 0.2
        sample[mut, rng.integers(0, m, int(mut.sum()))] = rng.integers(1, sigma + 1, int(mut.sum())).astype(np.uint8)
        s_off = np.arange(ks + 1, dtype=np.uint64) * m
        hip.stats_reset()
        gsp, gep = hip.search_batch(sample.reshape(-1), s_off)
        wsp, wep, wsteps = orc.search_batch(sample.reshape(-1), s_off, threads=cores)
        assert np.array_equal(gsp, wsp) and np.array_equal(gep, wep)
        assert hip.stats()["backward_steps"] == int(wsteps.sum())
        assert 0 < int((wsp >= wep).sum()) < ks          # both hits and misses were compared
        qc = rng.integers(0, sigma + 2, 50_000).astype(np.uint8)
        qi = rng.integers(-1, n, 50_000).astype(np.int64)
        assert np.array_equal(hip.occ_batch(qc, qi).astype(np.int64), orc.occ_batch(qc, qi))
        a_sp = rng.integers(0, n, 50_000).astype(np.uint64)
        a_ep = np.minimum(a_sp + rng.integers(0, 1 << 20, 50_000).astype(np.uint64), np.uint64(n))
        g1, g2 = hip.prev_range_batch(a_sp, a_ep, qc)
        w1, w2 = orc.prev_range_batch(a_sp, a_ep, qc)
        assert np.array_equal(g1, w1) and np.array_equal(g2, w2)
        orc.close()
    # hits and in-alphabet misses (one byte replaced), (sp, ep) and executed steps against the reference loop restated
    # over getPrevRange calls -- the value check that still works where the oracle's 32-bit lists stop (n > 2^32)
    kk = min(k, 100_000)
    mixed = pats[:kk].copy()
    mut = rng.random(kk) < 0.5
    mixed[mut, rng.integers(0, m, int(mut.sum()))] = rng.integers(1, sigma + 1, int(mut.sum())).astype(np.uint8)
    hip.stats_reset()
    msp, mep = hip.search_batch(mixed.reshape(-1), np.arange(kk + 1, dtype=np.uint64) * m)
    msteps = hip.stats()["backward_steps"]
    wsp, wep, wsteps = reference_loop_by_prev_range(hip, mixed)
    assert np.array_equal(msp, wsp) and np.array_equal(mep, wep) and msteps == int(wsteps.sum())
    assert 0 < int((wsp == wep).sum()) < kk
    # partition property of one step
    for a, bnd in ((0, n), (int(sp[0]), int(ep[0])), (n // 2, n // 2 + 100000)):
        parts = hip.getIntervalPrevRange(a, bnd, 0, 255)
        assert sum(e - s for s, e in parts) == bnd - a
        srt = sorted(parts)
        assert all(srt[j][1] <= srt[j + 1][0] for j in range(len(srt) - 1))
    if sigma == 128:
        # ---- the budget at C3's own size: by default the derived tables take 4 + 32 + 128 GiB beside the
        # 81 GiB index and leave ~27 GiB of the device; under a budget of 110 GiB the row jump table is built as single
        # entries (4 + 32 + 64 GiB), a caller allocates 40 GiB afterwards, and the answers are what they were
        st = hip.stats()
        assert st["jump_bytes"] == 32 * n and st["row_bytes"] == 8 * n and st["tables_held_bytes"] > 160 << 30
        hip.drop_tables(jump=True, frontier=True, ktab=True)
        assert hip.stats()["tables_held_bytes"] == 0
        hip.prepare(ktab=True, jump=True, budget_bytes=110 << 30)
        st = hip.stats()
        assert st["jump_bytes"] == 16 * n and st["row_bytes"] == 8 * n and st["ktab_k"] == 4
        assert st["tables_held_bytes"] <= 110 << 30 and st["hbm_free_after_tables"] >= 40 << 30, st
        room = torch.empty(40 << 30, dtype=torch.uint8, device="cuda")
        room[:: 1 << 20] = 1
        torch.cuda.synchronize()
        sp2, ep2 = hip.search_batch(pats.reshape(-1), off)
        assert np.array_equal(sp2, sp) and np.array_equal(ep2, ep)
        print("[full_size C3] tables under a 110 GiB budget: %.1f GiB held, built in %.0f ms, %.1f GiB free afterwards"
              % (st["tables_held_bytes"] / 2**30, st["tables_build_ms"], st["hbm_free_after_tables"] / 2**30))
        del room


@pytest.mark.timeout(900)
def test_c4_full_size_regex_batch():
    """BASELINE config C4 at its own size: 100k seeded regexes (<= 32 Glushkov positions) over a 2^30-row
    sigma=28 index.  Properties of the answer that need no CPU index of that size: every result interval is
    non-empty and inside [0, n); for sampled results the rows really spell a string the regex matches (text
    extracted with nextSubstr = Psi walks, checked with Python's `re`); the per-regex counts add up; the same
    BWT opened in the bytes layout gives the identical result list; and the literal-only regexes of the batch give
    exactly what fmx_search_batch gives for the same strings."""
    import re as pyre
    import sys
    torch = _torch()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import regex_workload
    n, k = 1 << 30, 100_000
    g = torch.Generator(device="cuda")
    g.manual_seed(0xF1DE0004)
    alpha = torch.tensor([ord(c) for c in regex_workload.ALPHABET], dtype=torch.uint8, device="cuda")
    bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
    for a in range(0, n, 1 << 28):
        b = min(n, a + (1 << 28))
        bwt[a:b] = alpha[torch.randint(0, alpha.numel(), (b - a,), generator=g, device="cuda")]
    torch.cuda.synchronize()
    hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, n // 3, None)
    findex_amd.set_layout("bytes")                 # the same BWT in the compact layout, for the cross-check below
    findex_amd.set_checkpoints("superblock")
    try:
        hip_b = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, n // 3, None)
    finally:
        findex_amd.set_layout("auto")
        findex_amd.set_checkpoints("auto")
    # the bench's regex list (tools/regex_workload.py, seed 4): candidates compiled in one fmx_regex_compile_batch call
    import random
    prng = random.Random(4)
    cand = [regex_workload.gen_one(prng) for _ in range(k + 64)]
    cs = findex_amd.ReTree.compile_batch(cand)
    keep = np.nonzero(cs.ok())[0][:k]
    assert keep.size == k
    res = [cand[i] for i in keep]
    trees = cs.select(keep)
    assert max(len(trees[i].tables()["c"]) for i in range(500)) <= 32
    batch = findex_amd.ReTree.prepare_batch(hip, trees)
    hip.stats_reset()
    out, per = batch.match_raw(max_steps=64)
    st = hip.stats()
    assert out.size > 10_000 and int(per.sum()) == out.size
    assert (out["sp"] < out["ep"]).all() and (out["ep"] <= n).all() and (out["len"] >= 4).all() and (out["len"] <= 64).all()
    assert st["frontier_results"] == out.size and st["frontier_elements"] == st["backward_steps"]
    key = np.stack([out["regex"].astype(np.uint64), out["len"].astype(np.uint64), out["sp"], out["ep"]], axis=1)
    assert (np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0])) == np.arange(out.size)).all()     # canonical order
    rng = np.random.default_rng(1)
    pick = rng.integers(0, out.size, 300)
    for ln in np.unique(out["len"][pick]):                       # one batched extraction per match length
        sel = pick[out["len"][pick] == ln]
        for j, s in zip(sel, hip.nextSubstr_batch(out["sp"][sel], int(ln))):
            assert len(s) == ln and pyre.fullmatch(res[out[j]["regex"]].encode(), s, pyre.S), (res[out[j]["regex"]], s)
    # the oracle at C4's own size: ReTree._matchSA (re2/retree.scala:618-653) in C over this very BWT for the first
    # 12k regexes -- every match up to length 64 as a multiset, and the reference's own answer under its default limits
    # (1024 / 1000) as lists in its own order -- against the two GPU modes, bit for bit
    orc, cores = oracle_at_size(torch, bwt, n // 3, "c4_full_size")
    del bwt
    if orc is not None:
        ks = 12_000
        tables = [trees[i].tables() for i in range(ks)]
        want, pops, _ = orc.match_tables_batch(tables, max_len=64, threads=cores)
        got = out[out["regex"] < ks]
        assert got.size == want.size and all(np.array_equal(got[f], want[f]) for f in ("regex", "len", "sp", "ep"))
        sub = findex_amd.ReTree.prepare_batch(hip, trees.select(np.arange(ks)))
        hip.stats_reset()
        sub_out, _ = sub.match_raw(max_steps=64)
        assert hip.stats()["backward_steps"] == pops and np.array_equal(sub_out, got)
        hip.stats_reset()
        ref_out, ref_per = sub.match_raw(mode="reference", maxBranching=1024, maxIterations=1000)
        wref, rpops, _ = orc.match_tables_batch(tables, maxBranching=1024, maxIterations=1000, threads=cores, ordered=True)
        assert ref_out.size == wref.size and all(np.array_equal(ref_out[f], wref[f]) for f in ("regex", "len", "sp", "ep"))
        assert hip.stats()["backward_steps"] == rpops
        assert np.array_equal(ref_per, np.bincount(wref["regex"], minlength=ks).astype(np.uint32))
        orc.close()
        del sub
    # the other layout (octets, two lines per rank query, 64-bit superblock counts) must give the same list, byte for byte
    assert hip.stats()["layout"] == 0 and hip_b.stats()["layout"] == 1
    out_b, per_b = findex_amd.ReTree.prepare_batch(hip_b, trees).match_raw(max_steps=64)
    assert out_b.size == out.size and all(np.array_equal(out_b[f], out[f]) for f in ("regex", "len", "sp", "ep"))
    assert np.array_equal(per_b, per)
    del hip_b
    # regexes that are plain literals: the frontier's answer is the literal search's answer (the index is over
    # the reversed text and the regex engine walks forward with getPrevRange, so the literal is searched reversed)
    lit = [j for j, re in enumerate(res) if re.isalpha()][:2000]
    assert len(lit) > 100
    pats, off = pack_patterns([res[j].encode()[::-1] for j in lit])
    sp, ep = hip.search_batch(pats, off)
    starts = np.searchsorted(out["regex"], np.arange(k + 1))
    for i, j in enumerate(lit):
        rows = out[starts[j]:starts[j + 1]]
        if sp[i] < ep[i]:
            assert rows.size == 1 and rows[0]["len"] == len(res[j]) and rows[0]["sp"] == sp[i] and rows[0]["ep"] == ep[i]
        else:
            assert rows.size == 0


# ---------------------------------------------------------------- round 4: staged pattern spans, lean batch forms
def _dev_search(hip, torch, buf, off, shift=0, fixed_len=0, packed=False, escape_cap=0):
    """fmx_search_batch_ex_dev on device copies of (buf, off); the pattern bytes start `shift` bytes into their tensor so
    that the buffer the kernel sees is not 16-byte aligned."""
    dev = torch.device("cuda", 0)
    k = (buf.size // fixed_len) if fixed_len else off.size - 1
    d_pat = torch.zeros(buf.size + shift + 64, dtype=torch.uint8, device=dev)
    d_pat[shift:shift + buf.size] = torch.from_numpy(buf).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev) if not fixed_len else None
    words = hip.packed_words(k, escape_cap) if packed else k
    d_sp = torch.zeros(max(words, 1), dtype=torch.int64, device=dev)
    d_ep = torch.zeros(max(k, 1), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    hip.search_batch_ex_dev(d_pat.data_ptr() + shift, d_off.data_ptr() if d_off is not None else 0, d_sp.data_ptr(), d_ep.data_ptr(),
                            k, torch.cuda.current_stream().cuda_stream, fixed_len=fixed_len, packed=packed, escape_cap=escape_cap)
    torch.cuda.synchronize()
    if packed:
        return d_sp.cpu().numpy().astype(np.uint64)
    return d_sp[:k].cpu().numpy().astype(np.uint64), d_ep[:k].cpu().numpy().astype(np.uint64)


@pytest.mark.parametrize("layout", ["onehot", "bytes"])
def test_staged_pattern_spans_edges(layout):
    """k_search4 reads a wave's patterns from LDS, staged as ONE contiguous span of up to 1 KiB per batch (16 patterns; 8
    in the bytes layout), and from global memory when the span is longer: batches on both sides of that limit inside one
    call, spans that end exactly at 1024 bytes behind every alignment of the buffer, patterns that start inside the
    span's first dword, empty patterns between long ones, a last batch with one pattern -- each against the oracle,
    steps counted, with the row tables on and off."""
    torch = _torch()
    findex_amd.set_layout(layout)
    try:
        bwt, eof, counts = synth_bwt(250_000, 1, 6, 41)
        orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
        rng = np.random.default_rng(11)
        per = 8 if layout == "bytes" else 16
        for jump in ("auto", "off"):
            findex_amd.config_set("jump", jump)
            hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
            groups = []
            for m in (1, 3, 4, 5, 31, 32, 33, 60, 63, 64, 65, 70, 128, 200):      # whole batches of one length: spans of per * m bytes
                groups.append(lf_walk_patterns(orc, rng, 3 * per, m, 0.2, alphabet=[1, 2, 3, 4, 5, 6]))
            mixed = []
            for _ in range(40 * per):                                             # ragged: spans between 0 and ~2 KiB
                m = int(rng.choice([0, 1, 2, 3, 7, 16, 40, 64, 90, 130]))
                mixed += lf_walk_patterns(orc, rng, 1, m, 0.2, alphabet=[1, 2, 3, 4, 5, 6]) if m else [b""]
            pats = [p for g in groups for p in g] + mixed + lf_walk_patterns(orc, rng, 1, 9, 0.0)
            buf, off = pack_patterns(pats)
            wsp, wep, wsteps = orc.search_batch(buf, off)
            for shift in (0, 1, 7, 15):
                hip.stats_reset()
                gsp, gep = _dev_search(hip, torch, buf, off, shift=shift)
                assert np.array_equal(gsp, wsp) and np.array_equal(gep, wep), (layout, jump, shift)
                assert hip.stats()["backward_steps"] == int(wsteps.sum())
            # spans that end exactly on the limit: `per` patterns of 1024 / per bytes behind a buffer start of every alignment
            m = 1024 // per
            exact = lf_walk_patterns(orc, rng, 4 * per, m, 0.1, alphabet=[1, 2, 3, 4, 5, 6])
            buf, off = pack_patterns(exact)
            wsp, wep, _ = orc.search_batch(buf, off)
            for shift in range(0, 16, 3):
                gsp, gep = _dev_search(hip, torch, buf, off, shift=shift)
                assert np.array_equal(gsp, wsp) and np.array_equal(gep, wep), (layout, jump, "exact", shift)
            hip.close()
    finally:
        findex_amd.set_layout("auto")
        findex_amd.config_set("jump", "auto")


def test_search_ex_fixed_length_and_packed_forms():
    """fmx_search_batch_ex[_dev]: a batch of equal-length patterns without offsets, and the intervals in the 8-byte form --
    host and device entry points, small (one staged copy) and large (whole arrays) batches, against the plain call; wide
    intervals (empty and one-character patterns over 40 M rows) through the escape list, in place on the device, and
    the overflow status when the list is too short."""
    torch = _torch()
    from findex_amd.distributed import pack_intervals_np, unpack_intervals_np
    bwt, eof, counts = synth_bwt(40_000_000, 1, 2, 8)
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(3)
    for k, m in ((1, 5), (17, 1), (1000, 12), (150_000, 26), (140_000, 3)):
        buf = rng.integers(1, 3, size=k * m, dtype=np.uint8)
        off = np.arange(k + 1, dtype=np.uint64) * np.uint64(m)
        wsp, wep = hip.search_batch(buf, off)
        hip.stats_reset()
        gsp, gep = hip.search_batch_ex(buf, fixed_len=m)
        assert np.array_equal(gsp, wsp) and np.array_equal(gep, wep), (k, m, "fixed")
        dsp, dep = _dev_search(hip, torch, buf, off, shift=5, fixed_len=m)
        assert np.array_equal(dsp, wsp) and np.array_equal(dep, wep), (k, m, "fixed dev")
        cap = 64
        for fixed in (0, m):
            pk = hip.search_batch_ex(buf, off, fixed_len=fixed, packed=True, escape_cap=cap)
            assert pk.size == k + 1 + 2 * cap
            usp, uep = hip.unpack_intervals(pk, k, cap)
            assert np.array_equal(usp, wsp) and np.array_equal(uep, wep), (k, m, fixed, "packed")
            nsp, nep = unpack_intervals_np(pk, k, cap)
            assert np.array_equal(nsp, wsp) and np.array_equal(nep, wep)
        pk = _dev_search(hip, torch, buf, off, fixed_len=m, packed=True, escape_cap=cap)
        assert np.array_equal(pk[:k + 1], pack_intervals_np(wsp, wep, cap)[:k + 1])
    # wide intervals: lengths 0 and 1 among longer ones
    pats = [b"", b"\x01", b"\x02", b"\x01\x02"] * 10 + [bytes(rng.integers(1, 3, 30, dtype=np.uint8)) for _ in range(100)]
    pats = [pats[i] for i in rng.permutation(len(pats))]
    buf, off = pack_patterns(pats)
    k = len(pats)
    wsp, wep = hip.search_batch(buf, off)
    n_wide = int(((wep - wsp) >= np.uint64(0xFFFFFF)).sum())
    assert n_wide == 30
    pk = hip.search_batch_ex(buf, off, packed=True, escape_cap=40)
    assert int(pk[k]) == n_wide
    usp, uep = hip.unpack_intervals(pk, k, 40)
    assert np.array_equal(usp, wsp) and np.array_equal(uep, wep)
    ref = pack_intervals_np(wsp, wep, 40)
    assert np.array_equal(pk[:k + 1], ref[:k + 1])
    assert sorted(zip(pk[k + 1:k + 1 + 2 * n_wide:2].tolist(), pk[k + 2:k + 2 + 2 * n_wide:2].tolist())) == \
        sorted(zip(ref[k + 1:k + 1 + 2 * n_wide:2].tolist(), ref[k + 2:k + 2 + 2 * n_wide:2].tolist()))
    # the device round trip: pack, unpack into fresh arrays
    dev = torch.device("cuda", 0)
    d_sp = torch.from_numpy(wsp.astype(np.int64)).to(dev)
    d_ep = torch.from_numpy(wep.astype(np.int64)).to(dev)
    d_pk = torch.zeros(k + 1 + 80, dtype=torch.int64, device=dev)
    d_a = torch.zeros(k, dtype=torch.int64, device=dev)
    d_b = torch.zeros(k, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    hip.pack_intervals_dev(d_sp.data_ptr(), d_ep.data_ptr(), k, d_pk.data_ptr(), escape_cap=40, stream=st)
    hip.unpack_intervals_dev(d_pk.data_ptr(), k, d_a.data_ptr(), d_b.data_ptr(), escape_cap=40, stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(d_a.cpu().numpy().astype(np.uint64), wsp) and np.array_equal(d_b.cpu().numpy().astype(np.uint64), wep)
    # an escape list that is too short: the count says so, the host decode refuses
    pk = hip.search_batch_ex(buf, off, packed=True, escape_cap=7)
    assert int(pk[k]) == n_wide
    with pytest.raises(findex_amd.FmxError) as e:
        hip.unpack_intervals(pk, k, 7)
    assert e.value.code == 9
    with pytest.raises(OverflowError):
        unpack_intervals_np(pk, k, 7)
    hip.close()


def test_text_index_parity_literals_and_regexes():
    """An index over a TEXT with natural repeats (tools/text_bwt.py: words drawn with replacement, suffix-sorted on the
    device, reversed like findex's) -- where intervals stay wide for many characters and no frontier needs a length
    cap: the generator's BWT equals the naive sort's on a small text; on 2^20 bytes, literal patterns (stretches of the
    text -- REVERSED: SuffixAlgo.search on an index over reverse(T) finds p in reverse(T), SURVEY section 0.2 -- a fifth of
    them with one byte replaced; ragged lengths) and the text workload's regexes (literals cut from the text, read
    forward: the regex engines walk a regex forward with getPrevRange) against the oracle, bit for bit, steps
    counted, no truncation."""
    import random
    import sys
    torch = _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import regex_workload
    import text_bwt
    dev = torch.device("cuda", 0)
    small = text_bwt.make_text(torch, 3000, 5, dev)
    sb, se = text_bwt.bwt_of_reversed_text(torch, small)
    wb, we, _ = bwt_of_text(small.cpu().numpy()[::-1].tobytes())
    assert se == we and np.array_equal(sb.cpu().numpy(), wb)
    n = 1 << 20
    text = text_bwt.make_text(torch, n - 1, 7, dev)
    d_bwt, eof = text_bwt.bwt_of_reversed_text(torch, text)
    bwt = d_bwt.cpu().numpy()
    h_text = text.cpu().numpy()
    counts = np.bincount(h_text, minlength=256).astype(np.int64)
    hip, orc = pair_from_mem(bwt, eof, counts)
    rng = np.random.default_rng(17)
    alpha = [c for c in range(256) if counts[c]]
    pats = []
    for m in (1, 2, 3, 5, 8, 13, 21, 32, 47, 80):
        for a in rng.integers(0, n - 1 - m, 400):
            p = bytearray(h_text[a:a + m][::-1].tobytes())
            if rng.random() < 0.2:
                p[int(rng.integers(0, m))] = int(rng.choice(alpha))
            pats.append(bytes(p))
    pats = [pats[i] for i in rng.permutation(len(pats))]
    hits = check_search(hip, orc, pats)
    assert hits > 3000
    assert hip.search(h_text[1000:1012][::-1].tobytes()) is not None      # a stretch of the text, reversed, is found
    st = hip.stats()
    assert st["jump_lookups"] > 0 and st["row_lookups"] > 0
    prng = random.Random(3)

    def literal(r, nlit):
        while True:
            a = r.randrange(0, h_text.size - nlit)
            w = h_text[a:a + nlit].tobytes().decode("latin-1")
            if w[0] not in " \n":
                return w
    res = []
    while len(res) < 400:
        re_ = regex_workload.gen_one(prng, literal=literal)
        try:
            R.ReTree(R.re2post(re_)).tables()
            res.append(re_)
        except (R.MatchError, R.Re2PostSyntax):
            pass
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(r_)) for r_ in res]
    batch = findex_amd.ReTree.prepare_batch(hip, trees)
    got, _ = batch.match_raw(cap=1 << 20)
    assert not batch.truncated                                      # the frontier died by itself
    want, pops, trunc = orc.match_tables_batch([R.ReTree(R.re2post(r_)).tables() for r_ in res])
    assert got.size == want.size and got.size > 20
    for f in ("regex", "len", "sp", "ep"):
        assert np.array_equal(got[f], want[f]), f


def test_parked_walks_flush_inside_the_kernel():
    """With a row jump table the patterns a table lookup finds to miss are kept in a per-wave list in LDS (64 entries) and
    walked inside the search kernel -- when 48 have come together, and before the wave ends.  600 k patterns, most of
    them with a byte replaced inside their one-row part, so that every wave fills its list several times over: (sp, ep)
    at the failing step and the executed steps against the oracle, from offsets and as a fixed-length batch."""
    bwt, eof, counts = synth_bwt(300_000, 1, 6, 23)
    hip, orc = pair_from_mem(bwt, eof, counts)
    rng = np.random.default_rng(31)
    base = lf_walk_patterns(orc, rng, 6000, 24, 0.0, alphabet=[1, 2, 3, 4, 5, 6])
    pats2d = np.frombuffer(b"".join(base), dtype=np.uint8).reshape(6000, 24).copy()
    pats2d = np.tile(pats2d, (100, 1))                               # 600 000 patterns
    k, m = pats2d.shape
    mut = rng.random(k) < 0.8
    pos = rng.integers(0, 14, k)                                      # the bytes read at steps 10 .. 23: inside the one-row part
    pats2d[mut, pos[mut]] = rng.integers(1, 7, int(mut.sum())).astype(np.uint8)
    buf = pats2d.reshape(-1)
    off = np.arange(k + 1, dtype=np.uint64) * np.uint64(m)
    wsp, wep, wsteps = orc.search_batch(buf, off, threads=8)
    hip.stats_reset()
    gsp, gep = hip.search_batch(buf, off)
    st = hip.stats()
    assert np.array_equal(gsp, wsp) and np.array_equal(gep, wep)
    assert st["backward_steps"] == int(wsteps.sum()) and st["jump_lookups"] > k // 2
    assert 0.3 < float((wsp == wep).mean()) < 0.8                      # most replaced bytes made a miss
    fsp, fep = hip.search_batch_ex(buf, fixed_len=m)
    assert np.array_equal(fsp, wsp) and np.array_equal(fep, wep)
    pk = hip.search_batch_ex(buf, fixed_len=m, packed=True, escape_cap=16)
    usp, uep = hip.unpack_intervals(pk, k, 16)
    assert np.array_equal(usp, wsp) and np.array_equal(uep, wep)


@pytest.mark.parametrize("layout", ["onehot", "bytes"])
def test_search_grid_follows_the_residency_census(layout):
    """k_search4's grid is sized by what was resident, not by what the occupancy query answers (fmx_search.hip,
    Residency).  Round 5: the census is taken by calibration launches where the tables are built -- fmx_prepare, or (this
    session: "tables_after" = 0) the handle's first search -- never by a later call: once the first search has returned,
    fmx_stats.search_residency carries a confirmed number (bit 8) -- the query's, or up to two below it -- every later
    launch is sized by it, and the intervals are the oracle's."""
    findex_amd.set_layout(layout)
    try:
        bwt, eof, counts = synth_bwt(3_000_000, 97, 120, 17)
        hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
        rng = np.random.default_rng(5)
        k, m = 1_000_000, 24
        walk, _ = hip.lf_walk_batch(rng.integers(0, bwt.size, size=k), m)     # hit patterns: every wave's batches take their full time
        pats = np.ascontiguousarray(walk[:, ::-1])
        off = np.arange(0, (k + 1) * m, m, dtype=np.uint64)
        first = hip.search_batch(pats.reshape(-1), off)
        r0 = hip.stats()["search_residency"]
        assert r0 & 0x100, "the search that built the tables did not calibrate its kernel: %#x" % r0
        assert 1 <= (r0 & 0xFF) <= 8
        for _ in range(4):
            sp, ep = hip.search_batch(pats.reshape(-1), off)
            assert np.array_equal(sp, first[0]) and np.array_equal(ep, first[1])
            assert hip.stats()["search_residency"] == r0
        # a second handle of the same shape finds the instantiation calibrated: prepare has nothing to launch
        hip2 = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
        hip2.prepare(ktab=True, jump=True)
        assert hip2.stats()["search_residency"] & 0x100          # (of the instantiation calibrated last: a handle may select two, by batch size)
        sp2, ep2 = hip2.search_batch(pats.reshape(-1), off)
        assert hip2.stats()["search_residency"] == r0 and np.array_equal(sp2, first[0]) and np.array_equal(ep2, first[1])
        hip2.close()
        orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
        wsp, wep, _ = orc.search_batch(pats[:5000].reshape(-1), off[:5001])
        assert np.array_equal(sp[:5000], wsp) and np.array_equal(ep[:5000], wep)
    finally:
        findex_amd.set_layout("auto")


_CAPTURE_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, findex_amd
from helpers import synth_bwt
findex_amd.set_layout(sys.argv[2])
bwt, eof, counts = synth_bwt(3_000_000, 97, 120, 17)
hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
hip.prepare(ktab=True, jump=True)
rng = np.random.default_rng(5)
k, m = 1_000_000, 24
walk, _ = hip.lf_walk_batch(rng.integers(0, bwt.size, size=k), m)
dev = torch.device("cuda", 0)
d_pat = torch.from_numpy(np.ascontiguousarray(walk[:, ::-1]).reshape(-1)).to(dev)
d_off = torch.arange(0, (k + 1) * m, m, dtype=torch.int64, device=dev)
sp = [torch.zeros(k, dtype=torch.int64, device=dev) for _ in range(3)]
ep = [torch.zeros(k, dtype=torch.int64, device=dev) for _ in range(3)]
def search(i):
    hip.search_batch_dev(d_pat.data_ptr(), d_off.data_ptr(), sp[i].data_ptr(), ep[i].data_ptr(), k, torch.cuda.current_stream().cuda_stream)
sys.stderr.flush()
r0 = hip.stats()["search_residency"]
assert r0 & 0x100, hex(r0)                  # fmx_prepare calibrated the kernel its tables select: nothing is left for a search to read
built = hip.stats()["tables_build_ms"]
print("FIRST-SEARCH follows", file=sys.stderr, flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):                   # the handle's FIRST search, on a stream that is being captured (torch: global mode --
    search(1)                               # a hipMalloc, a hipMemcpy or a hipStreamSynchronize in here fails the capture)
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
search(0)                                   # the plain call
torch.cuda.synchronize()
assert torch.equal(sp[1], sp[0]) and torch.equal(ep[1], ep[0]) and int((sp[0] < ep[0]).sum()) == k
r1 = hip.stats()["search_residency"]        # (of the instantiation a batch of this size takes: fmx_prepare calibrated it beside the quads')
assert r1 & 0x100, hex(r1)
for _ in range(6):
    search(2)
torch.cuda.synchronize()
assert torch.equal(sp[2], sp[0]) and torch.equal(ep[2], ep[0])
st = hip.stats()
r = st["search_residency"]
assert r == r1, (hex(r), hex(r1))
assert st["tables_build_ms"] == built, "a table was built after fmx_prepare"
print("ok", hex(r))
"""


@pytest.mark.parametrize("layout", ["onehot", "bytes"])
def test_search_inside_a_captured_graph_leaves_the_census_alone(layout):
    """The header's contract (fmx.h, fmx_prepare): after fmx_prepare the _dev entry points only enqueue work.  A handle's very
    FIRST fmx_search_batch_dev is made on a stream that is being captured into a torch CUDAGraph (global capture mode: any
    allocation, synchronous copy or stream synchronisation by this thread fails the capture) -- the capture succeeds, its
    replays give the intervals of the plain call, the grid is the one fmx_prepare's census confirmed, and no table is
    built afterwards.  Under FMX_TRACE the library reports every census launch: all of them must precede the first
    search.  (A child process: the census is per process and instantiation.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FMX_TRACE="1")
    r = subprocess.run([sys.executable, "-c", _CAPTURE_SCRIPT, root, layout], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    lines = [ln for ln in r.stderr.splitlines() if "census" in ln or "FIRST-SEARCH" in ln]      # (both on stderr: in order)
    assert any("census" in ln for ln in lines), "FMX_TRACE shows no census launch at all:\n" + r.stderr
    first_search = [i for i, ln in enumerate(lines) if "FIRST-SEARCH" in ln][0]
    assert not any("census" in ln for ln in lines[first_search:]), "a census was read after the first search:\n" + "\n".join(lines)


_TICKET_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, findex_amd, oracle
from helpers import synth_bwt
layout, mode = sys.argv[2], sys.argv[3]
findex_amd.set_layout(layout)
bwt, eof, counts = synth_bwt(3_000_000, 97, 120, 23)
hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
if mode == "pairs":
    hip.config_set("jump_pairs", "on")
else:
    hip.config_set("jump", "rows3")
hip.prepare(ktab=True, jump=True)
rng = np.random.default_rng(7)
k, m = 600_000, 28
walk, _ = hip.lf_walk_batch(rng.integers(0, bwt.size, size=k), m)
full = np.ascontiguousarray(walk[:, ::-1])
lens = rng.integers(6, m + 1, size=k)                      # ragged: the last `len` characters of each walk
miss = rng.random(k) < 0.1                                  # ... a tenth of them with one character changed
col = m - 1 - (rng.random(k) * lens).astype(np.int64)      # anywhere in the pattern
full[miss, col[miss]] = (full[miss, col[miss]] + 1) % 120 + 1
off = np.zeros(k + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
pats = np.concatenate([full[i, m - lens[i]:] for i in range(k)]) if k <= 1000 else None
if pats is None:
    idx = np.repeat(np.arange(k), lens) * m + (m - np.repeat(lens, lens)) + (np.arange(off[-1]) - np.repeat(off[:-1], lens))
    pats = full.reshape(-1)[idx]
dev = torch.device("cuda", 0)
d_pat = torch.from_numpy(pats).to(dev); d_off = torch.from_numpy(off).to(dev)
streams = [torch.cuda.Stream() for _ in range(20)]          # more streams than the handle has ticket areas: the last ones stride
outs = []
torch.cuda.synchronize()
for rep in range(3):                                        # one launch after the other on every stream: the counters set themselves back
    for s in streams[: (20 if rep == 0 else 3)]:
        sp = torch.full((k,), -1, dtype=torch.int64, device=dev); ep = torch.full((k,), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        hip.search_batch_dev(d_pat.data_ptr(), d_off.data_ptr(), sp.data_ptr(), ep.data_ptr(), k, s.cuda_stream)
        outs.append((sp, ep))
torch.cuda.synchronize()
for sp, ep in outs[1:]:
    assert torch.equal(sp, outs[0][0]) and torch.equal(ep, outs[0][1])
# two streams at once on one handle: each draws from its own area
a = [torch.full((k,), -1, dtype=torch.int64, device=dev) for _ in range(4)]
torch.cuda.synchronize()
for rep in range(2):
    hip.search_batch_dev(d_pat.data_ptr(), d_off.data_ptr(), a[0].data_ptr(), a[1].data_ptr(), k, streams[0].cuda_stream)
    hip.search_batch_dev(d_pat.data_ptr(), d_off.data_ptr(), a[2].data_ptr(), a[3].data_ptr(), k, streams[1].cuda_stream)
torch.cuda.synchronize()
assert torch.equal(a[0], outs[0][0]) and torch.equal(a[2], outs[0][0]) and torch.equal(a[1], outs[0][1]) and torch.equal(a[3], outs[0][1])
sp, ep = outs[0][0].cpu().numpy().view(np.uint64), outs[0][1].cpu().numpy().view(np.uint64)
orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
for lo, hi in ((0, 4000), (k - 60_000, k)):                 # the pool is the END of the batch list
    wsp, wep, _ = orc.search_batch(pats[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
    assert np.array_equal(sp[lo:hi], wsp) and np.array_equal(ep[lo:hi], wep), (lo, hi)
assert 0.85 < float((sp < ep).mean()) < 0.95
print("ok")
"""


@pytest.mark.parametrize("layout,mode", [("onehot", "rows3"), ("bytes", "rows3"), ("onehot", "pairs")])
def test_last_rounds_drawn_by_ticket(layout, mode):
    """k_search4's pool (fmx_search.hip, "The last rounds are DRAWN"): with a small grid (FMX_SEARCH_WGS=1: 1024 waves) 600 000
    ragged patterns are dozens of rounds, the last two-to-three of them drawn by ticket.  Launch after launch on one stream
    (the counters set themselves back), on twenty streams (the handle has sixteen areas: the others stride), two streams at
    once; all results equal, the first 4000 and the LAST 60 000 patterns -- the pool -- equal to the oracle's.  FMX_TRACE must
    show launches with a pool and, past the sixteenth stream, launches without."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FMX_TRACE="1", FMX_SEARCH_WGS="1", FMX_SEARCH_G2="1" if mode == "pairs" else "0")
    env.pop("FMX_SEARCH_TICKETS", None)
    r = subprocess.run([sys.executable, "-c", _TICKET_SCRIPT, root, layout, mode], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    import re
    drawn = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"the last (\d+) drawn from ticket area (\d+)", r.stderr)]
    assert len(drawn) >= 30, r.stderr[-2000:]
    assert sum(1 for d, a in drawn if d > 0 and a > 0) >= 22 and any(d == 0 for d, a in drawn), drawn[:40]
    assert {a for d, a in drawn if d > 0} == set(range(1, 17)), sorted({a for d, a in drawn})


@pytest.mark.parametrize("which", ["modes", "repeats", "walks"])
def test_pairs_of_lanes_with_single_entries(which, monkeypatch):
    """The pair-of-lanes kernel over a row jump table of SINGLE entries (k_search4<.., JT = 1, .., G2>: an index under 2^30 rows,
    or one whose budget has no room for pairs of entries): forced onto the small shapes of the row-table tests."""
    monkeypatch.delenv("FMX_JUMP_PAIRS", raising=False)
    monkeypatch.setenv("FMX_SEARCH_G2", "1")
    if which == "modes":
        test_row_jump_table_on_and_off_agree("onehot")
    elif which == "repeats":
        test_row_tables_on_repetitive_texts("onehot")
    else:
        test_parked_walks_flush_inside_the_kernel()


@pytest.mark.parametrize("which", ["modes", "repeats", "spans", "walks", "ragged"])
def test_pairs_of_lanes_search_kernel(which, monkeypatch):
    """k_search4<.., G2> (round 5): a pattern served by a PAIR of lanes, 32 patterns per wave, the dictionary's blocks
    fetched as two 32-byte halves, pattern spans of up to 2 KiB staged -- by default only for batches that give every wave
    several batches (the C3 tests at full size run it), forced here on the small ones (FMX_SEARCH_G2=1, with the pair table
    it needs: FMX_JUMP_PAIRS=1): the row-table tests in every mode and entry width, the repetitive texts (intervals of two
    rows: lane t looks up row sp + t), the staged spans' edges, the parked walks, and ragged lengths 0 .. 70 around the
    2 KiB span limit -- all against the oracle, executed steps included."""
    monkeypatch.setenv("FMX_JUMP_PAIRS", "1")
    monkeypatch.setenv("FMX_SEARCH_G2", "1")
    if which == "modes":
        test_row_jump_table_on_and_off_agree("onehot")
    elif which == "repeats":
        test_row_tables_on_repetitive_texts("onehot")
    elif which == "spans":
        test_staged_pattern_spans_edges("onehot")
    elif which == "walks":
        test_parked_walks_flush_inside_the_kernel()
    else:
        bwt, eof, counts = synth_bwt(500_000, 1, 12, 77)
        hip, orc = pair_from_mem(bwt, eof, counts)
        hip.prepare(ktab=True, jump=True)
        rng = np.random.default_rng(4)
        pats = []
        for m in list(range(0, 71)) * 12:                       # batches of 32 whose spans run from nothing to 2.2 KiB
            pats += lf_walk_patterns(orc, rng, 1, m, 0.25, alphabet=list(range(1, 13))) if m else [b""]
        rng.shuffle(pats)
        pats += lf_walk_patterns(orc, rng, 64, 63, 0.1, alphabet=list(range(1, 13)))      # 32 x 63 + alignment: on the limit
        pats += lf_walk_patterns(orc, rng, 64, 64, 0.1, alphabet=list(range(1, 13)))      # 32 x 64 = 2 KiB exactly: staged or not by the buffer's alignment
        pats += lf_walk_patterns(orc, rng, 64, 65, 0.1, alphabet=list(range(1, 13)))      # over it: read from global memory
        hip.stats_reset()
        check_search(hip, orc, pats)
        st = hip.stats()
        assert st["jump_lookups"] > 0 and st["jump_bytes"] == 32 * orc.n
        hip.close()


@pytest.mark.parametrize("kernel", ["quads", "pairs"])
def test_misses_as_none(kernel, monkeypatch):
    """FMX_SEARCH_MISS_NONE (fmx.h): a pattern that does not occur may come back as (0, 0) -- SuffixAlgo.search returns None for
    it either way (findex.scala:30) -- so the kernels with a row jump table park nothing to walk for it.  Hits bit-equal to the
    oracle's, every miss sp >= ep, and the count of the reference loop's steps (rank_queries) what the oracle counts, misses
    included; through the host form, the device form and the 8-byte form, with ragged lengths and a third of the patterns
    mutated anywhere; by quads and by pairs of lanes.  At least some misses must really have been cut short (0, 0)."""
    torch = _torch()
    monkeypatch.setenv("FMX_JUMP_PAIRS", "1")
    monkeypatch.setenv("FMX_SEARCH_G2", "1" if kernel == "pairs" else "0")
    bwt, eof, counts = synth_bwt(600_000, 97, 120, 31)
    hip, orc = pair_from_mem(bwt, eof, counts)
    hip.prepare(ktab=True, jump=True)
    rng = np.random.default_rng(12)
    pats = []
    for m in (3, 9, 17, 24, 32, 41):
        pats += lf_walk_patterns(orc, rng, 700, m, 0.35)
    rng.shuffle(pats)
    buf, off = pack_patterns(pats)
    k = len(pats)
    wsp, wep, wsteps = orc.search_batch(buf, off)
    hit = wsp < wep
    assert 0.5 < hit.mean() < 0.8
    hip.stats_reset()
    gsp, gep = hip.search_batch_ex(buf, off, miss_none=True)
    st = hip.stats()
    assert np.array_equal(gsp[hit], wsp[hit]) and np.array_equal(gep[hit], wep[hit]) and bool((gsp[~hit] >= gep[~hit]).all())
    assert st["backward_steps"] == int(wsteps.sum()) and st["rank_queries"] == 2 * int(wsteps.sum())
    cut = int(((gsp == 0) & (gep == 0) & ~hit).sum())
    assert cut > 100 and st["jump_lookups"] > 0, (cut, st["jump_lookups"])
    # the device form, and the 8-byte form: a miss is a word of width 0
    dev = torch.device("cuda", 0)
    d_pat, d_off = torch.from_numpy(buf).to(dev), torch.from_numpy(off.view(np.int64)).to(dev)
    d_sp, d_ep = torch.full((k,), -1, dtype=torch.int64, device=dev), torch.full((k,), -1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    hip.search_batch_ex_dev(d_pat.data_ptr(), d_off.data_ptr(), d_sp.data_ptr(), d_ep.data_ptr(), k, stream, miss_none=True)
    torch.cuda.synchronize()
    assert np.array_equal(d_sp.cpu().numpy().view(np.uint64), gsp) and np.array_equal(d_ep.cpu().numpy().view(np.uint64), gep)
    esc = 64
    d_pk = torch.zeros(hip.packed_words(k, esc), dtype=torch.int64, device=dev)
    hip.search_batch_ex_dev(d_pat.data_ptr(), d_off.data_ptr(), d_pk.data_ptr(), d_ep.data_ptr(), k, stream, packed=True, escape_cap=esc, miss_none=True)
    torch.cuda.synchronize()
    usp, uep = hip.unpack_intervals(d_pk.cpu().numpy().view(np.uint64), k, esc)
    assert np.array_equal(usp[hit], wsp[hit]) and np.array_equal(uep[hit], wep[hit]) and bool((usp[~hit] >= uep[~hit]).all())
    # unknown bits of fmx_search_opts.packed are refused
    import ctypes
    from findex_amd import _lib
    bad = _lib.fmx_search_opts(0, 4, 0)
    assert _lib.load().fmx_search_batch_ex_dev(hip.handle, ctypes.c_void_p(d_pat.data_ptr()), ctypes.c_void_p(d_off.data_ptr()), ctypes.c_void_p(d_sp.data_ptr()),
                                               ctypes.c_void_p(d_ep.data_ptr()), k, ctypes.byref(bad), ctypes.c_void_p(stream)) == 3
    hip.close()


def test_result_groups_of_every_size_are_ordered():
    """k_res_sort's paths side by side in one batch of 400 regexes (shuffled, so that every workgroup of 256 regexes
    holds a mix): groups of one result, of 2 .. 12, of 13 .. 64 (every result finds its own place), of 65 .. 1024 (bitonic
    sort by the workgroup), stretches of more than 1024 results per 256 regexes (not staged in LDS) -- each regex's list
    against the oracle's, in (len, sp, ep) order, through the host form and the device-resident form."""
    torch = _torch()
    from findex_amd.regex import RegexBatch, RESULT_DTYPE
    bwt, eof, counts = synth_bwt(400_000, 97, 100, 21)           # a..d
    hip = findex_amd.HipFMSearcher.from_mem(bwt, eof, counts)
    orc = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    rng = np.random.default_rng(11)
    abcd = "abcd"
    res = []
    for _ in range(220):                                          # one result each, or none
        res.append("".join(abcd[i] for i in rng.integers(0, 4, size=int(rng.integers(2, 9)))))
    for _ in range(100):                                          # 2 .. 12: a class or an option or two
        lit = "".join(abcd[i] for i in rng.integers(0, 4, size=3))
        res.append(lit + ["[ab]", "[a-c]", "[a-d]", "[ab][cd]", "[a-c][ab]", "(a|b)c?", "[ab]?[cd]?"][int(rng.integers(0, 7))])
    for _ in range(50):                                           # 13 .. 64
        lit = abcd[int(rng.integers(0, 4))]
        res.append(lit + ["[a-d][a-d]", "[a-d][a-c][ab]", "[ab][ab][ab][ab]", "[a-d][a-d][ab]", "[ab]?[a-d][a-d]"][int(rng.integers(0, 5))])
    for _ in range(28):                                           # 65 .. 1024
        res.append(abcd[int(rng.integers(0, 4))] + ["[a-d][a-d][a-d]", "[a-d][a-d][a-d][ab]", "[a-d][a-d][a-d][a-d]", "[ab][a-d][a-d][a-d][ab]?"][int(rng.integers(0, 4))])
    res += ["[a-d][a-d][a-d][a-d][a-d][ab]", "[a-d][a-d][a-d][a-d][a-c][ab]?"]      # more than 1024
    order = rng.permutation(len(res))
    res = [res[i] for i in order]
    trees = [findex_amd.ReTree(findex_amd.REParser.re2post(r)) for r in res]
    rb = RegexBatch(hip, trees)
    out, per = rb.match_raw(max_steps=12, cap=1 << 20)
    sizes = np.bincount(np.minimum(np.searchsorted([1, 2, 13, 65, 1025], per, side="right"), 5), minlength=6)
    assert all(sizes[1:] > 0), "the batch misses a size class: %s" % sizes.tolist()
    at = 0
    for i, re_ in enumerate(res):
        want, _ = oracle_results(orc, re_)
        got = [(int(r["len"]), int(r["sp"]), int(r["ep"])) for r in out[at:at + int(per[i])]]
        assert all(int(r["regex"]) == i for r in out[at:at + int(per[i])]) and got == want, re_
        at += int(per[i])
    assert at == out.size
    cap = 1 << 20
    d_out = torch.zeros(cap * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_per = torch.zeros(len(trees), dtype=torch.int32, device="cuda")
    for _ in range(2):                                            # (the second call replays whatever the first one set up)
        n = rb.match_dev(d_out.data_ptr(), cap, d_per.data_ptr(), max_steps=12)
        torch.cuda.synchronize()
        got = np.frombuffer(d_out[: n * RESULT_DTYPE.itemsize].cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
        assert n == out.size and got.tobytes() == out.tobytes()
        assert np.array_equal(d_per.cpu().numpy().astype(np.uint32), per)


@pytest.mark.parametrize("which", ["modes", "repeats", "spans", "walks", "census"])
def test_row_jump_table_as_pairs_of_entries(which, monkeypatch):
    """The row jump table as pairs J[r] | J[LF^jc r] (32 bytes per row, k_search4<.., JT = 2>: up to 2 x jump_chars steps
    per request; by default only for indexes of 2^30 rows and more -- the C3 tests at full size run it -- forced here on the
    small ones): the row-table tests in every mode and entry width, the repetitive texts (intervals of a few rows: the
    first entries only), the staged spans' edges, the parked walks (a pattern that agrees with a pair's first entry and
    not with its second jumps nine steps and parks behind them) and the residency census, all against the oracle."""
    monkeypatch.setenv("FMX_JUMP_PAIRS", "1")
    if which == "modes":
        test_row_jump_table_on_and_off_agree("onehot")
    elif which == "repeats":
        test_row_tables_on_repetitive_texts("onehot")
    elif which == "spans":
        test_staged_pattern_spans_edges("onehot")
    elif which == "walks":
        test_parked_walks_flush_inside_the_kernel()
    else:
        test_search_grid_follows_the_residency_census("onehot")
