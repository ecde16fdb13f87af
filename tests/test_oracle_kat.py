"""Pins the CPU oracle (oracle/) to the known answers and golden files the
reference's own test-suite holds for the rank / search / LF path.

Each test cites the reference test it transcribes:
  T = /root/reference/src/test/scala/org/fmindex/tests
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle.naive_bwt import NaiveBWTSearcher
from helpers import bwt_of_text


def sais(text: bytes):
    """SAISBuilder(fromString(text)) + build + buildOCC as the tests use it."""
    bwt, eof, counts = bwt_of_text(text)
    return oracle.SAISNaiveSearcher.from_mem(bwt, eof, counts)


# ---------------------------------------------------------------- T/Indexer.scala
def test_occ_cf_abracadabra():
    """T/Indexer.scala:247-294 'Naive SuffixAlgo test: occ/cf'"""
    sa = sais(b"abracadabra")
    assert sa.cf(0) == 0 and sa.cf(ord("a")) == 1 and sa.cf(ord("b")) == 6
    assert sa.bwt_read(0) == ord("a")
    assert sa.fm().tolist() == [3, 0, 6, 7, 8, 9, 10, 11, 5, 2, 1, 4]      # :272 (and :239)

    def row(c):
        return [sa.occ(c, i) for i in range(sa.n)]

    assert row(0) == [0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1]
    assert row(ord("a")) == [1, 1, 1, 1, 1, 1, 2, 3, 4, 5, 5, 5]
    assert row(ord("b")) == [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2]
    assert row(ord("c")) == [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1]
    assert row(ord("d")) == [0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]
    assert row(ord("r")) == [0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2]
    assert row(ord("x")) == [0] * 12


def test_plain_searching():
    """T/Indexer.scala:296-306"""
    assert sais(b"abracadabra").search(b"bra") == (6, 8)


def test_bwt_walk():
    """T/Indexer.scala:308-323 'BWT walki'"""
    sa = sais(b"abracadabra")
    assert sa.bwt_read(6) == ord("a")
    assert sa.getPrevI(6) == 2
    assert sa.bwt_read(sa.getPrevI(6)) == ord("d")
    assert sa.bwt_read(sa.getPrevI(2)) == ord("a")
    assert sa.getNextI(6) == 10 and sa.getNextI(10) == 1


def test_bwt_substrings():
    """T/Indexer.scala:324-341"""
    sa = sais(b"abracadabra")
    assert sa.nextSubstr(6, 4) == b"bra\0"
    assert sa.prevSubstr(6, 4) == b"cada"
    sa = sais(b"mmabcacadabbbca"[::-1])
    assert sa.nextSubstr(11, 3) == b"cba"
    assert sa.prevSubstr(11, 3) == b"aca"


def test_get_prev_range():
    """T/Indexer.scala:342-351"""
    sa = sais(b"mmabcacadabbbca"[::-1])
    assert sa.occ(ord("b"), 6) == 3
    assert sa.getPrevRange(0, 16, ord("a")) == (1, 6)
    assert sa.getPrevRange(1, 6, ord("b")) == (6, 8)


def test_naive_bwt_searcher_kats(golden):
    """T/Indexer.scala:685-743: 19 + 3 occ answers incl. the 0xff symbol and
    the EOF hole (inputs lifted by tests/golden/make_kat_fixtures.py)."""
    kat = json.load(open(os.path.join(golden, "naive_bwt_searcher_kat.json")))
    for case in ("case1", "case2"):
        k = kat[case]
        s = NaiveBWTSearcher(k["bwt"], k["bs"], k["rk0"])
        for c, key, want in k["occ"]:
            assert s.occ(c, key) == want, (case, c, key)
    assert kat["case2"]["bwt"].index(0xFF) == 721


@pytest.mark.parametrize("name", ["test1024", "test2048", "test2048-2", "test3072", "test", "test-part"])
def test_cmp_goldens_are_bwt_of_reversed_text(testdata, name):
    """T/Indexer.scala:638-648,745-820: BWTMerger2.merge(FileBWTReader(X.txt))
    must equal the little-endian X.cmp.{bwt,aux} written by the C bwtdisk tool.
    Construction is out of scope, but this pins what the goldens *are*: the BWT
    of the reversed file (bwtmerger.scala:1106-1108 copyReverse), which the
    parity tests' own index builder (helpers.bwt_of_text) must reproduce."""
    txt = open(os.path.join(testdata, name + ".txt"), "rb").read()
    bwt, size, eof = oracle.load_bwt_file(os.path.join(testdata, name + ".cmp.bwt"), bigEndian=False)
    aux = oracle.load_aux_file(os.path.join(testdata, name + ".cmp.aux"), bigEndian=False)
    assert size == len(txt) + 1
    mine, meof, mcounts = bwt_of_text(txt[::-1])
    assert meof == eof
    keep = np.arange(size) != eof
    assert np.array_equal(mine[keep], bwt[keep])
    assert np.array_equal(aux, mcounts) and aux[0] == 0 and aux.sum() == size - 1


@pytest.mark.parametrize("name,be", [("test1024.cmp", False), ("test2048.cmp", False), ("words", True)])
def test_fm_is_bwt2occ(testdata, name, be, tmp_path):
    """T/Indexer.scala:841-900 BWTCreatorTest: the .fm payload equals
    bwtstring.bwt2occ (util.scala:121-134) of the BWT with slot eof set to 0;
    plus the .fm wire format (bwtmerger.scala:252-267,483-485)."""
    sa = oracle.NaiveFMSearcher(os.path.join(testdata, name + ".bwt"), bigEndian=be)
    bwt, size, eof = oracle.load_bwt_file(os.path.join(testdata, name + ".bwt"), bigEndian=be)
    bwtd = bwt.copy()
    bwtd[eof] = 0
    want = np.argsort(bwtd, kind="stable").astype(np.uint32)        # == bwt2occ
    assert np.array_equal(sa.fm(), want)
    p = str(tmp_path / "x.fm")
    sa.write_fm(p)
    raw = np.fromfile(p, dtype=np.uint8)
    assert raw.size == 9 + 4 * size and raw[0] == 4
    assert int(raw[1:9].view(">i8")[0]) == size
    assert np.array_equal(raw[9:].view(">u4"), want)


def test_combined_indexing_test1024(testdata):
    """T/Indexer.scala:1076-1124 CombinedIndexingTest"""
    sa = oracle.NaiveFMSearcher(os.path.join(testdata, "test1024.cmp.bwt"), bigEndian=False)
    eof = sa.eof
    assert eof == 462
    assert [sa.bwt_read(i) for i in (0, 1, 2)] == [ord("u"), ord("b"), ord("x")]
    assert sa.bwt_read(eof) == 0
    assert sa.getPrevI(eof) == 0
    assert sa.bwt_read(sa.getPrevI(eof)) == ord("u")
    assert sa.getNextI(eof) == 517
    assert sa.bwt_read(sa.getNextI(eof)) == ord("l")
    assert sa.getPrevI(1) == 48 and sa.getPrevI(48) == 649
    assert sa.nextSubstr(1, 3) == b"haa"
    assert sa.bwt_read(1000) == ord("b")
    assert sa.nextSubstr(sa.getNextI(eof), 100) == (
        b"zajrtzbeqwbxdfpwjflmmsseewuudgfbtzqenjqafwzcnfanycigwsflfvxojxpqhhzekjdkhgsptqveavquuoqujbezdkarayom")
    assert sa.nextSubstr(eof, 100) == (
        b"ajrtzbeqwbxdfpwjflmmsseewuudgfbtzqenjqafwzcnfanycigwsflfvxojxpqhhzekjdkhgsptqveavquuoqujbezdkarayoml")
    assert sa.prevSubstr(1, 5) == b"bqxxa"
    assert sa.prevSubstr(eof, 5) == b"\0uexm"
    assert sa.prevSubstr(sa.getPrevI(eof), 4) == b"uexm"


# -------------------------------------------------------------- T/REParser.scala
def test_naive_fm_searcher_small2(testdata):
    """T/REParser.scala:236-291 'NaiveFMSearcher' on small2.txt ("ippisissim",
    indexed reversed = "missisippi")."""
    txt = open(os.path.join(testdata, "small2.txt"), "rb").read()
    assert txt == b"ippisissim"
    bwt, eof, counts = bwt_of_text(txt[::-1])
    assert bytes(bwt) == b"ipssmmpiisi" and eof == 5            # :273 comment
    sa = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    assert "".join(chr(sa.pos2char(i)) for i in range(11)) == "iiiimppssss"
    assert [sa.getNextI(i) for i in (0, 5, 4, 10, 9)] == [5, 4, 10, 9, 3]
    assert [sa.getPrevI(i) for i in (3, 9, 10)] == [9, 10, 4]
    assert sa.bwt_read(4) == ord("m")
    assert [sa.getPrevI(i) for i in (4, 5, 0)] == [5, 0, 1]


# --------------------------------------------------------- beyond the reference
def test_words_counts_match_text(testdata):
    """the index is over the reversed text, so search(reverse(p))
    counts p's occurrences in words.txt."""
    sa = oracle.NaiveFMSearcher(os.path.join(testdata, "words.bwt"))
    txt = open(os.path.join(testdata, "words.txt"), "rb").read()
    assert sa.n == len(txt) + 1 == 1916149 and sa.eof == 86533
    assert sa.search(b"aardvark"[::-1]) == (1044943, 1044945)
    assert sa.search(b"aardvark") is None
    for p in (b"aardvark", b"zebra", b"ing\r\n", b"e", b"qu", b"xyzzy"):
        r = sa.search(p[::-1])
        cnt = 0 if r is None else r[1] - r[0]
        # overlapping count
        want, k = 0, txt.find(p)
        while k >= 0:
            want += 1
            k = txt.find(p, k + 1)
        assert cnt == want, p


def test_search_edge_cases():
    sa = sais(b"abracadabra")
    assert sa.search(b"") == (0, sa.n)                 # loop never runs, findex.scala:20
    assert sa.search(b"zzz") is None
    found, sp, ep, steps = sa.search_raw(b"xbra")
    assert not found and sp == ep and steps == 4
    strict = oracle.SAISNaiveSearcher.from_mem(*bwt_of_text(b"abracadabra"), strict_signed=True)
    with pytest.raises(oracle.IndexOutOfBounds):       # signed Byte index, findex.scala:21,26
        strict.search(b"a\x80")
    with pytest.raises(oracle.IndexOutOfBounds):
        sa.occ(256, 0)


def test_interval_prev_range_descending():
    """findex.scala:37-51: inclusive cend, non-empty only, DESCENDING c."""
    sa = sais(b"mmabcacadabbbca"[::-1])
    got = sa.getIntervalPrevRange(0, sa.n, ord("a"), ord("d"))
    want = [sa.getPrevRange(0, sa.n, c) for c in (ord("d"), ord("c"), ord("b"), ord("a"))]
    assert got == [w for w in want if w is not None] and len(got) == 4
    assert sa.getIntervalPrevRange(1, 6, ord("c"), ord("z")) == [
        r for r in (sa.getPrevRange(1, 6, c) for c in range(ord("z"), ord("c") - 1, -1)) if r]


def test_occ_matches_bruteforce_random():
    rng = np.random.default_rng(7)
    bwt = rng.integers(1, 6, size=5000, dtype=np.uint8)
    eof = 1234
    counts = np.bincount(bwt, minlength=256).astype(np.int64)
    counts[bwt[eof]] -= 1
    sa = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    b2 = bwt.copy()
    b2[eof] = 0
    for _ in range(300):
        c = int(rng.integers(0, 7))
        i = int(rng.integers(-1, 5000))
        assert sa.occ(c, i) == int((b2[: i + 1] == c).sum())


def test_sampled_checkpoint_oracle_equals_the_inverted_lists(testdata):
    """The CPU baseline's second structure (oracle.SampledFMSearcher: symbol checkpoints every 256 positions + a scan of the
    BWT bytes -- BASELINE.md's "sampled popcount structure" for n >= 2^31, where the 32-bit inverted lists stop) computes
    the same cf / occ / search as the restatement of NaiveFMSearcher that the reference's vectors pin: on every fixture
    file, on block edges, with bytes >= 0x80, at the EOF row, and on random indexes."""
    import glob
    rng = np.random.default_rng(20)
    cases = []
    for f in sorted(glob.glob(os.path.join(testdata, "*.bwt"))):
        be = os.path.basename(f) == "words.bwt"
        bwt, _, eof = oracle.load_bwt_file(f, bigEndian=be)
        cases.append((os.path.basename(f), np.array(bwt, dtype=np.uint8), eof))
    assert len(cases) >= 7
    for n, lo, hi in ((1, 1, 1), (255, 1, 2), (256, 1, 2), (257, 1, 2), (5000, 1, 255), (70001, 97, 122), (3000, 200, 255)):
        cases.append(("random n=%d" % n, rng.integers(lo, hi + 1, n).astype(np.uint8), int(rng.integers(0, n))))
    for name, bwt, eof in cases:
        n = bwt.size
        counts = oracle.histogram(bwt, eof)
        a = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
        b = oracle.SampledFMSearcher(bwt, eof, threads=2)
        syms = sorted(set(int(c) for c in np.unique(bwt)) | {0, 1, 255})
        for c in syms:
            assert a.cf(c) == b.cf(c), (name, c)
        keys = np.unique(np.clip(np.concatenate([[-1, 0, 1, eof - 1, eof, eof + 1, n - 2, n - 1, n, n + 7, 254, 255, 256, 257, 511, 512],
                                                 rng.integers(-1, n + 2, 300)]), -1, n + 7))
        for c in syms[:12] + syms[-3:]:
            for i in keys:
                assert a.occ(c, int(i)) == b.occ(c, int(i)), (name, c, int(i))
        # patterns: stretches the LF walk reads (hits) and random ones
        pats = []
        for _ in range(200):
            r, m, s = int(rng.integers(0, n)), int(rng.integers(0, 9)), []
            for _ in range(m):
                c = a.bwt_read(r)
                s.append(c)
                r = a.getPrevI(r)
            pats.append(bytes(reversed(s)))
        pats += [bytes(rng.choice(syms, size=int(rng.integers(1, 5))).astype(np.uint8)) for _ in range(200)]
        buf = np.frombuffer(b"".join(pats), dtype=np.uint8)
        off = np.concatenate([[0], np.cumsum([len(p) for p in pats])]).astype(np.uint64)
        wsp, wep, wst = a.search_batch(buf, off)
        gsp, gep, gst = b.search_batch(buf, off, threads=3)
        assert np.array_equal(wsp, gsp) and np.array_equal(wep, gep) and np.array_equal(wst, gst), name
        assert int((wsp < wep).sum()) > 50, name
        b.close()
