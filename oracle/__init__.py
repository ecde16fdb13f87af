"""CPU oracle for the findex hot path -- TEST INFRASTRUCTURE ONLY.

`oracle/` restates the reference's algorithm (inverted position lists + binary
search `occ`, backward search, Glushkov frontier search) on the CPU so that the
HIP path can be checked bit for bit.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product package `findex_amd`
never does (tests/test_no_oracle_in_product.py enforces that).

Parity status: pinned by the reference's own fixtures and known answers
(tests/test_oracle_kat.py, tests/test_oracle_regex.py).  There is no
oracle/_ref: the reference is Scala/JVM and this image has no JVM, so it cannot
be built or run here.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "liboracle.so")
_SRC = os.path.join(_HERE, "fmx_oracle.c")

ORC_ERR_INDEX = -3

_lib = None


def build(force=False):
    """gcc the C restatement into oracle/_build/liboracle.so."""
    os.makedirs(_BUILD, exist_ok=True)
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    tmp = _SO + ".%d.tmp" % os.getpid()
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-std=c11", "-Wall",
                           "-o", tmp, _SRC])
    os.replace(tmp, _SO)
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        build()
    L = ctypes.CDLL(_SO)
    vp, u64, i64, i32, cp = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_char_p
    P = ctypes.POINTER
    L.orc_open_mem.restype = vp
    L.orc_open_mem.argtypes = [vp, u64, u64, vp, P(i32)]
    L.orc_open_mem_threads.restype = vp
    L.orc_open_mem_threads.argtypes = [vp, u64, u64, vp, i32, P(i32)]
    L.orc_open_files.restype = vp
    L.orc_open_files.argtypes = [cp, cp, i32, P(i32)]
    L.orc_close.argtypes = [vp]
    L.orc_n.restype = u64
    L.orc_n.argtypes = [vp]
    L.orc_eof.restype = u64
    L.orc_eof.argtypes = [vp]
    L.orc_fm_ptr.restype = vp
    L.orc_fm_ptr.argtypes = [vp]
    L.orc_bwt_ptr.restype = vp
    L.orc_bwt_ptr.argtypes = [vp]
    L.orc_write_fm.argtypes = [vp, cp]
    L.orc_cf.restype = i64
    L.orc_cf.argtypes = [vp, i32]
    L.orc_occ.restype = i64
    L.orc_occ.argtypes = [vp, i32, i64]
    L.orc_search.argtypes = [vp, vp, u64, i32, P(u64), P(u64), P(ctypes.c_uint32)]
    L.orc_get_prev_range.argtypes = [vp, i64, i64, i32, P(u64), P(u64)]
    L.orc_get_interval_prev_range.argtypes = [vp, i64, i64, i32, i32, vp, vp, vp]
    L.orc_bwt_read.argtypes = [vp, u64]
    L.orc_pos2char.argtypes = [vp, i64]
    L.orc_get_prev_i.restype = i64
    L.orc_get_prev_i.argtypes = [vp, i64]
    L.orc_get_next_i.restype = i64
    L.orc_get_next_i.argtypes = [vp, i64]
    L.orc_next_substr.restype = i64
    L.orc_next_substr.argtypes = [vp, i64, i64, vp]
    L.orc_prev_substr.restype = i64
    L.orc_prev_substr.argtypes = [vp, i64, i64, vp]
    L.orc_occ_batch.argtypes = [vp, vp, vp, vp, u64]
    L.orc_search_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp, i32]
    L.orc_prev_range_batch.argtypes = [vp, vp, vp, vp, vp, vp, u64]
    L.orc_match_sa.restype = i64
    L.orc_match_sa.argtypes = [vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp, ctypes.c_int32, i64, i64,
                               vp, vp, vp, i64, P(i64), P(i64)]
    L.orc_lf_chain.restype = i64
    L.orc_lf_chain.argtypes = [vp, i64, i64]
    L.orc_histogram.restype = None
    L.orc_histogram.argtypes = [vp, u64, u64, i32, vp]
    L.orc_match_sa_batch.restype = i64
    L.orc_match_sa_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32,
                                     vp, vp, vp, vp, i64, P(i64), P(i64)]
    L.orc_occ_chain.argtypes = [vp, vp, u64, i64, vp]
    L.orc_match_sa_batch_ordered.restype = i64
    L.orc_match_sa_batch_ordered.argtypes = L.orc_match_sa_batch.argtypes
    L.orc_sampled_open.restype = vp
    L.orc_sampled_open.argtypes = [vp, u64, u64, i32, P(i32)]
    L.orc_sampled_close.argtypes = [vp]
    L.orc_sampled_bytes.restype = u64
    L.orc_sampled_bytes.argtypes = [vp]
    L.orc_sampled_occ.restype = i64
    L.orc_sampled_occ.argtypes = [vp, i32, i64]
    L.orc_sampled_cf.restype = i64
    L.orc_sampled_cf.argtypes = [vp, i32]
    L.orc_sampled_search_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp, i32]
    _lib = L
    return L


class OracleError(Exception):
    pass


class IndexOutOfBounds(OracleError):
    """Where the reference throws ArrayIndexOutOfBoundsException."""


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def swap_ext(filename, ext):
    """BWTTempStorage.gen*Filename, bwtmerger.scala:17-48: strip the last
    extension, append the new one."""
    root, _ = os.path.splitext(filename)
    return root + ext


def load_bwt_file(path, bigEndian=True):
    """BWTLoader, bwtmerger.scala:144-174 -> (bytes ndarray, size, eof)"""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 16:
        raise OracleError("File %s too short" % path)
    bo = ">" if bigEndian else "<"
    size, eof = (int(x) for x in raw[:16].view(bo + "i8"))
    if size + 16 != raw.size:
        raise OracleError("File %s bad size %d != %d + 16" % (path, size, raw.size))
    return raw[16:].copy(), size, eof


def load_aux_file(path, bigEndian=True):
    """AUXLoader, bwtmerger.scala:130-142 -> int64[256]"""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size != 2048:
        raise OracleError("File %s bad aux size %d" % (path, raw.size))
    return raw.view((">" if bigEndian else "<") + "i8").astype(np.int64)


def histogram(bwt, eof, threads=1):
    """The .aux counts of an in-memory BWT (EOF slot excluded) -> int64[256]."""
    bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
    out = np.zeros(256, dtype=np.int64)
    lib().orc_histogram(_ptr(bwt), bwt.size, int(eof), int(threads), _ptr(out))
    return out


class NaiveFMSearcher:
    """Restates class NaiveFMSearcher (bwtmerger.scala:335-421) with the
    SuffixAlgo methods it inherits (findex.scala:9-52).  Positions are Python
    ints; `search`/`getPrevRange` return a tuple or None like the Scala Option."""

    def __init__(self, filename=None, bigEndian=True, _mem=None, strict_signed=False, threads=1):
        L = lib()
        err = ctypes.c_int(0)
        if _mem is not None:
            bwt, n, eof, counts = _mem
            bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
            counts = np.ascontiguousarray(counts, dtype=np.int64)
            assert bwt.size == n and counts.size == 256
            self._h = L.orc_open_mem_threads(_ptr(bwt), n, eof, _ptr(counts), int(threads), ctypes.byref(err))
        else:
            self._h = L.orc_open_files(swap_ext(filename, ".bwt").encode(), swap_ext(filename, ".aux").encode(),
                                       1 if bigEndian else 0, ctypes.byref(err))
        if not self._h:
            raise OracleError("oracle open failed: %d" % err.value)
        self._L = L
        self.n = int(L.orc_n(self._h))
        self.eof = int(L.orc_eof(self._h))
        self.strict_signed = strict_signed
        self.K = 256

    @classmethod
    def from_mem(cls, bwt, eof, counts, **kw):
        """threads=T sorts the position list on T cores (same list)."""
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        return cls(_mem=(bwt, int(bwt.size), int(eof), counts), **kw)

    def close(self):
        if self._h:
            self._L.orc_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- SuffixAlgo
    def cf(self, c):
        r = self._L.orc_cf(self._h, int(c))
        if r == ORC_ERR_INDEX:
            raise IndexOutOfBounds(c)
        return int(r)

    def occ(self, c, i):
        if not 0 <= int(c) < 256:
            raise IndexOutOfBounds(c)
        return int(self._L.orc_occ(self._h, int(c), int(i)))

    def search(self, pat):
        pat = np.frombuffer(bytes(pat), dtype=np.uint8)
        sp, ep, st = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32()
        r = self._L.orc_search(self._h, _ptr(pat) if pat.size else None, pat.size,
                               1 if self.strict_signed else 0, ctypes.byref(sp), ctypes.byref(ep), ctypes.byref(st))
        if r < 0:
            raise IndexOutOfBounds("byte >= 0x80 in pattern")
        return (sp.value, ep.value) if r == 1 else None

    def search_raw(self, pat):
        """(found, sp, ep, steps): the loop's final values, also for a miss."""
        pat = np.frombuffer(bytes(pat), dtype=np.uint8)
        sp, ep, st = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32()
        r = self._L.orc_search(self._h, _ptr(pat) if pat.size else None, pat.size, 0,
                               ctypes.byref(sp), ctypes.byref(ep), ctypes.byref(st))
        return r == 1, sp.value, ep.value, st.value

    def getPrevRange(self, sp, ep, c):
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        r = self._L.orc_get_prev_range(self._h, int(sp), int(ep), int(c), ctypes.byref(a), ctypes.byref(b))
        if r < 0:
            raise IndexOutOfBounds(c)
        return (a.value, b.value) if r == 1 else None

    def getIntervalPrevRange(self, sp, ep, cstart, cend):
        k = max(0, int(cend) - int(cstart) + 1)
        osp = np.zeros(max(k, 1), dtype=np.uint64)
        oep = np.zeros(max(k, 1), dtype=np.uint64)
        r = self._L.orc_get_interval_prev_range(self._h, int(sp), int(ep), int(cstart), int(cend),
                                                _ptr(osp), _ptr(oep), None)
        if r < 0:
            raise IndexOutOfBounds((cstart, cend))
        return [(int(osp[j]), int(oep[j])) for j in range(r)]

    # --- NaiveFMSearcher extras
    def bwt_read(self, i):
        r = self._L.orc_bwt_read(self._h, int(i))
        if r < 0:
            raise IndexOutOfBounds(i)
        return r

    def pos2char(self, key):
        return int(self._L.orc_pos2char(self._h, int(key)))

    def getPrevI(self, i):
        r = self._L.orc_get_prev_i(self._h, int(i))
        if r < 0:
            raise IndexOutOfBounds(i)
        return int(r)

    def getNextI(self, i):
        r = self._L.orc_get_next_i(self._h, int(i))
        if r < 0:
            raise IndexOutOfBounds(i)
        return int(r)

    def nextSubstr(self, sp, length):
        out = np.zeros(max(int(length), 1), dtype=np.uint8)
        k = self._L.orc_next_substr(self._h, int(sp), int(length), _ptr(out))
        if k < 0:
            raise IndexOutOfBounds(sp)
        return bytes(out[:k])

    def prevSubstr(self, sp, length):
        out = np.zeros(max(int(length), 1), dtype=np.uint8)
        k = self._L.orc_prev_substr(self._h, int(sp), int(length), _ptr(out))
        if k < 0:
            raise IndexOutOfBounds(sp)
        return bytes(out[:k])

    def lf_chain(self, row, steps):
        """`steps` dependent LF steps from `row` -> the row reached (one core, in C)."""
        return int(self._L.orc_lf_chain(self._h, int(row), int(steps)))

    def occ_chain(self, text, rank0):
        """calcGaps' rank chain (bwtmerger.scala:999-1001) over `text` from curRank = rank0, in C on one core."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        ranks = np.zeros(max(text.size, 1), dtype=np.int64)
        self._L.orc_occ_chain(self._h, _ptr(text), text.size, int(rank0), _ptr(ranks))
        return ranks[: text.size].astype(np.uint64)

    def fm(self):
        """The inverted list (= .fm payload) as a uint32 view copy."""
        p = self._L.orc_fm_ptr(self._h)
        a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32)), shape=(self.n,))
        return a.copy()

    def write_fm(self, path):
        if self._L.orc_write_fm(self._h, path.encode()) != 0:
            raise OracleError("cannot write %s" % path)

    # --- batches (bench.py cpu_baseline and parity tests)
    def occ_batch(self, c, i):
        c = np.ascontiguousarray(c, dtype=np.uint8)
        i = np.ascontiguousarray(i, dtype=np.int64)
        out = np.zeros(c.size, dtype=np.int64)
        self._L.orc_occ_batch(self._h, _ptr(c), _ptr(i), _ptr(out), c.size)
        return out

    def search_batch(self, pat, off, threads=1):
        pat = np.ascontiguousarray(pat, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        k = off.size - 1
        sp = np.zeros(k, dtype=np.uint64)
        ep = np.zeros(k, dtype=np.uint64)
        steps = np.zeros(k, dtype=np.uint32)
        self._L.orc_search_batch(self._h, _ptr(pat), _ptr(off), k, _ptr(sp), _ptr(ep), _ptr(steps), int(threads))
        return sp, ep, steps

    def prev_range_batch(self, sp, ep, c):
        sp = np.ascontiguousarray(sp, dtype=np.uint64)
        ep = np.ascontiguousarray(ep, dtype=np.uint64)
        c = np.ascontiguousarray(c, dtype=np.uint8)
        sp1 = np.zeros(sp.size, dtype=np.uint64)
        ep1 = np.zeros(sp.size, dtype=np.uint64)
        self._L.orc_prev_range_batch(self._h, _ptr(sp), _ptr(ep), _ptr(c), _ptr(sp1), _ptr(ep1), sp.size)
        return sp1, ep1

    def match_tables(self, t, maxBranching=1024, maxIterations=1000, cap=1 << 20):
        """C _matchSA over ReTree.tables(); returns (results newest-first as
        (len, sp, ep), leftover frontier size, getPrevRange calls)."""
        ns = len(t["c"])
        c = np.asarray(t["c"], dtype=np.uint8)
        num = np.asarray(t["num"], dtype=np.int32)
        last = np.asarray(t["isLast"], dtype=np.uint8)
        off = np.zeros(ns + 1, dtype=np.int32)
        for k, f in enumerate(t["follows"]):
            off[k + 1] = off[k] + len(f)
        fol = np.asarray([x for f in t["follows"] for x in f] or [0], dtype=np.int32)
        firsts = np.asarray(t["firsts"] or [0], dtype=np.int32)
        rl = np.zeros(cap, dtype=np.int64)
        rs = np.zeros(cap, dtype=np.uint64)
        re_ = np.zeros(cap, dtype=np.uint64)
        left, pops = ctypes.c_int64(), ctypes.c_int64()
        r = self._L.orc_match_sa(self._h, ns, _ptr(c), _ptr(num), _ptr(last), _ptr(off), _ptr(fol), _ptr(firsts),
                                 len(t["firsts"]), int(maxBranching), int(maxIterations),
                                 _ptr(rl), _ptr(rs), _ptr(re_), cap, ctypes.byref(left), ctypes.byref(pops))
        if r < 0:
            raise OracleError("orc_match_sa: %d" % r)
        k = min(int(r), cap)
        return [(int(rl[j]), int(rs[j]), int(re_[j])) for j in range(k)], left.value, pops.value


    def match_tables_batch(self, tables, maxBranching=1 << 40, maxIterations=0, max_len=0, threads=1, ordered=False):
        """C _matchSA over a list of ReTree.tables() dicts on `threads` cores; max_len > 0 caps the match
        length like the product's max_steps.  Returns (structured array of (regex, len, sp, ep) grouped by regex
        and sorted by (len, sp, ep) inside a group -- ordered=True: in the reference's own list order, newest
        first, what ReTree.matchSA returns when its limits bind --, getPrevRange calls made, regexes cut at
        max_len)."""
        k = len(tables)
        st_off = np.zeros(k + 1, dtype=np.int64)
        fol_base = np.zeros(k + 1, dtype=np.int64)
        first_off = np.zeros(k + 1, dtype=np.int64)
        for r, t in enumerate(tables):
            st_off[r + 1] = st_off[r] + len(t["c"])
            fol_base[r + 1] = fol_base[r] + sum(len(f) for f in t["follows"])
            first_off[r + 1] = first_off[r] + len(t["firsts"])
        ns = int(st_off[-1])
        st_c = np.zeros(max(ns, 1), dtype=np.uint8)
        st_num = np.zeros(max(ns, 1), dtype=np.int32)
        st_last = np.zeros(max(ns, 1), dtype=np.uint8)
        fol_off = np.zeros(ns + k + 1, dtype=np.int32)
        fol = np.zeros(max(int(fol_base[-1]), 1), dtype=np.int32)
        firsts = np.zeros(max(int(first_off[-1]), 1), dtype=np.int32)
        for r, t in enumerate(tables):
            a, b = int(st_off[r]), int(st_off[r + 1])
            st_c[a:b] = t["c"]
            st_num[a:b] = t["num"]
            st_last[a:b] = t["isLast"]
            lens = np.fromiter((len(f) for f in t["follows"]), dtype=np.int64, count=b - a)
            fol_off[a + r + 1:b + r + 1] = np.cumsum(lens)
            flat = [x for f in t["follows"] for x in f]
            fol[int(fol_base[r]):int(fol_base[r]) + len(flat)] = flat
            firsts[int(first_off[r]):int(first_off[r + 1])] = t["firsts"]
        res_start = np.zeros(k + 1, dtype=np.int64)
        cap = 1 << 16
        while True:
            out_len = np.zeros(cap, dtype=np.int64)
            out_sp = np.zeros(cap, dtype=np.uint64)
            out_ep = np.zeros(cap, dtype=np.uint64)
            pops, trunc = ctypes.c_int64(), ctypes.c_int64()
            fn = self._L.orc_match_sa_batch_ordered if ordered else self._L.orc_match_sa_batch
            got = fn(self._h, k, _ptr(st_off), _ptr(st_c), _ptr(st_num), _ptr(st_last),
                                             _ptr(fol_off), _ptr(fol_base), _ptr(fol), _ptr(first_off), _ptr(firsts),
                                             int(maxBranching), int(maxIterations), int(max_len), int(threads),
                                             _ptr(res_start), _ptr(out_len), _ptr(out_sp), _ptr(out_ep), cap,
                                             ctypes.byref(pops), ctypes.byref(trunc))
            if got < 0:
                raise OracleError("orc_match_sa_batch: %d" % got)
            if got <= cap:
                break
            cap = int(got)
        out = np.zeros(int(got), dtype=[("regex", np.uint32), ("len", np.uint32), ("sp", np.uint64), ("ep", np.uint64)])
        out["regex"] = np.repeat(np.arange(k, dtype=np.uint32), np.diff(res_start))
        out["len"] = out_len[:got]
        out["sp"] = out_sp[:got]
        out["ep"] = out_ep[:got]
        return out, int(pops.value), int(trunc.value)


class SampledFMSearcher:
    """occ / cf / search over per-256-position symbol checkpoints + a scan of the BWT bytes (fmx_oracle.c, "Sampled-checkpoint
    variant"): the CPU baseline for indexes the inverted lists cannot describe (n > 2^32, or no 6 n bytes of host memory) --
    BASELINE.md's "sampled popcount structure".  Same function as NaiveFMSearcher.occ / .search, another data structure;
    held to it by tests/test_oracle_kat.py.  The BWT array is NOT copied: keep it alive."""

    def __init__(self, bwt, eof, threads=1):
        self._L = lib()
        self._bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        err = ctypes.c_int(0)
        self._h = self._L.orc_sampled_open(_ptr(self._bwt), self._bwt.size, int(eof), int(threads), ctypes.byref(err))
        if not self._h:
            raise OracleError("orc_sampled_open failed (%d)" % err.value)
        self.n = int(self._bwt.size)
        self.eof = int(eof)

    def close(self):
        if getattr(self, "_h", None):
            self._L.orc_sampled_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bytes(self):
        return int(self._L.orc_sampled_bytes(self._h))

    def cf(self, c):
        return int(self._L.orc_sampled_cf(self._h, int(c)))

    def occ(self, c, i):
        return int(self._L.orc_sampled_occ(self._h, int(c), int(i)))

    def search_batch(self, pat, off, threads=1):
        pat = np.ascontiguousarray(pat, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        k = off.size - 1
        sp = np.zeros(k, dtype=np.uint64)
        ep = np.zeros(k, dtype=np.uint64)
        steps = np.zeros(k, dtype=np.uint32)
        self._L.orc_sampled_search_batch(self._h, _ptr(pat), _ptr(off), k, _ptr(sp), _ptr(ep), _ptr(steps), int(threads))
        return sp, ep, steps


class SAISNaiveSearcher(NaiveFMSearcher):
    """The in-memory searcher the reference's small-string tests use
    (SAISBuilder with NaiveSearcher, sais.scala:95-148, findex.scala:415-456):
    same cf/occ/search, but its substring walkers mirror NaiveFMSearcher's --
    prevSubstr reverses (sais.scala:110-118) and nextSubstr neither reverses nor
    stops at the EOF byte (sais.scala:140-148)."""

    def nextSubstr(self, sp, length):
        cp = self.getNextI(sp)
        out = bytearray()
        for _ in range(int(length)):
            out.append(self.bwt_read(cp))
            cp = self.getNextI(cp)
        return bytes(out)

    def prevSubstr(self, sp, length):
        return NaiveFMSearcher.prevSubstr(self, sp, length)[::-1]
