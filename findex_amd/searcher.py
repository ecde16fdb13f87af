"""HipFMSearcher: the drop-in for findex's NaiveFMSearcher (bwtmerger.scala:335-421) and the
SuffixAlgo / SuffixWalkingAlgo traits it implements (findex.scala:9-57), served by libfmx.so."""
import ctypes
import os

import numpy as np

from . import _lib


def _swap_ext(filename, ext):
    """BWTTempStorage.gen*Filename, bwtmerger.scala:17-48"""
    return os.path.splitext(filename)[0] + ext


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None


def _dp(x):
    return ctypes.c_void_p(int(x) if x else 0)


class PinnedArray:
    """A numpy view of page-locked host memory from fmx_host_alloc (freed with the object): batch buffers that the
    host-pointer entry points move by DMA at link speed."""

    def __init__(self, shape, dtype):
        self._L = _lib.load()
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        self._p = ctypes.c_void_p()
        _lib.check(self._L.fmx_host_alloc(max(n, 1) * self.dtype.itemsize, ctypes.byref(self._p)))
        buf = (ctypes.c_uint8 * (max(n, 1) * self.dtype.itemsize)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=n).reshape(shape)

    def __del__(self):
        try:
            if self._p:
                self.array = None
                self._L.fmx_host_free(self._p)
                self._p = None
        except Exception:
            pass


class HipFMSearcher:
    """`new NaiveFMSearcher(filename, bigEndian)`: opens X.bwt and X.aux next to `filename` and
    builds the rank dictionary in HBM.  The reference's X.fm is not needed; when it is there it must
    pass FMLoader's checks and hold the .bwt's n rows (fmx_open), or the open fails as the reference's does.

    Scalar methods keep the reference's names and Option-like results (tuple or None); each
    `*_batch` method is the batched form the kernels are built for.  Positions are Python ints /
    uint64 arrays."""

    def __init__(self, filename=None, bigEndian=True, device=0, _handle=None):
        self._L = _lib.load()
        self._h = ctypes.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _lib.check(self._L.fmx_open(_swap_ext(filename, ".bwt").encode(), _swap_ext(filename, ".aux").encode(),
                                        1 if bigEndian else 0, int(device), ctypes.byref(self._h)))
        v = ctypes.c_uint64()
        _lib.check(self._L.fmx_n(self._h, ctypes.byref(v)))
        self.n = int(v.value)
        _lib.check(self._L.fmx_eof(self._h, ctypes.byref(v)))
        self.eof = int(v.value)
        self.K = 256
        cf = np.zeros(256, dtype=np.uint64)
        for c in range(256):
            _lib.check(self._L.fmx_cf(self._h, c, ctypes.byref(v)))
            cf[c] = v.value
        self.bucketStarts = cf

    # ---- alternative constructors
    @classmethod
    def from_mem(cls, bwt, eof, counts, device=0):
        """fmx_open_mem: BWT bytes (filler at slot eof) + the .aux counts, from host memory."""
        L = _lib.load()
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        if counts.size != 256:
            raise ValueError("counts must have 256 entries")
        h = ctypes.c_void_p()
        _lib.check(L.fmx_open_mem(_ptr(bwt), bwt.size, int(eof), _ptr(counts), int(device), ctypes.byref(h)))
        return cls(_handle=h)

    @classmethod
    def from_device(cls, d_bwt_ptr, n, eof, counts=None, device=0, stream=0):
        """fmx_open_dev: BWT bytes already resident in HBM (e.g. tensor.data_ptr())."""
        L = _lib.load()
        c = None if counts is None else np.ascontiguousarray(counts, dtype=np.int64)
        h = ctypes.c_void_p()
        _lib.check(L.fmx_open_dev(_dp(d_bwt_ptr), int(n), int(eof), _ptr(c), int(device), _dp(stream),
                                  ctypes.byref(h)))
        return cls(_handle=h)

    @classmethod
    def from_block(cls, bwt, bucketStarts, rk0, device=0):
        """`new NaiveBWTSearcher(bwt, bucketStarts, rk0)` (findex.scala:459-506): one merge block's BWT, the caller's
        bucket starts, row rk0 skipped."""
        L = _lib.load()
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        bs = np.ascontiguousarray(bucketStarts, dtype=np.int64)
        if bs.size != 256:
            raise ValueError("bucketStarts must have 256 entries")
        h = ctypes.c_void_p()
        _lib.check(L.fmx_open_block(_ptr(bwt), bwt.size, _ptr(bs), int(rk0), int(device), ctypes.byref(h)))
        return cls(_handle=h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.fmx_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- SuffixAlgo (findex.scala:9-52)
    def cf(self, c):
        if not 0 <= int(c) < 256:
            raise IndexError("symbol %r (reference: ArrayIndexOutOfBoundsException)" % (c,))
        return int(self.bucketStarts[int(c)])

    def occ(self, c, i):
        if not 0 <= int(c) < 256:
            raise IndexError("symbol %r (reference: ArrayIndexOutOfBoundsException)" % (c,))
        return int(self.occ_batch(np.array([c], dtype=np.uint8), np.array([i], dtype=np.int64))[0])

    def search(self, pat):
        pat = bytes(pat)
        sp, ep = self.search_batch(np.frombuffer(pat, dtype=np.uint8), np.array([0, len(pat)], dtype=np.uint64))
        return (int(sp[0]), int(ep[0])) if sp[0] < ep[0] else None

    def getPrevRange(self, sp, ep, c):
        if not 0 <= int(c) < 256:
            raise IndexError("symbol %r (reference: ArrayIndexOutOfBoundsException)" % (c,))
        a, b = self.prev_range_batch(np.array([sp], dtype=np.uint64), np.array([ep], dtype=np.uint64),
                                     np.array([c], dtype=np.uint8))
        return (int(a[0]), int(b[0])) if a[0] < b[0] else None

    def getIntervalPrevRange(self, sp, ep, cstart, cend):
        k = max(int(cend) - int(cstart) + 1, 1)
        osp = np.zeros(k, dtype=np.uint64)
        oep = np.zeros(k, dtype=np.uint64)
        n_out = ctypes.c_size_t()
        _lib.check(self._L.fmx_interval_prev_range(self._h, int(sp), int(ep), int(cstart), int(cend), _ptr(osp),
                                                   _ptr(oep), None, ctypes.byref(n_out)))
        return [(int(osp[j]), int(oep[j])) for j in range(n_out.value)]

    # ---- NaiveFMSearcher walkers (bwtmerger.scala:376-419)
    def getPrevI(self, i):
        _, end = self.lf_walk_batch(np.array([i], dtype=np.uint64), 1, want_bytes=False)
        return int(end[0])

    def getNextI(self, i):
        return int(self.psi_batch(np.array([i], dtype=np.uint64))[0])

    def bwt_read(self, i):
        """BWTLoader.read (bwtmerger.scala:155-162): 0 at the EOF slot."""
        b, _ = self.lf_walk_batch(np.array([i], dtype=np.uint64), 1)
        return int(b[0, 0])

    def nextSubstr(self, sp, length):
        out = np.zeros(max(int(length), 1), dtype=np.uint8)
        w = ctypes.c_uint32()
        _lib.check(self._L.fmx_next_substr(self._h, int(sp), int(length), _ptr(out), ctypes.byref(w)))
        return bytes(out[: w.value])

    def nextSubstr_batch(self, rows, length):
        """nextSubstr for many rows in one call -> list of bytes (what SAResult.toString prints per result)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros((rows.size, max(int(length), 1)), dtype=np.uint8)
        w = np.zeros(max(rows.size, 1), dtype=np.uint32)
        _lib.check(self._L.fmx_next_substr_batch(self._h, _ptr(rows), rows.size, int(length), _ptr(out), _ptr(w)))
        if int(length) == 0:
            return [b""] * rows.size
        flat = out.reshape(-1)
        return [bytes(flat[q * int(length): q * int(length) + int(w[q])]) for q in range(rows.size)]

    def prevSubstr(self, sp, length):
        out = np.zeros(max(int(length), 1), dtype=np.uint8)
        _lib.check(self._L.fmx_prev_substr(self._h, int(sp), int(length), _ptr(out)))
        return bytes(out[: int(length)])

    @staticmethod
    def search_batch_multi(searchers, pat, off):
        """fmx_search_batch_multi: one process, one replica handle per GPU; the batch is cut by pattern bytes."""
        pat = np.ascontiguousarray(pat, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        k = off.size - 1
        sp = np.zeros(max(k, 0), dtype=np.uint64)
        ep = np.zeros(max(k, 0), dtype=np.uint64)
        arr = (ctypes.c_void_p * len(searchers))(*[s._h for s in searchers])
        _lib.check(_lib.load().fmx_search_batch_multi(arr, len(searchers), _ptr(pat), _ptr(off), _ptr(sp), _ptr(ep), max(k, 0)))
        return sp, ep

    def extract(self, row, length, direction=1):
        """fmx_extract: direction > 0 = nextSubstr, < 0 = prevSubstr."""
        out = np.zeros(max(int(length), 1), dtype=np.uint8)
        w = ctypes.c_uint32()
        _lib.check(self._L.fmx_extract(self._h, int(row), int(length), int(direction), _ptr(out), ctypes.byref(w)))
        return bytes(out[: w.value])

    def write_fm(self, path):
        """FMCreator.create (bwtmerger.scala:452-532): write the reference's .fm file."""
        _lib.check(self._L.fmx_write_fm(self._h, str(path).encode()))

    # ---- batched forms (host arrays in, host arrays out)
    def occ_batch(self, c, i):
        c = np.ascontiguousarray(c, dtype=np.uint8)
        i = np.ascontiguousarray(i, dtype=np.int64)
        if c.size != i.size:
            raise ValueError("c and i differ in length")
        out = np.zeros(c.size, dtype=np.uint64)
        _lib.check(self._L.fmx_occ_batch(self._h, _ptr(c), _ptr(i), _ptr(out), c.size))
        return out

    def search_batch(self, pat, off, out=None):
        """out = (sp, ep) uint64 arrays to fill (e.g. PinnedArray views), else new ones."""
        pat = np.ascontiguousarray(pat, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        k = off.size - 1
        if k < 0:
            raise ValueError("off needs k+1 entries")
        if k and int(off[-1]) > pat.size:
            raise ValueError("offsets run past the pattern buffer")
        if out is not None:
            sp, ep = out
            if sp.size != k or ep.size != k or sp.dtype != np.uint64 or ep.dtype != np.uint64:
                raise ValueError("out arrays must be uint64 of length k")
        else:
            sp = np.zeros(k, dtype=np.uint64)
            ep = np.zeros(k, dtype=np.uint64)
        _lib.check(self._L.fmx_search_batch(self._h, _ptr(pat), _ptr(off), _ptr(sp), _ptr(ep), k))
        return sp, ep

    def search_batch_ex(self, pat, off=None, fixed_len=0, packed=False, escape_cap=0, miss_none=False):
        """fmx_search_batch_ex: the lean forms of search_batch -- `fixed_len` > 0: k = len(pat) // fixed_len patterns of
        that length, no offsets travel; `packed`: the intervals come back in the 8-byte form (one uint64 array of
        fmx_packed_words(k, escape_cap) words: unpack_intervals).  Returns (sp, ep) or the packed array."""
        pat = np.ascontiguousarray(pat, dtype=np.uint8)
        if fixed_len:
            if pat.size % int(fixed_len):
                raise ValueError("the pattern buffer is not a whole number of patterns")
            k, offp = pat.size // int(fixed_len), None
        else:
            off = np.ascontiguousarray(off, dtype=np.uint64)
            k, offp = off.size - 1, _ptr(off)
            if k and int(off[-1]) > pat.size:
                raise ValueError("offsets run past the pattern buffer")
        opts = _lib.fmx_search_opts(int(fixed_len), (1 if packed else 0) | (2 if miss_none else 0), int(escape_cap))
        if packed:
            out = np.zeros(int(self._L.fmx_packed_words(k, int(escape_cap))), dtype=np.uint64)
            _lib.check(self._L.fmx_search_batch_ex(self._h, _ptr(pat), offp, _ptr(out), None, k, ctypes.byref(opts)))
            return out
        sp = np.zeros(k, dtype=np.uint64)
        ep = np.zeros(k, dtype=np.uint64)
        _lib.check(self._L.fmx_search_batch_ex(self._h, _ptr(pat), offp, _ptr(sp), _ptr(ep), k, ctypes.byref(opts)))
        return sp, ep

    def unpack_intervals(self, packed, k, escape_cap=0):
        """fmx_unpack_intervals (host): the 8-byte form -> (sp, ep); raises FMX_ERR_OVERFLOW when more intervals were wide
        than the escape list holds."""
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        sp = np.zeros(int(k), dtype=np.uint64)
        ep = np.zeros(int(k), dtype=np.uint64)
        _lib.check(self._L.fmx_unpack_intervals(_ptr(packed) if packed.size else None, int(k), int(escape_cap), _ptr(sp), _ptr(ep)))
        return sp, ep

    def prev_range_batch(self, sp, ep, c):
        sp = np.ascontiguousarray(sp, dtype=np.uint64)
        ep = np.ascontiguousarray(ep, dtype=np.uint64)
        c = np.ascontiguousarray(c, dtype=np.uint8)
        if not (sp.size == ep.size == c.size):
            raise ValueError("sp, ep and c differ in length")
        sp1 = np.zeros(sp.size, dtype=np.uint64)
        ep1 = np.zeros(sp.size, dtype=np.uint64)
        _lib.check(self._L.fmx_prev_range_batch(self._h, _ptr(sp), _ptr(ep), _ptr(c), _ptr(sp1), _ptr(ep1), sp.size))
        return sp1, ep1

    def lf_walk_batch(self, rows, length, want_bytes=True):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros((rows.size, int(length)), dtype=np.uint8) if want_bytes else None
        end = np.zeros(rows.size, dtype=np.uint64)
        _lib.check(self._L.fmx_lf_walk_batch(self._h, _ptr(rows), rows.size, int(length),
                                             _ptr(out) if want_bytes else None, _ptr(end)))
        return out, end

    def psi_batch(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(rows.size, dtype=np.uint64)
        _lib.check(self._L.fmx_psi_batch(self._h, _ptr(rows), _ptr(out), rows.size))
        return out

    # ---- device-pointer forms (inputs resident in HBM; only enqueue on `stream`)
    def search_batch_dev(self, d_pat, d_off, d_sp, d_ep, k, stream=0):
        _lib.check(self._L.fmx_search_batch_dev(self._h, _dp(d_pat), _dp(d_off), _dp(d_sp), _dp(d_ep), int(k),
                                                _dp(stream)))

    def search_batch_ex_dev(self, d_pat, d_off, d_sp, d_ep, k, stream=0, fixed_len=0, packed=False, escape_cap=0, miss_none=False):
        """fmx_search_batch_ex_dev: d_off may be 0 with fixed_len; with `packed` d_sp receives the packed words (it needs
        fmx_packed_words(k, escape_cap) of them) and d_ep is k words of scratch."""
        opts = _lib.fmx_search_opts(int(fixed_len), (1 if packed else 0) | (2 if miss_none else 0), int(escape_cap))
        _lib.check(self._L.fmx_search_batch_ex_dev(self._h, _dp(d_pat), _dp(d_off), _dp(d_sp), _dp(d_ep), int(k),
                                                   ctypes.byref(opts), _dp(stream)))

    def packed_words(self, k, escape_cap=0):
        return int(self._L.fmx_packed_words(int(k), int(escape_cap)))

    def pack_intervals_dev(self, d_sp, d_ep, k, d_packed, escape_cap=0, stream=0):
        _lib.check(self._L.fmx_pack_intervals_dev(self._h, _dp(d_sp), _dp(d_ep), int(k), int(escape_cap), _dp(d_packed), _dp(stream)))

    def unpack_intervals_dev(self, d_packed, k, d_sp, d_ep, escape_cap=0, stream=0):
        _lib.check(self._L.fmx_unpack_intervals_dev(self._h, _dp(d_packed), int(k), int(escape_cap), _dp(d_sp), _dp(d_ep), _dp(stream)))

    def occ_batch_dev(self, d_c, d_i, d_out, k, stream=0):
        _lib.check(self._L.fmx_occ_batch_dev(self._h, _dp(d_c), _dp(d_i), _dp(d_out), int(k), _dp(stream)))

    def prev_range_batch_dev(self, d_sp, d_ep, d_c, d_sp1, d_ep1, k, stream=0):
        _lib.check(self._L.fmx_prev_range_batch_dev(self._h, _dp(d_sp), _dp(d_ep), _dp(d_c), _dp(d_sp1), _dp(d_ep1),
                                                    int(k), _dp(stream)))

    def lf_walk_batch_dev(self, d_rows, k, length, d_out_bytes, d_end_rows, stream=0):
        _lib.check(self._L.fmx_lf_walk_batch_dev(self._h, _dp(d_rows), int(k), int(length), _dp(d_out_bytes),
                                                 _dp(d_end_rows), _dp(stream)))

    def psi_batch_dev(self, d_rows, d_out, k, stream=0):
        _lib.check(self._L.fmx_psi_batch_dev(self._h, _dp(d_rows), _dp(d_out), int(k), _dp(stream)))

    def next_substr_batch_dev(self, d_rows, k, length, d_out, d_out_len, stream=0):
        _lib.check(self._L.fmx_next_substr_batch_dev(self._h, _dp(d_rows), int(k), int(length), _dp(d_out),
                                                     _dp(d_out_len), _dp(stream)))

    # ---- BWTMerger2.calcGaps' rank loop (bwtmerger.scala:981-1023): one dependent chain, answered on the host
    def occ_host(self, c, i):
        """fmx_occ_host: occ(c, i) from the host-side copy of the dictionary (same answer as occ)."""
        if not 0 <= int(c) < 256:
            raise IndexError("symbol %r (reference: ArrayIndexOutOfBoundsException)" % (c,))
        v = ctypes.c_uint64()
        _lib.check(self._L.fmx_occ_host(self._h, int(c), int(i), ctypes.byref(v)))
        return int(v.value)

    def calc_gaps_chain(self, text, rank0=0, last_char=-1, rklst=0):
        """fmx_calc_gaps_chain: (ranks, done) -- ranks[j] = calcGaps' curRank after byte j; done < len(text) when the
        chain stopped at a rank equal to rklst for the caller to decide (bwtmerger.scala:1004-1010)."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        ranks = np.zeros(max(text.size, 1), dtype=np.uint64)
        done = ctypes.c_size_t()
        _lib.check(self._L.fmx_calc_gaps_chain(self._h, _ptr(text), text.size, int(rank0), int(last_char), int(rklst),
                                               _ptr(ranks), ctypes.byref(done)))
        return ranks[: text.size], int(done.value)

    def prepare(self, ktab=True, select=False, jump=False, frontier=False, search=False, budget_bytes=0):
        """fmx_prepare[_ex]: build the k-mer jump table / the select directory / the literal search's row tables (J, R3) / the
        regex frontier's row table now instead of at the threshold or at first use, and calibrate the search kernel they
        select (`search` alone: only that).  budget_bytes != 0: the handle's "table_budget" first (fmx_prepare_ex)."""
        what = (1 if ktab else 0) | (2 if select else 0) | (4 if jump else 0) | (8 if frontier else 0) | (16 if search else 0)
        if budget_bytes:
            _lib.check(self._L.fmx_prepare_ex(self._h, what, int(budget_bytes)))
        else:
            _lib.check(self._L.fmx_prepare(self._h, what))

    def drop_tables(self, jump=True, frontier=True, ktab=False):
        """fmx_drop_tables: free the row jump table and the three-step row table / the frontier's row table / the k-mer table."""
        _lib.check(self._L.fmx_drop_tables(self._h, (4 if jump else 0) | (8 if frontier else 0) | (1 if ktab else 0)))

    def config_set(self, key, value):
        """fmx_index_config_set: this handle's own table policy ("ktab", "jump", "jump_pairs", "jump_chars", "tables_after",
        "table_budget"); tables that exist stay until drop_tables."""
        _lib.check(self._L.fmx_index_config_set(self._h, key.encode(), str(value).encode()))

    # ---- statistics
    def stats(self):
        s = _lib.fmx_stats_t()
        _lib.check(self._L.fmx_stats(self._h, ctypes.byref(s)))
        return {f: getattr(s, f) for f, _ in s._fields_}

    def last_kernel_ms(self):
        v = ctypes.c_double()
        _lib.check(self._L.fmx_last_kernel_ms(self._h, ctypes.byref(v)))
        return float(v.value)

    def stats_reset(self):
        _lib.check(self._L.fmx_stats_reset(self._h))
