"""REParser / ReTree / SAResult: the regex-search surface of findex (re2/re2.scala:9-205,
re2/retree.scala) over libfmx.so.  Parsing and the Glushkov tables are built by the library's
C++ host code; the SA-interval frontier is expanded on the GPU."""
import ctypes

import numpy as np

from . import _lib


class PostfixRe:
    """REParser.PostfixRe (re2.scala:49): the postfix token list, kept as the source string the
    library re-parses; str() gives re2poststr."""

    def __init__(self, src, lineOnly):
        self.src = src
        self.lineOnly = bool(lineOnly)
        raw = src.encode("latin-1")
        buf = ctypes.create_string_buffer(600 * len(raw) + 1024)     # '.' can print as 1 char, sets as many
        _lib.check(_lib.load().fmx_regex_post_string(raw, 1 if lineOnly else 0, buf, len(buf)))
        self.text = buf.value.decode("utf-8")

    def __str__(self):
        return self.text


class PostfixString(PostfixRe):
    """REParser.post2re(str) (re2.scala:188-205): a postfix string, '.' = concat."""

    def __init__(self, src):
        self.src = src
        self.lineOnly = False
        self.text = src
        self.is_postfix = True


class NFA:
    """REParser.createNFA(postfix) (re2.scala:264-334): the Thompson NFA handle for
    REParser.matchSA.  Raises MatchError where the reference throws scala.MatchError."""

    def __init__(self, postfix):
        self._L = _lib.load()
        self._h = ctypes.c_void_p()
        _lib.check(self._L.fmx_nfa_compile(postfix.src.encode("latin-1"), 1 if postfix.lineOnly else 0,
                                           1 if getattr(postfix, "is_postfix", False) else 0, ctypes.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._L.fmx_regex_free(self._h)
                self._h = None
        except Exception:
            pass


class DFA:
    """class DFA(nstates, nchars) (dfa.scala:114-289) filled with addLink / finishStates, then
    compileBuckets() and matchSA(sa) as in the reference."""

    def __init__(self, nstates, nchars=256):
        self.nstates, self.nchars = int(nstates), int(nchars)
        self.moves = np.full((self.nstates, self.nchars), -1, dtype=np.int32)
        self.finishStates = set()
        self._h = None
        self._L = _lib.load()

    def addLink(self, frm, to, ch):
        self.moves[int(frm), int(ch)] = int(to)
        self._h = None

    def compileBuckets(self):
        fin = np.zeros(self.nstates, dtype=np.uint8)
        for s in self.finishStates:
            fin[s] = 1
        h = ctypes.c_void_p()
        _lib.check(self._L.fmx_dfa_compile(self.moves.ctypes.data_as(ctypes.c_void_p), self.nstates, self.nchars,
                                           fin.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)))
        self._h = h

    def matchSA(self, sa, max_steps=0, cap=1 << 20):
        """DFA.matchSA (dfa.scala:261-289) -> list of SAResult (DFAResult there), sorted."""
        if self._h is None:
            self.compileBuckets()
        return ReTree.matchSA_batch(sa, [self], max_steps=max_steps, cap=cap)[0]

    def __del__(self):
        try:
            if self._h:
                self._L.fmx_regex_free(self._h)
                self._h = None
        except Exception:
            pass


class REParser:
    @staticmethod
    def post2re(s):
        """REParser.post2re, re2.scala:188-205"""
        return PostfixString(s)

    @staticmethod
    def createNFA(postfix):
        """REParser.createNFA, re2.scala:264-334"""
        return NFA(postfix)

    @staticmethod
    def matchSA(nfa, sa, debugLevel=0, maxIterations=0, maxLength=0, cap=1 << 20):
        """REParser.matchSA (re2.scala:568-693) -> list of SAResult, sorted by (len, sp, ep): the
        reference's result multiset with its default maxIterations = 0 (unbounded)."""
        if maxIterations:
            raise NotImplementedError("REParser.matchSA's maxIterations cut depends on the reference's queue order")
        return ReTree.matchSA_batch(sa, [nfa], max_steps=maxLength, cap=cap)[0]

    @staticmethod
    def re2post(s, lineOnly=False):
        """REParser.re2post, re2.scala:50-185 (raises Re2PostSyntax like the reference's Exception)."""
        return PostfixRe(s, lineOnly)

    @staticmethod
    def re2poststr(s):
        """re2.scala:187"""
        return str(PostfixRe(s, False))


class SAResult:
    """SAResult(sa, len, sp, ep), re2.scala:9-19."""

    def __init__(self, sa, length, sp, ep):
        self.sa, self.len, self.sp, self.ep = sa, int(length), int(sp), int(ep)
        self.cnt = self.ep - self.sp

    @property
    def strResult(self):
        if self.cnt == 1:
            return self.sa.nextSubstr(self.sp, self.len).decode("latin-1")
        if self.cnt > 0:
            return "[%d Results] %s" % (self.cnt, self.sa.nextSubstr(self.sp, self.len).decode("latin-1"))
        return "[no results]"

    def __str__(self):
        return self.strResult

    def __repr__(self):
        return "SAResult(len=%d, sp=%d, ep=%d)" % (self.len, self.sp, self.ep)

    def key(self):
        return (self.len, self.sp, self.ep)


class ReTree:
    """ReTree(postfix) (retree.scala:156-370): raises MatchError for the operand shapes the
    reference has no case for."""

    def __init__(self, postfix):
        if isinstance(postfix, str):
            postfix = PostfixRe(postfix, False)
        self._L = _lib.load()
        self.postfix = postfix
        self._h = ctypes.c_void_p()
        _lib.check(self._L.fmx_regex_compile(postfix.src.encode("latin-1"), 1 if postfix.lineOnly else 0,
                                             ctypes.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._L.fmx_regex_free(self._h)
                self._h = None
        except Exception:
            pass

    def tables(self):
        """The flat Glushkov tables (c, num, isLast, follows, firsts) as Python lists."""
        ns, nf, nfi = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        _lib.check(self._L.fmx_regex_tables(self._h, ctypes.byref(ns), None, None, None, None, ctypes.byref(nf), None,
                                            ctypes.byref(nfi), None))
        c = np.zeros(max(ns.value, 1), dtype=np.uint8)
        num = np.zeros(max(ns.value, 1), dtype=np.int32)
        last = np.zeros(max(ns.value, 1), dtype=np.uint8)
        off = np.zeros(ns.value + 1, dtype=np.int32)
        fol = np.zeros(max(nf.value, 1), dtype=np.int32)
        firsts = np.zeros(max(nfi.value, 1), dtype=np.int32)

        def p(a):
            return a.ctypes.data_as(ctypes.c_void_p)

        _lib.check(self._L.fmx_regex_tables(self._h, None, p(c), p(num), p(last), p(off), None, p(fol), None,
                                            p(firsts)))
        n = ns.value
        return {"c": c[:n].tolist(), "num": num[:n].tolist(), "isLast": [bool(x) for x in last[:n]],
                "follows": [fol[off[k]:off[k + 1]].tolist() for k in range(n)],
                "firsts": firsts[: nfi.value].tolist()}

    last_truncated = False      # True when the last frontier call stopped at max_steps (FMX_TRUNCATED)

    def matchSA(self, sa, debugLevel=0, maxBranching=1024, maxIterations=1000, cap=1 << 20):
        """ReTree.matchSA (retree.scala:570-617) with the reference's own limits, pop order and
        result order: list of SAResult, newest first, exactly `ret` of the first _matchSA pass."""
        return ReTree.matchSA_batch(sa, [self], mode="reference", maxBranching=maxBranching,
                                    maxIterations=maxIterations, cap=cap)[0]

    def matchAll(self, sa, max_steps=0, max_frontier=0, cap=1 << 20):
        """Every match (frontier mode): what matchSA returns when its limits do not bind, sorted by
        (len, sp, ep)."""
        return ReTree.matchSA_batch(sa, [self], max_steps=max_steps, max_frontier=max_frontier, cap=cap)[0]

    @staticmethod
    def compile_batch(res, lineOnly=False):
        """ReTree(REParser.re2post(re, lineOnly)) for every string of `res` in one library call on all host
        cores (fmx_regex_compile_batch) -> CompiledRegexes."""
        return CompiledRegexes(res, lineOnly)

    @staticmethod
    def prepare_batch(sa, trees):
        """Make a batch of compiled regexes resident on sa's device (fmx_regex_batch_create)."""
        return RegexBatch(sa, trees)

    @staticmethod
    def matchSA_batch(sa, trees, max_steps=0, max_frontier=0, cap=1 << 20, mode="frontier", maxBranching=1024,
                      maxIterations=1000):
        """A batch of regexes in one call.  mode="frontier": breadth-first over the whole batch,
        every match; mode="reference": ReTree._matchSA's queue replayed per regex with
        maxBranching / maxIterations."""
        L = _lib.load()
        arr, k, _keep = _handle_array(trees)
        lim = _lib.fmx_limits(int(max_steps), _lib.FMX_MATCH_REFERENCE if mode == "reference" else _lib.FMX_MATCH_FRONTIER,
                              int(max_frontier), int(maxBranching), int(maxIterations))
        out = (_lib.fmx_result * cap)()
        n_out = ctypes.c_size_t()
        per = np.zeros(max(k, 1), dtype=np.uint32)
        rc = _lib.check(L.fmx_regex_match_batch(sa.handle, arr, k, ctypes.byref(lim), out, cap, ctypes.byref(n_out),
                                                per.ctypes.data_as(ctypes.c_void_p)))
        ReTree.last_truncated = rc == _lib.FMX_TRUNCATED
        res = [[] for _ in range(k)]
        for j in range(n_out.value):
            r = out[j]
            res[r.regex].append(SAResult(sa, r.len, r.sp, r.ep))
        return res


RESULT_DTYPE = np.dtype([("regex", np.uint32), ("len", np.uint32), ("sp", np.uint64), ("ep", np.uint64)])


class _TreeView:
    """One regex of a CompiledRegexes set, with ReTree's read-only surface (tables())."""

    def __init__(self, owner, h):
        self._owner, self._h, self._L = owner, h, owner._L      # the owner keeps the handle alive

    tables = ReTree.tables


class CompiledRegexes:
    """fmx_regex_compile_batch: REParser.re2post + ReTree(...) for a list of regex strings in ONE library call,
    compiled on all the host cores the process may use -- no Python object and no ctypes call per regex.
    `status[i]` is 0 or the code ReTree(REParser.re2post(res[i])) would raise with (7 = Re2PostSyntax,
    8 = MatchError); regexes that failed have no handle.  `select(idx)` gives a view of some of them (what a
    workload generator keeps); RegexBatch / RegexBatchMulti / matchSA_batch take either form."""

    def __init__(self, res, lineOnly=False, _view=None):
        self._L = _lib.load()
        if _view is not None:
            self._owner, self.handles, self.status, self.k = _view
            return
        self._owner = None
        self.k = len(res)
        # the k C strings as ONE buffer and an array of pointers into it, made with array operations (a ctypes
        # c_char_p per regex costs more than compiling it)
        if all(isinstance(r, str) for r in res):
            blob = ("\0".join(res) + "\0").encode("latin-1")
            lens = np.fromiter(map(len, res), dtype=np.int64, count=self.k)
        else:
            raw = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in res]
            blob = b"\0".join(raw) + b"\0"
            lens = np.fromiter(map(len, raw), dtype=np.int64, count=self.k)
        if blob.count(b"\0") != self.k + (0 if self.k else 1):
            raise ValueError("a regex holds a NUL character")
        buf = ctypes.create_string_buffer(blob, len(blob))
        ptrs = np.zeros(max(self.k, 1), dtype=np.uint64)
        if self.k:
            ptrs[0] = 0
            np.cumsum(lens[:-1] + 1, out=ptrs[1:self.k].view(np.int64))
            ptrs[: self.k] += np.uint64(ctypes.addressof(buf))
        self.handles = (ctypes.c_void_p * max(self.k, 1))()
        self.status = np.zeros(max(self.k, 1), dtype=np.int32)
        _lib.check(self._L.fmx_regex_compile_batch(ptrs.ctypes.data_as(ctypes.c_void_p), self.k, 1 if lineOnly else 0,
                                                   self.handles, self.status.ctypes.data_as(ctypes.c_void_p)))
        self.status = self.status[: self.k]

    def __len__(self):
        return self.k

    def ok(self):
        """Boolean mask of the regexes that compiled."""
        return self.status == 0

    def select(self, idx):
        """A view holding the handles res[idx] (all must have compiled); the parent owns them."""
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size and (self.status[idx] != 0).any():
            raise ValueError("select: some of the chosen regexes did not compile")
        h = (ctypes.c_void_p * max(idx.size, 1))(*[self.handles[int(i)] for i in idx])
        return CompiledRegexes(None, _view=(self._owner or self, h, np.zeros(idx.size, dtype=np.int32), int(idx.size)))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.select(np.arange(self.k)[i])
        if self.status[i] != 0:
            raise (Re2PostSyntaxOrMatch(int(self.status[i])))
        return _TreeView(self._owner or self, ctypes.c_void_p(self.handles[i]))

    def __del__(self):
        try:
            if self._owner is None and self.k:
                self._L.fmx_regex_free_batch(self.handles, self.k)
                self.k = 0
        except Exception:
            pass


def Re2PostSyntaxOrMatch(code):
    return (_lib.Re2PostSyntax if code == 7 else _lib.MatchError if code == 8 else _lib.FmxError)(code, "regex did not compile")


def _handle_array(trees):
    """(ctypes array of fmx_regex handles, k, object to keep alive) for a list of ReTree / NFA / DFA objects or a
    CompiledRegexes set."""
    if isinstance(trees, CompiledRegexes):
        if (trees.status != 0).any():
            raise ValueError("the set holds regexes that did not compile: select() the good ones")
        return trees.handles, trees.k, trees
    k = len(trees)
    return (ctypes.c_void_p * max(k, 1))(*[t._h for t in trees]), k, list(trees)


class RegexBatch:
    """A regex batch resident on the device: build once, match many times (serving shape of
    ReTree.matchSA over many regexes)."""

    def __init__(self, sa, trees):
        self._L = _lib.load()
        self.sa = sa
        arr, self.k, self._trees = _handle_array(trees)      # keeps the handles alive
        self._h = ctypes.c_void_p()
        _lib.check(self._L.fmx_regex_batch_create(sa.handle, arr, self.k, ctypes.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._L.fmx_regex_batch_free(self._h)
                self._h = None
        except Exception:
            pass

    def info(self):
        """fmx_regex_batch_info: {"regexes", "states", "follows", "firsts"} of the resident batch."""
        v = [ctypes.c_uint64() for _ in range(4)]
        _lib.check(self._L.fmx_regex_batch_info(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("regexes", "states", "follows", "firsts"), (int(x.value) for x in v)))

    def match_raw(self, max_steps=0, max_frontier=0, cap=1 << 22, mode="frontier", maxBranching=1024, maxIterations=1000,
                  copy=True):
        """-> (results as a structured array, per-regex counts).  mode="frontier": every match, sorted by
        (regex, len, sp, ep); mode="reference": ReTree._matchSA's own queue and limits, per regex in the
        reference's list order.  The library writes into page-locked buffers the batch keeps between calls
        (fmx_host_alloc: the device copies straight into them); copy=False returns views of those buffers, valid
        until the next match on this batch -- the serving loop's form: a fresh 1.2 MB array per call costs more in
        page faults than the copy itself."""
        from .searcher import PinnedArray
        lim = _lib.fmx_limits(int(max_steps), _lib.FMX_MATCH_REFERENCE if mode == "reference" else _lib.FMX_MATCH_FRONTIER,
                              int(max_frontier), int(maxBranching), int(maxIterations))
        if getattr(self, "_out", None) is None or self._out.array.size < cap:
            self._out = PinnedArray((cap,), RESULT_DTYPE)
            self._per = PinnedArray((max(self.k, 1),), np.uint32)
            self._out_p = self._out.array.ctypes.data_as(ctypes.c_void_p)
            self._per_p = self._per.array.ctypes.data_as(ctypes.c_void_p)
        n_out = ctypes.c_size_t()
        rc = _lib.check(self._L.fmx_regex_batch_match(self.sa.handle, self._h, ctypes.byref(lim), self._out_p, cap,
                                                      ctypes.byref(n_out), self._per_p))
        self.truncated = rc == _lib.FMX_TRUNCATED
        out, per = self._out.array[: n_out.value], self._per.array[: self.k]
        return (out.copy(), per.copy()) if copy else (out, per)


def _match_dev(self, d_out, cap, d_per=None, max_steps=0, max_frontier=0):
    """fmx_regex_batch_match_dev: results stay in HBM.  d_out = device pointer (int) to room for `cap` 24-byte
    records (RESULT_DTYPE), d_per = device pointer to k uint32 counts or None; returns the number of results."""
    lim = _lib.fmx_limits(int(max_steps), _lib.FMX_MATCH_FRONTIER, int(max_frontier), 1024, 1000)
    n_out = ctypes.c_size_t()
    rc = _lib.check(self._L.fmx_regex_batch_match_dev(self.sa.handle, self._h, ctypes.byref(lim), ctypes.c_void_p(int(d_out)),
                                                      int(cap), ctypes.byref(n_out), ctypes.c_void_p(int(d_per)) if d_per else None))
    self.truncated = rc == _lib.FMX_TRUNCATED
    return int(n_out.value)


RegexBatch.match_dev = _match_dev


class RegexBatchMulti:
    """fmx_regex_batch_create_multi / _match_multi: one process, one replica handle per GPU, the regex batch cut by
    estimated frontier work; same results as RegexBatch on one handle."""

    def __init__(self, searchers, trees):
        self._L = _lib.load()
        self.searchers = list(searchers)
        arr, self.k, self._trees = _handle_array(trees)
        idxs = (ctypes.c_void_p * len(self.searchers))(*[s.handle for s in self.searchers])
        self._h = ctypes.c_void_p()
        _lib.check(self._L.fmx_regex_batch_create_multi(idxs, len(self.searchers), arr, self.k, ctypes.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._L.fmx_regex_batch_free_multi(self._h)
                self._h = None
        except Exception:
            pass

    def match_raw(self, max_steps=0, max_frontier=0, cap=1 << 22, copy=True):
        """Same contract as RegexBatch.match_raw (copy=False: views of buffers the object keeps between calls)."""
        lim = _lib.fmx_limits(int(max_steps), _lib.FMX_MATCH_FRONTIER, int(max_frontier), 1024, 1000)
        from .searcher import PinnedArray
        if getattr(self, "_out", None) is None or self._out.array.size < cap:
            self._out = PinnedArray((cap,), RESULT_DTYPE)      # page-locked and kept between calls, as in RegexBatch
            self._per = PinnedArray((max(self.k, 1),), np.uint32)
        n_out = ctypes.c_size_t()
        rc = _lib.check(self._L.fmx_regex_batch_match_multi(self._h, ctypes.byref(lim), self._out.array.ctypes.data_as(ctypes.c_void_p),
                                                            cap, ctypes.byref(n_out), self._per.array.ctypes.data_as(ctypes.c_void_p)))
        self.truncated = rc == _lib.FMX_TRUNCATED
        out, per = self._out.array[: n_out.value], self._per.array[: self.k]
        return (out.copy(), per.copy()) if copy else (out, per)
