"""Multi-GPU form of the search path: one process per GPU, every rank holds a full replica of the
index (each LF step jumps to an arbitrary row, so the BWT itself cannot be split), the pattern /
regex batch is cut into contiguous shards, each rank searches its shard with no data-path
collective, and ONE gather of the result intervals (16 bytes per pattern) closes the call --
`torch.distributed` all_gather: RCCL over xGMI with the "nccl" backend, gloo on CPU.

The reference has no distributed layer at all; this module is the host logic the
BASELINE config C5 asks for.  It is backend-agnostic: `searcher` is anything with the
HipFMSearcher batch methods, so the CPU tests drive it with gloo.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover - torch is plumbing; importing this module needs it
    torch = None
    dist = None


def shard_bounds(off, world):
    """Cut k patterns (offsets `off`, k+1 entries) into `world` contiguous slices balanced by
    total pattern bytes.  Returns world+1 pattern indices; slice r is [b[r], b[r+1])."""
    off = np.asarray(off, dtype=np.uint64)
    k = off.size - 1
    if k <= 0:
        return [0] * (world + 1)
    total = int(off[-1] - off[0])
    if total == 0:                       # all patterns empty: balance by count
        return [(k * r) // world for r in range(world + 1)]
    rel = (off - off[0]).astype(np.float64)
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(rel, total * r / world, side="left")))
    cuts.append(k)
    for r in range(1, world + 1):        # keep monotone
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts


def _group_info(group):
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def all_gather_varlen(t, group=None):
    """all_gather of 1-D tensors whose lengths differ per rank: sizes first, then the payload
    padded to the longest shard (one collective each).  Returns the list of per-rank tensors."""
    rank, world = _group_info(group)
    if world == 1:
        return [t]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes + [1])
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:s] for o, s in zip(out, sizes)]


PACK_WIDE = 0xFFFFFF      # width field of a packed interval that stands for "look in the escape list" (include/fmx.h)


def packed_words(k, escape_cap):
    return int(k) + 1 + 2 * int(escape_cap)


def pack_intervals_np(sp, ep, escape_cap):
    """The 8-byte form of a batch's intervals (include/fmx.h, fmx_pack_intervals_dev) on host arrays: word q = sp |
    min(ep - sp, 0xFFFFFF) << 40; intervals of 0xFFFFFF rows or more leave their ep in the escape list behind the k words
    (word k = how many there are, then (q, ep) pairs, the first escape_cap of them).  What the gloo ranks of the CPU
    tests exchange; the device kernel's output differs at most in the ORDER of the escape entries."""
    sp = np.ascontiguousarray(sp, dtype=np.uint64)
    ep = np.ascontiguousarray(ep, dtype=np.uint64)
    k = sp.size
    w = ep - sp
    out = np.zeros(packed_words(k, escape_cap), dtype=np.uint64)
    out[:k] = sp | (np.minimum(w, np.uint64(PACK_WIDE)) << np.uint64(40))
    wide = np.nonzero(w >= np.uint64(PACK_WIDE))[0]
    out[k] = wide.size
    keep = wide[: int(escape_cap)]
    out[k + 1: k + 1 + 2 * keep.size: 2] = keep.astype(np.uint64)
    out[k + 2: k + 2 + 2 * keep.size: 2] = ep[keep]
    return out


def unpack_intervals_np(packed, k, escape_cap):
    """Inverse of pack_intervals_np; raises OverflowError when more intervals were wide than the escape list holds
    (the caller then exchanges the 16-byte form)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    k = int(k)
    sp = packed[:k] & np.uint64((1 << 40) - 1)
    ep = sp + (packed[:k] >> np.uint64(40))
    cnt = int(packed[k])
    if cnt > int(escape_cap):
        raise OverflowError("%d intervals of 2^24 - 1 rows or more, the escape list holds %d" % (cnt, escape_cap))
    q = packed[k + 1: k + 1 + 2 * cnt: 2].astype(np.int64)
    ep = ep.copy()
    ep[q] = packed[k + 2: k + 2 + 2 * cnt: 2]
    return sp.copy(), ep


def gather_varlen_to_root(t, root=0, group=None):
    """Root-only delivery of 1-D tensors whose lengths differ per rank: the sizes are all-gathered (8 bytes per rank),
    the payload padded to the longest goes to `root` alone (dist.gather: point-to-point sends under RCCL).  Returns the
    list of per-rank tensors on the root, None elsewhere."""
    rank, world = _group_info(group)
    if world == 1:
        return [t]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes + [1])
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    out = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
    dist.gather(pad, out, dst=dist.get_global_rank(group, root) if group is not None else root, group=group)
    return [o[:s] for o, s in zip(out, sizes)] if rank == root else None


def search_batch_sharded(searcher, pat, off, group=None, form="packed", delivery="all", root=0):
    """SuffixAlgo.search over a pattern batch sharded across the ranks of `group`.
    Every rank passes the same (pat, off).  form = "packed" (default): the exchange carries the 8-byte form of each
    interval (sp in 40 bits + the width in 24, wider intervals in an escape list: 8 bytes per pattern + 16 per escape);
    "pairs": (sp, ep) as 16 bytes per pattern.  delivery = "all": every rank gets the full (sp, ep) back (all-gather);
    "root": only rank `root` does (returns None elsewhere) -- the "final gather of hit intervals" when one rank
    consumes the answer."""
    if form not in ("packed", "pairs") or delivery not in ("all", "root"):
        raise ValueError("form is packed or pairs, delivery is all or root")
    rank, world = _group_info(group)
    pat = np.ascontiguousarray(pat, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    cuts = shard_bounds(off, world)
    a, b = cuts[rank], cuts[rank + 1]
    sp, ep = searcher.search_batch(pat, off[a:b + 1])
    if world == 1:
        return sp, ep
    if form == "packed":
        n_wide = int(((ep - sp) >= np.uint64(PACK_WIDE)).sum())
        mine = pack_intervals_np(sp, ep, n_wide)              # exactly as long as it has to be: the lengths travel anyway
    else:
        mine = np.concatenate([sp, ep])
    t = torch.from_numpy(mine.astype(np.int64))               # same bits as uint64
    if dist.get_backend(group) == "nccl":
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    parts = all_gather_varlen(t, group) if delivery == "all" else gather_varlen_to_root(t, root, group)
    if parts is None:
        return None
    sps, eps = [], []
    for r, p in enumerate(parts):
        q = p.cpu().numpy().astype(np.uint64)
        kr = cuts[r + 1] - cuts[r]
        if form == "packed":
            s_, e_ = unpack_intervals_np(q, kr, (q.size - kr - 1) // 2)
        else:
            s_, e_ = q[:kr], q[kr:]
        sps.append(s_)
        eps.append(e_)
    return np.concatenate(sps), np.concatenate(eps)


RESULT_DTYPE = np.dtype([("regex", np.uint32), ("len", np.uint32), ("sp", np.uint64), ("ep", np.uint64)])


def work_bounds(weights, world):
    """Cut k regexes into `world` contiguous slices of about equal total weight (an estimate of each regex's
    frontier work: SURVEY.md 8e).  Returns world+1 indices."""
    w = np.asarray(weights, dtype=np.float64)
    k = w.size
    if k == 0:
        return [0] * (world + 1)
    cum = np.cumsum(np.maximum(w, 1e-9))
    # slice r ends with the first regex that brings the running weight to r/world of the total
    cuts = [0] + [min(k, int(np.searchsorted(cum, cum[-1] * r / world, side="left")) + 1) for r in range(1, world)] + [k]
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts


def gather_results(res, device=None, group=None):
    """The regex path's exchange: every rank contributes its result list (structured array, 24 bytes per
    result) and receives all of them, concatenated in rank order.  Two collectives: the sizes, then the payload
    padded to the longest list (all_gather_varlen)."""
    rank, world = _group_info(group)
    res = np.ascontiguousarray(res, dtype=RESULT_DTYPE)
    if world == 1:
        return res
    t = torch.from_numpy(res.view(np.int64).reshape(-1).copy())      # 3 words per result, same bits
    if dist.get_backend(group) == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    parts = [p.cpu().numpy().view(RESULT_DTYPE) for p in all_gather_varlen(t, group)]
    return np.concatenate(parts)


def group_results(res, k):
    """Structured result array -> k lists of (len, sp, ep) tuples, by regex id (array operations only)."""
    res = res[np.argsort(res["regex"], kind="stable")]
    bounds = np.searchsorted(res["regex"], np.arange(k + 1))
    cols = np.stack([res["len"].astype(np.uint64), res["sp"], res["ep"]], axis=1)
    return [list(map(tuple, cols[bounds[r]:bounds[r + 1]].tolist())) for r in range(k)]


MATCH_KW = ("max_steps", "max_frontier", "cap")      # what the sharded regex match passes on to a rank's search


def exchange_result_words(t, first_regex, group=None):
    """The regex path's exchange on tensors, device or host alike (what the RCCL branch of match_batch_sharded runs
    on HBM-resident results): `t` = this rank's result list as int64 words, three per result (word 0 = regex |
    len << 32, then sp, ep -- RESULT_DTYPE's bits), regex ids local to the rank's slice.  Makes the ids global
    (+ first_regex), all-gathers the lists (sizes, then the payload padded to the longest: all_gather_varlen) and
    returns the concatenation in rank order as a structured host array."""
    if t.numel() % 3:
        raise ValueError("result words come three per result")
    t = t.clone()
    t[0::3] += int(first_regex)
    parts = all_gather_varlen(t, group)
    if not parts:
        return np.zeros(0, dtype=RESULT_DTYPE)
    return np.concatenate([np.ascontiguousarray(p.cpu().numpy()).view(RESULT_DTYPE) for p in parts])


def match_batch_sharded(sa, trees, group=None, match_fn=None, weights=None, **kw):
    """ReTree.matchSA over a regex batch sharded across the ranks: contiguous slices balanced by `weights`
    (estimated frontier work per regex; by count when None), each rank matches its slice, one gather of the
    result lists.  Returns the structured array of all results, regex ids global, on every rank.
    `match_fn(sa, trees_slice)` defaults to the GPU frontier search on a resident batch; it returns a structured
    array with slice-local regex ids (or, per regex, a list of (len, sp, ep) tuples / SAResult objects).
    Keyword arguments for the search: max_steps, max_frontier, cap (every match, frontier mode) -- the same on every
    backend; anything else is refused."""
    unknown = sorted(set(kw) - set(MATCH_KW))
    if unknown:
        raise TypeError("match_batch_sharded: unsupported keyword(s) %s (the sharded match runs the frontier mode: %s)"
                        % (", ".join(unknown), ", ".join(MATCH_KW)))
    rank, world = _group_info(group)
    k = len(trees)
    cuts = work_bounds(weights if weights is not None else np.ones(k), world)
    if match_fn is None and dist is not None and dist.is_initialized() and dist.get_backend(group) == "nccl":
        # RCCL: the results never leave HBM before the exchange (fmx_regex_batch_match_dev): the regex ids are
        # made global on the device, the lists are all-gathered, and one copy brings the whole answer to the host
        from .regex import ReTree
        kw = dict(kw)
        cap = int(kw.pop("cap", 1 << 22))
        dev = torch.device("cuda", torch.cuda.current_device())
        d_out = torch.empty(3 * cap, dtype=torch.int64, device=dev)
        n_res = ReTree.prepare_batch(sa, trees[cuts[rank]:cuts[rank + 1]]).match_dev(d_out.data_ptr(), cap, None, **kw)
        return exchange_result_words(d_out[: 3 * n_res], cuts[rank], group)
    if match_fn is None:
        from .regex import ReTree

        def match_fn(sa_, ts):
            return ReTree.prepare_batch(sa_, ts).match_raw(**kw)[0]
    mine = match_fn(sa, trees[cuts[rank]:cuts[rank + 1]])
    if not isinstance(mine, np.ndarray):      # per-regex lists (CPU stand-ins in the tests)
        rows = [(j, r.len, r.sp, r.ep) if hasattr(r, "len") else (j,) + tuple(r) for j, res in enumerate(mine) for r in res]
        mine = np.array(rows, dtype=RESULT_DTYPE) if rows else np.zeros(0, dtype=RESULT_DTYPE)
    mine = np.ascontiguousarray(mine, dtype=RESULT_DTYPE)
    words = torch.from_numpy(mine.view(np.int64).reshape(-1).copy())
    return exchange_result_words(words, cuts[rank], group)


def gather_intervals_dev(sp, ep, group=None):
    """The 16-byte form in one call: equal-sized int64 device tensors sp, ep (k each) ->
    one all_gather_into_tensor of 16 B per pattern; returns a (world, 2, k) tensor."""
    rank, world = _group_info(group)
    mine = torch.stack([sp, ep])
    if dist is None or not dist.is_initialized():
        return mine.unsqueeze(0)
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=group)
    return out


class IntervalGather:
    """Pipelined form of the path's one exchange, for callers that search batch after batch (bench.py): the
    gather of batch i's intervals runs on the collective's own stream while batch i+1 is being searched.

    `depth` slots, each a (2, k) int64 tensor the search writes its sp / ep rows into (no staging copy), a send buffer
    and a receive buffer.  Per batch: `sp, ep = slot(i)` (waits, on the current stream, for the collective that last
    used the slot), search into them, `launch(i)`; `finish()` waits for everything outstanding.

    form = "packed" (default): the send buffer is the 8-byte form of the intervals (include/fmx.h: k + 1 + 2 * escape_cap
    words, packed by fmx_pack_intervals_dev on the current stream -- `searcher` must then be the rank's HipFMSearcher;
    host tensors are packed with numpy), "pairs": the (2, k) tensor itself, 16 bytes per pattern.
    delivery = "root" (default): every rank sends to rank `root`, which alone receives (world, words) -- one slice per
    xGMI link into the root, nothing into the others; "all": all-gather, every rank receives everything.
    With the nccl backend (RCCL over xGMI) a batch costs max(search, gather), not their sum.  What that is on 8 GPUs
    for a million 32-character patterns per rank (C3): the search 0.2 ms; the payload 8 MB per rank packed (16 MB as
    pairs), i.e. 8 MB over each of the root's seven links or, for the all-gather, 56 MB into every rank -- at ~50 GB/s per
    link direction ~0.16 ms packed / ~0.33 ms as pairs before RCCL's own latencies.  bench.py measures it
    (`exchange.gather_ms`)."""

    def __init__(self, k, device, group=None, depth=2, form="packed", delivery="root", root=0, escape_cap=None, searcher=None):
        if form not in ("packed", "pairs") or delivery not in ("all", "root"):
            raise ValueError("form is packed or pairs, delivery is all or root")
        self.rank, self.world = _group_info(group)
        self.group, self.k, self.form, self.delivery, self.root = group, int(k), form, delivery, int(root)
        self.live = dist is not None and dist.is_initialized()
        self.cap = max(16, self.k // 256) if escape_cap is None else int(escape_cap)
        self.searcher = searcher
        self.words = packed_words(self.k, self.cap) if form == "packed" else 2 * self.k
        self.mine = [torch.empty((2, self.k), dtype=torch.int64, device=device) for _ in range(depth)]
        self.send = [torch.empty(self.words, dtype=torch.int64, device=device) if form == "packed" else self.mine[j].view(-1)
                     for j in range(depth)]
        recv_here = delivery == "all" or self.rank == self.root or not self.live
        self.out = [torch.empty((self.world, self.words), dtype=torch.int64, device=device) if recv_here else None
                    for _ in range(depth)]
        self.work = [None] * depth
        self.depth = depth
        self.payload_bytes = 8 * self.words

    def slot(self, i):
        j = i % self.depth
        if self.work[j] is not None:
            self.work[j].wait()
            self.work[j] = None
        return self.mine[j][0], self.mine[j][1]

    def search_into(self, i, searcher, d_pat, d_off, stream=0, fixed_len=0):
        """Slot i's search, written where the exchange sends from: in the packed form the search kernels emit the 8-byte
        words themselves (fmx_search_batch_ex_dev, packed) -- no (sp, ep) arrays, no packing pass behind the search; call
        launch(i, stream, packed_already=True) behind it.  Waits for the slot like slot(i)."""
        sp, ep = self.slot(i)
        j = i % self.depth
        if self.form == "packed":
            searcher.search_batch_ex_dev(d_pat, d_off, self.send[j].data_ptr(), ep.data_ptr(), self.k, stream, fixed_len=fixed_len,
                                         packed=True, escape_cap=self.cap)
        else:
            searcher.search_batch_ex_dev(d_pat, d_off, sp.data_ptr(), ep.data_ptr(), self.k, stream, fixed_len=fixed_len)

    def pack(self, i, stream=0):
        """The send buffer of slot i from its (sp, ep) rows (a no-op for form "pairs")."""
        j = i % self.depth
        if self.form != "packed":
            return
        sp, ep = self.mine[j][0], self.mine[j][1]
        if sp.device.type == "cpu":
            w = pack_intervals_np(sp.numpy().view(np.uint64), ep.numpy().view(np.uint64), self.cap)
            self.send[j].copy_(torch.from_numpy(w.view(np.int64)))
        else:
            if self.searcher is None:
                raise ValueError("IntervalGather(form='packed') on device tensors needs searcher=")
            self.searcher.pack_intervals_dev(sp.data_ptr(), ep.data_ptr(), self.k, self.send[j].data_ptr(), escape_cap=self.cap,
                                             stream=stream)

    def launch(self, i, stream=0, packed_already=False):
        j = i % self.depth
        if not packed_already:
            self.pack(i, stream)
        if not self.live:
            self.out[j][0].copy_(self.send[j])
        elif self.delivery == "all":
            self.work[j] = dist.all_gather_into_tensor(self.out[j].view(-1), self.send[j], group=self.group, async_op=True)
        else:
            dst = dist.get_global_rank(self.group, self.root) if self.group is not None else self.root
            outs = list(self.out[j].unbind(0)) if self.rank == self.root else None
            self.work[j] = dist.gather(self.send[j], outs, dst=dst, group=self.group, async_op=True)
        return self.out[j]

    def finish(self):
        for j in range(self.depth):
            if self.work[j] is not None:
                self.work[j].wait()
                self.work[j] = None

    def overflow(self, i):
        """A collective every rank calls for slot i (after its launch): True ON EVERY RANK when any rank's batch held more
        wide intervals (2^24 - 1 rows or more: patterns of a character or two) than the escape list of the packed form has
        room for -- word k of a rank's send buffer counts them all, the list keeps the first escape_cap.  The intervals
        beyond it are NOT in the payload: every rank then repeats the slot in the 16-byte form (IntervalGather(form="pairs")).
        Until round 4 only the root found out, by an OverflowError while decoding, after the exchange.
        One all-reduce of one word; synchronises this rank's slot.  Always False for form "pairs"."""
        j = i % self.depth
        if self.form != "packed":
            return False
        if self.work[j] is not None:
            self.work[j].wait()
        over = (self.send[j][self.k: self.k + 1] > self.cap).to(torch.int64)
        if self.live and self.world > 1:
            dist.all_reduce(over, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(over.item()))

    def intervals(self, out, r):
        """(sp, ep) of rank r's batch as uint64 host arrays from a received buffer (None where nothing was delivered)."""
        if out is None:
            return None
        row = out[r].cpu().numpy().view(np.uint64)
        if self.form == "packed":
            return unpack_intervals_np(row, self.k, self.cap)
        return row[: self.k].copy(), row[self.k:].copy()
