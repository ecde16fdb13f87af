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


def search_batch_sharded(searcher, pat, off, group=None):
    """SuffixAlgo.search over a pattern batch sharded across the ranks of `group`.
    Every rank passes the same (pat, off); every rank gets the full (sp, ep) back."""
    rank, world = _group_info(group)
    pat = np.ascontiguousarray(pat, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    cuts = shard_bounds(off, world)
    a, b = cuts[rank], cuts[rank + 1]
    sp, ep = searcher.search_batch(pat, off[a:b + 1])
    if world == 1:
        return sp, ep
    both = torch.from_numpy(np.concatenate([sp, ep]).astype(np.int64))      # same bits as uint64
    dev = None
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        both = both.to(dev)
    parts = all_gather_varlen(both, group)
    sps, eps = [], []
    for p in parts:
        h = p.numel() // 2
        q = p.cpu().numpy().astype(np.uint64)
        sps.append(q[:h])
        eps.append(q[h:])
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
    """Device form used by bench.py: equal-sized int64 device tensors sp, ep (k each) ->
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
    all-gather of batch i's intervals runs on the collective's own stream while batch i+1 is being searched.

    `depth` slots, each a (2, k) int64 tensor the search writes its sp / ep rows into (no staging copy) and a
    (world, 2, k) tensor the gather fills.  Per batch: `sp, ep = slot(i)` (waits, on the current stream, for
    the collective that last used the slot), search into them, `launch(i)`; `finish()` waits for everything
    outstanding.  With the nccl backend (RCCL over xGMI) a batch then costs max(search, gather) instead of
    their sum: 16 MB per rank and million patterns is ~0.4 ms on 8 GPUs against ~0.7 ms of search."""

    def __init__(self, k, device, group=None, depth=2):
        self.rank, self.world = _group_info(group)
        self.group = group
        self.live = dist is not None and dist.is_initialized()
        self.mine = [torch.empty((2, k), dtype=torch.int64, device=device) for _ in range(depth)]
        self.out = [torch.empty((self.world, 2, k), dtype=torch.int64, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.depth = depth

    def slot(self, i):
        j = i % self.depth
        if self.work[j] is not None:
            self.work[j].wait()
            self.work[j] = None
        return self.mine[j][0], self.mine[j][1]

    def launch(self, i):
        j = i % self.depth
        if self.live:
            self.work[j] = dist.all_gather_into_tensor(self.out[j].view(-1), self.mine[j].view(-1), group=self.group,
                                                       async_op=True)
        else:
            self.out[j][0].copy_(self.mine[j])
        return self.out[j]

    def finish(self):
        for j in range(self.depth):
            if self.work[j] is not None:
                self.work[j].wait()
                self.work[j] = None
