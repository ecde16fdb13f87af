#!/usr/bin/env python3
"""A REAL-TEXT index at benchmark scale, for bench.py's `c3text` / `c4text` workloads.

SURVEY 8(d) prescribes i.i.d. byte strings as the C2-C5 "BWTs" -- any byte string is a valid LF permutation -- and
the headline numbers keep that input.  But an i.i.d. string is the best case for the derived tables of rounds 2-3 (the
interval of a hit is one row after log_sigma n characters) and it has LF cycles no text has (a starred class never runs
dry).  This module makes the other kind of input: a text with the repeats of natural language -- the words of the
reference's own fixture `words.txt`, sampled with replacement, separated by spaces and newlines (the C4 alphabet) -- and
its true BWT, by a prefix-doubling suffix sort written with torch tensor operations (a bench-input generator that runs
on the GPU box; not product code).  The index is over the REVERSED text, as findex builds it
(bwtmerger.scala:1106-1108 copyReverse), so patterns and regexes read forward in the text.

    python tools/text_bwt.py [log2n]        # self-check against a naive sort on a small text
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORDS = os.path.join(ROOT, "tests", "golden", "testdata", "words.txt")


def make_text(torch, n_bytes, seed, device):
    """n_bytes of text: words of words.txt (lowercase entries only) drawn with replacement, each followed by a space --
    by a newline every ~12th word.  Returns a uint8 tensor (no byte 0)."""
    words = [w for w in open(WORDS, "rb").read().split() if w.isalpha() and w.islower()]
    lens = torch.tensor([len(w) for w in words], dtype=torch.int64, device=device)
    flat = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).to(device)
    starts = torch.cumsum(lens, 0) - lens
    g = torch.Generator(device=device)
    g.manual_seed(0x7E870000 + seed)
    mean = float(lens.double().mean().item()) + 1.0
    count = int(n_bytes / mean * 1.02) + 1024
    pick = torch.randint(0, len(words), (count,), generator=g, device=device)
    wl = lens[pick] + 1                                     # the word and its separator
    ends = torch.cumsum(wl, 0)
    assert int(ends[-1].item()) >= n_bytes
    pos = torch.arange(n_bytes, dtype=torch.int64, device=device)
    wi = torch.searchsorted(ends, pos, right=True)          # the word position p falls into
    inword = pos - (ends[wi] - wl[wi])
    is_sep = inword == lens[pick[wi]]
    src = starts[pick[wi]] + torch.where(is_sep, torch.zeros_like(inword), inword)
    text = flat[src]
    nl = torch.rand(count, generator=g, device=device) < (1.0 / 12.0)
    sep = torch.where(nl[wi], torch.full_like(text, 10), torch.full_like(text, 32))
    return torch.where(is_sep, sep, text)


def suffix_sort(torch, s, log=None):
    """Suffix array of the byte tensor s, whose LAST byte is a unique smallest sentinel (0): prefix doubling -- ranks by
    the first h characters, sort the pairs (rank[i], rank[i + h]), h doubles until every rank is distinct.  O(n log n)
    per round in torch.sort, ~log2(longest repeat) rounds."""
    n = s.numel()
    dev = s.device
    rank = s.to(torch.int64)
    maxr = 255
    h = 1
    rounds = 0
    while True:
        key = rank * (maxr + 2)
        key[: n - h] += rank[h:] + 1                        # beyond the end: 0, smaller than any rank + 1
        key, sa = torch.sort(key)
        flag = torch.ones(n, dtype=torch.int64, device=dev)
        flag[1:] = (key[1:] != key[:-1]).to(torch.int64)
        del key
        nr = torch.cumsum(flag, 0) - 1
        del flag
        rank = torch.empty(n, dtype=torch.int64, device=dev)
        rank[sa] = nr
        maxr = int(nr[-1].item())
        del nr
        rounds += 1
        if log:
            log("suffix sort: round %d (h = %d): %d of %d ranks distinct" % (rounds, h, maxr + 1, n))
        if maxr == n - 1:
            return sa
        h *= 2


def bwt_of_reversed_text(torch, text, log=None):
    """(bwt uint8[n + 1], eof) in the layout findex's merger writes (helpers.bwt_of_text): the index over reverse(text) +
    EOF; row i holds the byte before suffix SA[i], the row of the whole string is the EOF slot (filled with a neighbour's
    byte)."""
    s = torch.cat([torch.flip(text, dims=[0]), torch.zeros(1, dtype=torch.uint8, device=text.device)])
    sa = suffix_sort(torch, s, log)
    n = s.numel()
    prev = sa - 1
    eof = int(torch.nonzero(sa == 0)[0].item())
    prev[eof] = sa[eof - 1] - 1 if eof > 0 else sa[eof + 1] - 1      # a neighbour's byte as filler (bwtmerger.scala:799-806)
    del sa
    bwt = s[prev]
    return bwt, eof


if __name__ == "__main__":
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import bwt_of_text
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    t = make_text(torch, (1 << log2n) - 1, 1, dev)
    bwt, eof = bwt_of_reversed_text(torch, t, print)
    if log2n <= 14:
        want, weof, _ = bwt_of_text(bytes(t.cpu().numpy()[::-1].tobytes()))
        assert weof == eof and np.array_equal(want, bwt.cpu().numpy()), "differs from the naive sort"
        print("equal to the naive sort: n = %d, eof = %d" % (bwt.numel(), eof))
    print(bytes(t[:200].cpu().numpy().tobytes()))
