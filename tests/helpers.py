"""Shared test helpers (CPU only, numpy)."""
import numpy as np


def bwt_of_text(text: bytes):
    """BWT of `text` + EOF in the layout findex's merger writes to .bwt/.aux
    (reference: bwtmerger.scala:782-810 sa2BWT, :841-856 aux): row i holds the
    byte before suffix SA[i]; the row whose suffix is the whole text is the EOF
    slot (filled with a neighbour's byte); counts exclude the EOF symbol.
    The text must not contain byte 0 (the readers escape it,
    bwtreader.scala:136-155).  Naive O(n^2 log n) sort: small inputs only.
    Returns (bwt uint8[n+1], eof, counts int64[256])."""
    assert 0 not in text
    s = bytes(text) + b"\0"
    n = len(s)
    sa = sorted(range(n), key=lambda i: s[i:])
    bwt = np.zeros(n, dtype=np.uint8)
    eof = -1
    for i, p in enumerate(sa):
        if p == 0:
            eof = i
        else:
            bwt[i] = s[p - 1]
    if eof > 0:
        bwt[eof] = bwt[eof - 1]
    elif n != 1:
        bwt[eof] = bwt[eof + 1]
    counts = np.bincount(np.frombuffer(bytes(text), dtype=np.uint8), minlength=256).astype(np.int64)
    return bwt, eof, counts


def synth_bwt(n, sigma_lo, sigma_hi, seed, eof=None):
    """i.i.d. uniform symbols in [sigma_lo, sigma_hi]; any byte string is a valid
    BWT for rank / backward-search purposes.  Returns (bwt, eof, counts)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bwt = rng.integers(sigma_lo, sigma_hi + 1, size=n, dtype=np.uint8 if sigma_hi < 256 else np.int64).astype(np.uint8)
    if eof is None:
        eof = n // 3
    counts = np.bincount(bwt, minlength=256).astype(np.int64)
    counts[bwt[eof]] -= 1
    return bwt, eof, counts


def lf_walk_patterns(sa, rng, k, m, mutate_frac=0.1, alphabet=None):
    """Hit patterns by LF walk: from a random row take c=L[r],
    r=LF(r) m times and emit the bytes reversed, so every backward step has a
    non-empty interval; `mutate_frac` of them get one random byte replaced."""
    pats = []
    for _ in range(k):
        r = int(rng.integers(0, sa.n))
        cs = []
        for _ in range(m):
            cs.append(sa.bwt_read(r))
            r = sa.getPrevI(r)
        p = bytearray(reversed(cs))
        if rng.random() < mutate_frac and m > 0:
            j = int(rng.integers(0, m))
            p[j] = int(rng.choice(alphabet)) if alphabet is not None else int(rng.integers(1, 256))
        pats.append(bytes(p))
    return pats


def pack_patterns(pats):
    off = np.zeros(len(pats) + 1, dtype=np.uint64)
    for i, p in enumerate(pats):
        off[i + 1] = off[i] + len(p)
    buf = np.frombuffer(b"".join(pats), dtype=np.uint8).copy() if pats and int(off[-1]) else np.zeros(0, dtype=np.uint8)
    return buf, off
