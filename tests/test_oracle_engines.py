"""Pins oracle/engines.py (Thompson-NFA and DFA SA-interval engines) to the reference's vectors."""
import os

import oracle
from oracle import engines as E
from oracle import retree as R
from helpers import bwt_of_text


class _Idx:
    def __init__(self, sa):
        self.sa, self.n = sa, sa.n

    def getPrevRange(self, sp, ep, c):
        return self.sa.getPrevRange(sp, ep, c)


def render(sa, res):
    """SAResult.toString, re2.scala:9-19 (SAISBuilder.nextSubstr for the in-memory searcher)."""
    out = []
    for ln, sp, ep in res:
        s = sa.nextSubstr(sp, ln).decode("latin-1")
        out.append(s if ep - sp == 1 else "[%d Results] %s" % (ep - sp, s))
    return out


def test_match_sa_basics():
    """T/REParser.scala:219-234"""
    sa = oracle.SAISNaiveSearcher.from_mem(*bwt_of_text(b"mmabcacamabbbca"[::-1]))
    r = E.nfa_matchSA(E.createNFA(R.post2re("ma.b.")), _Idx(sa))
    assert render(sa, r) == ["[2 Results] bam"]
    r = E.nfa_matchSA(E.createNFA(R.post2re("ba|c.")), _Idx(sa))
    assert sorted(render(sa, r)) == sorted(["ca", "[2 Results] cb"])


def test_match_sa_fmindex(testdata):
    """T/REParser.scala:292-307: (b|a|d|e)c over the file-backed index of test1024.txt"""
    sa = oracle.NaiveFMSearcher(os.path.join(testdata, "test1024.cmp.bwt"), bigEndian=False)
    r = E.nfa_matchSA(E.createNFA(R.post2re("ba|d|e|c.")), _Idx(sa))
    got = set()
    for ln, sp, ep in r:
        s = sa.nextSubstr(sp, ln).decode()
        got.add(s if ep - sp == 1 else "[%d Results] %s" % (ep - sp, s))
    assert got == {"ec", "dc", "[2 Results] ac", "bc"}


def test_dfa_match_sa_basics():
    """T/dfa.scala:110-122: the hand-built automaton s -a-> a -b-> b -b-> b -c-> f over reversed
    'mmabcacadabbbca' gives 'cbbba' and 'cba'."""
    sa = oracle.SAISNaiveSearcher.from_mem(*bwt_of_text(b"mmabcacadabbbca"[::-1]))
    d = E.DFA(4)
    d.addLink(0, 1, ord("a"))
    d.addLink(1, 2, ord("b"))
    d.addLink(2, 2, ord("b"))
    d.addLink(2, 3, ord("c"))
    d.finishStates = {3}
    d.compileBuckets()
    r = d.matchSA(_Idx(sa))
    assert len(r) == 2
    assert sorted(render(sa, r)) == ["cba", "cbbba"]


def test_dfa_buckets():
    """T/dfa.scala:96-108: runs of equal targets become buckets, single characters DFAChar."""
    d = E.DFA(4)
    for ch in "cdfmkl":
        d.addLink(0, 1, ord(ch))
    d.addLink(1, 2, ord("b"))
    d.addLink(2, 2, ord("b"))
    d.addLink(2, 3, ord("c"))
    d.compileBuckets()
    assert d.buckets[0] == [("bucket", 1, ord("c"), ord("d")), ("char", 1, ord("f")), ("bucket", 1, ord("k"), ord("m"))]
    assert d.buckets[1] == [("char", 2, ord("b"))]
    assert d.buckets[2] == [("char", 2, ord("b")), ("char", 3, ord("c"))]
    assert d.buckets[3] == []


def test_order_dependent_cuts_are_not_reproducible_without_the_jvm():
    """VERDICT r1 item 10.  REParser.matchSA's maxIterations (re2.scala:612) and DFA.matchSA's 500-iteration cap
    (dfa.scala:268) cut the search after a number of pops, so what they return depends on the pop order -- and that
    order is not in the reference's source: the NFA engine seeds its queue from an immutable Set of state objects
    hashed by identity and pops equal-length elements in heap-layout order; the DFA engine takes `head` of an
    immutable HashSet.  Two orders the source allows give different results as soon as the cut binds, and the same
    multiset when it does not: that is why the product serves these two engines through the frontier mode only
    (every match; include/fmx.h) and reproduces limits exactly only for ReTree, whose queue order IS defined by
    its source (state.num + Scala's binary heap)."""
    from collections import Counter
    from helpers import bwt_of_text
    from oracle import engines as E
    import oracle
    text = (b"abcabdabeabcabcabd" * 6)[::-1]
    sa = oracle.NaiveFMSearcher.from_mem(*bwt_of_text(text))
    nfa = E.createNFA(E.R.re2post("ab(c|d|e)a"))
    free_a = Counter(E.nfa_matchSA(nfa, sa, tie="first"))
    free_b = Counter(E.nfa_matchSA(nfa, sa, tie="last"))
    assert free_a == free_b and sum(free_a.values()) >= 3            # not binding: one multiset
    cut_a = Counter(E.nfa_matchSA(nfa, sa, maxIterations=5, tie="first"))
    cut_b = Counter(E.nfa_matchSA(nfa, sa, maxIterations=5, tie="last"))
    assert cut_a != cut_b                                              # binding: the order decides
    assert not (cut_a - free_a) and not (cut_b - free_a)               # both are sub-multisets of the full answer
    # the DFA engine: 0 -a-> 0, 0 -b-> 1, 1 -c-> 0, 1 -d-> 0, every state final (each pop reports its interval)
    d = E.DFA(2, 256)
    d.addLink(0, 0, ord("a"))
    d.addLink(0, 1, ord("b"))
    d.addLink(1, 0, ord("c"))
    d.addLink(1, 0, ord("e"))
    d.finishStates = {0, 1}
    d.compileBuckets()
    full_a = Counter(d.matchSA(sa, cap=10 ** 6, take="first"))
    full_b = Counter(d.matchSA(sa, cap=10 ** 6, take="last"))
    assert full_a == full_b and sum(full_a.values()) > 0
    short_a = Counter(d.matchSA(sa, cap=6, take="first"))
    short_b = Counter(d.matchSA(sa, cap=6, take="last"))
    assert short_a != short_b
