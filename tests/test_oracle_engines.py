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
