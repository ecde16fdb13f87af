"""Pins oracle/retree.py (re2post + ReTree + _matchSA) to the reference's own
known answers.  T = /root/reference/src/test/scala/org/fmindex/tests/REParser.scala
"""
import pytest

import oracle
from oracle import retree as R
from helpers import bwt_of_text


# ------------------------------------------------------------------- re2post
def test_re2post_strings():
    """T:10-26"""
    assert R.re2poststr("abc") == "ab·c·"
    assert R.re2poststr("a(bb)+a") == "abb·+·a·"
    assert R.re2poststr("(a|b)") == "ab|"
    assert R.re2poststr("((a|b)*aba*)*(a|b)(a|b)") == "ab|*a·b·a*·*ab|·ab|·"


def test_re2post5_reference_vector_is_stale():
    """T:27-31 expects 'a.*·(·b·[a..z]·]·.*·c·', i.e. a stray ']' literal after
    the set.  The reference's own code cannot produce it: processAltChar
    consumes the ']' (re2.scala:96-108) and the caller resumes after it
    (:154-155,172); and REAnalys anal29 (T:466-471) only parses because no such
    token is emitted.  The oracle follows the code."""
    assert R.re2poststr("a.*\\(b[a-z].*c") == "a.*·(·b·[abcdefghijklmnopqrstuvwxyz]·.*·c·"


def test_re2post_syntax_errors():
    for bad in ("|a", "a)", "*a", "(a", "[a", "[a-]", "[-a]", "[z-a]", "a||b"):
        with pytest.raises(R.Re2PostSyntax):
            R.re2post(bad)


def test_re2post_classes():
    """re2.scala:66-70: \\w = ['A','z'), \\d = ['0','9'), '.' = [2,255) or
    [0x20,255) with lineOnly -- END-EXCLUSIVE once ReTree expands them
    (retree.scala:165-173)."""
    (p,) = R.re2post("\\d")
    assert (p.start, p.end) == (ord("0"), ord("9"))
    t = R.ReTree(R.re2post("\\d"))
    assert sorted(n.c for n in t.char_nodes()) == list(range(ord("0"), ord("9")))
    t = R.ReTree(R.re2post("."))
    assert sorted(n.c for n in t.char_nodes()) == list(range(2, 255))
    t = R.ReTree(R.re2post(".", lineOnly=True))
    assert sorted(n.c for n in t.char_nodes()) == list(range(0x20, 255))


# ------------------------------------------------------------------- ReTree
MUST_PARSE = [  # T:319-478 REAnalys anal1..anal30, T:544-556,567-572
    "abcd", "abcd*", "abc*d", "a*bcd", "a*b*c*d*", "(ab)*", "(ab)*cd", "(ab)*(cd*)*", "(a|b)", "(a|b|d|c)",
    "(a|b*|d|c)", "(a|b*|d|c)*|(abc)", "(a|b|c)|(c|d|e)", "[a-c]", "a[b-d]e", "a[b-d]*e", "a[x.]e", "a\\de",
    "a+", "a****", "a+b", "a+((b|c)+|d)", "a*+", "a+*", "a+*+*++*", "a?", "(abc)?+|a?|bcd", "ab(cd|ef)+gh",
    "(10\\.[0-9]|[1-9][0-9]|[1-2][0-5][0-5]\\.[0-9]|[1-9][0-9]|[1-2][0-5][0-5]\\.[0-9]|[1-9][0-9]|[1-2][0-5][0-5])",
    "ab(cd)*ef", "ab*(cd)*(gh)*ij", "a(cd|ef)*j",
    ".*ab(cd)*(m(k|l)|tm*)(a|abc)(a*|(abc)*)ef(a*b*c*dg*)*gh",
    "a.*(b|c)d.*f",
]


@pytest.mark.parametrize("re", MUST_PARSE)
def test_must_parse(re):
    t = R.ReTree(R.re2post(re))
    t.tables()


def test_shapes_the_reference_has_no_case_for():
    """retree.scala:243-295: ConcatPoint has no case for (OrNode, CharNode),
    (OrNode, UnarOp) or (anything, FollowNode); :184-239: OrPoint none for
    (OrNode, CharNode) etc.  Those regexes throw scala.MatchError there."""
    for re in ("(a|b)c", "[ab]c", "a(bc)", "(ab)(cd)", "[ab]c*", "(a|b)|c"):
        with pytest.raises(R.MatchError):
            R.ReTree(R.re2post(re))


def test_anal1_parents():
    """T:481-486"""
    t = R.ReTree(R.re2post("a"))
    assert repr(t.root.parent) == "<<<ROOT>>>"
    assert repr(t.root.childs[0].parent) == "F[a]"


def test_anal2_star_parent():
    """T:487-493"""
    t = R.ReTree(R.re2post("ab*"), removeNulls=False)
    assert repr(t.root.childs[1].childs[0].parent) == "*[b]"


def test_remove_border_nulls():
    """T:494-510"""
    t = R.ReTree(R.re2post("a*(b|a)*bB*cd*e*"), removeNulls=True)
    assert len(t.root.childs) == 3
    t = R.ReTree(R.re2post("a*(b|a)*b?B*c?d*e*"), removeNulls=True)
    assert len(t.root.childs) == 0 and t.root.isNull


def test_nums():
    """T:511-515,574-588"""
    t = R.ReTree(R.re2post("abcdef"))
    assert t.root.childs[3].num == 4
    t = R.ReTree(R.re2post("(a|bX|cYZ)(a|b|c)"))
    assert t.root.childs[1].childs[1].num == 4
    t = R.ReTree(R.re2post("(a|b|c)(a|b|c)"))
    assert t.root.childs[1].childs[1].num == 2


def _same(a, b):
    return len(a) == len(b) and all(x is y for x, y in zip(a, b))


def test_follows_abc_cde_star_ef():
    """T:517-542 'anal4.follows'"""
    t = R.ReTree(R.re2post("abc(cde)*ef"))
    F = t.root
    assert F.follows == []
    a, b, c, cdeS, e, f = F.childs
    assert _same(a.follows, [b]) and _same(b.follows, [c])
    assert _same(cdeS.follows, [e]) and _same(e.follows, [f]) and f.follows == []
    cdeSF = cdeS.childs[0]
    cc, cd, ce = cdeSF.childs
    assert _same(cdeSF.follows, [cc, e])
    assert _same(cc.follows, [cd]) and _same(cd.follows, [ce]) and _same(ce.follows, [cc, e])


def test_follows_question():
    """T:557-565"""
    t = R.ReTree(R.re2post("ab?j"))
    F = t.root
    assert F.childs[1].isNull
    assert _same(F.childs[0].follows, [F.childs[2], F.childs[1].childs[0]])


# ------------------------------------------------------------------- matchSA
class _PyIndex:
    """getPrevRange through the C oracle, as a SuffixWalkingAlgo stand-in."""

    def __init__(self, sa):
        self.sa, self.n = sa, sa.n

    def getPrevRange(self, sp, ep, c):
        return self.sa.getPrevRange(sp, ep, c)


def test_retree_match_sa_anal1():
    """T:591-605 REAnalys3 anal1: '.*(a|b)ca' over reversed 'mmabcacamabbbca'
    returns exactly 2 results.  (removeBorderNulls strips the leading '.*',
    retree.scala:371-385, so the search is '(a|b)ca': 2 firsts + 2 'c' + 2 'a'
    pops.)"""
    bwt, eof, counts = bwt_of_text(b"mmabcacamabbbca"[::-1])
    sa = oracle.SAISNaiveSearcher.from_mem(bwt, eof, counts)
    t = R.ReTree(R.re2post(".*(a|b)ca"))
    ret = t.matchSA(_PyIndex(sa))
    assert len(ret) == 2
    # same engine in C (limits bind the same way)
    cret, left, pops = sa.match_tables(t.tables())
    assert cret == ret and pops == 6 and left == 0


def test_c_and_python_match_sa_agree_when_limits_bind():
    bwt, eof, counts = bwt_of_text((b"the quick brown fox jumps over the lazy dog " * 6)[::-1])
    sa = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    for re in ("o.*e", "a[b-z]*e", ".*ab", "(a|e|o)+r", "th?e*"):
        t = R.ReTree(R.re2post(re))
        for mb, mi in ((1024, 1000), (16, 50), (4, 0), (1 << 20, 0) if "." not in re else (64, 300)):
            py, front, pops = t._matchSA(_PyIndex(sa), mb, mi)
            c, left, cpops = sa.match_tables(t.tables(), mb, mi)
            assert c == py and left == len(front) and cpops == pops, (re, mb, mi)


def test_match_sa_finds_all_occurrences_unbounded():
    """With limits that do not bind, the result multiset is every distinct
    (len, interval) the regex matches; check against Python's re on the text."""
    import re as pyre
    text = b"abcabdabeacdacd xabcabd"
    bwt, eof, counts = bwt_of_text(text[::-1])
    sa = oracle.NaiveFMSearcher.from_mem(bwt, eof, counts)
    t = R.ReTree(R.re2post("ab[cde]"))
    res, left, _ = sa.match_tables(t.tables(), 1 << 20, 0)
    assert left == 0
    total = sum(ep - sp for _, sp, ep in res)
    assert total == len(pyre.findall(rb"ab[cde]", text))
    assert all(ln == 3 for ln, _, _ in res)
