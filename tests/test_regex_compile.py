"""Host logic of the product (C++ re2post + ReTree inside libfmx.so, no GPU needed) against
the oracle's independent Python restatement and the reference's own known answers."""
import random

import pytest

import findex_amd
from findex_amd import REParser, ReTree
from oracle import retree as R
from test_oracle_regex import MUST_PARSE


def both(re, lineOnly=False):
    want = R.ReTree(R.re2post(re, lineOnly)).tables()
    got = ReTree(REParser.re2post(re, lineOnly)).tables()
    return got, want


@pytest.mark.parametrize("re", MUST_PARSE + ["a", ".", "\\w\\d", "x[abc]*y?z+", "a.*(b|c)da.*f", "99*0", "a\\.b\\\\c"])
def test_tables_equal_oracle(re):
    got, want = both(re)
    assert got == want


def test_line_only_dot():
    got, want = both("a.*(b|c)da.*f", lineOnly=True)    # T/REParser.scala:630 WordsDB
    assert got == want


def test_re2post_strings():
    """T/REParser.scala:10-26 through the product's parser"""
    assert REParser.re2poststr("abc") == "ab·c·"
    assert REParser.re2poststr("a(bb)+a") == "abb·+·a·"
    assert REParser.re2poststr("(a|b)") == "ab|"
    assert REParser.re2poststr("((a|b)*aba*)*(a|b)(a|b)") == "ab|*a·b·a*·*ab|·ab|·"
    assert REParser.re2poststr("a.*\\(b[a-z].*c") == R.re2poststr("a.*\\(b[a-z].*c")


def test_reference_known_answers_on_product_tables():
    """follows / num / isLast answers of T/REParser.scala:511-588 read off the flat tables."""
    t = ReTree(REParser.re2post("abc(cde)*ef")).tables()
    # nodes in tree order: a b c (c d e)* e f
    assert t["c"] == [ord(x) for x in "abccdeef"]
    a, b, c, cc, cd, ce, e, f = range(8)
    assert t["follows"][a] == [b] and t["follows"][b] == [c]
    assert t["follows"][cc] == [cd] and t["follows"][cd] == [ce] and t["follows"][ce] == [cc, e]
    assert t["follows"][e] == [f] and t["follows"][f] == []
    assert t["isLast"] == [False] * 7 + [True] and t["firsts"] == [a]
    t = ReTree(REParser.re2post("ab?j")).tables()
    assert t["follows"][0] == [2, 1]
    t = ReTree(REParser.re2post("abcdef")).tables()
    assert t["num"][3] == 4
    t = ReTree(REParser.re2post("(a|bX|cYZ)(a|b|c)")).tables()
    assert t["num"][-3:] == [4, 4, 4]
    t = ReTree(REParser.re2post("(a|b|c)(a|b|c)")).tables()
    assert t["num"][-3:] == [2, 2, 2]
    t = ReTree(REParser.re2post("a*(b|a)*b?B*c?d*e*")).tables()       # removeBorderNulls -> empty
    assert t["c"] == [] and t["firsts"] == []


def test_error_parity():
    for bad in ("|a", "a)", "*a", "(a", "[a", "[a-]", "[-a]", "[z-a]", "a||b"):
        with pytest.raises(R.Re2PostSyntax):
            R.re2post(bad)
        with pytest.raises(findex_amd.Re2PostSyntax):
            REParser.re2post(bad)
    for bad in ("(a|b)c", "[ab]c", "a(bc)", "(ab)(cd)", "[ab]c*", "(a|b)|c"):
        with pytest.raises(R.MatchError):
            R.ReTree(R.re2post(bad))
        with pytest.raises(findex_amd.MatchError):
            ReTree(REParser.re2post(bad))


def random_regex(rng, depth=0):
    """Random strings over the regex syntax: most are valid, some are not -- both sides must
    agree either way."""
    atoms = "abcde"
    out = []
    for _ in range(rng.randint(1, 5)):
        r = rng.random()
        if r < 0.45:
            out.append(rng.choice(atoms))
        elif r < 0.55:
            out.append("[" + "".join(rng.sample(atoms, rng.randint(1, 3))) + "]")
        elif r < 0.62:
            out.append("[a-" + rng.choice("bcde") + "]")
        elif r < 0.67:
            out.append(rng.choice([".", "\\d", "\\w"]))
        elif r < 0.85 and depth < 3:
            alts = [random_regex(rng, depth + 1) for _ in range(rng.randint(1, 3))]
            out.append("(" + "|".join(alts) + ")")
        else:
            out.append(rng.choice(atoms))
        if rng.random() < 0.3:
            out.append(rng.choice("*+?"))
    return "".join(out)


def test_random_regexes_agree_with_oracle():
    rng = random.Random(1234)
    ok = bad = 0
    for _ in range(600):
        re = random_regex(rng)
        try:
            want = R.ReTree(R.re2post(re)).tables()
        except R.Re2PostSyntax:
            with pytest.raises(findex_amd.Re2PostSyntax):
                ReTree(REParser.re2post(re))
            bad += 1
            continue
        except R.MatchError:
            with pytest.raises(findex_amd.MatchError):
                ReTree(REParser.re2post(re))
            bad += 1
            continue
        assert ReTree(REParser.re2post(re)).tables() == want, re
        ok += 1
    assert ok > 100 and bad > 50


def test_compile_batch_matches_one_by_one_and_the_oracle():
    """fmx_regex_compile_batch (all host cores inside the library): per regex the same tables and the same
    FMX_ERR_SYNTAX / FMX_ERR_MATCH status as fmx_regex_compile, which the tests above pin to the oracle's
    restatement of re2post / ReTree.apply (re2/re2.scala:50-185, re2/retree.scala:156-423)."""
    import numpy as np
    rng = random.Random(99)
    res = [random_regex(rng) for _ in range(3000)] + ["|a", "a)", "*a", "(a", "[a", "[a-]", "a||b", "(a|b)c", "a", "ab?j", "\xe9[\xe8-\xea]+z"]
    for threads in (b"1", b"3", b"0"):
        assert findex_amd.load().fmx_config_set(b"threads", threads) == 0
        cs = ReTree.compile_batch(res)
        assert len(cs) == len(res)
        n_ok = n_syntax = n_match = 0
        for i, re in enumerate(res):
            try:
                want = R.ReTree(R.re2post(re)).tables()
                code = 0
            except R.Re2PostSyntax:
                code = 7
            except R.MatchError:
                code = 8
            assert cs.status[i] == code, (re, cs.status[i], code)
            if code == 0:
                if i % 7 == 0 or i >= 3000:
                    assert cs[i].tables() == want, re
                n_ok += 1
            else:
                with pytest.raises(findex_amd.Re2PostSyntax if code == 7 else findex_amd.MatchError):
                    cs[i]
                n_syntax += code == 7
                n_match += code == 8
        assert n_ok > 500 and n_syntax >= 7 and n_match > 100
        good = cs.select(np.nonzero(cs.ok())[0])
        assert len(good) == n_ok and good[0].tables() == cs[int(np.nonzero(cs.ok())[0][0])].tables()
        # the first failure's message, as the one-regex entry point leaves it
        assert b"regex" in findex_amd.load().fmx_last_error()
    assert len(ReTree.compile_batch([])) == 0
    with pytest.raises(ValueError):
        ReTree.compile_batch(["a\0b"])
    with pytest.raises(findex_amd.FmxError):
        findex_amd._lib.check(findex_amd.load().fmx_config_set(b"threads", b"many"))
