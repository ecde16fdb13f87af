#!/usr/bin/env python3
"""Seeded regex generator for BASELINE config C4: regexes over a lowercase + space +
newline alphabet -- 4-12 literal chars, up to two small sets, at most one '?', one (x|y) group and
one x* / x+ on a small set; no '.'; at most 32 Glushkov positions.  Only shapes the reference's
ReTree.apply accepts are kept (it has no case for e.g. a set followed by a literal at the start,
re2/retree.scala:243-295): candidates are filtered through a `compiles(re) -> bool` callback."""
import random

ALPHABET = "abcdefghijklmnopqrstuvwxyz \n"


def gen_one(rng, letters="abcdefghijklmnopqrstuvwxyz", literal=None):
    """literal(rng, nlit) -> nlit characters: where the literal part comes from (default: random letters; the text
    workloads pass a function that cuts a stretch out of their text, so that the regexes have matches there)."""
    nlit = rng.randint(4, 12)
    toks = list(literal(rng, nlit)) if literal else [rng.choice(letters) for _ in range(nlit)]
    extras = []
    for _ in range(rng.randint(0, 2)):
        extras.append("[" + "".join(rng.sample(letters, rng.randint(2, 3))) + "]")
    if rng.random() < 0.4:
        extras.append(rng.choice(letters) + "?")
    if rng.random() < 0.4:
        extras.append("(" + rng.choice(letters) + "|" + rng.choice(letters) + rng.choice(letters) + ")")
    if rng.random() < 0.4:
        extras.append("[" + "".join(rng.sample(letters, rng.randint(1, 3))) + "]" + rng.choice("*+"))
    for e in extras:
        toks.insert(rng.randint(1, len(toks)), e)      # never first: the reference cannot parse that
    return "".join(toks)


def generate(k, seed, compiles):
    rng = random.Random(seed)
    out = []
    while len(out) < k:
        re = gen_one(rng)
        if compiles(re):
            out.append(re)
    return out


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ".")
    from oracle import retree as R

    def ok(re):
        try:
            R.ReTree(R.re2post(re)).tables()
            return True
        except (R.MatchError, R.Re2PostSyntax):
            return False
    for r in generate(10, 4, ok):
        print(repr(r))
