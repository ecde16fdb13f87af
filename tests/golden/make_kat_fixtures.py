#!/usr/bin/env python3
"""Transcribes the data arrays of the reference's NaiveBWTSearcher known-answer
tests into tests/golden/naive_bwt_searcher_kat.json.

Run in the build container only (it READS the reference's test file as text to
lift the literal input arrays; nothing of the reference is executed):
    python tests/golden/make_kat_fixtures.py

Source of the vectors: /root/reference/src/test/scala/org/fmindex/tests/Indexer.scala
  :685-726  "BWTMerger2 test2048 occ searcher"       (t1v, bs, gtEof, 19 occ answers)
  :727-743  "BWTMerger2 test2048 occ 0xff searcher"  (t1v with one 0xff, gtEof, 3 answers + bwt.indexOf(0xff)==721)
The block BWT those tests search is what BWTMerger2.calcSAStatistic
(bwtmerger.scala:934-952) derives from (t1v, gtEof): remapAlphabet (:679-733),
suffix sort of the remapped string, sa2BWT (:782-810).  That derivation is
construction (out of scope for the product) and is restated here only to obtain
the KAT's input; the reference's own assertion indexOf(0xff)==721 checks it.
"""
import json
import os
import re

SRC = "/root/reference/src/test/scala/org/fmindex/tests/Indexer.scala"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "naive_bwt_searcher_kat.json")


def ints(s):
    return [int(x) for x in s.split(",")]


def remap_alphabet(t, gt_eof):
    n1 = len(t) - 1
    tn = t[n1]
    occ = [0] * 258
    for i in range(n1):
        ti = t[i]
        if ti < tn or (ti == tn and (i + 1) not in gt_eof):
            occ[ti] += 1
        else:
            occ[ti + 2] += 1
    occ[tn + 1] += 1
    mp = [0] * 258
    asize = 1
    for i in range(258):
        if occ[i] > 0:
            mp[i] = asize
            asize += 1
        else:
            mp[i] = 258
    newt = []
    for i in range(len(t)):
        if i == n1:
            c = t[i] + 1
        elif t[i] < tn or (t[i] == tn and (i + 1) not in gt_eof):
            c = t[i]
        else:
            c = t[i] + 2
        newt.append(mp[c])
    newt.append(0)
    return newt, asize


def calc_sa_statistic(t, gt_eof):
    remapped, _ = remap_alphabet(t, gt_eof)
    full = sorted(range(len(remapped)), key=lambda i: remapped[i:])
    sa = full[1:]                                  # drop the sentinel's row
    n = len(sa)
    bwt = [0] * n
    rank0 = -1
    for i in range(n):
        j = sa[i] - 1
        if j < 0:
            rank0 = i
            j = n - 1
        bwt[i] = t[j]
    if rank0 > 0:
        bwt[rank0] = bwt[rank0 - 1]
    elif n != 1:
        bwt[rank0] = bwt[rank0 + 1]
    return bwt, sa.index(0)


def main():
    text = open(SRC).read()
    arrays = [ints(m) for m in re.findall(r"val t1v = Array\[Byte\]\(([-0-9,]+)\)", text)]
    bs = ints(re.search(r"val bs = Array\[Long\]\(([0-9,]+)\)", text).group(1))
    gts = [ints(m) for m in re.findall(r"val gtEof=BitSet\(([0-9,]+)\)", text)]
    assert len(arrays) == 2 and len(gts) == 2 and len(bs) == 256
    t1 = [x & 0xFF for x in arrays[0]]
    t2 = [x & 0xFF for x in arrays[1]]
    bwt1, rk1 = calc_sa_statistic(t1, set(gts[0]))
    bwt2, rk2 = calc_sa_statistic(t2, set(gts[1]))
    assert bwt2.index(0xFF) == 721          # Indexer.scala:737-738
    # bucket starts for case 2: calcBs(calcOcc(t1v)) == plain exclusive prefix sum of byte counts
    cnt = [0] * 256
    for x in t2:
        cnt[x] += 1
    bs2, tot = [], 0
    for c in range(256):
        bs2.append(tot)
        tot += cnt[c]
    j, d, q, t_, e = ord("j"), ord("d"), ord("q"), ord("t"), ord("e")
    kat1 = [[j, 130, 3], [j, 131, 4], [j, 132, 5], [j, 133, 5], [j, 600, 16], [j, 954, 29], [j, 968, 30],
            [j, 1007, 30], [d, 30, 0], [d, 31, 0], [d, 32, 1], [d, 998, 36], [d, 999, 37], [d, 1000, 37],
            [d, 1004, 37], [q, 730, 31], [t_, 661, 25], [e, 411, 13], [-1, 411, 0]]     # Indexer.scala:703-725
    kat2 = [[-1, 721, 1], [-1, 720, 0], [-1, 722, 1]]                                   # Indexer.scala:740-742
    json.dump({"source": "Indexer.scala:685-743",
               "case1": {"t1v": t1, "gtEof": gts[0], "bs": bs, "bwt": bwt1, "rk0": rk1, "occ": kat1},
               "case2": {"t1v": t2, "gtEof": gts[1], "bs": bs2, "bwt": bwt2, "rk0": rk2, "occ": kat2}},
              open(OUT, "w"))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
