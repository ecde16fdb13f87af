#!/usr/bin/env python3
"""Soak of the literal search: the bench's batch searched N times through the device entry points -- plain (sp, ep)
arrays and the 8-byte packed form in turn -- every call's output compared with the first call's on the device.
    python tools/soak_c3.py [calls] [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, findex_amd
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
log2n, sigma, k, m, seed = bench.LITERAL[wl]
n = 1 << log2n
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
bwt, eof = bench.make_bwt(torch, n, sigma, seed, dev); torch.cuda.synchronize()
hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None, device=0, stream=stream)
hip.prepare(ktab=True, jump=True)
del bwt
RING = 3          # distinct batches, rotated (round 5): every call's output is compared with its batch's first
ring = [bench.make_patterns(torch, hip, n, sigma, k, m, seed * 1000 + 7919 * j, dev, stream) for j in range(RING)]
esc = 4096
ref = []
for bp, bo in ring:
    rsp, rep_ = torch.zeros(k, dtype=torch.int64, device=dev), torch.zeros(k, dtype=torch.int64, device=dev)
    rpk = torch.zeros(hip.packed_words(k, esc), dtype=torch.int64, device=dev)
    scratch = torch.zeros(k, dtype=torch.int64, device=dev)
    hip.search_batch_dev(bp.data_ptr(), bo.data_ptr(), rsp.data_ptr(), rep_.data_ptr(), k, stream)
    hip.search_batch_ex_dev(bp.data_ptr(), bo.data_ptr(), rpk.data_ptr(), scratch.data_ptr(), k, stream, packed=True, escape_cap=esc)
    torch.cuda.synchronize()
    ref.append((rsp, rep_, rpk))
nesc_ref = [int(r[2][k]) for r in ref]
sp = [None, torch.zeros(k, dtype=torch.int64, device=dev)]
ep = [None, torch.zeros(k, dtype=torch.int64, device=dev)]
pk = [None, torch.zeros(hip.packed_words(k, esc), dtype=torch.int64, device=dev)]
bad = 0
t0 = time.time()
for i in range(calls):
    pats, off = ring[(i // 2) % RING]
    sp[0], ep[0], pk[0] = ref[(i // 2) % RING]
    nesc = nesc_ref[(i // 2) % RING]
    if i & 1:
        pk[1].zero_()
        hip.search_batch_ex_dev(pats.data_ptr(), off.data_ptr(), pk[1].data_ptr(), ep[1].data_ptr(), k, stream, packed=True, escape_cap=esc)
        # the words of the patterns and the count; the escape list's ORDER is whatever the waves' appends made it
        ok = torch.equal(pk[1][: k + 1], pk[0][: k + 1]) and torch.equal(torch.sort(pk[1][k + 1: k + 1 + 2 * nesc].view(-1, 2)[:, 0])[0],
                                                                         torch.sort(pk[0][k + 1: k + 1 + 2 * nesc].view(-1, 2)[:, 0])[0])
    else:
        sp[1].zero_(); ep[1].zero_()
        hip.search_batch_dev(pats.data_ptr(), off.data_ptr(), sp[1].data_ptr(), ep[1].data_ptr(), k, stream)
        ok = torch.equal(sp[1], sp[0]) and torch.equal(ep[1], ep[0])
    if not ok:
        bad += 1
        print("call %d differs" % i, flush=True)
    if (i + 1) % 1000 == 0:
        print("%d calls, %d differ, %.1fs" % (i + 1, bad, time.time() - t0), flush=True)
st = hip.stats()
print("soak %s: %d calls rotating through %d batches of %d patterns (%d hits in the last, %d wide intervals in the packed form), %d differ; search_residency 0x%x"
      % (wl, calls, RING, k, int((sp[0] < ep[0]).sum()), nesc, bad, st["search_residency"]))
sys.exit(1 if bad else 0)
