#!/bin/bash
O=gpurun_out/${1:-r03p}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
FMX_TRACE=1 timeout -k 10 300 python - 2>&1 <<'PY' | grep -v amdgpu.ids | tee $O/jump_build.txt
import time, torch, bench, findex_amd
dev = torch.device("cuda", 0)
for log2n, sigma in ((32, 128), (30, 28), (28, 4)):
    n = 1 << log2n
    bwt, eof = bench.make_bwt(torch, n, sigma if sigma != 28 else bench.C4_ALPHABET, 3, dev)
    torch.cuda.synchronize()
    hip = findex_amd.HipFMSearcher.from_device(bwt.data_ptr(), n, eof, None)
    del bwt; torch.cuda.empty_cache()
    t0 = time.time(); findex_amd._lib.check(hip._L.fmx_prepare(hip._h, 4)); t1 = time.time()
    st = hip.stats()
    print("n=2^%d sigma=%s: jump table %.1f GiB built in %.3f s (tables_build_ms %.0f)" % (log2n, sigma, st["jump_bytes"] / 2**30, t1 - t0, st["tables_build_ms"]), flush=True)
    hip.close()
PY
