#!/usr/bin/env python3
"""How long hipMalloc takes on this box for the sizes the row jump table asks for, beside 117 GiB already held (the C3 index
without that table): one 128 GiB allocation, two of 64 GiB, one of 64 GiB.   python tools/alloc_time.py"""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def malloc(nbytes):
    p = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes))
    hip.hipDeviceSynchronize()
    return rc, p, time.perf_counter() - t0
def free(p):
    hip.hipFree(p)
G = 1 << 30
rc, held, dt = malloc(117 * G); print("117 GiB (held): rc %d, %.3f s" % (rc, dt))
for label, sizes in (("128 GiB", [128 * G]), ("2 x 64 GiB", [64 * G, 64 * G]), ("64 GiB", [64 * G]), ("128 GiB again", [128 * G])):
    ps, tot = [], 0.0
    for s in sizes:
        rc, p, dt = malloc(s); ps.append(p); tot += dt
        if rc: print(label, "failed rc", rc)
    print("%-14s %.3f s" % (label, tot))
    for p in ps: free(p)
free(held)
