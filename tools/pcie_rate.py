#!/usr/bin/env python3
"""Host<->device copy rates of this box (pinned and pageable, each direction, and both at once): the floor of any
host-pointer entry point."""
import time
import torch
n = 40 << 20
dev = torch.device("cuda", 0)
d = torch.empty(n, dtype=torch.uint8, device=dev)
d2 = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
hp = torch.empty(n, dtype=torch.uint8).pin_memory()
hq = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
hg = torch.empty(n, dtype=torch.uint8)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: d.copy_(hp, non_blocking=True))
print("H2D pinned   40 MiB: %.3f ms = %.1f GB/s" % (t * 1e3, n / t / 1e9))
t = timed(lambda: hq.copy_(d2, non_blocking=True))
print("D2H pinned   16 MiB: %.3f ms = %.1f GB/s" % (t * 1e3, (16 << 20) / t / 1e9))
t = timed(lambda: d.copy_(hg))
print("H2D pageable 40 MiB: %.3f ms = %.1f GB/s" % (t * 1e3, n / t / 1e9))


def both():
    with torch.cuda.stream(s1):
        d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2):
        hq.copy_(d2, non_blocking=True)


t = timed(both)
print("both at once (40 MiB up, 16 MiB down, two streams): %.3f ms" % (t * 1e3))
