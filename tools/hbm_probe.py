#!/usr/bin/env python3
"""HBM bandwidth probe for DESIGN.md (SURVEY.md section 8d asks for the box's own number next to the 8 TB/s
spec): device-to-device copy and a read-only reduction over a 4 GiB fp64 buffer, best of 10."""
import time
import torch

dev = torch.device("cuda", 0)
n = 1 << 29                               # 4 GiB of fp64
a = torch.ones(n, dtype=torch.float64, device=dev)
b = torch.empty_like(a)
for name, fn, nbytes in (("copy (read + write)", lambda: b.copy_(a), 2 * n * 8), ("sum (read only)", lambda: a.sum(), n * 8)):
    best = 1e9
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%-20s %.2f TB/s" % (name, nbytes / best / 1e12))
