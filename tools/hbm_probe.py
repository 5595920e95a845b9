#!/usr/bin/env python3
"""HBM bandwidth probe for DESIGN.md (SURVEY.md section 8d asks for the box's own number next to the 8 TB/s
spec): device-to-device copy, a read-only reduction and a 2-reads-1-write add (the read : write mix of the SpMM's
measured traffic, 1.0 GB : 0.45 GB) over 4 GiB fp64 buffers, best of 10, timed with events."""
import time
import torch

dev = torch.device("cuda", 0)
n = 1 << 29                               # 4 GiB of fp64
a = torch.ones(n, dtype=torch.float64, device=dev)
b = torch.empty_like(a)
c = torch.empty_like(a)
for name, fn, nbytes in (("copy (read + write)", lambda: b.copy_(a), 2 * n * 8), ("sum (read only)", lambda: a.sum(), n * 8),
                         ("add (2 reads + 1 write)", lambda: torch.add(a, b, out=c), 3 * n * 8)):
    best = 1e9
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    print("%-26s %.2f TB/s" % (name, nbytes / best / 1e12))
