"""Dev probe: how the n = 32 kernel time scales with nonzeros per row (per-entry vs per-wave cost)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crp_spmm_amd import gen, hip
dev = torch.device("cuda", 0)
m = 217918
cases = {"51/row": gen.PWTK_OFFSETS,
         "25/row": tuple(range(1, 8)) + tuple(range(1200, 1203)) + tuple(range(36000, 36003)),
         "13/row": tuple(range(1, 4)) + (1200, 1201) + (36000, 36001),
         "5/row": (1, 36000)}
for n in (32, 256):
    for name, offs in cases.items():
        rp, ci, va = gen.banded_fem(m, offsets=offs)
        A = hip.CsrDev(m, m, rp, ci, va)
        B = torch.ones((m, n), dtype=torch.float64, device=dev)
        Cd = torch.empty((m, n), dtype=torch.float64, device=dev)
        for _ in range(30):
            hip.spmm_csr(A, B, Cd, n=n, variant=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            hip.spmm_csr(A, B, Cd, n=n, variant=3)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print("n=%d %s nnz=%d: %.4f ms" % (n, name, rp[-1], dt * 1e3), flush=True)
        A.free()
