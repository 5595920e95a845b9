"""Development check of the team kernel (variant 4) against the oracle on a few shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle as orc
from crp_spmm_amd import gen, hip
dev = torch.device("cuda", 0)
cases = []
cases.append(("random 777x1234", gen.random_csr(777, 1234, 70, seed=3, empty_every=13), 1234))
cases.append(("banded 5000", gen.banded_fem(5000, offsets=(1, 2, 3, 40, 41, 900), seed=2), 5000))
nx, ny, nz = 300, 8, 5
m = nx * ny * nz
cases.append(("lattice %d" % m, gen.banded_fem(m, offsets=(1, 2, 3, nx, nx + 1, nx * ny, nx * ny + 1), seed=3), m))
cases.append(("tiny 5x9", gen.random_csr(5, 9, 4, seed=1), 9))
for name, (rp, ci, va), k in cases:
    mm = len(rp) - 1
    for n in (256, 130, 200):
        B = np.random.default_rng(n).uniform(-1, 1, size=(k, n))
        ref = orc.spmm_csr(rp, ci, va, B, fast=True)
        A = hip.CsrDev(mm, k, rp, ci, va)
        Bd = torch.from_numpy(B).to(dev)
        Cd = torch.full((mm, n), float("nan"), dtype=torch.float64, device=dev)
        hip.spmm_csr(A, Bd, Cd, n=n, variant=4)
        torch.cuda.synchronize()
        err = orc.rel_fro_err(ref, Cd.cpu().numpy())
        print(name, "n=%d" % n, "rel err %.2e" % err, flush=True)
        assert err <= 1e-12, (name, n)
        A.free()
print("TEAM_CHECK_OK")
