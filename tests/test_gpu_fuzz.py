"""GPU: randomized parity sweep of the device-level product (crp_spmm_csr_f64) against the oracle --
shapes, widths, leading dimensions, duplicates, empty rows, two-source column codes, every kernel
variant and processing order.  Seeds are fixed: a failure names its case."""
import numpy as np
import pytest

from conftest import FP64_TOL

pytestmark = pytest.mark.gpu


def _case(rng):
    m = int(rng.choice([1, 7, 64, 257, 1000, 4099, 9000]))
    k = int(rng.choice([1, 5, 300, 1000, 5000]))
    maxdeg = int(rng.choice([1, 4, 30, 90]))
    deg = rng.integers(0, min(maxdeg, k) + 1, size=m)
    if rng.random() < 0.5:
        deg[rng.integers(0, m, size=max(1, m // 6))] = 0
    rp = np.zeros(m + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    ci = np.empty(int(rp[-1]), np.int32)
    for i in range(m):
        if deg[i]:
            c = rng.integers(0, k, size=deg[i])                 # duplicates allowed
            ci[rp[i]:rp[i + 1]] = np.sort(c)
    va = rng.uniform(-1, 1, size=ci.size)
    return m, k, rp, ci, va


@pytest.mark.parametrize("seed", range(6))
def test_random_products(crp, orc, gpu, seed, monkeypatch):
    import torch
    from crp_spmm_amd import hip
    rng = np.random.default_rng(1000 + seed)
    monkeypatch.setenv("CRPSPMM_PANEL_ORDER", str(rng.choice(["0", "1", "2", "3"])))
    for rep in range(5):
        m, k, rp, ci, va = _case(rng)
        n = int(rng.choice([1, 3, 24, 31, 32, 33, 64, 100, 128, 130, 200, 256, 300]))
        split = rng.random() < 0.4 and k > 4                     # two-source: columns >= k0 come from B1
        k0 = k // 2 if split else k
        B = rng.uniform(-1, 1, size=(k, n))
        ref = orc.spmm_csr(rp, ci, va, B, fast=True)
        cdev = np.where(ci >= k0, ~(ci - k0), ci).astype(np.int32) if split else ci
        # codes must stay in key order (receive-buffer rows first): re-sort inside the rows
        if split:
            key = np.where(cdev < 0, ~cdev, cdev.astype(np.int64) + (1 << 31))
            order = np.lexsort((key, np.repeat(np.arange(m), np.diff(rp))))
            cdev, vdev = cdev[order], va[order]
        else:
            vdev = va
        pad = int(rng.choice([0, 1, 2, 6]))
        A = hip.CsrDev(m, k0, rp, cdev, vdev)
        B0 = torch.zeros((max(k0, 1), n + pad), dtype=torch.float64, device=gpu)
        B0[:k0, :n] = torch.from_numpy(B[:k0]).to(gpu)
        B1 = None
        if split:
            B1 = torch.zeros((k - k0, n + pad), dtype=torch.float64, device=gpu)
            B1[:, :n] = torch.from_numpy(B[k0:]).to(gpu)
        for variant in (0, 1, 2, 3, 5):
            Cd = torch.full((m, n + pad), float("nan"), dtype=torch.float64, device=gpu)
            hip.spmm_csr(A, B0[:, :n] if pad else B0, Cd[:, :n] if pad else Cd, n=n, B1=(B1[:, :n] if pad else B1) if split else None,
                         variant=variant)
            torch.cuda.synchronize()
            out = Cd.cpu().numpy()
            err = orc.rel_fro_err(ref, out[:, :n]) if np.abs(ref).sum() > 0 else float(np.abs(out[:, :n]).max())
            assert err <= FP64_TOL, (seed, rep, m, k, n, pad, split, variant, err)
            if pad:
                assert np.isnan(out[:, n:]).all(), (seed, rep, "padding written")
        A.free()
