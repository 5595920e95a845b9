"""GPU (MI355X): parity of the HIP hot path, called through the C ABI, against the
oracle, the MKL golden fixtures and size-independent properties.
Tolerance: relative Frobenius error <= 1e-12 (BASELINE.json north_star, fp64);
byte-moving kernels (gather / scatter / transpose) must be bit-exact."""
import os

import numpy as np
import pytest

from conftest import FP64_TOL, GOLDEN, GOLDEN_NAMES, load_golden

pytestmark = pytest.mark.gpu


def _t(x, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _spmm(crp, dev, rp, ci, va, k, B, n, layout=0, B1=None, ldpad=0, variant=0):
    import torch
    from crp_spmm_amd import hip
    A = hip.CsrDev(len(rp) - 1, k, rp, ci, va)
    m = len(rp) - 1
    if layout == 0:
        Bd = torch.zeros((B.shape[0], n + ldpad), dtype=torch.float64, device=dev)
        Bd[:, :n] = _t(B[:, :n], dev)
        Cd = torch.full((m, n + ldpad), float("nan"), dtype=torch.float64, device=dev)
        B1d = None
        if B1 is not None:
            B1d = _t(B1[:, :n], dev)
        hip.spmm_csr(A, Bd[:, :n] if ldpad else Bd, Cd[:, :n] if ldpad else Cd, n=n, B1=B1d, variant=variant)
        torch.cuda.synchronize()
        out = Cd.cpu().numpy()
        if ldpad:
            assert np.isnan(out[:, n:]).all()       # padding between rows is never written
        A.free()
        return out[:, :n]
    # column-major: operands as (n, ld)
    ldb, ldc = B.shape[0] + 3, m + 2
    Bd = torch.zeros((n, ldb), dtype=torch.float64, device=dev)
    Bd[:, :B.shape[0]] = _t(B[:, :n].T, dev)
    Cd = torch.full((n, ldc), float("nan"), dtype=torch.float64, device=dev)
    hip.spmm_csr(A, Bd, Cd, n=n, layout=1)
    torch.cuda.synchronize()
    out = Cd.cpu().numpy()
    assert np.isnan(out[:, m:]).all()
    A.free()
    return out[:, :m].T


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 100, 128, 129, 256, 300, 512])
def test_kernel_vs_oracle_all_widths(crp, orc, gpu, n):
    from crp_spmm_amd import gen
    m, k = 777, 1234
    rp, ci, va = gen.random_csr(m, k, 70, seed=n, empty_every=13)
    B = np.random.default_rng(n).uniform(-2, 2, size=(k, n))
    ref = orc.spmm_csr(rp, ci, va, B)
    for variant in (0, 1, 2, 3, 5):                    # auto, csr-rowgroup, rowpanel-R4, rowpanel-R8, team2-R8
        for ldpad in (0, 1, 2):
            got = _spmm(crp, gpu, rp, ci, va, k, B, n, ldpad=ldpad, variant=variant)
            assert orc.rel_fro_err(ref, got) <= FP64_TOL, (n, ldpad, variant)
            assert not got[::13].any()                 # empty rows: exact zeros


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_kernel_variants_banded_and_nonfinite(crp, orc, gpu, variant):
    """Banded (column-sharing) matrix through every kernel family, duplicates included, and
    the 0 * Inf corner: an Inf / NaN in a B row must only reach the rows that reference it."""
    from crp_spmm_amd import gen
    m = k = 2500
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, 4, 60, 61, 900), seed=4)
    ci = ci.copy()
    ci[rp[10] + 1] = ci[rp[10]]                        # a duplicated column inside row 10
    n = 256
    B = np.random.default_rng(3).uniform(-1, 1, size=(k, n))
    ref = orc.spmm_csr(rp, ci, va, B)
    got = _spmm(crp, gpu, rp, ci, va, k, B, n, variant=variant)
    assert orc.rel_fro_err(ref, got) <= FP64_TOL
    B[1234, 7] = np.inf
    B[1300, 9] = np.nan
    ref = orc.spmm_csr(rp, ci, va, B)
    got = _spmm(crp, gpu, rp, ci, va, k, B, n, variant=variant)
    assert np.array_equal(np.isfinite(ref), np.isfinite(got))
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    fin = np.isfinite(ref)
    assert np.abs(ref[fin] - got[fin]).max() <= 1e-12 * np.abs(ref[fin]).max()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_kernel_vs_mkl_golden(crp, orc, gpu, name):
    g, s = load_golden(name, "csr"), load_golden(name, "spmm")
    k = int(g["k"])
    for n in (4, 33):
        got = _spmm(crp, gpu, g["rowptr"], g["colidx"], g["val"], k, orc.fill_B(0, k, 0, n), n)
        assert orc.rel_fro_err(s["C_fillB_n%d" % n], got) <= FP64_TOL
    got = _spmm(crp, gpu, g["rowptr"], g["colidx"], g["val"], k, s["B2"], 6)
    assert orc.rel_fro_err(s["C_B2"], got) <= FP64_TOL
    got = _spmm(crp, gpu, g["rowptr"], g["colidx"], g["val"], k, orc.fill_B(0, k, 0, 5), 5, layout=1)
    assert orc.rel_fro_err(s["C_fillB_colmajor_n5"].T, got) <= FP64_TOL


def test_kernel_two_source_index(crp, orc, gpu):
    """c >= 0 reads the local block, c < 0 reads row ~c of the receive buffer."""
    from crp_spmm_amd import gen
    m, k = 500, 900
    rp, ci, va = gen.random_csr(m, k, 30, seed=2)
    n = 48
    B = np.random.default_rng(0).normal(size=(k, n))
    lo, hi = 300, 650                                   # rows [lo, hi) are "local", the rest "received"
    remote_rows = np.concatenate([np.arange(0, lo), np.arange(hi, k)])
    pos = np.full(k, -1)
    pos[remote_rows] = np.arange(remote_rows.size)
    c2 = np.where((ci >= lo) & (ci < hi), ci - lo, ~pos[ci]).astype(np.int32)
    for n_, variant in ((n, 1), (200, 1), (200, 2), (200, 3), (48, 3), (48, 2)):
        B = np.random.default_rng(n_).normal(size=(k, n_))
        got = _spmm(crp, gpu, rp, c2, va, hi - lo, B[lo:hi], n_, B1=B[remote_rows], variant=variant)
        assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), got) <= FP64_TOL, (n_, variant)


def test_kernel_edge_shapes(crp, orc, gpu):
    import torch
    from crp_spmm_amd import hip
    # zero rows, zero columns of B/C, zero nnz
    A = hip.CsrDev(0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    hip.spmm_csr(A, torch.zeros((5, 4), dtype=torch.float64, device=gpu), torch.zeros((0, 4), dtype=torch.float64, device=gpu), n=4)
    A.free()
    A = hip.CsrDev(6, 5, np.zeros(7, np.int32), np.zeros(0, np.int32), np.zeros(0))
    Cd = torch.full((6, 4), 7.0, dtype=torch.float64, device=gpu)
    hip.spmm_csr(A, torch.ones((5, 4), dtype=torch.float64, device=gpu), Cd, n=4)
    torch.cuda.synchronize()
    assert not Cd.cpu().numpy().any()                   # beta = 0: C overwritten with zeros
    A.free()
    # one long row (> 64 nonzeros, several broadcast rounds) and duplicates / explicit zeros
    k = 400
    ci = np.sort(np.concatenate([np.arange(k), np.array([3, 3, 77])])).astype(np.int32)
    va = np.linspace(-1, 1, ci.size)
    va[5] = 0.0
    rp = np.array([0, ci.size], dtype=np.int32)
    B = np.random.default_rng(4).normal(size=(k, 20))
    got = _spmm(crp, gpu, rp, ci, va, k, B, 20)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), got) <= FP64_TOL
    # bad arguments are reported, not executed
    lib = crp.load()
    assert lib.crp_spmm_csr_f64(None, 0, 4, None, 0, None, 0, None, 0, 0, None) != 0


def test_row_kernels_bit_exact(crp, gpu):
    import torch
    from crp_spmm_amd import hip
    rng = np.random.default_rng(1)
    for n in (1, 6, 7, 256):
        src = rng.normal(size=(300, n))
        idx = rng.integers(0, 300, size=411).astype(np.int32)
        sd, id_ = _t(src, gpu), _t(idx, gpu)
        dst = torch.zeros((411, n), dtype=torch.float64, device=gpu)
        hip.gather_rows(id_, sd, dst)
        assert np.array_equal(dst.cpu().numpy(), src[idx])
        perm = rng.permutation(300).astype(np.int32)
        out = torch.zeros((300, n), dtype=torch.float64, device=gpu)
        hip.gather_rows(_t(perm, gpu), sd, out, scatter=True)
        exp = np.zeros_like(src)
        exp[perm] = src
        assert np.array_equal(out.cpu().numpy(), exp)
        # column-major gather: operands (n, ld)
        srcT = _t(np.ascontiguousarray(src.T), gpu)
        dstT = torch.zeros((n, 411), dtype=torch.float64, device=gpu)
        hip.gather_rows(id_, srcT, dstT, layout=1)
        assert np.array_equal(dstT.cpu().numpy(), src[idx].T)
    for (r, c) in [(1, 1), (33, 65), (300, 7), (64, 64)]:
        x = rng.normal(size=(r, c))
        y = torch.zeros((c, r), dtype=torch.float64, device=gpu)
        hip.transpose(_t(x, gpu), y)
        assert np.array_equal(y.cpu().numpy(), x.T)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("where", ["device", "host"])
def test_rp_engine_single_rank(crp, orc, gpu, layout, where):
    """rp_spmm protocol of examples/test_rp_spmm.c:120-145 at one rank, B = fill_B."""
    import torch
    from crp_spmm_amd import comm, engine, gen
    m = k = 3000
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 5, 90, 700), seed=3)
    n = 40
    B = orc.fill_B(0, k, 0, n)
    ref = orc.spmm_csr(rp, ci, va, B)
    sc = comm.SelfComm()
    e = engine.RpSpmm(0, m, rp, ci, va, [0, k], n, sc)
    if layout == 0:
        Bh, Ch = B.copy(), np.full((m, n), np.nan)
    else:
        Bh, Ch = np.ascontiguousarray(B.T), np.full((n, m), np.nan)
    if where == "device":
        Bx, Cx = _t(Bh, gpu), _t(Ch, gpu)
    else:
        Bx, Cx = Bh, Ch
    e.exec(layout, Bx, Cx)          # warm-up, like the reference driver
    e.clear_stat()
    for _ in range(2):
        e.exec(layout, Bx, Cx)
    if where == "device":
        torch.cuda.synchronize()
        Cx = Cx.cpu().numpy()
    got = Cx if layout == 0 else Cx.T
    assert orc.rel_fro_err(ref, got) <= FP64_TOL
    v = e.plan()
    assert v["n_exec"] == 2 and v["t_exec"] > 0 and v["t_spmm"] > 0 and v["t_unpack"] == 0
    e.print_stat()
    e.free()
    sc.free()


def test_engine_update_values(crp, orc, gpu):
    """New values on the same pattern (the deprecated engine's calling convention) refresh the CSR
    and the derived row-panel format."""
    import torch
    from crp_spmm_amd import comm, engine, gen
    m = k = 2000
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, 40), seed=6)
    n = 128
    B = orc.fill_B(0, k, 0, n)
    sc = comm.SelfComm()
    e = engine.RpSpmm(0, m, rp, ci, np.zeros_like(va), [0, k], n, sc)
    Bd, Cd = _t(B, gpu), torch.empty((m, n), dtype=torch.float64, device=gpu)
    for trial in range(2):
        v2 = va * (trial + 1) + trial
        e.update_values(v2)
        for variant in (1, 2, 3):
            e.set_variant(variant)
            e.exec(0, Bd, Cd)
            torch.cuda.synchronize()
            assert orc.rel_fro_err(orc.spmm_csr(rp, ci, v2, B), Cd.cpu().numpy()) <= FP64_TOL, (trial, variant)
    e.free()
    sc.free()


@pytest.mark.parametrize("kind", ["kkt3d", "fem3d", "er"])
def test_baseline_config_standins(crp, orc, gpu, kind):
    """Stand-ins for the other BASELINE configs (nlpkkt240 n=128, Queen_4147 n=1024, Erdos-Renyi n=64)
    at sizes the oracle finishes in seconds: every kernel variant through the engine, plus linearity
    (C(B1 + 2 B2) = C(B1) + 2 C(B2)) as the size-independent property."""
    import torch
    from crp_spmm_amd import comm, engine, gen
    if kind == "kkt3d":
        rp, ci, va = gen.kkt3d(20)
        n = 128
    elif kind == "fem3d":
        rp, ci, va = gen.fem3d(12)
        n = 1024
    else:
        rp, ci, va = gen.erdos_renyi(20000, 20000, 32, seed=1)
        n = 64
    m = len(rp) - 1
    B = np.random.default_rng(5).uniform(-1, 1, size=(m, n))
    ref = orc.spmm_csr(rp, ci, va, B, fast=True)
    sc = comm.SelfComm()
    e = engine.RpSpmm(0, m, rp, ci, va, [0, m], n, sc)
    Bd, Cd = _t(B, gpu), torch.empty((m, n), dtype=torch.float64, device=gpu)
    for variant in (0, 1, 2, 3):
        e.set_variant(variant)
        Cd.fill_(float("nan"))
        e.exec(0, Bd, Cd)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(ref, Cd.cpu().numpy()) <= FP64_TOL, (kind, variant)
    e.set_variant(0)
    B2 = _t(np.random.default_rng(6).uniform(-1, 1, size=(m, n)), gpu)
    C1, C2, C3 = torch.empty_like(Cd), torch.empty_like(Cd), torch.empty_like(Cd)
    e.exec(0, Bd, C1)
    e.exec(0, B2, C2)
    e.exec(0, Bd + 2.0 * B2, C3)
    torch.cuda.synchronize()
    assert orc.rel_fro_err((C1 + 2.0 * C2).cpu().numpy(), C3.cpu().numpy()) <= 1e-13
    e.free()
    sc.free()


def test_b_block_beyond_4gib(crp, orc, gpu):
    """B rows addressed past 4 GiB (what one GPU sees for nlpkkt240 x n=256): the row-panel kernels
    switch from 32-bit buffer offsets to 64-bit addresses.  Reached cheaply with a 4 MiB leading
    dimension: 1100 rows span 4.6 GB while only n columns of each row hold data."""
    import torch
    from crp_spmm_amd import gen, hip
    k, n, ld = 1100, 256, 1 << 19
    m = 1500
    rp, ci, va = gen.random_csr(m, k, 40, seed=21)
    ci = ci.copy()
    ci[::5] = k - 1 - (ci[::5] % 7)                      # make sure the last rows (beyond 4 GiB) are hit
    for r in range(m):                                   # keep columns ascending inside a row
        ci[rp[r]:rp[r + 1]] = np.sort(ci[rp[r]:rp[r + 1]])
    B = np.random.default_rng(3).uniform(-1, 1, size=(k, n))
    ref = orc.spmm_csr(rp, ci, va, B)
    Bbig = torch.empty((k, ld), dtype=torch.float64, device=gpu)
    assert Bbig.numel() * 8 > (1 << 32)
    Bbig[:, :n] = _t(B, gpu)
    A = hip.CsrDev(m, k, rp, ci, va)
    for variant in (1, 2, 3):
        Cd = torch.full((m, n), float("nan"), dtype=torch.float64, device=gpu)
        hip.spmm_csr(A, Bbig[:, :n], Cd, n=n, variant=variant)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(ref, Cd.cpu().numpy()) <= FP64_TOL, variant
    A.free()
    del Bbig
    torch.cuda.empty_cache()


@pytest.mark.parametrize("order", ["default", "0", "1", "2", "3"])
def test_lattice_matrix_team_paths(crp, orc, gpu, monkeypatch, order):
    """A stride-lattice matrix (two nested far strides) through every processing order of the
    row-panel kernels (default = team schedule with the per-round workgroup barrier) and through the LDS-sharing
    team kernel (variant 5), at widths on both sides of its 256-column tile; then new values on the same pattern
    (both formats refresh)."""
    import torch
    from crp_spmm_amd import comm, engine, gen
    if order != "default":
        monkeypatch.setenv("CRPSPMM_PANEL_ORDER", order)
    nx, ny, nz = 300, 7, 5
    m = nx * ny * nz + 13                                      # ragged last tooth / partial teams
    if order in ("default", "3"):                              # 27-point-like: the outer stride is a group of three clusters
        offs = (1, 2, nx - 1, nx, nx + 1, nx * ny - nx, nx * ny - 1, nx * ny, nx * ny + 1, nx * ny + nx)
    else:
        offs = (1, 2, 3, nx, nx + 1, nx * ny, nx * ny + 1)
    rp, ci, va = gen.banded_fem(m, offsets=offs, seed=3)
    sc = comm.SelfComm()
    for n in (256, 200, 64, 512):
        B = np.random.default_rng(n).uniform(-1, 1, size=(m, n))
        ref = orc.spmm_csr(rp, ci, va, B, fast=True)
        e = engine.RpSpmm(0, m, rp, ci, va, [0, m], n, sc)
        Bd, Cd = _t(B, gpu), torch.empty((m, n), dtype=torch.float64, device=gpu)
        for variant in (3, 5, 2):
            e.set_variant(variant)
            Cd.fill_(float("nan"))
            e.exec(0, Bd, Cd)
            torch.cuda.synchronize()
            assert orc.rel_fro_err(ref, Cd.cpu().numpy()) <= FP64_TOL, (order, n, variant)
            first = Cd.clone()
            e.exec(0, Bd, Cd)
            torch.cuda.synchronize()
            assert torch.equal(first, Cd), (order, n, variant, "not reproducible")
        e.update_values(va * 3.0)
        for variant in (3, 5):
            e.set_variant(variant)
            e.exec(0, Bd, Cd)
            torch.cuda.synchronize()
            assert orc.rel_fro_err(3.0 * ref, Cd.cpu().numpy()) <= FP64_TOL, (order, n, variant, "update_values")
        e.free()
    sc.free()


def test_row_subset_matrices(crp, orc, gpu):
    """crp_csr_dev_set_rowmap: two row subsets of A write disjoint rows of one C; every kernel
    variant; the untouched rows keep their content."""
    import ctypes as C
    import torch
    from crp_spmm_amd import gen
    lib = crp.load()
    _IP, _DP = C.POINTER(C.c_int), C.POINTER(C.c_double)
    m = k = 3001
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, 60), seed=13)
    n = 96
    B = orc.fill_B(0, k, 0, n)
    C_ref = orc.spmm_csr(rp, ci, va, B)
    sel = np.zeros(m, dtype=bool)
    sel[100:1500] = True
    sel[2000::3] = True
    parts = [np.nonzero(sel)[0].astype(np.int32), np.nonzero(~sel)[0].astype(np.int32)]
    Bd = _t(B, gpu)
    for variant in (1, 2, 3, 5):
        Cd = torch.full((m, n), -3.0, dtype=torch.float64, device=gpu)
        for pi, rows in enumerate(parts):
            cnt = (rp[rows + 1] - rp[rows]).astype(np.int64)
            sub_rp = np.zeros(rows.size + 1, dtype=np.int32)
            sub_rp[1:] = np.cumsum(cnt)
            idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
            sub_ci, sub_va = np.ascontiguousarray(ci[idx]), np.ascontiguousarray(va[idx])
            h = C.c_void_p()
            assert lib.crp_csr_dev_create(rows.size, k, sub_rp.ctypes.data_as(_IP), sub_ci.ctypes.data_as(_IP),
                                          sub_va.ctypes.data_as(_DP), C.byref(h)) == 0
            assert lib.crp_csr_dev_set_rowmap(h, rows.ctypes.data_as(_IP), int(rows.max())) == -2      # a row outside C
            assert lib.crp_csr_dev_set_rowmap(h, rows.ctypes.data_as(_IP), m) == 0
            assert lib.crp_spmm_csr_f64(h, 0, n, Bd.data_ptr(), n, None, 0, Cd.data_ptr(), n, variant, None) == 0
            torch.cuda.synchronize()
            got = Cd.cpu().numpy()
            assert orc.rel_fro_err(C_ref[rows], got[rows]) <= FP64_TOL, (variant, pi)
            if pi == 0:
                assert (got[parts[1]] == -3.0).all()
            lib.crp_csr_dev_destroy(C.byref(h))
        assert orc.rel_fro_err(C_ref, Cd.cpu().numpy()) <= FP64_TOL, variant
    # the narrow team kernels (variants 6, 7: 24 .. 64 columns); variant 7 keeps the C rows of its panels in a table that must
    # follow a row map set AFTER the first product
    n2 = 48
    B2 = orc.fill_B(0, k, 0, n2)
    C2_ref = orc.spmm_csr(rp, ci, va, B2)
    B2d = _t(B2, gpu)
    rows = parts[0]
    cnt = (rp[rows + 1] - rp[rows]).astype(np.int64)
    sub_rp = np.zeros(rows.size + 1, dtype=np.int32)
    sub_rp[1:] = np.cumsum(cnt)
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
    sub_ci, sub_va = np.ascontiguousarray(ci[idx]), np.ascontiguousarray(va[idx])
    for variant in (7,):
        h = C.c_void_p()
        assert lib.crp_csr_dev_create(rows.size, k, sub_rp.ctypes.data_as(_IP), sub_ci.ctypes.data_as(_IP), sub_va.ctypes.data_as(_DP), C.byref(h)) == 0
        Cd = torch.full((m, n2), -3.0, dtype=torch.float64, device=gpu)
        assert lib.crp_spmm_csr_f64(h, 0, n2, B2d.data_ptr(), n2, None, 0, Cd.data_ptr(), n2, variant, None) == 0      # no map: rows 0 .. rows.size - 1
        torch.cuda.synchronize()
        got = Cd.cpu().numpy()
        assert orc.rel_fro_err(C2_ref[rows], got[:rows.size]) <= FP64_TOL, variant
        assert (got[rows.size:] == -3.0).all()
        assert lib.crp_csr_dev_last_variant(h) == variant
        Cd.fill_(-3.0)
        assert lib.crp_csr_dev_set_rowmap(h, rows.ctypes.data_as(_IP), m) == 0
        assert lib.crp_spmm_csr_f64(h, 0, n2, B2d.data_ptr(), n2, None, 0, Cd.data_ptr(), n2, variant, None) == 0
        torch.cuda.synchronize()
        got = Cd.cpu().numpy()
        assert orc.rel_fro_err(C2_ref[rows], got[rows]) <= FP64_TOL, variant
        assert (got[parts[1]] == -3.0).all()
        lib.crp_csr_dev_destroy(C.byref(h))


def test_crpspmm_engine_single_rank(crp, orc, gpu):
    """The older all-in-one API at one rank: host operands in, host C out, values passed per exec
    (deprecated/src/crpspmm.h:89-122); C through a padded leading dimension."""
    from crp_spmm_amd import comm, engine, gen
    m, k, n = 1300, 1700, 40
    rp, ci, va = gen.random_csr(m, k, 18, seed=12)
    B = orc.fill_B(0, k, 0, n)
    sc = comm.SelfComm()
    e = engine.CrpspmmEngine(m, n, k, 0, m, rp, ci, 0, k, 0, n, 0, m, 0, n, sc)
    v = e.view()
    assert (v["np_row"], v["np_col"], v["loc_A_nrow"], v["loc_B_ncol"]) == (1, 1, m, n)
    Cpad = np.full((m, n + 3), -7.0)
    for trial in range(2):
        v2 = va * (1.0 + trial)
        e.exec(v2, B, Cpad[:, :n])
        assert orc.rel_fro_err(orc.spmm_csr(rp, ci, v2, B), Cpad[:, :n]) <= FP64_TOL, trial
        assert (Cpad[:, n:] == -7.0).all()
    assert e.view()["n_exec"] == 2
    e.print_stat()
    e.free()
    sc.free()


def test_para2d_engine_single_rank_no_deadlock(crp, orc, gpu):
    """para2d at one rank (the reference self-sends and hangs: src/para2d_spmm.c:102-109)."""
    import torch
    from crp_spmm_amd import comm, engine, gen, planner
    m, k = 900, 900
    rp, ci, va = gen.random_csr(m, k, 25, seed=8)
    n = 16
    rb = planner.csr_mat_row_partition(rp, 1)
    pl = planner.calc_spmm_part2d_from_1d(1, m, n, k, rb, rp, ci)
    assert (pl["pm"], pl["pn"]) == (1, 1)
    sc = comm.SelfComm()
    e = engine.Para2dSpmm(sc, 1, 1, pl["A0_rowptr"], pl["B_rowptr"], pl["AC_rowptr"], pl["BC_colptr"], rp, ci, va)
    B = orc.fill_B(0, k, 0, n)
    Cd = torch.zeros((m, n), dtype=torch.float64, device=gpu)
    e.exec(0, _t(B, gpu), Cd)
    torch.cuda.synchronize()
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), Cd.cpu().numpy()) <= FP64_TOL
    assert e.rA_cost == 0
    e.print_stat()
    e.free()
    sc.free()


def test_full_size_properties_pwtk_standin(crp, orc, gpu):
    """BASELINE config 2 size (pwtk stand-in, n = 256): closed form for fill_B, linearity,
    and a sampled-row comparison with the oracle (the full oracle product would take minutes)."""
    import torch
    from crp_spmm_amd import gen, hip
    m = k = 217918
    n = 256
    rp, ci, va = gen.banded_fem(m)
    assert rp[-1] == 11102984
    A = hip.CsrDev(m, k, rp, ci, va)
    i = torch.arange(k, dtype=torch.float64, device=gpu)[:, None]
    j = torch.arange(n, dtype=torch.float64, device=gpu)[None, :]
    B = i * 0.19 + j * 0.24                                   # fill_B on the device
    C1 = torch.empty((m, n), dtype=torch.float64, device=gpu)
    hip.spmm_csr(A, B, C1)
    rows = np.repeat(np.arange(m), np.diff(rp))
    s1 = np.bincount(rows, weights=va * ci, minlength=m)
    s0 = np.bincount(rows, weights=va, minlength=m)
    expect = 0.19 * s1[:, None] + 0.24 * np.arange(n)[None, :] * s0[:, None]
    got = C1.cpu().numpy()
    assert np.linalg.norm(got - expect) / np.linalg.norm(expect) <= 1e-12
    # sampled rows against the oracle
    pick = np.unique(np.random.default_rng(0).integers(0, m, size=400))
    sub_rp = np.concatenate([[0], np.cumsum(rp[pick + 1] - rp[pick])]).astype(np.int32)
    sel = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in pick])
    ref = orc.spmm_csr(sub_rp, ci[sel], va[sel], B.cpu().numpy())
    assert orc.rel_fro_err(ref, got[pick]) <= FP64_TOL
    # linearity: A(2X - 3Y) == 2AX - 3AY
    g = torch.Generator(device=gpu).manual_seed(1)
    X = torch.rand((k, n), dtype=torch.float64, device=gpu, generator=g)
    Y = torch.rand((k, n), dtype=torch.float64, device=gpu, generator=g)
    CX, CY, CZ = (torch.empty((m, n), dtype=torch.float64, device=gpu) for _ in range(3))
    hip.spmm_csr(A, X, CX)
    hip.spmm_csr(A, Y, CY)
    hip.spmm_csr(A, 2 * X - 3 * Y, CZ)
    lin = 2 * CX - 3 * CY
    assert (torch.linalg.norm(CZ - lin) / torch.linalg.norm(lin)).item() <= 1e-12
    A.free()


@pytest.mark.parametrize("values", ["compact", "full"])
@pytest.mark.parametrize("n", [24, 64, 100, 128, 130, 200, 256, 300, 520])
def test_team2_kernel_vs_oracle(crp, orc, gpu, monkeypatch, n, values):
    """Variant 5 (LDS-sharing team kernel, csrc/team2_kernel.hip): random matrix with empty rows (teams of 8
    consecutive panels, ragged last team), padded leading dimensions, and a stride-lattice matrix (teams of
    4 x 2 teeth); widths on both tile shapes (128 / 256 columns) and beyond one tile; both kernel instances -- value
    blocks without the holes (taken by itself for panels filled under 40 %) and with 8 values per part."""
    from crp_spmm_amd import gen
    monkeypatch.setenv("CRPSPMM_TEAM2_COMPACT", "1" if values == "compact" else "0")
    m, k = 777, 1234
    rp, ci, va = gen.random_csr(m, k, 70, seed=n, empty_every=13)
    B = np.random.default_rng(n).uniform(-2, 2, size=(k, n))
    ref = orc.spmm_csr(rp, ci, va, B)
    for ldpad in (0, 2):
        got = _spmm(crp, gpu, rp, ci, va, k, B, n, ldpad=ldpad, variant=5)
        assert orc.rel_fro_err(ref, got) <= FP64_TOL, (n, ldpad)
        assert not got[::13].any()
    nx, ny, nz = 300, 6, 5
    mm = nx * ny * nz
    rp, ci, va = gen.banded_fem(mm, offsets=(1, 2, 3, 4, 5, nx, nx + 1, nx * ny, nx * ny + 1), seed=4)
    B = np.random.default_rng(n + 1).uniform(-2, 2, size=(mm, n))
    got = _spmm(crp, gpu, rp, ci, va, mm, B, n, variant=5)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), got) <= FP64_TOL, ("lattice", n)


@pytest.mark.parametrize("values", ["compact", "full"])
def test_team2_two_source_nonfinite_update_rowmap(crp, orc, gpu, monkeypatch, values):
    """Variant 5: two-source column index (B1 = receive buffer), absent pairs never multiplied (an Inf in a B
    row that a panel-mate reads must not leak NaNs into rows that do not have that column), value updates,
    row maps, bit-identical repeats."""
    monkeypatch.setenv("CRPSPMM_TEAM2_COMPACT", "1" if values == "compact" else "0")
    import torch
    from crp_spmm_amd import gen, hip
    lib = crp.load()
    m, k = 500, 900
    rp, ci, va = gen.random_csr(m, k, 30, seed=2)
    lo, hi = 300, 650
    remote_rows = np.concatenate([np.arange(0, lo), np.arange(hi, k)])
    pos = np.full(k, -1)
    pos[remote_rows] = np.arange(remote_rows.size)
    c2 = np.where((ci >= lo) & (ci < hi), ci - lo, ~pos[ci]).astype(np.int32)
    for n_ in (48, 100, 200):                  # the half-piece, one-piece and two-piece instances
        B = np.random.default_rng(n_).normal(size=(k, n_))
        got = _spmm(crp, gpu, rp, c2, va, hi - lo, B[lo:hi], n_, B1=B[remote_rows], variant=5)
        assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), got) <= FP64_TOL, n_
    # non-finite B rows
    for n in (40, 104, 136):
        B = np.random.default_rng(5).normal(size=(k, n))
        used = np.unique(ci)
        B[used[::17]] = np.inf
        B[used[5::29]] = np.nan
        ref = orc.spmm_csr(rp, ci, va, B)
        got = _spmm(crp, gpu, rp, ci, va, k, B, n, ldpad=2 if n == 104 else 0, variant=5)
        assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(np.isinf(ref), np.isinf(got)), n
        fin = np.isfinite(ref)
        assert np.abs(ref[fin] - got[fin]).max() <= 1e-12 * np.abs(ref[fin]).max(), n
    # value update + repeats + row map
    for n in (40, 104, 136):
        _team2_update_and_repeat(crp, orc, gpu, hip, lib, m, k, rp, ci, va, n)


def _team2_update_and_repeat(crp, orc, gpu, hip, lib, m, k, rp, ci, va, n):
    import torch
    A = hip.CsrDev(m, k, rp, ci, va)
    Bf = np.random.default_rng(6).normal(size=(k, n))
    Bd = _t(Bf, gpu)
    Cd = torch.empty((m, n), dtype=torch.float64, device=gpu)
    hip.spmm_csr(A, Bd, Cd, n=n, variant=5)
    torch.cuda.synchronize()
    first = Cd.clone()
    hip.spmm_csr(A, Bd, Cd, n=n, variant=5)
    torch.cuda.synchronize()
    assert torch.equal(first, Cd)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, Bf), Cd.cpu().numpy()) <= FP64_TOL
    v2 = -2.5 * va
    assert lib.crp_csr_dev_update_values(A.handle, v2.ctypes.data, None) == 0
    hip.spmm_csr(A, Bd, Cd, n=n, variant=5)
    torch.cuda.synchronize()
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, v2, Bf), Cd.cpu().numpy()) <= FP64_TOL
    A.free()


def test_locality_order_all_variants(crp, orc, gpu, monkeypatch):
    """Formats built on the locality order of the rows (csrc/locality.cpp; forced with CRPSPMM_REORDER=1): every
    panel / team variant still writes every C row where the caller expects it, with and without a caller row map,
    and value updates reach the re-ordered slots."""
    import ctypes as C
    import torch
    from crp_spmm_amd import gen, hip
    lib = crp.load()
    _IP = C.POINTER(C.c_int)
    rp, ci, va = gen.shell_fem(nc=24, nl=40, m=24 * 40 * 6 - 3, seam_to=30)
    m = len(rp) - 1
    n = 136
    B = np.random.default_rng(9).normal(size=(m, n))
    ref = orc.spmm_csr(rp, ci, va, B)
    monkeypatch.setenv("CRPSPMM_REORDER", "1")
    A = hip.CsrDev(m, m, rp, ci, va)
    assert lib.crp_csr_dev_reordered(A.handle) == 1
    Bd = _t(B, gpu)
    for variant in (0, 1, 2, 3, 5):
        Cd = torch.full((m, n), float("nan"), dtype=torch.float64, device=gpu)
        hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(ref, Cd.cpu().numpy()) <= FP64_TOL, variant
    # caller row map on top (rows scattered into a taller C), then new values
    rowmap = (np.arange(m, dtype=np.int32) * 2 + 1)
    assert lib.crp_csr_dev_set_rowmap(A.handle, rowmap.ctypes.data_as(_IP), 2 * m + 1) == 0
    v2 = 0.5 * va + 1.0
    assert lib.crp_csr_dev_update_values(A.handle, v2.ctypes.data, None) == 0
    ref2 = orc.spmm_csr(rp, ci, v2, B)
    for variant in (1, 3, 5):
        Cd = torch.full((2 * m + 1, n), 7.0, dtype=torch.float64, device=gpu)
        hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
        torch.cuda.synchronize()
        out = Cd.cpu().numpy()
        assert orc.rel_fro_err(ref2, out[1::2]) <= FP64_TOL, variant
        assert (out[0::2] == 7.0).all()
    A.free()
    # auto: a jittered shell mesh of 100 rings takes the locality order, the stride-lattice matrix keeps the caller's
    monkeypatch.delenv("CRPSPMM_REORDER")
    rp2, ci2, va2 = gen.shell_fem(nc=100, nl=100, m=60000, seam_to=80)
    A = hip.CsrDev(60000, 60000, rp2, ci2, va2)
    assert lib.crp_csr_dev_reordered(A.handle) == 1
    A.free()
    rp3, ci3, va3 = gen.banded_fem(6000, offsets=(1, 2, 3, 300, 301, 1800, 1801), seed=2)
    A = hip.CsrDev(6000, 6000, rp3, ci3, va3)
    assert lib.crp_csr_dev_reordered(A.handle) == 0
    A.free()


@pytest.mark.parametrize("n", [1, 8, 16, 23])
def test_locality_order_narrow_fallback_keeps_row_order(crp, orc, gpu, monkeypatch, n):
    """A re-ordered matrix multiplied by fewer than 24 columns falls back to the CSR kernel on the caller's row order:
    the C row map of the derived formats must not be applied to it (round-2 advisor finding, hip_api.hip).  auto and
    forced variants 2/3/4/5, with and without a caller row map."""
    import ctypes as C
    import torch
    from crp_spmm_amd import gen, hip
    lib = crp.load()
    _IP = C.POINTER(C.c_int)
    rp, ci, va = gen.shell_fem(nc=24, nl=40, m=24 * 40 * 6 - 3, seam_to=30)
    m = len(rp) - 1
    B = np.random.default_rng(10 + n).normal(size=(m, n))
    ref = orc.spmm_csr(rp, ci, va, B)
    monkeypatch.setenv("CRPSPMM_REORDER", "1")
    A = hip.CsrDev(m, m, rp, ci, va)
    assert lib.crp_csr_dev_reordered(A.handle) == 1
    Bd = _t(B, gpu)
    for variant in (0, 2, 3, 5, 1):
        Cd = torch.full((m, n), float("nan"), dtype=torch.float64, device=gpu)
        hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(ref, Cd.cpu().numpy()) <= FP64_TOL, variant
    rowmap = (np.arange(m, dtype=np.int32)[::-1] * 2).copy()
    assert lib.crp_csr_dev_set_rowmap(A.handle, rowmap.ctypes.data_as(_IP), 2 * m) == 0
    for variant in (0, 3, 5):
        Cd = torch.full((2 * m, n), 7.0, dtype=torch.float64, device=gpu)
        hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
        torch.cuda.synchronize()
        out = Cd.cpu().numpy()
        assert orc.rel_fro_err(ref, out[rowmap]) <= FP64_TOL, variant
        assert (out[1::2] == 7.0).all()
    A.free()


@pytest.mark.parametrize("reorder", ["0", "1"])
def test_update_values_device_pointer_before_first_product(crp, orc, gpu, monkeypatch, reorder):
    """crp_csr_dev_update_values with a DEVICE pointer before the (lazily built) derived formats exist: every variant's
    first product must see the new values (round-2 advisor finding: the formats were built from the stale host copy)."""
    import torch
    from crp_spmm_amd import gen, hip
    lib = crp.load()
    monkeypatch.setenv("CRPSPMM_REORDER", reorder)
    rp, ci, va = gen.shell_fem(nc=24, nl=40, m=24 * 40 * 6 - 3, seam_to=30)
    m = len(rp) - 1
    n = 136
    B = np.random.default_rng(21).normal(size=(m, n))
    v2 = np.cos(np.arange(len(va))) + 2.0
    ref2 = orc.spmm_csr(rp, ci, v2, B)
    ref1 = orc.spmm_csr(rp, ci, va, B)
    Bd = _t(B, gpu)
    v2d = _t(v2, gpu)
    for variant in (0, 3, 5, 2, 1):
        A = hip.CsrDev(m, m, rp, ci, va)
        assert lib.crp_csr_dev_update_values(A.handle, v2d.data_ptr(), None) == 0
        Cd = torch.full((m, n), float("nan"), dtype=torch.float64, device=gpu)
        hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(ref2, Cd.cpu().numpy()) <= FP64_TOL, variant
        if variant in (0, 5):
            # ... and the fp32 value streams derived afterwards
            Cf = torch.empty((m, n), dtype=torch.float32, device=gpu)
            hip.spmm_csr_f32(A, Bd.to(torch.float32), Cf, n=n, variant=5)
            torch.cuda.synchronize()
            assert orc.rel_fro_err(ref2, Cf.cpu().numpy().astype(np.float64)) <= 1e-5
        # a host update afterwards is authoritative again
        assert lib.crp_csr_dev_update_values(A.handle, va.ctypes.data, None) == 0
        hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(ref1, Cd.cpu().numpy()) <= FP64_TOL, variant
        A.free()


@pytest.mark.parametrize("n", [128, 512])
def test_grid_2x4_rankwise(crp, orc, gpu, n):
    """BASELINE configs[2] (nlpkkt240, n = 128, 2 x 4 grid on 8 GPUs) on the HIP path without eight processes on one
    card: for each of the 8 ranks of a forced 2 x 4 grid on the nlpkkt stand-in kkt3d(20) the test builds what
    para2d_spmm_init leaves on that rank (/root/reference/src/para2d_spmm.c:20-127: the replicated row panel of its grid
    row, its n / 4 columns of B, the B rows it needs from the other grid row in owner-then-row order), assembles the
    receive buffer on the host, runs crp_spmm_csr_f64 with the two-source column index on the GPU and compares the C
    block with the oracle.  n = 128 gives the narrow-operand kernels (n_local = 32), n = 512 the team kernel with a
    second B source (n_local = 128)."""
    import torch
    from crp_spmm_amd import gen, hip, planner
    rp, ci, va = gen.kkt3d(20)
    m = k = len(rp) - 1
    B = np.random.default_rng(5).normal(size=(k, n))
    C_ref = orc.spmm_csr(rp, ci, va, B)
    P, pm, pn = 8, 2, 4
    rb = planner.csr_mat_row_partition(rp, P)
    ac = np.array([rb[i * pn] for i in range(pm + 1)], dtype=np.int64)       # (src/spmat_part.c:169-202 for a forced grid)
    bc = planner.even_displs(n, pn)
    for rank in range(P):
        pi, pj = rank // pn, rank % pn
        lo, hi = int(ac[pi]), int(ac[pi + 1])
        c0, c1 = int(bc[pj]), int(bc[pj + 1])
        prp = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
        pci = ci[rp[lo]:rp[hi]].astype(np.int64)
        pva = np.ascontiguousarray(va[rp[lo]:rp[hi]])
        local = (pci >= lo) & (pci < hi)
        remote = np.unique(pci[~local])                   # ascending global row = by owner, then by row
        assert remote.size > 0                            # the KKT coupling crosses the two grid rows
        two = np.empty(pci.size, dtype=np.int32)
        two[local] = (pci[local] - lo).astype(np.int32)
        two[~local] = ~np.searchsorted(remote, pci[~local]).astype(np.int32)
        B0 = _t(B[lo:hi, c0:c1], gpu)
        B1 = _t(B[remote][:, c0:c1], gpu)                 # the receive buffer, rows in final position
        A = hip.CsrDev(hi - lo, hi - lo, prp, two, pva)
        for variant in (0, 1, 3, 5):
            Cd = torch.full((hi - lo, c1 - c0), float("nan"), dtype=torch.float64, device=gpu)
            hip.spmm_csr(A, B0, Cd, n=c1 - c0, B1=B1, variant=variant)
            torch.cuda.synchronize()
            assert orc.rel_fro_err(C_ref[lo:hi, c0:c1], Cd.cpu().numpy()) <= FP64_TOL, (rank, variant)
        A.free()


FP32_TOL = 1e-5      # fp32 path vs the fp64 oracle: relative Frobenius error (there is no fp32 reference: src/rowpara_spmm.h:28)


@pytest.mark.parametrize("n", [1024, 300, 128, 52, 7])
def test_fp32_path_vs_fp64_oracle(crp, orc, gpu, n):
    """crp_spmm_csr_f32 (BASELINE configs[3]: Queen-class FEM matrix, n = 1024, fp32): the fem3d stand-in and a random
    matrix, every variant (auto, CSR row-group, team kernel), against the fp64 oracle at FP32_TOL; widths that are not
    multiples of 4 take the row-group kernel."""
    import torch
    from crp_spmm_amd import gen, hip
    cases = [("fem3d", gen.fem3d(12)), ("random", gen.random_csr(777, 1234, 70, seed=n, empty_every=13))]
    for name, (rp, ci, va) in cases:
        m = len(rp) - 1
        k = max(int(ci.max()) + 1, m) if name == "fem3d" else 1234
        B = np.random.default_rng(n).uniform(-1, 1, size=(k, n))
        ref = orc.spmm_csr(rp, ci, va, B)
        A = hip.CsrDev(m, k, rp, ci, va)
        Bd = _t(B.astype(np.float32), gpu)
        for variant in (0, 1, 5):
            Cd = torch.full((m, n), float("nan"), dtype=torch.float32, device=gpu)
            hip.spmm_csr_f32(A, Bd, Cd, n=n, variant=variant)
            torch.cuda.synchronize()
            got = Cd.cpu().numpy().astype(np.float64)
            assert orc.rel_fro_err(ref, got) <= FP32_TOL, (name, n, variant, orc.rel_fro_err(ref, got))
        A.free()


def test_fp32_two_source_and_update(crp, orc, gpu):
    """fp32 path: two-source column index through both kernels, and value updates reaching the fp32 copies."""
    import torch
    from crp_spmm_amd import gen, hip
    lib = crp.load()
    m, k = 500, 900
    rp, ci, va = gen.random_csr(m, k, 30, seed=2)
    lo, hi = 300, 650
    remote_rows = np.concatenate([np.arange(0, lo), np.arange(hi, k)])
    pos = np.full(k, -1)
    pos[remote_rows] = np.arange(remote_rows.size)
    c2 = np.where((ci >= lo) & (ci < hi), ci - lo, ~pos[ci]).astype(np.int32)
    n = 200
    B = np.random.default_rng(1).normal(size=(k, n))
    A = hip.CsrDev(m, hi - lo, rp, c2, va)
    B0 = _t(B[lo:hi].astype(np.float32), gpu)
    B1 = _t(B[remote_rows].astype(np.float32), gpu)
    for vals in (va, 0.5 * va - 1.0):
        if vals is not va:
            assert lib.crp_csr_dev_update_values(A.handle, vals.ctypes.data, None) == 0
        ref = orc.spmm_csr(rp, ci, vals, B)
        for variant in (1, 5):
            Cd = torch.full((m, n), float("nan"), dtype=torch.float32, device=gpu)
            hip.spmm_csr_f32(A, B0, Cd, n=n, B1=B1, variant=variant)
            torch.cuda.synchronize()
            assert orc.rel_fro_err(ref, Cd.cpu().numpy().astype(np.float64)) <= FP32_TOL, variant
    A.free()


@pytest.mark.parametrize("n", [24, 26, 30, 32, 48, 64])
def test_narrow_kernel(crp, orc, gpu, monkeypatch, n):
    """The narrow-operand kernel (csrc/narrow_kernel.hip: row-panel format, four entries per instruction; variant 3 at
    24 <= n <= 32, even n, 16-byte aligned operands): random / banded / empty-row matrices with padded leading dimensions,
    the two-source column index (general addressing path), non-finite B rows next to absent pairs, value updates,
    bit-identical repeats, and B rows past 4 GiB (64-bit addressing path)."""
    import torch
    from crp_spmm_amd import gen, hip
    lib = crp.load()
    # (the values of the panels without their holes are taken when under 60 % of the (row, entry) pairs exist: the kkt case below;
    #  n = 48, 64: the two-piece instance, taken by itself only for panels that are mostly holes, forced here)
    if n > 32:
        monkeypatch.setenv("CRPSPMM_NARROW_MAX", "64")
    cases = [gen.random_csr(777, 1234, 70, seed=n, empty_every=13), gen.banded_fem(5000, offsets=(1, 2, 3, 40, 41, 900), seed=n),
             gen.random_csr(13, 40, 5, seed=1), gen.kkt3d(10)]
    for rp, ci, va in cases:
        m = len(rp) - 1
        k = max(int(ci.max()) + 1, 1) if ci.size else 1
        B = np.random.default_rng(n).uniform(-2, 2, size=(k, n))
        ref = orc.spmm_csr(rp, ci, va, B)
        for ldpad in (0, 2, 6):
            got = _spmm(crp, gpu, rp, ci, va, k, B, n, ldpad=ldpad, variant=3)
            assert orc.rel_fro_err(ref, got) <= FP64_TOL, (m, ldpad)
    # two-source column index
    m, k = 500, 900
    rp, ci, va = gen.random_csr(m, k, 30, seed=2)
    lo, hi = 300, 650
    remote_rows = np.concatenate([np.arange(0, lo), np.arange(hi, k)])
    pos = np.full(k, -1)
    pos[remote_rows] = np.arange(remote_rows.size)
    c2 = np.where((ci >= lo) & (ci < hi), ci - lo, ~pos[ci]).astype(np.int32)
    B = np.random.default_rng(n + 1).normal(size=(k, n))
    got = _spmm(crp, gpu, rp, c2, va, hi - lo, B[lo:hi], n, B1=B[remote_rows], variant=3)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), got) <= FP64_TOL
    # non-finite B rows: an Inf that a panel-mate reads must not leak NaNs into rows without that column
    used = np.unique(ci)
    B[used[::17]] = np.inf
    B[used[5::29]] = np.nan
    ref = orc.spmm_csr(rp, ci, va, B)
    got = _spmm(crp, gpu, rp, ci, va, k, B, n, variant=3)
    assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(np.isinf(ref), np.isinf(got))
    fin = np.isfinite(ref)
    assert np.abs(ref[fin] - got[fin]).max() <= 1e-12 * np.abs(ref[fin]).max()
    # repeats are bit-identical; value updates reach the panel format
    A = hip.CsrDev(m, k, rp, ci, va)
    Bf = np.random.default_rng(6).normal(size=(k, n))
    Bd = _t(Bf, gpu)
    Cd = torch.empty((m, n), dtype=torch.float64, device=gpu)
    hip.spmm_csr(A, Bd, Cd, n=n, variant=3)
    torch.cuda.synchronize()
    first = Cd.clone()
    hip.spmm_csr(A, Bd, Cd, n=n, variant=3)
    torch.cuda.synchronize()
    assert torch.equal(first, Cd)
    v2 = -2.5 * va
    assert lib.crp_csr_dev_update_values(A.handle, v2.ctypes.data, None) == 0
    hip.spmm_csr(A, Bd, Cd, n=n, variant=3)
    torch.cuda.synchronize()
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, v2, Bf), Cd.cpu().numpy()) <= FP64_TOL
    A.free()
    if n == 32:
        # B rows addressed past 4 GiB: the 64-bit addressing path
        kb, ld = 1100, 1 << 19
        rpb, cib, vab = gen.random_csr(1500, kb, 40, seed=21)
        cib = cib.copy()
        cib[::5] = kb - 1 - (cib[::5] % 7)
        for r in range(1500):
            cib[rpb[r]:rpb[r + 1]] = np.sort(cib[rpb[r]:rpb[r + 1]])
        Bb = np.random.default_rng(3).uniform(-1, 1, size=(kb, n))
        Bbig = torch.empty((kb, ld), dtype=torch.float64, device=gpu)
        Bbig[:, :n] = _t(Bb, gpu)
        Ab = hip.CsrDev(1500, kb, rpb, cib, vab)
        Cb = torch.full((1500, n), float("nan"), dtype=torch.float64, device=gpu)
        hip.spmm_csr(Ab, Bbig[:, :n], Cb, n=n, variant=3)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(orc.spmm_csr(rpb, cib, vab, Bb), Cb.cpu().numpy()) <= FP64_TOL
        Ab.free()
        del Bbig
        torch.cuda.empty_cache()


@pytest.mark.parametrize("variant", [7])
@pytest.mark.parametrize("n", [24, 30, 32, 34, 48, 64])
def test_team2r_kernel(crp, orc, gpu, n, variant):
    """Variant 7, the team kernel for 24 <= n <= 64 columns (B rows shared through LDS; csrc/team2r_kernel.hip): lane groups own rows,
    a step = every row's next nonzero, padding = 0.0 x a slice of zeros.  Random / banded / lattice / KKT / tiny matrices with
    padded leading dimensions, the two-source column index, non-finite B rows next to absent pairs, value updates (host and device
    pointers), row maps through the locality order, bit-identical repeats."""
    vname = {7: b"team2r-R8"}[variant]
    import torch
    from crp_spmm_amd import gen, hip
    lib = crp.load()
    nx, ny, nz = 300, 6, 5
    cases = [gen.random_csr(777, 1234, 70, seed=n, empty_every=13), gen.banded_fem(5000, offsets=(1, 2, 3, 40, 41, 900), seed=n),
             gen.banded_fem(nx * ny * nz, offsets=(1, 2, 3, 4, 5, nx, nx + 1, nx * ny, nx * ny + 1), seed=4),
             gen.random_csr(13, 40, 5, seed=1), gen.kkt3d(10), gen.random_csr(64, 40, 36, seed=7)]
    for rp, ci, va in cases:
        m = len(rp) - 1
        k = max(int(ci.max()) + 1, 1) if ci.size else 1
        B = np.random.default_rng(n).uniform(-2, 2, size=(k, n))
        ref = orc.spmm_csr(rp, ci, va, B)
        for ldpad in (0, 2, 6):
            got = _spmm(crp, gpu, rp, ci, va, k, B, n, ldpad=ldpad, variant=variant)
            assert orc.rel_fro_err(ref, got) <= FP64_TOL, (m, ldpad)
    # the variant really ran
    m, k = 500, 900
    rp, ci, va = gen.random_csr(m, k, 30, seed=2)
    A = hip.CsrDev(m, k, rp, ci, va)
    Bf = np.random.default_rng(6).normal(size=(k, n))
    Bd = _t(Bf, gpu)
    Cd = torch.empty((m, n), dtype=torch.float64, device=gpu)
    hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
    torch.cuda.synchronize()
    assert lib.crp_spmm_variant_name(lib.crp_csr_dev_last_variant(A.handle)) == vname
    # repeats are bit-identical; value updates (host pointer, then device pointer) reach the streams
    first = Cd.clone()
    hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
    torch.cuda.synchronize()
    assert torch.equal(first, Cd)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, Bf), Cd.cpu().numpy()) <= FP64_TOL
    v2 = -2.5 * va
    assert lib.crp_csr_dev_update_values(A.handle, v2.ctypes.data, None) == 0
    hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
    torch.cuda.synchronize()
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, v2, Bf), Cd.cpu().numpy()) <= FP64_TOL
    v3 = _t(0.5 * va, gpu)
    assert lib.crp_csr_dev_update_values(A.handle, v3.data_ptr(), None) == 0
    hip.spmm_csr(A, Bd, Cd, n=n, variant=variant)
    torch.cuda.synchronize()
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, 0.5 * va, Bf), Cd.cpu().numpy()) <= FP64_TOL
    A.free()
    # two-source column index
    lo, hi = 300, 650
    remote_rows = np.concatenate([np.arange(0, lo), np.arange(hi, k)])
    pos = np.full(k, -1)
    pos[remote_rows] = np.arange(remote_rows.size)
    c2 = np.where((ci >= lo) & (ci < hi), ci - lo, ~pos[ci]).astype(np.int32)
    B = np.random.default_rng(n + 1).normal(size=(k, n))
    got = _spmm(crp, gpu, rp, c2, va, hi - lo, B[lo:hi], n, B1=B[remote_rows], variant=variant)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), got) <= FP64_TOL
    # non-finite B rows: an Inf that a team-mate reads must not leak NaNs into rows without that column
    used = np.unique(ci)
    B[used[::17]] = np.inf
    B[used[5::29]] = np.nan
    ref = orc.spmm_csr(rp, ci, va, B)
    got = _spmm(crp, gpu, rp, ci, va, k, B, n, variant=variant)
    assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(np.isinf(ref), np.isinf(got))
    fin = np.isfinite(ref)
    assert np.abs(ref[fin] - got[fin]).max() <= 1e-12 * np.abs(ref[fin]).max()
    if variant == 7:
        # what variant 0 takes for panels that are mostly holes (a KKT system: 1.9 of 8 rows per panel entry) at these widths
        rp, ci, va = gen.kkt3d(14)
        m = len(rp) - 1
        A = hip.CsrDev(m, m, rp, ci, va)
        # (up to 32 columns; above, the half-piece instances of the team kernel, variant 5, since round 4)
        want = 7 if n <= 32 else 5
        assert lib.crp_csr_dev_resolved_variant(A.handle, n) == want
        Bk = np.random.default_rng(n + 3).normal(size=(m, n))
        Cd = torch.empty((m, n), dtype=torch.float64, device=gpu)
        hip.spmm_csr(A, _t(Bk, gpu), Cd, n=n, variant=0)
        torch.cuda.synchronize()
        assert lib.crp_csr_dev_last_variant(A.handle) == want
        assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, Bk), Cd.cpu().numpy()) <= FP64_TOL
        A.free()
        # ... and not for filled panels
        rp, ci, va = gen.banded_fem(6000, offsets=(1, 2, 3, 4, 5, 6, 40, 41, 42), seed=3)
        A = hip.CsrDev(6000, 6000, rp, ci, va)
        assert lib.crp_csr_dev_resolved_variant(A.handle, n) != 7
        A.free()
    # a square matrix that the locality order re-orders (row map of the formats), and a caller row map on top
    rp, ci, va = gen.fem3d(12)
    m = len(rp) - 1
    B = np.random.default_rng(n + 2).normal(size=(m, n))
    got = _spmm(crp, gpu, rp, ci, va, m, B, n, variant=variant)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), got) <= FP64_TOL


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_comm_size_on_device_vs_host_planner(crp, orc, gpu, name):
    """csr_mat_row_part_comm_size (src/spmat_part.c:38-64) on the device-resident CSR (crp_csr_dev_row_part_comm_size: one bitmap
    per block) against the host function -- itself bit-exact against the reference build -- and the oracle, for the
    planner's nnz-balanced row blocks and for ragged ones (empty blocks, x partition different from the row partition); integers:
    exact."""
    from crp_spmm_amd import gen, hip, planner
    g = load_golden(name, "csr")
    cases = [(int(g["m"]), int(g["k"]), g["rowptr"], g["colidx"], g["val"])]
    rp, ci, va = gen.banded_fem(20011, offsets=(1, 2, 3, 40, 41, 900, 5000), seed=7)
    cases.append((20011, 20011, rp, ci, va))
    rp, ci, va = gen.random_csr(3000, 7001, 23, seed=11, empty_every=7)
    cases.append((3000, 7001, rp, ci, va))
    for m, k, rp, ci, va in cases:
        A = hip.CsrDev(m, k, rp, ci, va)
        for P in (1, 2, 3, 4, 6, 8, 13):
            rb = planner.csr_mat_row_partition(rp, P)
            xd = rb if m == k else planner.even_displs(k, P)
            ref_sizes, ref_tot = planner.csr_mat_row_part_comm_size(k, rp, ci, rb, xd)
            got_sizes, got_tot = A.row_part_comm_size(rb, xd)
            assert np.array_equal(ref_sizes, got_sizes) and ref_tot == got_tot, (name, m, P)
            o_sizes, o_tot = orc.csr_row_part_comm_size(k, rp, ci, rb, xd)
            assert np.array_equal(np.asarray(o_sizes), got_sizes) and int(o_tot) == got_tot, (name, m, P, "oracle")
        # ragged: empty blocks at both ends, an x partition unrelated to the rows
        rb = np.array([0, 0, m // 3, m // 3, m, m], dtype=np.int32)
        xd = np.array([0, k // 5, k // 5, k // 2, k - 1, k], dtype=np.int32)
        ref_sizes, ref_tot = planner.csr_mat_row_part_comm_size(k, rp, ci, rb, xd)
        got_sizes, got_tot = A.row_part_comm_size(rb, xd)
        assert np.array_equal(ref_sizes, got_sizes) and ref_tot == got_tot, (name, m, "ragged")
        A.free()


def test_create_with_device_values(crp, orc, gpu):
    """crp_csr_dev_create_dv: a matrix whose values are in device memory already (the panel a device all-gather replicated,
    src/para2d_spmm.c:56-86): the whole matrix (values in order) and a row subset (src_start per row), through the CSR kernel, the
    row-panel kernels and the team kernel (derived formats are built from the host values; both copies agree here by construction),
    against the oracle."""
    import ctypes as C
    import torch
    from crp_spmm_amd import gen
    lib = crp.load()
    _IP, _DP = C.POINTER(C.c_int), C.POINTER(C.c_double)
    rp, ci, va = gen.banded_fem(6000, offsets=(1, 2, 3, 40, 41, 900), seed=9)
    m = k = 6000
    n = 128
    B = np.random.default_rng(3).uniform(-1, 1, size=(k, n))
    ref = orc.spmm_csr(rp, ci, va, B)
    va_dev = torch.from_numpy(va).to(gpu)
    Bd = _t(B, gpu)
    h = C.c_void_p()
    assert lib.crp_csr_dev_create_dv(m, k, rp.ctypes.data_as(_IP), ci.ctypes.data_as(_IP), va.ctypes.data_as(_DP), C.c_void_p(va_dev.data_ptr()), None,
                                     C.byref(h)) == 0
    for variant in (1, 3, 5):
        Cd = torch.full((m, n), float("nan"), dtype=torch.float64, device=gpu)
        assert lib.crp_spmm_csr_f64(h, 0, n, C.c_void_p(Bd.data_ptr()), n, None, 0, C.c_void_p(Cd.data_ptr()), n, variant, None) == 0
        torch.cuda.synchronize()
        assert orc.rel_fro_err(ref, Cd.cpu().numpy()) <= FP64_TOL, variant
    lib.crp_csr_dev_destroy(C.byref(h))
    # a row subset: every third row, values gathered on the device from the full matrix's values
    rows = np.arange(0, m, 3, dtype=np.int32)
    cnt = (rp[rows + 1] - rp[rows]).astype(np.int64)
    sub_rp = np.zeros(rows.size + 1, dtype=np.int32)
    sub_rp[1:] = np.cumsum(cnt)
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
    sub_ci, sub_va = np.ascontiguousarray(ci[idx]), np.ascontiguousarray(va[idx])
    start = np.ascontiguousarray(rp[rows], dtype=np.int32)
    assert lib.crp_csr_dev_create_dv(rows.size, k, sub_rp.ctypes.data_as(_IP), sub_ci.ctypes.data_as(_IP), sub_va.ctypes.data_as(_DP),
                                     C.c_void_p(va_dev.data_ptr()), start.ctypes.data_as(_IP), C.byref(h)) == 0
    Cd = torch.full((rows.size, n), float("nan"), dtype=torch.float64, device=gpu)
    assert lib.crp_spmm_csr_f64(h, 0, n, C.c_void_p(Bd.data_ptr()), n, None, 0, C.c_void_p(Cd.data_ptr()), n, 1, None) == 0
    torch.cuda.synchronize()
    assert orc.rel_fro_err(ref[rows], Cd.cpu().numpy()) <= FP64_TOL
    assert lib.crp_csr_dev_create_dv(m, k, rp.ctypes.data_as(_IP), ci.ctypes.data_as(_IP), va.ctypes.data_as(_DP), None, None, C.byref(h)) == -1
    lib.crp_csr_dev_destroy(C.byref(h))
