"""CPU: host logic of the product library (planner, ingest, exchange plan, C ABI
surface) against the oracle and the golden fixtures.  No device calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, GOLDEN_NAMES, ROOT, load_golden

PS = (1, 2, 3, 4, 6, 8)
NS = (1, 4, 64, 128, 512)


def test_library_exports_every_declared_symbol(crp):
    """Every function declared in include/*.h is exported by the C-ABI library and
    bound in _lib.SIGNATURES (MPI-typed facade headers are checked against libcrpspmm.so)."""
    from crp_spmm_amd import _lib
    inc = os.path.join(ROOT, "include")
    core = ["crpspmm_hip.h", "crp_comm.h", "crp_engine.h", "utils.h", "spmat_part.h", "mmio_utils.h", "dev_type.h", "crp_rccl.h"]
    pat = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\(", re.M)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    declared = set()
    for h in core:
        txt = open(os.path.join(inc, h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"#define[^\n]*(\\\n[^\n]*)*", "", txt)
        body = txt.replace("\n", " ")
        for stmt in body.split(";"):
            mm = re.match(r"\s*(?:extern\s+\"C\"\s*\{)?\s*((?:const\s+)?(?:unsigned\s+)?[A-Za-z_]\w*(?:\s+[A-Za-z_]\w*)*[\s\*]+)(\w+)\s*\(", stmt)
            if mm and "(*" not in stmt.split("(")[0] and mm.group(2) not in ("defined",) and "typedef" not in stmt.split("(")[0]:
                declared.add(mm.group(2))
    declared = {d for d in declared if not d.startswith("__")}
    assert len(declared) > 60, sorted(declared)
    missing = sorted(d for d in declared if d not in exported)
    assert not missing, "declared but not exported: %s" % missing
    unbound = sorted(d for d in declared if d not in _lib.SIGNATURES)
    assert not unbound, "declared but not bound in _lib.SIGNATURES: %s" % unbound


def test_mpi_facade_exports_reference_api(crp):
    """libcrpspmm.so (built when mpi.h is present) exports the reference's exact entry points
    (src/rowpara_spmm.h:60-87, src/para2d_spmm.h:42-75, src/mat_redist.h:69-100,
    deprecated/src/crpspmm.h:89-130)."""
    from crp_spmm_amd import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcrpspmm.so")
    if not os.path.exists(path):
        pytest.skip("no MPI on this machine: facade not built")
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    for fn in ("rp_spmm_init", "rp_spmm_free", "rp_spmm_exec", "rp_spmm_print_stat", "rp_spmm_clear_stat",
               "para2d_spmm_init", "para2d_spmm_free", "para2d_spmm_exec", "para2d_spmm_print_stat",
               "para2d_spmm_clear_stat", "mat_redist_engine_init", "mat_redist_engine_attach_workbuf",
               "mat_redist_engine_exec", "mat_redist_engine_free",
               # deprecated/src/crpspmm.h:89-130
               "crpspmm_engine_init", "crpspmm_engine_attach_workbuf", "crpspmm_engine_exec", "crpspmm_engine_free",
               "crpspmm_engine_print_stat", "crpspmm_engine_clear_stat",
               # include/crp_mpi.h
               "crp_mpi_comm_wrap", "crp_mpi_comm_uses_rccl"):
        assert fn in exported, fn


def test_missing_library_fails_loudly(crp, monkeypatch):
    from crp_spmm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcrpspmm_hip.so")
    with pytest.raises(_lib.CrpLibraryError):
        _lib.load()


def test_product_does_not_reference_oracle():
    """The product tree never imports / links the oracle."""
    pkg = os.path.join(ROOT, "crp-spmm_amd")
    for base, _dirs, files in os.walk(pkg):
        if os.sep + "build" in base or os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "import oracle" not in txt and "liborc" not in txt and "from oracle" not in txt, f


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_ingest_vs_golden(crp, name):
    from crp_spmm_amd import mmio
    g = load_golden(name, "csr")
    m, k, rp, ci, cv = mmio.read_mtx_csr(os.path.join(GOLDEN, name + ".mtx"), verbose=False)
    assert (m, k) == (int(g["m"]), int(g["k"]))
    assert np.array_equal(rp, g["rowptr"]) and np.array_equal(ci, g["colidx"])
    # values: identical except the order of exact duplicates (same row, same column), which the
    # reference's unstable quicksort leaves unspecified -> compare per-(row, col) multisets
    rows = np.repeat(np.arange(m), np.diff(rp))
    a = np.lexsort((cv, ci, rows))
    b = np.lexsort((g["val"], g["colidx"], rows))
    assert np.array_equal(cv[a], g["val"][b])


def test_ingest_rejects_what_the_reference_rejects(crp, tmp_path, orc):
    from crp_spmm_amd import mmio
    cases = {
        "complex.mtx": "%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 0.0\n",
        "skew.mtx": "%%MatrixMarket matrix coordinate real skew-symmetric\n2 2 1\n2 1 1.0\n",
        "herm.mtx": "%%MatrixMarket matrix coordinate real hermitian\n2 2 1\n2 1 1.0\n",
        "array.mtx": "%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n",
        "nobanner.mtx": "2 2 1\n1 1 1.0\n",
        "badtype.mtx": "%%MatrixMarket matrix coordinate quaternion general\n2 2 1\n1 1 1.0\n",
    }
    for fn, txt in cases.items():
        p = tmp_path / fn
        p.write_text(txt)
        assert mmio.mm_read_sparse_RPI(str(p))[0] == -1, fn
        assert orc.mm_read(str(p))[0] == -1, fn
        if orc.ref() is not None:
            assert orc.ref_mm_read(str(p))[0] == -1, fn
    assert mmio.mm_read_sparse_RPI(str(tmp_path / "does_not_exist.mtx"))[0] == -1
    # need_symm on a general file (METIS path of the drivers, examples/test_rp_spmm.c:21-22)
    assert mmio.mm_read_sparse_RPI(os.path.join(GOLDEN, "g_gen.mtx"), need_symm=1)[0] == -1
    # accepted: upper-case banner tokens, comment lines, blank line before the size line
    ok = tmp_path / "ok.mtx"
    ok.write_text("%%MatrixMarket MATRIX Coordinate REAL Symmetric\n% c1\n% c2\n\n3 3 2\n2 1 1.5\n3 3 -2\n")
    st, m, k, r, c, v = mmio.mm_read_sparse_RPI(str(ok))
    assert st == 0 and (m, k) == (3, 3)
    assert list(r) == [1, 2, 0] and list(c) == [0, 2, 1] and list(v) == [1.5, -2.0, 1.5]
    assert [list(x) for x in orc.mm_read(str(ok))[3:]] == [[1, 2, 0], [0, 2, 1], [1.5, -2.0, 1.5]]


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_planner_vs_golden(crp, name):
    from crp_spmm_amd import planner
    g, p = load_golden(name, "csr"), load_golden(name, "plan")
    m, k = int(g["m"]), int(g["k"])
    for P in PS:
        rb = planner.csr_mat_row_partition(g["rowptr"], P)
        assert np.array_equal(rb, p["rb_P%d" % P])
        for n in NS:
            r = planner.calc_spmm_part2d_from_1d(P, m, n, k, rb, g["rowptr"], g["colidx"])
            key = "P%d_n%d_" % (P, n)
            assert [r["pm"], r["pn"]] == list(p[key + "grid"]) and r["comm_cost"] == int(p[key + "cost"][0])
            for a in ("A0_rowptr", "B_rowptr", "AC_rowptr", "BC_colptr"):
                assert np.array_equal(r[a], p[key + a]), (P, n, a)


def test_planner_vs_oracle_random(crp, orc):
    from crp_spmm_amd import gen, planner
    for seed, (m, k) in enumerate([(2000, 2000), (700, 1900), (1900, 700), (50, 50)]):
        rp, ci, va = gen.random_csr(m, k, 14, seed=seed + 20, empty_every=11 if seed % 2 else 0)
        for P in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
            a = planner.csr_mat_row_partition(rp, P)
            assert np.array_equal(a, orc.csr_row_partition(rp, P))
            x = a if m == k else planner.even_displs(k, P)
            s1, t1 = planner.csr_mat_row_part_comm_size(k, rp, ci, a, x)
            s2, t2 = orc.csr_row_part_comm_size(k, rp, ci, a, x)
            assert np.array_equal(s1, s2) and t1 == t2
            for n in (1, 8, 100, 1024):
                for rA in (1, 3):
                    r1 = planner.calc_spmm_part2d_from_1d(P, m, n, k, a, rp, ci, rA=rA)
                    r2 = orc.part2d_from_1d(P, m, n, k, a, rp, ci, rA=rA)
                    for key in r1:
                        assert np.array_equal(np.asarray(r1[key]), np.asarray(r2[key])), (P, n, key)
    assert planner.prime_factorization(360) == [2, 2, 2, 3, 3, 5] and planner.prime_factorization(1) == []
    assert planner.prime_factorization(97) == [97]
    for length, nblk in [(10, 3), (7, 7), (5, 8), (0, 2)]:
        for i in range(-1, nblk + 2):
            assert planner.calc_block_spos_size(length, nblk, i) == orc.block_spos(length, nblk, i)


def test_utils_abi(crp, orc):
    lib = crp.load()
    x0 = np.linspace(-1, 2, 1001)
    x1 = x0 + 1e-9 * np.cos(np.arange(1001))
    a, e = C.c_double(), C.c_double()
    lib.calc_err_2norm(1001, x0.ctypes.data_as(C.POINTER(C.c_double)), x1.ctypes.data_as(C.POINTER(C.c_double)),
                       C.byref(a), C.byref(e))
    assert (a.value, e.value) == orc.err_2norm(x0, x1)
    assert lib.calc_2norm(1001, x0.ctypes.data_as(C.POINTER(C.c_double))) == a.value
    src = np.arange(35, dtype=np.float64).reshape(5, 7)
    dst = np.zeros((5, 9))
    for omp in (0, 1):
        dst[:] = 0
        lib.copy_matrix(8, 5, 4, src.ctypes.data, 7, dst.ctypes.data, 9, omp)
        assert np.array_equal(dst[:, :4], src[:, :4]) and not dst[:, 4:].any()
    t0 = lib.get_wtime_sec()
    assert lib.get_wtime_sec() >= t0 > 1e9
    p = lib.malloc_aligned(100, 64)
    assert p % 64 == 0
    lib.free_aligned(p)


def test_rp_plan_single_rank_vs_oracle(crp, orc):
    from crp_spmm_amd import comm, engine, gen
    sc = comm.SelfComm()
    for seed, (m, k) in enumerate([(800, 800), (300, 900)]):
        rp, ci, va = gen.random_csr(m, k, 12, seed=seed + 3, empty_every=17)
        e = engine.RpSpmm(0, m, rp, ci, va, [0, k], 8, sc, plan_only=True)
        p = e.plan()
        o = orc.rp_plan_all([(rp, ci, va)], [0, k], 8)[0]
        for key in ("A_rowptr", "A_colidx", "A_val", "rB_nrow", "rB_self_nrow", "rB_self_src_offset",
                    "rB_self_dst_offset", "rB_self_src_ridxs", "rB_sridxs", "rB_rridxs", "rB_rcnts", "rB_scnts",
                    "rB_rdispls", "rB_sdispls", "rB_recv_size"):
            assert np.array_equal(np.asarray(p[key]), np.asarray(o[key])), key
        assert np.array_equal(p["dev_colidx"], ci)         # one rank: every column is a local B row
        assert e.alg_bytes() == 12 * ci.size + 4 * (m + 1) + 8 * 8 * np.unique(ci).size + 8 * 8 * m
        e.free()
    # empty matrix / empty rank
    e = engine.RpSpmm(0, 5, np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0), [0, 9], 4, sc, plan_only=True)
    p = e.plan()
    assert p["rB_nrow"] == 0 and p["rB_self_nrow"] == 0 and p["A_colidx"].size == 0
    e.free()
    sc.free()


def test_rp_plan_env_knobs(crp, orc, monkeypatch, capfd):
    from crp_spmm_amd import comm, engine, gen
    rp, ci, va = gen.random_csr(200, 200, 6, seed=1)
    ci = ci.copy()
    sel = ci < 20
    ci[sel] += 20                                 # leave a hole at the low end so reidx=0 differs
    order = np.concatenate([np.sort(ci[rp[i]:rp[i + 1]]) for i in range(200)])
    sc = comm.SelfComm()
    monkeypatch.setenv("RP_SPMM_REIDX", "0")
    monkeypatch.setenv("RP_SPMM_P2P", "0")
    e = engine.RpSpmm(0, 200, rp, order, va, [0, 200], 4, sc, plan_only=True)
    p = e.plan()
    o = orc.rp_plan_all([(rp, order, va)], [0, 200], 4, reidx=0)[0]
    assert p["rB_reidx"] == 0 and p["rB_p2p"] == 0
    for key in ("A_colidx", "rB_nrow", "rB_self_dst_offset", "rB_self_src_offset", "rB_self_nrow"):
        assert np.array_equal(np.asarray(p[key]), np.asarray(o[key])), key
    out = capfd.readouterr().out
    assert "Overriding parameter rB_reidx: 1 (default) --> 0 (runtime)" in out     # src/utils.h:79-83
    assert "Overriding parameter rB_p2p: 1 (default) --> 0 (runtime)" in out
    e.free()
    monkeypatch.setenv("RP_SPMM_REIDX", "7")      # out of range -> default, no message
    e = engine.RpSpmm(0, 200, rp, order, va, [0, 200], 4, sc, plan_only=True)
    assert e.plan()["rB_reidx"] == 1
    e.free()
    sc.free()


@pytest.mark.parametrize("R", [4, 8])
def test_panel_format_host(crp, orc, R):
    """The row-panel format is a lossless regrouping of the CSR: expanding it entry by entry
    (mask-selected rows only, in entry order) reproduces A * B, including duplicates,
    explicit zeros, empty rows / panels and the two-source column encoding."""
    from crp_spmm_amd import gen, hip
    for seed, (m, k) in enumerate([(103, 90), (800, 800), (37, 500)]):
        rp, ci, va = gen.random_csr(m, k, 9, seed=seed, empty_every=5 if seed else 0)
        ci, va = ci.copy(), va.copy()
        if seed == 1:
            rp2, ci2, va2 = gen.banded_fem(m, offsets=(1, 2, 3, 10), seed=2)
            rp, ci, va = rp2, ci2.copy(), va2.copy()
            va[::7] = 0.0                                     # explicit zeros stay entries
        if seed == 0:                                          # duplicates: repeat a column inside some rows
            for r in range(0, m, 3):
                if rp[r + 1] - rp[r] >= 2:
                    ci[rp[r] + 1] = ci[rp[r]]
        if seed == 2:                                          # two-source encoding: columns >= 300 come from B1
            ci = np.where(ci >= 300, ~(ci - 300), ci).astype(np.int32)
        f = hip.panel_format_host(rp, ci, va, R)
        assert f["npanel"] == (m + R - 1) // R and (f["pptr"] % 8 == 0).all()
        n = 3
        B0 = np.random.default_rng(seed).normal(size=(k, n))
        B1 = np.random.default_rng(seed + 9).normal(size=(k, n))
        stacked = np.vstack([B0, B1])
        cc = np.where(ci >= 0, ci, k + (~ci))
        ref = orc.spmm_csr(rp, cc.astype(np.int32), va, stacked)
        got = np.zeros((f["npanel"] * R, n))
        nnz_seen = 0
        for p in range(f["npanel"]):
            for q in range(f["pptr"][p], f["pptr"][p + 1]):
                mask = (int(f["pmask4"][q >> 2]) >> (8 * (q & 3))) & 0xFF
                c = int(f["pcol"][q])
                brow = B0[c] if c >= 0 else B1[~c]
                for r in range(R):
                    if mask >> r & 1:
                        got[p * R + r] += f["pval"][q, r] * brow
                        nnz_seen += 1
                    else:
                        assert f["pval"][q, r] == 0.0
        assert nnz_seen == rp[-1]
        assert not got[m:].any()
        assert orc.rel_fro_err(ref, got[:m]) <= 1e-14


def test_panel_locality_order(crp, monkeypatch):
    """The processing order is a permutation made of whole groups of consecutive panels; a narrow
    band reproduces the natural order; for a 3D-mesh-like matrix (bands at +-1, +-nx, +-nx*ny) the
    panels that share B rows across the far bands end up close together."""
    from crp_spmm_amd import gen, hip
    rp, ci, va = gen.banded_fem(4000, offsets=(1, 2), seed=1)
    f = hip.panel_format_host(rp, ci, va, 4)
    assert np.array_equal(f["porder"], np.arange(f["npanel"]))
    nx, ny, nz = 64, 8, 6
    m = nx * ny * nz
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, nx, nx * ny), seed=2)
    monkeypatch.setenv("CRPSPMM_PANEL_ORDER", "1")               # breadth-first groups of 16 consecutive panels
    f = hip.panel_format_host(rp, ci, va, 4)
    po = f["porder"]
    assert np.array_equal(np.sort(po), np.arange(f["npanel"]))
    assert np.array_equal(po.reshape(-1, 16)[:, 1], po.reshape(-1, 16)[:, 0] + 1)    # groups stay whole
    pos = np.empty_like(po)
    pos[po] = np.arange(po.size)
    far = (nx * ny) // 4                                   # panels one z-plane apart
    d_nat = far
    d_ord = np.median(np.abs(pos[far:] - pos[:-far]))
    assert d_ord < 0.75 * d_nat, (d_ord, d_nat)    # (small mesh: BFS levels are wide; pwtk-size meshes gain far more)
    monkeypatch.setenv("CRPSPMM_PANEL_ORDER", "0")
    assert np.array_equal(hip.panel_format_host(rp, ci, va, 4)["porder"], np.arange(f["npanel"]))


def test_panel_stride_lattice_order(crp, monkeypatch):
    """Two nested far strides (3D mesh in natural order): the order is a permutation in which every
    XCD block of positions holds neighbouring teeth swept in lockstep -- panels one stride apart sit
    within a few positions of each other instead of thousands; matrices without that shape keep the
    breadth-first / natural order."""
    from crp_spmm_amd import gen, hip
    nx, ny, nz = 600, 32, 5                                  # strides 600 and 19200 rows
    m = nx * ny * nz
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, nx, nx + 1, nx * ny, nx * ny + 1), seed=3)
    R = 8
    monkeypatch.setenv("CRPSPMM_PANEL_ORDER", "2")            # single panels on the lattice (what R = 4 gets by default)
    po = hip.panel_format_host(rp, ci, va, R)["porder"]
    monkeypatch.delenv("CRPSPMM_PANEL_ORDER")
    npanel = (m + R - 1) // R
    assert np.array_equal(np.sort(po), np.arange(npanel))
    pos = np.empty_like(po)
    pos[po] = np.arange(po.size)
    chunk = ((((npanel + 3) // 4) + 7) // 8) * 4
    for stride in (nx, nx * ny):
        sp_ = stride // R
        a, b = pos[:-sp_], pos[sp_:]
        same_block = (a // chunk) == (b // chunk)
        assert same_block.mean() > 0.6, (stride, same_block.mean())          # tooth-mates share an XCD ...
        assert np.median(np.abs(a - b)[same_block]) <= 2 * (ny * nz), stride  # ... and run together
    monkeypatch.setenv("CRPSPMM_PANEL_ORDER", "1")
    assert not np.array_equal(hip.panel_format_host(rp, ci, va, R)["porder"], po)
    monkeypatch.delenv("CRPSPMM_PANEL_ORDER")
    # one stride only, and no structure at all: not a lattice
    rp1, ci1, va1 = gen.banded_fem(m, offsets=(1, 2, 3, nx), seed=3)
    monkeypatch.setenv("CRPSPMM_PANEL_ORDER", "2")
    assert np.array_equal(hip.panel_format_host(rp1, ci1, va1, R)["porder"], np.arange(npanel))
    rp2, ci2, va2 = gen.erdos_renyi(8192, 8192, 12, seed=4)
    assert np.array_equal(hip.panel_format_host(rp2, ci2, va2, R)["porder"], np.arange(1024))


def _expand_panels(f, R, B0, B1):
    """C rows computed straight from a row-panel format (mask-selected rows, entry order)."""
    got = np.zeros((f["npanel"] * R, B0.shape[1]))
    for p in range(f["npanel"]):
        for q in range(f["pptr"][p], f["pptr"][p + 1]):
            mask = (int(f["pmask4"][q >> 2]) >> (8 * (q & 3))) & 0xFF
            c = int(f["pcol"][q])
            brow = B0[c] if c >= 0 else B1[~c]
            for r in range(R):
                if mask >> r & 1:
                    got[p * R + r] += f["pval"][q, r] * brow
    return got


def test_team_schedule_and_team_format(crp, orc, monkeypatch):
    """Stride-lattice matrix, R = 8 (the default there): workgroups of four tooth-mate panels.
    The processing order has 4 positions per team (-1 = none) and covers every panel once; a
    team's panels sit one stride apart; the re-ordered entries still expand to A * B; every wave
    meets a column it shares with a team-mate after (nearly) the same number of its own entries.
    The team format lists every panel entry exactly once per wave, in that same order."""
    from crp_spmm_amd import gen, hip
    nx, ny, nz = 600, 8, 4
    m = nx * ny * nz
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, nx, nx + 1, nx * ny, nx * ny + 1), seed=3)
    R = 8
    npanel = (m + R - 1) // R
    monkeypatch.delenv("CRPSPMM_PANEL_ORDER", raising=False)
    # workgroups of four (2 x 2 teeth)
    f = hip.panel_format_host(rp, ci, va, R)
    po = f["porder"]
    assert po.size % 4 == 0 and po.size >= npanel
    assert np.array_equal(np.sort(po[po >= 0]), np.arange(npanel))
    teams = po.reshape(-1, 4)
    full = teams[(teams >= 0).all(axis=1)]
    assert full.shape[0] > 0.8 * teams.shape[0]
    d1, d2 = nx / R, nx * ny / R
    assert np.abs(np.median(full[:, 1] - full[:, 0]) - d1) <= 2 and np.abs(np.median(full[:, 2] - full[:, 0]) - d2) <= 2
    n = 2
    B0 = np.random.default_rng(1).normal(size=(m, n))
    ref = orc.spmm_csr(rp, ci, va, B0)
    assert orc.rel_fro_err(ref, _expand_panels(f, R, B0, B0)[:m]) <= 1e-14
    # shared columns are met together: own-entry index of a shared column differs by a few entries at most
    worst = []
    for t in full[:200]:
        pos = {}
        for w, panel in enumerate(t):
            cols = f["pcol"][f["pptr"][panel]:f["pptr"][panel + 1]]
            msk = [(int(f["pmask4"][q >> 2]) >> (8 * (q & 3))) & 0xFF for q in range(f["pptr"][panel], f["pptr"][panel + 1])]
            for k_, (c, mk) in enumerate(zip(cols, msk)):
                if mk:
                    pos.setdefault(int(c), []).append(k_)
        worst += [max(v) - min(v) for v in pos.values() if len(v) > 1]
    assert len(worst) > 0 and np.percentile(worst, 95) <= 6, np.percentile(worst, [50, 95, 100])
    # natural-order format for comparison: same matrix, CRPSPMM_PANEL_ORDER=0
    monkeypatch.setenv("CRPSPMM_PANEL_ORDER", "0")
    f0 = hip.panel_format_host(rp, ci, va, R)
    assert np.array_equal(f0["porder"], np.arange(npanel))
    monkeypatch.delenv("CRPSPMM_PANEL_ORDER")

    t = hip.team_format_host(rp, ci, va)
    assert t["lattice"] and np.array_equal(np.sort(t["tpanel"][t["tpanel"] >= 0]), np.arange(npanel))
    assert (t["tptr"] % 8 == 0).all() and np.array_equal(np.sort(t["torder"]), np.arange(t["nteam"]))
    for g in range(0, t["nteam"], 37):
        for w in range(4):
            panel = t["tpanel"][g, w]
            sl = slice(t["tptr"][g], t["tptr"][g + 1])
            mine = ((t["tmask"][sl] >> (8 * w)) & 0xFF) != 0
            if panel < 0:
                assert not mine.any()
                continue
            q0, q1 = f0["pptr"][panel], f0["pptr"][panel + 1]
            real = np.array([(int(f0["pmask4"][q >> 2]) >> (8 * (q & 3))) & 0xFF for q in range(q0, q1)]) != 0
            assert sorted(t["tcol"][sl][mine].tolist()) == sorted(f0["pcol"][q0:q1][real].tolist()), (g, w)
            # the team-scheduled panel format holds the wave's entries in exactly this order
            assert t["tcol"][sl][mine].tolist() == f["pcol"][f["pptr"][panel]:f["pptr"][panel + 1]][:mine.sum()].tolist()
    # the outer stride may be a group of clusters (27-point stencil: nx*ny - nx, nx*ny, nx*ny + nx); one stride
    # alone, or a third one further out, is not a lattice
    gx, gy, gz = 300, 10, 6
    mm = gx * gy * gz
    o27 = (1, 2, gx - 1, gx, gx + 1, gx * gy - gx, gx * gy - gx + 1, gx * gy - 1, gx * gy, gx * gy + 1, gx * gy + gx - 1, gx * gy + gx)
    assert hip.team_format_host(*gen.banded_fem(mm, offsets=o27, seed=1))["lattice"]
    assert not hip.team_format_host(*gen.banded_fem(mm, offsets=(1, 2, gx), seed=1))["lattice"]
    assert not hip.team_format_host(*gen.banded_fem(mm, offsets=(1, 2, gx, gx * gy, 3 * gx * gy + 7), seed=1))["lattice"]
    # matrices without a lattice: four consecutive panels per team
    rp2, ci2, va2 = gen.random_csr(300, 300, 12, seed=5)
    t2 = hip.team_format_host(rp2, ci2, va2)
    assert not t2["lattice"] and np.array_equal(t2["tpanel"].reshape(-1)[:38], np.arange(38))


def test_crpspmm_grid_rule_matches_oracle(crp, orc):
    """Grid rule of the older all-in-one engine (deprecated/src/crpspmm.c:136-195): library vs the
    oracle's line-by-line restatement, on matrices that meet the reference's assumptions (sorted
    indices, no empty rows); both M-, N- and mixed splits must occur."""
    import scipy.sparse as sp
    from crp_spmm_amd import planner
    rng = np.random.default_rng(0)
    kinds = set()
    for trial in range(120):
        m, k = int(rng.integers(20, 400)), int(rng.integers(20, 400))
        if trial % 3 == 0:
            A = sp.diags([1.0] * 5, [-2, -1, 0, 1, 2], shape=(m, m), format="csr")
            k = m
        else:
            A = sp.random(m, k, rng.uniform(0.01, 0.2), format="csr", random_state=int(rng.integers(1 << 30)))
            A = sp.csr_matrix(A + sp.csr_matrix((np.ones(m), (np.arange(m), np.arange(m) % k)), shape=(m, k)))
        A.sort_indices()
        P = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 7, 9, 30]))
        n = int(rng.choice([1, 2, 8, 32, 256]))
        try:
            a = orc.crpspmm_plan_grid(P, m, n, k, A.indptr, A.indices)
        except IndexError:
            continue                      # a candidate panel starts at row m: the reference reads past its arrays
        b = planner.crpspmm_plan_grid(P, m, n, k, A.indptr, A.indices)
        assert (a[0], a[1]) == (b[0], b[1]), (trial, P, n, a, b)
        ai = a[2].copy()
        ai[-1] = m                        # the library closes the last panel at m (trailing empty rows)
        assert np.array_equal(ai, b[2]), (trial, a, b)
        kinds.add((a[0] > 1, a[1] > 1))
    assert kinds == {(False, False), (True, False), (False, True), (True, True)}


def test_amortized_planner_is_exhaustive_minimum(crp, orc):
    """crp_spmm_part2d_amortized: cheapest of ALL pm x pn under the reference's cost terms
    (src/spmat_part.c:143-145) with rA applied to every candidate; arrays laid out like the
    reference planner's for the same grid."""
    from crp_spmm_amd import gen, planner
    m, n = 6000, 48
    rp, ci, _ = gen.banded_fem(m, offsets=(1, 2, 3, 40, 41, 900), seed=2)
    for P in (1, 2, 4, 6, 8, 12):
        rb = planner.csr_mat_row_partition(rp, P)
        for rA in (1, 7, 500):
            got = planner.spmm_part2d_amortized(P, m, n, m, rb, rp, ci, rA)
            costs = {}
            for pn in [d for d in range(1, P + 1) if P % d == 0 and (d == 1 or d <= n)]:
                pm = P // pn
                rows = np.array([rb[i * pn] for i in range(pm + 1)], dtype=np.int32)
                _sz, vol = orc.csr_row_part_comm_size(m, rp, ci, rows, rows)
                costs[pn] = int(float(rp[-1]) * (pn - 1) * 1.5) + rA * vol * n
            best_pn = min(costs, key=lambda d: (costs[d], d))
            assert (got["pm"], got["pn"], got["comm_cost"]) == (P // best_pn, best_pn, costs[best_pn]), (P, rA, costs, got)
            ref = planner.calc_spmm_part2d_from_1d(P, m, n, m, rb, rp, ci, rA=1)
            if rA == 1:
                assert got["comm_cost"] <= ref["comm_cost"]
            if (got["pm"], got["pn"]) == (ref["pm"], ref["pn"]):
                for key in ("A0_rowptr", "B_rowptr", "AC_rowptr", "BC_colptr"):
                    assert np.array_equal(got[key], ref[key]), (P, rA, key)
    # many reuses of a matrix with a far band: replicate A, exchange nothing
    got = planner.spmm_part2d_amortized(8, m, n, m, planner.csr_mat_row_partition(rp, 8), rp, ci, 1000)
    assert (got["pm"], got["pn"]) == (1, 8)


def test_timed_planner_link_model(crp, orc):
    """crp_spmm_part2d_timed: grid by a time model of point-to-point links.  Checked against a
    restatement of the model in numpy for every candidate grid, and on three regimes: a far-banded matrix reused often
    (replicate A, exchange nothing: 1 x P), a block-diagonal matrix multiplied once (nothing to exchange and nothing to
    replicate: P x 1), and a capacity limit that rules out the replicating grids."""
    from crp_spmm_amd import gen, planner

    def frac(nl):
        return 0.44 if nl >= 256 else 0.43 if nl >= 96 else 0.40 if nl > 32 else 0.50 if nl >= 24 else 0.13

    def model(P, m, n, rp, ci, rb, rA, link=64e9, hbm=8000e9):
        out = {}
        for pn in [d for d in range(1, P + 1) if P % d == 0 and (d == 1 or d <= n)]:
            pm = P // pn
            rows = np.array([rb[i * pn] for i in range(pm + 1)])
            nl = -(-n // pn)
            t_rep = t_exch = t_comp = 0.0
            for b in range(pm):
                cols = np.unique(ci[rp[rows[b]]:rp[rows[b + 1]]])
                owner = np.searchsorted(rows, cols, side="right") - 1
                worst = max([int(np.sum(owner == q)) for q in range(pm) if q != b], default=0)
                pnnz = float(rp[rows[b + 1]] - rp[rows[b]])
                prow = float(rows[b + 1] - rows[b])
                t_rep = max(t_rep, 12.0 * pnnz / pn / link if pn > 1 else 0.0)
                t_exch = max(t_exch, 8.0 * worst * nl / link)
                t_comp = max(t_comp, (12.0 * pnnz + 4.0 * (prow + 1) + 8.0 * nl * len(cols) + 8.0 * nl * prow) / (hbm * frac(nl)))
            out[pn] = (t_rep / rA + max(t_comp, t_exch), t_rep, t_exch, t_comp)
        return out

    m = 6000
    rp, ci, _ = gen.banded_fem(m, offsets=(1, 2, 3, 40, 41, 900), seed=2)
    for P in (2, 4, 8):
        rb = planner.csr_mat_row_partition(rp, P)
        for n, rA in ((256, 1), (256, 500), (32, 3), (1024, 50)):
            got = planner.spmm_part2d_timed(P, m, n, m, rb, rp, ci, rA)
            mod = model(P, m, n, rp, ci, rb, rA)
            best = min(mod, key=lambda d: (mod[d][0], d))
            assert (got["pm"], got["pn"]) == (P // best, best), (P, n, rA, mod, got)
            assert np.allclose(got["times"], mod[best][1:], rtol=1e-12, atol=0)
    got = planner.spmm_part2d_timed(8, m, 256, m, planner.csr_mat_row_partition(rp, 8), rp, ci, 1000)
    assert (got["pm"], got["pn"]) == (1, 8) and got["times"][1] == 0.0
    # block diagonal, one multiply: rows split, nothing moves
    blk = 512
    rows = np.repeat(np.arange(4096), 8)
    cols = (rows // blk) * blk + (np.arange(rows.size) * 37) % blk
    order = np.lexsort((cols, rows))
    ci2 = cols[order].astype(np.int32)
    rp2 = np.arange(0, rows.size + 1, 8, dtype=np.int32)
    got = planner.spmm_part2d_timed(8, 4096, 256, 4096, planner.csr_mat_row_partition(rp2, 8), rp2, ci2, 1)
    assert (got["pm"], got["pn"]) == (8, 1) and got["times"][0] == 0.0 and got["times"][1] == 0.0
    # a GPU too small for a replicated panel: only grids with pn = 1 remain
    tiny = 4.0 * 12.0 * float(rp[-1]) / 8 * 1.5 / 0.9
    got = planner.spmm_part2d_timed(8, m, 8, m, planner.csr_mat_row_partition(rp, 8), rp, ci, 1000, hbm_bytes=tiny)
    assert got["pn"] == 1


def test_parallel_ingest_and_cache(crp, orc, tmp_path):
    """Files above 200k entries take the line-parallel parser: same COO as the reference's reader
    (oracle/_ref) entry for entry; an entry spread over two lines or a line with extra tokens falls
    back to the token reader and still matches; the binary CSR cache round-trips bit-exactly."""
    import time
    from crp_spmm_amd import gen, mmio
    m = 40000
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, 50, 700), seed=8)
    rows = np.repeat(np.arange(m), np.diff(rp))
    keep = ci <= rows                                         # lower triangle, symmetric file
    r, c, v = rows[keep], ci[keep], va[keep]
    assert r.size > 200000
    body = "\n".join("%d %d %.17g" % (a + 1, b + 1, x) for a, b, x in zip(r, c, v))
    f1 = tmp_path / "big.mtx"
    f1.write_text("%%MatrixMarket matrix coordinate real symmetric\n% generated\n" + "%d %d %d\n" % (m, m, r.size) + body + "\n")
    t0 = time.time()
    st, nr, nc, row, col, val = mmio.mm_read_sparse_RPI(f1)
    t_par = time.time() - t0
    assert st == 0 and (nr, nc) == (m, m)
    if orc.ref() is not None:
        rr = orc.ref_mm_read(str(f1))
        assert rr[0] == 0 and np.array_equal(row, rr[3]) and np.array_equal(col, rr[4]) and np.array_equal(val, rr[5])
    # entry 7 split over two lines + trailing junk lines: token reader, same result
    lines = body.split("\n")
    a, b, x = lines[7].split()
    lines[7] = a + " " + b + "\n   " + x
    f2 = tmp_path / "split.mtx"
    f2.write_text("%%MatrixMarket matrix coordinate real symmetric\n" + "%d %d %d\n" % (m, m, r.size) + "\n".join(lines) + "\n\n")
    st2, _, _, row2, col2, val2 = mmio.mm_read_sparse_RPI(f2)
    assert st2 == 0 and np.array_equal(row, row2) and np.array_equal(col, col2) and np.array_equal(val, val2)
    # cache
    mm, kk, rp1, ci1, va1 = mmio.read_mtx_csr(f1, verbose=False, cache=True)
    assert (tmp_path / "big.mtx.crpcsr").exists()
    mm2, kk2, rp2, ci2, va2 = mmio.read_mtx_csr(f1, verbose=False, cache=True)
    assert (mm, kk) == (mm2, kk2) and np.array_equal(rp1, rp2) and np.array_equal(ci1, ci2) and np.array_equal(va1, va2)
    assert np.array_equal(rp1, rp) and np.array_equal(ci1, ci) and np.allclose(va1, va, rtol=0, atol=0)
    (tmp_path / "junk.crpcsr").write_bytes(b"not a cache file")
    assert mmio.csr_cache_read(tmp_path / "junk.crpcsr") is None
    print("parallel ingest of %d entries: %.3f s" % (r.size, t_par))


def _range_of_code(code):
    for first in range(8):
        base = first * 8 - first * (first - 1) // 2
        if base <= code < base + (8 - first):
            return first, code - base + 1
    raise AssertionError("bad range code %d" % code)


def _replay_team2(t, m, B, va=None):
    """Replays the team2 streams the way csrc/team2_kernel.hip walks them: per team and wave, round by round;
    the column behind ring slot e of round r is what wave e fetched for that round (tpro for the first 3
    rounds, the record of round r - 3 afterwards); the parts of round r take their values from the wave's stream
    at the offset the record of round r - 3 (or tpro) announced.  W = 8 or 16 waves per team."""
    W, P, compact = t["waves"], t.get("panels_per_wave", 1), t.get("compact", True)
    sbits, fbase = (3, 16) if W == 8 else (4, 20)
    C_out = np.zeros((m, B.shape[1]))
    written = np.zeros(m, dtype=bool)
    rec = t["trec"].reshape(-1, 8, W, 4)          # [block][round in block][wave][word]
    for g in range(t["nteam"]):
        nr, blk0 = int(t["tinfo"][g, 0]), int(t["tinfo"][g, 1])
        cols = np.zeros((nr, W), dtype=np.int64)
        voffs = np.zeros((nr, W), dtype=np.int64)
        for r in range(nr):
            for w in range(W):
                if r < 3:
                    cols[r, w], voffs[r, w] = t["tpro"][g, r, w]
                else:
                    cols[r, w] = np.int32(rec[blk0 + ((r - 3) >> 3), (r - 3) & 7, w, 3])
                    voffs[r, w] = rec[blk0 + ((r - 3) >> 3), (r - 3) & 7, w, 2]
        parts_total = 0
        for w in range(W):
            panels = [int(t["tpanel"][g, w * P + j]) for j in range(P)]
            panel = max(panels)                                         # (-1 only when the wave owns nothing)
            k0 = int(t["tvoff"][W * g + w])
            k = 0
            acc = np.zeros((P, 8, B.shape[1]))
            for r in range(nr):
                x, y = int(rec[blk0 + (r >> 3), r & 7, w, 0]), int(rec[blk0 + (r >> 3), r & 7, w, 1])
                cnt = x & 7
                assert cnt <= 4 and (x & 8) == 0
                # flags: ISSUE while a round r + 3 exists, TAIL near the end, LAST on the last round
                assert bool(x >> fbase & 1) == (r + 3 < nr) and bool(x >> (fbase + 2) & 1) == (r == nr - 1)
                assert bool(x >> (fbase + 1) & 1) == (r + 2 >= nr)
                if panel < 0:
                    assert cnt == 0
                z = int(rec[blk0 + (r >> 3), r & 7, w, 2])
                if cnt:
                    assert (voffs[r, w] & 0xFFFFF) == k, (g, w, r)         # the announced offset (units of 4 values) is where the stream stands
                prefix = 0
                for i in range(cnt):
                    slot = (x >> (4 + sbits * i)) & (W - 1)
                    code = (y >> (6 * i)) & 63
                    first, ln = code >> 3, (code & 7) + 1
                    assert first + ln <= 8
                    pos = ((x >> (fbase + 5)) & 63, (y >> 24) & 63, (z >> 20) & 63, (z >> 26) & 63)[i]
                    # compact blocks: the part's values follow those of the parts before it; full groups: 8 per part
                    assert pos == ((prefix + 7 - first) if compact else (8 * i + 7)), (g, w, r, i)
                    c = int(cols[r, slot])
                    assert 0 <= c < B.shape[0]
                    bank = (x >> (fbase + 11 + i)) & 1
                    assert bank < P and panels[bank] >= 0
                    for rr in range(first, first + ln):
                        # what lane rr reads: the value at (pos - 7 + rr) of the round's block
                        acc[bank, rr] += t["tval"][4 * (k0 + k) + pos - 7 + rr] * B[c]
                    prefix += ln if compact else 8
                if r >= 3 and cnt:
                    # the size class announced three rounds earlier covers this round's values
                    y3 = int(rec[blk0 + ((r - 3) >> 3), (r - 3) & 7, w, 1])
                    assert 8 * ((y3 >> 30) + 1) >= prefix > 8 * (y3 >> 30), (g, w, r)
                k += (prefix + 3) // 4
                parts_total += cnt
            if panel >= 0:
                assert k0 + k == int(t["tvoff"][W * g + w + 1]), (g, w)
            for j in range(P):
                if panels[j] < 0:
                    continue
                lo, hi = panels[j] * 8, min(m, panels[j] * 8 + 8)
                C_out[lo:hi] = acc[j, :hi - lo]
                assert not written[lo:hi].any()
                written[lo:hi] = True
        assert parts_total == int(t["tinfo"][g, 2])
    assert written.all()
    if va is not None:                             # the value-update map names every nonzero's slot
        assert np.array_equal(t["tval"].reshape(-1)[t["vmap"]], va)
    return C_out


@pytest.mark.parametrize("order", ["default", "compact", "full-groups"])
def test_team2_streams_replay(crp, orc, monkeypatch, order):
    """The streams of the LDS-sharing kernel (variant 5), replayed in numpy: every row is produced once and
    equals the oracle's product -- for a stride-lattice matrix (teams of 4 x 2 teeth), a random matrix (8
    consecutive panels per team; duplicates, empty rows), and sizes that leave ragged last panels / teams."""
    from crp_spmm_amd import gen, hip
    if order in ("compact", "full-groups"):
        # value blocks without the holes / with 8 values per part, whatever the fill (the default picks by fill)
        monkeypatch.setenv("CRPSPMM_TEAM2_COMPACT", "1" if order == "compact" else "0")
    rng = np.random.default_rng(2)
    cases = []
    # pwtk-like bands (a near band of 14, two far bands of 6): the tooth-shaped lattice teams are kept
    offs = tuple(range(1, 15)) + tuple(range(304, 310)) + tuple(range(3040, 3046))
    cases.append(("lattice",) + gen.banded_fem(9120, offsets=offs, seed=3))
    # a thin 3-D stencil: also a lattice, but teams clustered by shared columns need fewer B rows and win
    nx, ny, nz = 300, 5, 3
    cases.append(("clustered",) + gen.banded_fem(nx * ny * nz, offsets=(1, 2, 3, nx, nx + 1, nx * ny, nx * ny + 1), seed=3))
    cases.append(("random",) + gen.random_csr(611, 611, 14, seed=5, empty_every=9))
    # enough clustered teams (>= 128) for the bisection order and the generation-wide absolute schedule; primal rows
    # of 34 and dual rows of 7 nonzeros give teams of very different length (rounds with empty slots, idle waves)
    cases.append(("kkt",) + gen.kkt3d(16))
    # rows with many scattered nonzeros: with two panels per wave a column can need more than 4 parts of one wave and is
    # split over two rounds (the scheduler once looped forever on exactly this)
    cases.append(("random40",) + gen.random_csr(777, 1234, 40, seed=3))
    cases.append(("tiny",) + gen.random_csr(13, 40, 5, seed=1))
    rp, ci, va = gen.random_csr(200, 64, 6, seed=8)
    ci2 = ci.copy()
    ci2[1::7] = ci2[0::7][:ci2[1::7].size]             # duplicate columns inside rows (kept, like the reference ingest)
    cases.append(("dups", rp, ci2, va))
    for name, rp, ci, va in cases:
        m = len(rp) - 1
        k = int(ci.max()) + 1 if ci.size else 1
        t = hip.team2_format_host(rp, ci, va)
        assert np.array_equal(np.sort(t["tpanel"][t["tpanel"] >= 0]), np.arange((m + 7) // 8)), name
        assert np.array_equal(np.sort(t["torder"]), np.arange(t["nteam"])), name
        if name in ("lattice", "clustered"):
            assert t["lattice"] == (name == "lattice")
        if name in ("clustered", "random"):
            # clustered teams are not runs of consecutive panels
            tp = t["tpanel"]
            assert any(np.any(np.diff(np.sort(row[row >= 0])) != 1) for row in tp), name
        B = rng.uniform(-1, 1, size=(k, 3))
        got = _replay_team2(t, m, B, va)
        ref = orc.spmm_csr(rp, ci, va, B)
        assert orc.rel_fro_err(ref, got) <= 1e-13, name


def _replay_team2r(t, m, B, va):
    """What csrc/team2r_kernel.hip does with the streams of crp_team2r_format_host, in numpy (B1-less)."""
    G, rd = t["G"], t["rowdma"]
    S, slotb, perw, zero = 8 * G * rd, 1024 // G, G * rd, 8192 * rd
    out = np.zeros((m, B.shape[1]))
    done = np.zeros(m, dtype=np.int64)
    assert np.array_equal(t["tval"][t["vmap"]], va)
    u16 = t["tval"].view(np.uint16)
    u32 = t["tval"].view(np.uint32)
    seen_teams = []
    for en, g in enumerate(t["tgrid"].ravel()):
        ent = t["tent"][en]
        if g < 0:
            assert not ent[:, 0].any()                                    # rounds = 0: the run of this XCD ends here
            continue
        seen_teams.append(int(g))
        nr, r0 = (int(x) for x in t["tinfo"][g])
        assert nr >= 1
        for w in range(8):                                                # the entry table: what a workgroup reads when it turns to this team
            assert int(ent[w, 0]) == nr and int(np.int32(ent[w, 1])) == int(t["tpanel"][g, w])
            assert int(ent[w, 2]) + (int(ent[w, 3]) << 32) == int(t["tvoff"][8 * g + w])
            assert np.array_equal(ent[w, 4:14], t["trec"][r0, w, :10])
            if nr > 1:
                assert np.array_equal(ent[w, 14:24], t["trec"][r0 + 1, w, :10])
        acc = np.zeros((8, 8, B.shape[1]))
        for r in range(nr):
            recs = t["trec"][r0 + r]
            cols = np.array([[int(np.int32(recs[w, 2 + j])) for j in range(perw)] for w in range(8)]).ravel()   # slot w * perw + j
            assert np.all(cols >= 0) and np.all(cols < B.shape[0])
            for w in range(8):
                Lp, at16 = int(recs[w, 0]), int(recs[w, 1])
                assert Lp % 2 == 0 and Lp <= 12
                w0 = 2 * (int(t["tvoff"][8 * g + w]) + at16)              # first 8-byte word of the block
                assert w0 + 10 * Lp + 8 <= 2 * int(t["tvoff"][8 * g + w + 1])
                vals = t["tval"][w0:w0 + 8 * Lp].reshape(8, Lp)
                offs = u16[4 * (w0 + 8 * Lp):4 * (w0 + 8 * Lp) + 8 * Lp].reshape(8, Lp)
                hdr = u32[2 * (w0 + 10 * Lp):2 * (w0 + 10 * Lp) + 16]
                if r + 2 < nr:
                    assert np.array_equal(hdr, t["trec"][r0 + r + 2, w])  # the record of round r + 2 rides behind the block of round r
                else:
                    assert not hdr.any()
                for rr in range(8):
                    for st in range(Lp):
                        o = int(offs[rr, st])
                        if o == zero:
                            assert vals[rr, st] == 0.0                    # padding: the slice of zeros, never a B row
                            continue
                        assert o % slotb == 0 and o // slotb < S
                        acc[w, rr] += vals[rr, st] * B[cols[o // slotb]]
        for w in range(8):
            p = int(t["tpanel"][g, w])
            if p < 0:
                assert not acc[w].any()
                continue
            for rr in range(8):
                row = p * 8 + rr
                if row < m:
                    out[row] = acc[w, rr]
                    done[row] += 1
                else:
                    assert not acc[w, rr].any()
    assert sorted(seen_teams) == list(range(t["nteam"]))
    assert np.all(done == 1)
    return out


@pytest.mark.parametrize("G", [4, 2])
def test_team2r_streams_replay(crp, orc, G):
    """The streams of the row-owner team kernel (variant 7), replayed in numpy: every row is produced once and equals the oracle's
    product; padding steps carry the value 0.0 and the offset of the slice of zeros."""
    from crp_spmm_amd import gen, hip
    rng = np.random.default_rng(4)
    cases = []
    offs = tuple(range(1, 15)) + tuple(range(304, 310)) + tuple(range(3040, 3046))
    cases.append(("lattice",) + gen.banded_fem(9120, offsets=offs, seed=3))
    nx, ny, nz = 300, 5, 3
    cases.append(("clustered",) + gen.banded_fem(nx * ny * nz, offsets=(1, 2, 3, nx, nx + 1, nx * ny, nx * ny + 1), seed=3))
    cases.append(("random",) + gen.random_csr(611, 611, 14, seed=5, empty_every=9))
    cases.append(("kkt",) + gen.kkt3d(16))
    cases.append(("random40",) + gen.random_csr(777, 1234, 40, seed=3))
    cases.append(("dense64",) + gen.random_csr(64, 40, 36, seed=7))         # rows that pass 16 nonzeros on a round's slots: rounds close early
    cases.append(("tiny",) + gen.random_csr(13, 40, 5, seed=1))
    rp, ci, va = gen.random_csr(200, 64, 6, seed=8)
    ci2 = ci.copy()
    ci2[1::7] = ci2[0::7][:ci2[1::7].size]
    cases.append(("dups", rp, ci2, va))
    for name, rp, ci, va in cases:
        m = len(rp) - 1
        k = int(ci.max()) + 1 if ci.size else 1
        t = hip.team2r_format_host(rp, ci, va, G=G)
        assert np.array_equal(np.sort(t["tpanel"][t["tpanel"] >= 0]), np.arange((m + 7) // 8)), name
        B = rng.uniform(-1, 1, size=(k, 3))
        got = _replay_team2r(t, m, B, va)
        ref = orc.spmm_csr(rp, ci, va, B)
        assert orc.rel_fro_err(ref, got) <= 1e-13, name


def test_lattice_team_order_search(crp, orc, monkeypatch):
    """The processing order of lattice teams (csrc/team_order.cpp, lattice_block_order): for a 3-D stencil in natural order the order
    found by the search against the L2 model fetches clearly fewer B rows than the strips along the teeth in tools/l2sim.py's replay
    (the same model, restated in Python: dispatch in order, 64 resident teams per XCD, an LRU of 2048 row slices); both launch grids
    name every team once; the streams' product does not depend on the order."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import l2sim
    from crp_spmm_amd import gen, hip
    rp, ci, va = gen.fem3d(32)
    m = len(rp) - 1
    res = {}
    for lo in ("0", "1"):
        monkeypatch.setenv("CRPSPMM_T2_LATORDER", lo)
        t = hip.team2_format_host(rp, ci, va)
        assert t["lattice"] and t["nteam"] >= 1024
        tg = t["tgrid"].reshape(-1)
        assert np.array_equal(np.sort(tg[tg >= 0]), np.arange(t["nteam"]))
        req, miss = l2sim.simulate(l2sim.team_rounds(t), None, 2048, 64, grid=t["tgrid"])
        res[lo] = (miss, req, t)
    assert res["0"][1] == res["1"][1]                                   # the same rounds, another order
    assert res["1"][0] < 0.85 * res["0"][0], (res["0"][0], res["1"][0])
    B = np.random.default_rng(5).uniform(-1, 1, size=(m, 2))
    got = _replay_team2(res["1"][2], m, B, va)
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B, fast=True), got) <= 1e-13
    # a lattice with few team columns (the pwtk stand-in's shape, scaled down): whatever the search picks is no worse in the model
    offs = tuple(range(1, 15)) + tuple(range(304, 310)) + tuple(range(3040, 3046))
    rp, ci, va = gen.banded_fem(9120 * 8, offsets=offs, seed=3)
    miss = []
    for lo in ("0", "1"):
        monkeypatch.setenv("CRPSPMM_T2_LATORDER", lo)
        t = hip.team2_format_host(rp, ci, va)
        miss.append(l2sim.simulate(l2sim.team_rounds(t), None, 2048, 64, grid=t["tgrid"])[1])
    assert miss[1] <= 1.02 * miss[0], miss
