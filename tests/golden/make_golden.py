#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.  Run in the BUILD container
only (needs /root/reference for oracle/_ref and MKL's runtime for the C values):

    MKL_THREADING_LAYER=SEQUENTIAL python tests/golden/make_golden.py

What the fixtures pin, and with what:
  * <name>.mtx            small Matrix Market inputs written by this script
                          (data, not reference text);
  * <name>.csr.npz        rowptr / colidx / val / bandwidth produced by the
                          REFERENCE's own reader + coo2csr (oracle/_ref, compiled
                          unmodified from examples/mmio.c, examples/mmio_utils.c);
  * <name>.plan.npz       csr_mat_row_partition + calc_spmm_part2d_from_1d
                          outputs of the REFERENCE's own planner (oracle/_ref,
                          src/spmat_part.c) for P in {1,2,3,4,6,8} x n in {1,4,64,128,512};
  * <name>.spmm.npz       C = A * fill_B(0.19, 0.24) computed by Intel MKL's
                          mkl_sparse_d_mm (libmkl_rt from the image, called through
                          ctypes with exactly the argument set of the reference call
                          site src/rowpara_spmm.c:388-408 / examples/test_utils.c:157-179),
                          row-major for n in {4, 33} and column-major for n = 5,
                          plus a second operand B2 with irregular values.
MKL must run with MKL_THREADING_LAYER=SEQUENTIAL or GNU: the default Intel layer
next to libgomp returns wrong numbers.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")

import oracle  # noqa: E402

PS = (1, 2, 3, 4, 6, 8)
NS = (1, 4, 64, 128, 512)


def write_mtx(path, m, k, entries, field, symmetry, comments=()):
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n" % (field, symmetry))
        for c in comments:
            f.write("%% %s\n" % c)
        f.write("%d %d %d\n" % (m, k, len(entries)))
        for e in entries:
            if field == "pattern":
                f.write("%d %d\n" % (e[0] + 1, e[1] + 1))
            elif field == "integer":
                f.write("%d %d %d\n" % (e[0] + 1, e[1] + 1, int(e[2])))
            else:
                f.write("%d %d %.17g\n" % (e[0] + 1, e[1] + 1, e[2]))


def make_inputs():
    rng = np.random.default_rng(20261004)
    # g_symm: real symmetric banded, lower triangle stored
    m = 300
    ent = []
    for i in range(m):
        ent.append((i, i, 30.0 + rng.random()))
        for d in (1, 2, 3, 7, 40, 41):
            if i - d >= 0:
                ent.append((i, i - d, rng.uniform(-1, 1)))
    write_mtx(os.path.join(HERE, "g_symm.mtx"), m, m, ent, "real", "symmetric", ["synthetic banded symmetric"])
    # g_gen: real general, non-square, explicit zeros, an empty row block, duplicates, unsorted order
    m, k = 120, 200
    ent = []
    for i in range(m):
        if 40 <= i < 44:
            continue                      # empty rows
        deg = int(rng.integers(1, 9))
        for c in rng.choice(k, size=deg, replace=False):
            ent.append((i, int(c), float(rng.uniform(-2, 2))))
    ent.append((5, 17, 0.0))              # explicit zero
    ent.append((6, 3, 0.0))
    ent.append((7, 11, 1.25))             # duplicate pair (same row, same column)
    ent.append((7, 11, -0.5))
    perm = rng.permutation(len(ent))
    ent = [ent[i] for i in perm]
    write_mtx(os.path.join(HERE, "g_gen.mtx"), m, k, ent, "real", "general", ["non-square, zeros, duplicates"])
    # g_pat: pattern general
    m = 64
    ent = sorted({(int(rng.integers(0, m)), int(rng.integers(0, m))) for _ in range(400)})
    write_mtx(os.path.join(HERE, "g_pat.mtx"), m, m, [(a, b, 1.0) for a, b in ent], "pattern", "general")
    # g_int: integer symmetric
    m = 50
    ent = []
    for i in range(m):
        ent.append((i, i, int(rng.integers(1, 9))))
        for d in (1, 5, 20):
            if i - d >= 0 and rng.random() < 0.8:
                ent.append((i, i - d, int(rng.integers(-9, 10))))
    write_mtx(os.path.join(HERE, "g_int.mtx"), m, m, ent, "integer", "symmetric")
    return ["g_symm", "g_gen", "g_pat", "g_int"]


class MatrixDescr(C.Structure):
    _fields_ = [("type", C.c_int), ("mode", C.c_int), ("diag", C.c_int)]


def mkl_spmm(rowptr, colidx, val, k, B, n, layout, ldB, ldC, m):
    """mkl_sparse_d_create_csr + mkl_sparse_d_mm + mkl_sparse_destroy, arguments as
    src/rowpara_spmm.c:398-408 (alpha 1, beta 0, GENERAL / FULL / NON_UNIT, base 0)."""
    mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so")
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    if ci.size == 0:
        ci, va = np.zeros(1, np.int32), np.zeros(1, np.float64)
    h = C.c_void_p()
    ip = C.POINTER(C.c_int)
    st = mkl.mkl_sparse_d_create_csr(C.byref(h), C.c_int(0), C.c_int(m), C.c_int(k), rp.ctypes.data_as(ip),
                                     rp[1:].ctypes.data_as(ip), ci.ctypes.data_as(ip),
                                     va.ctypes.data_as(C.POINTER(C.c_double)))
    assert st == 0, st
    B = np.ascontiguousarray(B, dtype=np.float64)
    Cm = np.full((m, ldC) if layout == 0 else (n, ldC), np.nan)
    mkl.mkl_sparse_d_mm.argtypes = [C.c_int, C.c_double, C.c_void_p, MatrixDescr, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_double, C.c_void_p, C.c_int]
    st = mkl.mkl_sparse_d_mm(10, 1.0, h, MatrixDescr(20, 42, 50), 101 if layout == 0 else 102, B.ctypes.data, n, ldB,
                             0.0, Cm.ctypes.data, ldC)
    assert st == 0, st
    mkl.mkl_sparse_destroy(h)
    return Cm


def main():
    oracle.build()
    assert oracle.ref() is not None, "oracle/_ref not built (needs /root/reference)"
    names = make_inputs()
    for name in names:
        path = os.path.join(HERE, name + ".mtx")
        m, k, rp, ci, cv, bw = oracle.read_mtx_csr(path, use_ref=True)
        np.savez_compressed(os.path.join(HERE, name + ".csr.npz"), m=m, k=k, rowptr=rp, colidx=ci, val=cv, bandwidth=bw)
        plan = {}
        for P in PS:
            rb = oracle.csr_row_partition(rp, P, use_ref=True)
            plan["rb_P%d" % P] = rb
            for n in NS:
                r = oracle.part2d_from_1d(P, m, n, k, rb, rp, ci, rA=1, use_ref=True)
                key = "P%d_n%d_" % (P, n)
                plan[key + "grid"] = np.array([r["pm"], r["pn"]], dtype=np.int64)
                plan[key + "cost"] = np.array([r["comm_cost"]], dtype=np.uint64)
                for a in ("A0_rowptr", "B_rowptr", "AC_rowptr", "BC_colptr"):
                    plan[key + a] = r[a]
        np.savez_compressed(os.path.join(HERE, name + ".plan.npz"), **plan)
        out = {}
        rng = np.random.default_rng(7)
        for n in (4, 33):
            B = oracle.fill_B(0, k, 0, n)
            Cm = mkl_spmm(rp, ci, cv, k, B, n, 0, n, n, m)
            naive = oracle.spmm_csr(rp, ci, cv, B)
            err = oracle.rel_fro_err(Cm, naive)
            assert err < 1e-14, (name, n, err)
            out["C_fillB_n%d" % n] = Cm
        B2 = rng.uniform(-3, 3, size=(k, 6))
        out["B2"] = B2
        out["C_B2"] = mkl_spmm(rp, ci, cv, k, B2, 6, 0, 6, 6, m)
        # column-major: B stored (n, ldB) with ldB = k + 3, C (n, ldC) with ldC = m + 2
        n = 5
        Bc = np.zeros((n, k + 3))
        Bc[:, :k] = oracle.fill_B(0, k, 0, n).T
        Cc = mkl_spmm(rp, ci, cv, k, Bc, n, 1, k + 3, m + 2, m)
        out["C_fillB_colmajor_n5"] = Cc[:, :m].copy()
        np.savez_compressed(os.path.join(HERE, name + ".spmm.npz"), **out)
        print("%-7s m=%d k=%d nnz=%d bw=%d  MKL-vs-naive ok" % (name, m, k, rp[-1], bw))


if __name__ == "__main__":
    main()
