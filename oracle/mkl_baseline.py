#!/usr/bin/env python3
"""TEST / BENCH INFRASTRUCTURE (oracle/): the reference's own CPU path, timed for bench.py's cpu_baseline.

What /root/reference/src/rowpara_spmm.c:398-408 does for every multiply, through the MKL runtime the image carries
(libmkl_rt, no headers -- called with ctypes exactly like tests/golden/make_golden.py does):

    mkl_sparse_d_create_csr(&h, BASE_ZERO, m, k, rowptr, rowptr + 1, colidx, val)
    mkl_sparse_d_mm(NON_TRANSPOSE, 1.0, h, {GENERAL, FULL, NON_UNIT}, ROW_MAJOR, B, n, ldB, 0.0, C, ldC)
    mkl_sparse_destroy(h)

all three inside the timed region, per call, as the reference has them.  Must run in a fresh process with
MKL_THREADING_LAYER=GNU (the default Intel layer mixed with libgomp returns garbage); bench.py starts it
that way.  Prints one JSON line: {"value": GFLOP/s, "unit", "kind": "reference", "engine": "mkl", "sample": ...} -- "reference":
the reference's own arithmetic (its three MKL calls per multiply), not this repository's restatement ("port").

usage: mkl_baseline.py <npz with rp, ci, va, k, n> <seconds>
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np


class MatrixDescr(C.Structure):
    _fields_ = [("type", C.c_int), ("mode", C.c_int), ("diag", C.c_int)]


def main():
    d = np.load(sys.argv[1])
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
    rp = np.ascontiguousarray(d["rp"], dtype=np.int32)
    ci = np.ascontiguousarray(d["ci"], dtype=np.int32)
    va = np.ascontiguousarray(d["va"], dtype=np.float64)
    k, n = int(d["k"]), int(d["n"])
    m = rp.size - 1
    mkl = None
    for name in ("/opt/conda/lib/libmkl_rt.so", "libmkl_rt.so", "libmkl_rt.so.2", "libmkl_rt.so.1"):
        try:
            mkl = C.CDLL(name)
            break
        except OSError:
            continue
    if mkl is None:
        sys.exit(3)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    mkl.mkl_sparse_d_mm.argtypes = [C.c_int, C.c_double, C.c_void_p, MatrixDescr, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_double, C.c_void_p, C.c_int]
    # B = fill_B(0.19, 0.24) (examples/test_utils.c:121-154), first touched by many threads is MKL's business
    B = 0.19 * np.arange(k, dtype=np.float64)[:, None] + 0.24 * np.arange(n, dtype=np.float64)[None, :]
    Cm = np.empty((m, n))

    def once():
        h = C.c_void_p()
        st = mkl.mkl_sparse_d_create_csr(C.byref(h), C.c_int(0), C.c_int(m), C.c_int(k), rp.ctypes.data_as(ip),
                                         rp[1:].ctypes.data_as(ip), ci.ctypes.data_as(ip), va.ctypes.data_as(dp))
        if st != 0:
            sys.exit(4)
        st = mkl.mkl_sparse_d_mm(10, 1.0, h, MatrixDescr(20, 42, 50), 101, B.ctypes.data, n, n, 0.0, Cm.ctypes.data, n)
        if st != 0:
            sys.exit(5)
        mkl.mkl_sparse_destroy(h)

    once()                                               # warm-up (thread pool, page faults of C)
    # the number MKL computes must be the product (a wrong threading layer gives garbage without an error)
    rows = np.repeat(np.arange(m), np.diff(rp))
    s1 = np.bincount(rows, weights=va * ci, minlength=m)
    s0 = np.bincount(rows, weights=va, minlength=m)
    probe = np.arange(0, m, max(1, m // 997))
    expect = 0.19 * s1[probe, None] + 0.24 * np.arange(n)[None, :] * s0[probe, None]
    err = np.linalg.norm(Cm[probe] - expect) / max(np.linalg.norm(expect), 1e-300)
    if not err <= 1e-10:
        sys.exit(6)
    reps, t0 = 0, time.time()
    while True:
        once()
        reps += 1
        if time.time() - t0 > budget or reps >= 200:
            break
    dt = (time.time() - t0) / reps
    print(json.dumps({"value": 2.0 * ci.size * n / dt / 1e9, "unit": "GFLOP/s", "kind": "reference", "engine": "mkl",
                      "sample": "full workload (%d rows, %d nnz, n=%d), %d calls of mkl_sparse_d_create_csr + mkl_sparse_d_mm + "
                                "mkl_sparse_destroy (src/rowpara_spmm.c:398-408), %.4f s each, MKL_THREADING_LAYER=%s, %s threads"
                                % (m, ci.size, n, reps, dt, os.environ.get("MKL_THREADING_LAYER", "?"),
                                   os.environ.get("OMP_NUM_THREADS", "?"))}))


if __name__ == "__main__":
    main()
