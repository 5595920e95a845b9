"""Matrix Market ingest (C++ in csrc/mmio_utils.cpp) -- Python view of
include/mmio_utils.h (/root/reference/examples/mmio_utils.h:17-33) and of the
harness helper read_mtx_csr (/root/reference/examples/test_utils.c:21-55)."""
import ctypes as C
import time

import numpy as np

from . import _lib as L


def _take(ptr, n, dtype):
    out = np.ctypeslib.as_array(ptr, (max(n, 1),))[:n].astype(dtype).copy()
    L.c_free(C.cast(ptr, C.c_void_p))
    return out


def mm_read_sparse_RPI(fname, need_symm=0):
    """-> (status, nrow, ncol, row, col, val); status -1 = rejected / unreadable."""
    nrow, ncol, nnz = C.c_int(), C.c_int(), C.c_int()
    r, c, v = L.c_int_p(), L.c_int_p(), L.c_dbl_p()
    st = L.load().mm_read_sparse_RPI(str(fname).encode(), need_symm, C.byref(nrow), C.byref(ncol), C.byref(nnz),
                                     C.byref(r), C.byref(c), C.byref(v))
    if st != 0:
        return st, 0, 0, None, None, None
    z = nnz.value
    return 0, nrow.value, ncol.value, _take(r, z, np.int32), _take(c, z, np.int32), _take(v, z, np.float64)


def coo2csr(nrow, ncol, row, col, val):
    row = np.ascontiguousarray(row, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    nnz = row.size
    rp, ci, cv = L.c_int_p(), L.c_int_p(), L.c_dbl_p()
    L.load().coo2csr(nrow, ncol, nnz, row.ctypes.data_as(L.c_int_p), col.ctypes.data_as(L.c_int_p),
                     val.ctypes.data_as(L.c_dbl_p), C.byref(rp), C.byref(ci), C.byref(cv))
    return _take(rp, nrow + 1, np.int32), _take(ci, nnz, np.int32), _take(cv, nnz, np.float64)


def csr_cache_write(fname, nrow, ncol, rowptr, colidx, val):
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    if ci.size == 0:
        ci, va = np.zeros(1, np.int32), np.zeros(1)
    return L.load().crp_csr_cache_write(str(fname).encode(), nrow, ncol, rp.ctypes.data_as(L.c_int_p),
                                        ci.ctypes.data_as(L.c_int_p), va.ctypes.data_as(L.c_dbl_p))


def csr_cache_read(fname):
    """-> (nrow, ncol, rowptr, colidx, val) or None when the file is absent / not a cache file."""
    nrow, ncol = C.c_int(), C.c_int()
    rp, ci, va = L.c_int_p(), L.c_int_p(), L.c_dbl_p()
    if L.load().crp_csr_cache_read(str(fname).encode(), C.byref(nrow), C.byref(ncol), C.byref(rp), C.byref(ci),
                                   C.byref(va)) != 0:
        return None
    rowptr = _take(rp, nrow.value + 1, np.int32)
    nnz = int(rowptr[-1])
    return nrow.value, ncol.value, rowptr, _take(ci, nnz, np.int32), _take(va, nnz, np.float64)


def read_mtx_csr(fname, need_symm=0, glb_n=None, verbose=True, cache=False):
    """examples/test_utils.c:21-55: read + convert + the 'A size = ...' banner line.
    -> (m, k, rowptr, colidx, val).  cache=True keeps / reuses a binary copy "<fname>.crpcsr"
    (only when it is newer than the .mtx)."""
    if verbose and glb_n is not None:
        print("B has %d columns" % glb_n)
    t0 = time.time()
    if cache:
        import os
        cf = str(fname) + ".crpcsr"
        if os.path.exists(cf) and os.path.getmtime(cf) >= os.path.getmtime(fname):
            got = csr_cache_read(cf)
            if got is not None:
                m, k, rp, ci, cv = got
                if verbose:
                    print("Rank 0 read matrix A from cache %s used %.2f s" % (cf, time.time() - t0))
                return m, k, rp, ci, cv
    st, m, k, row, col, val = mm_read_sparse_RPI(fname, need_symm)
    if st != 0:
        raise ValueError("cannot ingest Matrix Market file %s" % fname)
    rp, ci, cv = coo2csr(m, k, row, col, val)
    if cache:
        csr_cache_write(str(fname) + ".crpcsr", m, k, rp, ci, cv)
    t1 = time.time()
    bw = int(np.abs(row.astype(np.int64) - col.astype(np.int64)).max()) if row.size else 0
    if verbose:
        print("Rank 0 read matrix A from file %s used %.2f s" % (fname, t1 - t0))
        print("A size = %d * %d, nnz = %d, nnz/row = %d, bandwidth = %d\n" % (m, k, row.size, row.size // max(m, 1), bw))
    return m, k, rp, ci, cv


def write_mtx(fname, m, k, row, col, val, field="real", symmetry="general", comment=None):
    """Write a coordinate Matrix Market file (1-based)."""
    with open(fname, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n" % (field, symmetry))
        if comment:
            f.write("%% %s\n" % comment)
        f.write("%d %d %d\n" % (m, k, len(row)))
        for i in range(len(row)):
            if field == "pattern":
                f.write("%d %d\n" % (row[i] + 1, col[i] + 1))
            elif field == "integer":
                f.write("%d %d %d\n" % (row[i] + 1, col[i] + 1, int(val[i])))
            else:
                f.write("%d %d %.17g\n" % (row[i] + 1, col[i] + 1, val[i]))
