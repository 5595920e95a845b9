"""Partition planner (host, C++ in csrc/spmat_part.cpp) -- Python view of
include/spmat_part.h; same functions as /root/reference/src/spmat_part.h:19-76."""
import ctypes as C

import numpy as np

from . import _lib as L


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ip(a):
    return a.ctypes.data_as(L.c_int_p)


def _take(ptr, n):
    out = np.ctypeslib.as_array(ptr, (n,)).copy()
    L.c_free(C.cast(ptr, C.c_void_p))
    return out


def calc_block_spos_size(length, nblk, iblk):
    s, z = C.c_int(), C.c_int()
    L.load().calc_block_spos_size(length, nblk, iblk, C.byref(s), C.byref(z))
    return s.value, z.value


def even_displs(length, nblk):
    return np.array([calc_block_spos_size(length, nblk, i)[0] for i in range(nblk + 1)], dtype=np.int32)


def csr_mat_row_partition(rowptr, nblk):
    rowptr = _i32(rowptr)
    out = np.zeros(nblk + 1, dtype=np.int32)
    L.load().csr_mat_row_partition(rowptr.size - 1, _ip(rowptr), nblk, _ip(out))
    return out


def prime_factorization(n):
    p = L.c_int_p()
    k = L.load().prime_factorization(n, C.byref(p))
    return [int(x) for x in _take(p, k)] if k > 0 else (L.c_free(C.cast(p, C.c_void_p)) or [])


def csr_mat_row_part_comm_size(ncol, rowptr, colidx, rblk_ptr, x_displs):
    rowptr, colidx, rblk_ptr, x_displs = _i32(rowptr), _i32(colidx), _i32(rblk_ptr), _i32(x_displs)
    nblk = rblk_ptr.size - 1
    sizes = np.zeros(nblk, dtype=np.int32)
    tot = C.c_int()
    L.load().csr_mat_row_part_comm_size(rowptr.size - 1, ncol, _ip(rowptr), _ip(colidx), nblk, _ip(rblk_ptr),
                                        _ip(x_displs), _ip(sizes), C.byref(tot))
    return sizes, tot.value


def calc_spmm_part2d_from_1d(nproc, m, n, k, rb_displs0, rowptr, colidx, rA=1, dbg_print=0):
    """-> dict(pm, pn, comm_cost, A0_rowptr, B_rowptr, AC_rowptr, BC_colptr)."""
    rb, rowptr, colidx = _i32(rb_displs0), _i32(rowptr), _i32(colidx)
    pm, pn, cost = C.c_int(), C.c_int(), C.c_size_t()
    a0, br, ac, bc = L.c_int_p(), L.c_int_p(), L.c_int_p(), L.c_int_p()
    L.load().calc_spmm_part2d_from_1d(nproc, m, n, k, _ip(rb), _ip(rowptr), _ip(colidx), rA, C.byref(pm),
                                      C.byref(pn), C.byref(cost), C.byref(a0), C.byref(br), C.byref(ac),
                                      C.byref(bc), dbg_print)
    return dict(pm=pm.value, pn=pn.value, comm_cost=int(cost.value), A0_rowptr=_take(a0, nproc + 1),
                B_rowptr=_take(br, pm.value + 1), AC_rowptr=_take(ac, pm.value + 1),
                BC_colptr=_take(bc, pn.value + 1))


def spmm_part2d_amortized(nproc, m, n, k, rb_displs0, rowptr, colidx, rA):
    """Grid for an A that is multiplied rA times (crp_spmm_part2d_amortized, include/crp_engine.h):
    every pm x pn priced with the reference's own cost terms, rA applied consistently.
    -> the same dict as calc_spmm_part2d_from_1d."""
    rb, rowptr, colidx = _i32(rb_displs0), _i32(rowptr), _i32(colidx)
    pm, pn, cost = C.c_int(), C.c_int(), C.c_size_t()
    a0, br, ac, bc = L.c_int_p(), L.c_int_p(), L.c_int_p(), L.c_int_p()
    L.load().crp_spmm_part2d_amortized(nproc, m, n, k, _ip(rb), _ip(rowptr), _ip(colidx), rA, C.byref(pm),
                                       C.byref(pn), C.byref(cost), C.byref(a0), C.byref(br), C.byref(ac), C.byref(bc))
    return dict(pm=pm.value, pn=pn.value, comm_cost=int(cost.value), A0_rowptr=_take(a0, nproc + 1),
                B_rowptr=_take(br, pm.value + 1), AC_rowptr=_take(ac, pm.value + 1),
                BC_colptr=_take(bc, pn.value + 1))


def spmm_part2d_timed(nproc, m, n, k, rb_displs0, rowptr, colidx, rA, link_GBs=None, hbm_GBs=None, hbm_bytes=None):
    """Grid by the time model of crp_spmm_part2d_timed (include/crp_engine.h): point-to-point links, the kernels'
    measured roofline fraction at n / pn columns, HBM capacity.  -> the planner dict + times = (t_rep, t_exch, t_comp) s."""
    rb, rowptr, colidx = _i32(rb_displs0), _i32(rowptr), _i32(colidx)
    pm, pn = C.c_int(), C.c_int()
    a0, br, ac, bc = L.c_int_p(), L.c_int_p(), L.c_int_p(), L.c_int_p()
    mm = np.array([link_GBs or 0.0, hbm_GBs or 0.0, hbm_bytes or 0.0], dtype=np.float64)
    times = np.zeros(3)
    L.load().crp_spmm_part2d_timed(nproc, m, n, k, _ip(rb), _ip(rowptr), _ip(colidx), rA, mm.ctypes.data_as(L.c_dbl_p), C.byref(pm),
                                   C.byref(pn), times.ctypes.data_as(L.c_dbl_p), C.byref(a0), C.byref(br), C.byref(ac), C.byref(bc))
    return dict(pm=pm.value, pn=pn.value, times=tuple(times), A0_rowptr=_take(a0, nproc + 1),
                B_rowptr=_take(br, pm.value + 1), AC_rowptr=_take(ac, pm.value + 1), BC_colptr=_take(bc, pn.value + 1))


def crpspmm_plan_grid(nproc, m, n, k, rowptr, colidx):
    """Grid rule of the older all-in-one engine (/root/reference/deprecated/src/crpspmm.c:136-195):
    returns (np_row, np_col, m_split_idx).  rowptr / colidx: the global CSR pattern."""
    lib = L.load()
    rowptr, colidx = _i32(rowptr), _i32(colidx)
    cse = np.zeros(2 * max(m, 1), np.int32)
    for i in range(m):
        s, e = rowptr[i], rowptr[i + 1]
        cse[2 * i], cse[2 * i + 1] = (colidx[s:e].min(), colidx[s:e].max()) if e > s else (1, 0)
    pr, pc = C.c_int(), C.c_int()
    idx = np.zeros(nproc + 1, np.int32)
    lib.crp_crpspmm_plan_grid(nproc, m, n, k, _ip(rowptr), _ip(cse), C.byref(pr), C.byref(pc), _ip(idx))
    return pr.value, pc.value, idx[:pr.value + 1].copy()
