"""TEST INFRASTRUCTURE -- CPU oracle for the CRP-SpMM hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  It wraps

* ``oracle/liborc.so``  (``crp_oracle.c``): the plain-C restatement of the
  reference's algorithms, every function citing the reference file:line;
* ``oracle/_ref/libcrpref.so``: the reference's OWN planner / ingest sources
  compiled unmodified (``oracle/Makefile``), used to validate the restatement
  where the reference is buildable in this image;
* a numpy restatement of ``rp_spmm_init``'s exchange plan
  (/root/reference/src/rowpara_spmm.c:20-190) for all ranks at once.

Pin status: planner + ingest pinned by ``_ref`` and the golden fixtures;
SpMM values pinned by MKL-generated golden C matrices and the closed-form
answer for ``fill_B``; the ``rp_spmm_init`` plan arrays are **parity
unpinned** (their translation unit needs ``mkl.h``, absent from the image)
beyond the end-to-end C check that consumes them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c_int_p = C.POINTER(C.c_int)
_c_dbl_p = C.POINTER(C.c_double)


def build(verbose=False):
    """Compile liborc.so (and _ref when /root/reference exists)."""
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)


def _ip(a):
    return a.ctypes.data_as(_c_int_p)


def _dp(a):
    return a.ctypes.data_as(_c_dbl_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_part2d_from_1d.restype = None
        _lib.orc_prime_factorization.restype = C.c_int
        _lib.orc_symm_expand.restype = C.c_int
    return _lib


def ref():
    """The reference's own compiled planner/ingest, or None when not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libcrpref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        _ref.prime_factorization.restype = C.c_int
        _ref.mm_read_sparse_RPI.restype = C.c_int
    return _ref


# --------------------------------------------------------------------------- SpMM
def spmm_csr(rowptr, colidx, val, B, n=None, layout=0, ldB=None, ldC=None, fast=False):
    """C = A @ B by the naive restatement (rowpara_spmm.c:388-408). Returns C (2-D)."""
    rowptr, colidx, val = _i32(rowptr), _i32(colidx), _f64(val)
    m = rowptr.size - 1
    B = _f64(B)
    if layout == 0:
        k, nn = B.shape
        n = nn if n is None else n
        ldB_ = B.shape[1] if ldB is None else ldB
        Cm = np.empty((m, n if ldC is None else ldC), dtype=np.float64)
        if fast:
            lib().orc_spmm_csr_f64_rm_fast(C.c_int(m), C.c_int(n), _ip(rowptr), _ip(colidx), _dp(val), _dp(B),
                                           C.c_longlong(ldB_), _dp(Cm), C.c_longlong(Cm.shape[1]))
        else:
            lib().orc_spmm_csr_f64(C.c_int(m), C.c_int(n), _ip(rowptr), _ip(colidx), _dp(val), C.c_int(0), _dp(B),
                                   C.c_longlong(ldB_), _dp(Cm), C.c_longlong(Cm.shape[1]))
        return Cm[:, :n]
    # column-major: B given as a (n, ldB) C-contiguous array holding B^T
    nn, ldB_ = B.shape
    n = nn if n is None else n
    ldC_ = m if ldC is None else ldC
    Cm = np.empty((n, ldC_), dtype=np.float64)
    lib().orc_spmm_csr_f64(C.c_int(m), C.c_int(n), _ip(rowptr), _ip(colidx), _dp(val), C.c_int(1), _dp(B),
                           C.c_longlong(ldB_), _dp(Cm), C.c_longlong(ldC_))
    return Cm[:, :m]


def fill_B(srow, nrow, scol, ncol, fi=0.19, fj=0.24):
    """examples/test_utils.c:121-154: B[i][j] = (srow+i)*fi + (scol+j)*fj (row-major)."""
    i = (np.arange(nrow, dtype=np.int32) + np.int32(srow)).astype(np.float64)[:, None]
    j = (np.arange(ncol, dtype=np.int32) + np.int32(scol)).astype(np.float64)[None, :]
    return i * fi + j * fj


def err_2norm(x0, x1):
    """src/utils.c:75-89 -> (||x0||, ||x0 - x1||) with the reference's naive sums."""
    x0, x1 = _f64(x0).ravel(), _f64(x1).ravel()
    a, e = C.c_double(), C.c_double()
    lib().orc_err_2norm(C.c_longlong(x0.size), _dp(x0), _dp(x1), C.byref(a), C.byref(e))
    return a.value, e.value


def rel_fro_err(c_ref, c):
    a, e = err_2norm(c_ref, c)
    return e / a if a > 0 else e


# --------------------------------------------------------------------------- ingest
def coo2csr(nrow, row, col, val, use_ref=False):
    row, col, val = _i32(row), _i32(col), _f64(val)
    nnz = row.size
    if use_ref:
        r = ref()
        rp, ci, cv = _c_int_p(), _c_int_p(), _c_dbl_p()
        r.coo2csr(C.c_int(nrow), C.c_int(0), C.c_int(nnz), _ip(row), _ip(col), _dp(val),
                  C.byref(rp), C.byref(ci), C.byref(cv))
        out = (np.ctypeslib.as_array(rp, (nrow + 1,)).copy(), np.ctypeslib.as_array(ci, (max(nnz, 1),))[:nnz].copy(),
               np.ctypeslib.as_array(cv, (max(nnz, 1),))[:nnz].copy())
        libc = C.CDLL(None)
        for p in (rp, ci, cv):
            libc.free(p)
        return out
    rp = np.zeros(nrow + 1, dtype=np.int32)
    ci = np.zeros(max(nnz, 1), dtype=np.int32)
    cv = np.zeros(max(nnz, 1), dtype=np.float64)
    lib().orc_coo2csr(C.c_int(nrow), C.c_int(nnz), _ip(row), _ip(col), _dp(val), _ip(rp), _ip(ci), _dp(cv))
    return rp, ci[:nnz], cv[:nnz]


def mm_read(path):
    """Restatement of mm_read_sparse_RPI (examples/mmio_utils.c:11-125) + banner
    rules of examples/mmio.c:96-179 in pure Python (small files only).
    Returns (status, nrow, ncol, row, col, val); status -1 on rejection."""
    with open(path, "r") as f:
        lines = f.read().split("\n")
    if not lines or not lines[0]:
        return -1, 0, 0, None, None, None
    tok = lines[0].split()
    if len(tok) < 5 or not tok[0].startswith("%%MatrixMarket"):
        return -1, 0, 0, None, None, None
    mtx, crd, dtype, sym = (t.lower() for t in tok[1:5])
    if mtx != "matrix" or crd not in ("coordinate", "array"):
        return -1, 0, 0, None, None, None
    if dtype not in ("real", "complex", "pattern", "integer") or \
            sym not in ("general", "symmetric", "hermitian", "skew-symmetric"):
        return -1, 0, 0, None, None, None
    if dtype not in ("real", "pattern", "integer") or crd != "coordinate" or sym not in ("general", "symmetric"):
        return -1, 0, 0, None, None, None
    li = 1
    while li < len(lines) and lines[li].startswith("%"):
        li += 1
    toks = " ".join(lines[li:]).split()
    nrow, ncol, nnz = int(toks[0]), int(toks[1]), int(toks[2])
    toks = toks[3:]
    per = 2 if dtype == "pattern" else 3
    row = np.empty(nnz * 2, dtype=np.int32)
    col = np.empty(nnz * 2, dtype=np.int32)
    val = np.empty(nnz * 2, dtype=np.float64)
    for i in range(nnz):
        row[i] = int(toks[per * i]) - 1
        col[i] = int(toks[per * i + 1]) - 1
        if dtype == "real":
            val[i] = float(toks[per * i + 2])
        elif dtype == "integer":
            val[i] = float(int(toks[per * i + 2]))
        else:
            val[i] = 1.0
    if sym == "symmetric":
        nnz = lib().orc_symm_expand(C.c_int(nnz), _ip(row), _ip(col), _dp(val))
    return 0, nrow, ncol, row[:nnz].copy(), col[:nnz].copy(), val[:nnz].copy()


def ref_mm_read(path, need_symm=0):
    """The reference's own mm_read_sparse_RPI (compiled in _ref)."""
    r = ref()
    nrow, ncol, nnz = C.c_int(), C.c_int(), C.c_int()
    rp, cp, vp = _c_int_p(), _c_int_p(), _c_dbl_p()
    st = r.mm_read_sparse_RPI(path.encode(), C.c_int(need_symm), C.byref(nrow), C.byref(ncol), C.byref(nnz),
                              C.byref(rp), C.byref(cp), C.byref(vp))
    if st != 0:
        return st, 0, 0, None, None, None
    z = nnz.value
    out = (np.ctypeslib.as_array(rp, (max(z, 1),))[:z].copy(), np.ctypeslib.as_array(cp, (max(z, 1),))[:z].copy(),
           np.ctypeslib.as_array(vp, (max(z, 1),))[:z].copy())
    return 0, nrow.value, ncol.value, out[0], out[1], out[2]


def read_mtx_csr(path, use_ref=False):
    """examples/test_utils.c:21-55 -> (m, k, rowptr, colidx, val, bandwidth)."""
    st, m, k, row, col, val = (ref_mm_read(path) if use_ref else mm_read(path))
    if st != 0:
        raise ValueError("unsupported Matrix Market file: %s" % path)
    rp, ci, cv = coo2csr(m, row, col, val, use_ref=use_ref)
    bw = int(np.abs(row.astype(np.int64) - col.astype(np.int64)).max()) if row.size else 0
    return m, k, rp, ci, cv, bw


# --------------------------------------------------------------------------- planner
def block_spos(length, nblk, iblk, use_ref=False):
    s, z = C.c_int(), C.c_int()
    f = ref().calc_block_spos_size if use_ref else lib().orc_block_spos_size
    f(C.c_int(length), C.c_int(nblk), C.c_int(iblk), C.byref(s), C.byref(z))
    return s.value, z.value


def even_displs(length, nblk, use_ref=False):
    return np.array([block_spos(length, nblk, i, use_ref)[0] for i in range(nblk + 1)], dtype=np.int32)


def csr_row_partition(rowptr, nblk, use_ref=False):
    rowptr = _i32(rowptr)
    out = np.zeros(nblk + 1, dtype=np.int32)
    f = ref().csr_mat_row_partition if use_ref else lib().orc_csr_row_partition
    f(C.c_int(rowptr.size - 1), _ip(rowptr), C.c_int(nblk), _ip(out))
    return out


def csr_row_part_comm_size(ncol, rowptr, colidx, rblk_ptr, x_displs, use_ref=False):
    rowptr, colidx, rblk_ptr, x_displs = _i32(rowptr), _i32(colidx), _i32(rblk_ptr), _i32(x_displs)
    nblk = rblk_ptr.size - 1
    sizes = np.zeros(nblk, dtype=np.int32)
    tot = C.c_int()
    f = ref().csr_mat_row_part_comm_size if use_ref else lib().orc_csr_row_part_comm_size
    f(C.c_int(rowptr.size - 1), C.c_int(ncol), _ip(rowptr), _ip(colidx), C.c_int(nblk), _ip(rblk_ptr),
      _ip(x_displs), _ip(sizes), C.byref(tot))
    return sizes, tot.value


def prime_factorization(n, use_ref=False):
    if use_ref:
        p = _c_int_p()
        k = ref().prime_factorization(C.c_int(n), C.byref(p))
        out = [p[i] for i in range(k)]
        C.CDLL(None).free(p)
        return out
    fac = np.zeros(32, dtype=np.int32)
    k = lib().orc_prime_factorization(C.c_int(n), _ip(fac))
    return [int(x) for x in fac[:k]]


def part2d_from_1d(nproc, m, n, k, rb_displs0, rowptr, colidx, rA=1, use_ref=False):
    """src/spmat_part.c:85-210 -> dict(pm, pn, comm_cost, A0_rowptr, B_rowptr, AC_rowptr, BC_colptr)."""
    rb, rowptr, colidx = _i32(rb_displs0), _i32(rowptr), _i32(colidx)
    pm, pn = C.c_int(), C.c_int()
    if use_ref:
        cost = C.c_size_t()
        a0, br, ac, bc = _c_int_p(), _c_int_p(), _c_int_p(), _c_int_p()
        ref().calc_spmm_part2d_from_1d(C.c_int(nproc), C.c_int(m), C.c_int(n), C.c_int(k), _ip(rb), _ip(rowptr),
                                       _ip(colidx), C.c_int(rA), C.byref(pm), C.byref(pn), C.byref(cost),
                                       C.byref(a0), C.byref(br), C.byref(ac), C.byref(bc), C.c_int(0))
        out = dict(pm=pm.value, pn=pn.value, comm_cost=int(cost.value),
                   A0_rowptr=np.ctypeslib.as_array(a0, (nproc + 1,)).copy(),
                   B_rowptr=np.ctypeslib.as_array(br, (pm.value + 1,)).copy(),
                   AC_rowptr=np.ctypeslib.as_array(ac, (pm.value + 1,)).copy(),
                   BC_colptr=np.ctypeslib.as_array(bc, (pn.value + 1,)).copy())
        libc = C.CDLL(None)
        for p in (a0, br, ac, bc):
            libc.free(p)
        return out
    cost = C.c_ulonglong()
    a0 = np.zeros(nproc + 1, dtype=np.int32)
    br = np.zeros(nproc + 1, dtype=np.int32)
    ac = np.zeros(nproc + 1, dtype=np.int32)
    bc = np.zeros(nproc + 1, dtype=np.int32)
    lib().orc_part2d_from_1d(C.c_int(nproc), C.c_int(m), C.c_int(n), C.c_int(k), _ip(rb), _ip(rowptr), _ip(colidx),
                             C.c_int(rA), C.byref(pm), C.byref(pn), C.byref(cost), _ip(a0), _ip(br), _ip(ac), _ip(bc))
    return dict(pm=pm.value, pn=pn.value, comm_cost=int(cost.value), A0_rowptr=a0,
                B_rowptr=br[:pm.value + 1].copy(), AC_rowptr=ac[:pm.value + 1].copy(),
                BC_colptr=bc[:pn.value + 1].copy())


# --------------------------------------------------------------------------- rp_spmm_init plan
def rp_plan_all(A_parts, B_row_displs, glb_n, reidx=1):
    """numpy restatement of rp_spmm_init (src/rowpara_spmm.c:20-190) for ALL ranks
    of one communicator at once.  A_parts[r] = (rowptr_slice, colidx, val) exactly
    as rank r would pass them (rowptr slice keeps global nnz offsets).
    Returns a list of dicts with the struct's fields (src/rowpara_spmm.h:8-40)."""
    P = len(A_parts)
    displs = np.asarray(B_row_displs, dtype=np.int64)
    glb_k = int(displs[P])
    plans = []
    for me, (rp_, ci_, va_) in enumerate(A_parts):
        rp_ = np.asarray(rp_, dtype=np.int64)
        nrow = rp_.size - 1
        nnz = int(rp_[nrow] - rp_[0])
        ci = np.asarray(ci_, dtype=np.int64)[:nnz]
        va = np.asarray(va_, dtype=np.float64)[:nnz]
        d = {}
        # step 1 (:47-86)
        srow = int(ci.min()) if nnz else 2 ** 31 - 1
        erow = int(ci.max()) if nnz else 0
        d["A_rowptr"] = (rp_ - rp_[0]).astype(np.int32)
        flag = np.zeros(glb_k, dtype=np.int32)
        flag[ci] = 1
        rB_nrow = erow - srow + 1
        if reidx:
            rowmap = np.arange(max(rB_nrow, 0), dtype=np.int64)
            nz = np.flatnonzero(flag)
            rowmap[nz - srow] = np.arange(nz.size)
            colidx1 = rowmap[ci - srow]
            rB_nrow = int(nz.size)
        else:
            rowmap = None
            colidx1 = ci - srow
        d["A_colidx"] = colidx1.astype(np.int32)
        d["A_val"] = va.copy()
        d["rB_nrow"] = rB_nrow
        # step 2 (:89-117) self rows
        lo, hi = int(displs[me]), int(displs[me + 1])
        self_rows = lo + np.flatnonzero(flag[lo:hi])
        d["rB_self_nrow"] = int(self_rows.size)
        if self_rows.size:
            d["rB_self_src_offset"] = int(self_rows[0] - lo)
            dst = int(self_rows[0] - srow)
            d["rB_self_dst_offset"] = int(rowmap[dst]) if reidx else dst
        else:
            d["rB_self_src_offset"] = 0
            d["rB_self_dst_offset"] = 0
        d["rB_self_src_ridxs"] = self_rows.astype(np.int32)
        flag[lo:hi] = 0
        # step 3 (:120-149) rows needed from every owner
        need = np.flatnonzero(flag)
        owner_cnt = np.array([np.count_nonzero((need >= displs[q]) & (need < displs[q + 1])) for q in range(P)],
                             dtype=np.int64)
        d["_need_glb"] = need                     # global row ids, ordered by owner (monotone)
        d["_rcnt_rows"] = owner_cnt
        d["rB_recv_size"] = int(owner_cnt.sum() - owner_cnt[me])
        d["_srow"] = srow
        d["_rowmap"] = rowmap
        plans.append(d)
    # step 4 (:152-165) alltoall of counts and row ids
    for me, d in enumerate(plans):
        scnt = np.array([plans[q]["_rcnt_rows"][me] for q in range(P)], dtype=np.int64)
        sidx = []
        for q in range(P):
            nd = plans[q]["_need_glb"]
            sidx.append(nd[(nd >= displs[me]) & (nd < displs[me + 1])])
        sidx = np.concatenate(sidx) if sidx else np.zeros(0, dtype=np.int64)
        d["_scnt_rows"] = scnt
        d["rB_sridxs"] = (sidx - displs[me]).astype(np.int32)        # step 5 (:174)
    for d in plans:
        need, srow, rowmap = d.pop("_need_glb"), d.pop("_srow"), d.pop("_rowmap")
        rr = need - srow
        if reidx:
            rr = rowmap[rr] if rr.size else rr
        d["rB_rridxs"] = rr.astype(np.int32)
        rc, sc = d.pop("_rcnt_rows"), d.pop("_scnt_rows")
        d["rB_rcnts"] = (rc * glb_n).astype(np.int64)
        d["rB_scnts"] = (sc * glb_n).astype(np.int64)
        d["rB_rdispls"] = np.concatenate([[0], np.cumsum(rc)]).astype(np.int64) * glb_n
        d["rB_sdispls"] = np.concatenate([[0], np.cumsum(sc)]).astype(np.int64) * glb_n
        d["glb_n"] = glb_n
    return plans


def rp_exec_all(plans, B_parts, n):
    """numpy restatement of rp_spmm_exec (src/rowpara_spmm.c:212-422), row-major,
    for all ranks: pack -> exchange -> unpack -> self copy -> local SpMM.
    B_parts[r]: (loc_B_nrow, n) array.  Returns [C_r]."""
    P = len(plans)
    sendbufs = []
    for me, d in enumerate(plans):
        sendbufs.append(np.asarray(B_parts[me], dtype=np.float64)[d["rB_sridxs"], :n])
    out = []
    for me, d in enumerate(plans):
        rB = np.full((d["rB_nrow"], n), np.nan)
        for q in range(P):
            r0, r1 = d["rB_rdispls"][q] // n, d["rB_rdispls"][q + 1] // n
            if r1 == r0:
                continue
            s0 = plans[q]["rB_sdispls"][me] // n
            rows = sendbufs[q][s0:s0 + (r1 - r0)]
            rB[d["rB_rridxs"][r0:r1]] = rows
        ns = d["rB_self_nrow"]
        if ns:
            src = d["rB_self_src_ridxs"] - d["rB_self_src_ridxs"][0] + d["rB_self_src_offset"]
            # reidx=1 destination is dst_offset + i; reidx=0 dst_offset + row_i (rowpara_spmm.c:359-363)
            if d.get("reidx", 1):
                dst = d["rB_self_dst_offset"] + np.arange(ns)
            else:
                dst = d["rB_self_dst_offset"] + (d["rB_self_src_ridxs"] - d["rB_self_src_ridxs"][0])
            rB[dst] = np.asarray(B_parts[me], dtype=np.float64)[src, :n]
        out.append(spmm_csr(d["A_rowptr"], d["A_colidx"], d["A_val"], rB, n=n))
    return out


def crpspmm_plan_grid(P, m, n, k, rowptr, colidx):
    """The deprecated all-in-one engine's grid rule, restated in plain Python
    (/root/reference/deprecated/src/crpspmm.c:104-195): per row the (first, last) column, then per
    prime factor of P (largest first) split M or N by comparing
        split N: floor(nnz * n_split * 1.5) * p + current B volume   (SIZE_MAX when n_split*p > n)
        split M: floor(nnz * n_split * 1.5) + sum over candidate panels of (hull width) * n
    Returns (np_row, np_col, m_split_idx[:np_row + 1]).  Parity unpinned by reference OUTPUT (the
    deprecated engine needs mkl.h / MPI to build); the restatement follows the source line by line
    and assumes what that code assumes: sorted column indices, no empty rows, panels that
    start before row m."""
    rowptr = np.asarray(rowptr, np.int64)
    colidx = np.asarray(colidx, np.int64)
    first = colidx[rowptr[:-1]]              # crpspmm.c:110-115
    last = colidx[rowptr[1:] - 1]
    SIZE_MAX = 2 ** 64 - 1
    m_split, n_split = 1, 1
    idx = [0, m]
    copy_B = k * n
    nnz = int(rowptr[m])
    for p in reversed(prime_factorization(P)):
        a1 = int(float(nnz) * float(n_split) * 1.5)
        cost_n = a1 * p + copy_B
        if n_split * p > n:
            cost_n = SIZE_MAX
        ms = m_split * p
        idx2, srow, copy_B2 = [0], 0, 0
        for j in range(ms):
            target = nnz if j == ms - 1 else nnz // ms * (j + 1)
            erow = srow + 1
            lo, hi = int(first[srow]), int(last[srow])
            while rowptr[erow] < target:
                lo, hi = min(lo, int(first[erow])), max(hi, int(last[erow]))
                erow += 1
            copy_B2 += (hi - lo + 1) * n
            idx2.append(erow)
            srow = erow
        cost_m = a1 + copy_B2
        if cost_m < cost_n:
            m_split, copy_B, idx = ms, copy_B2, idx2
        else:
            n_split *= p
    return m_split, n_split, np.asarray(idx[:m_split + 1], np.int32)
