"""Device-level wrappers over include/crpspmm_hip.h for torch tensors
(data_ptr plumbing only; all arithmetic happens in the HIP kernels)."""
import ctypes as C

import numpy as np

from . import _lib as L


class CsrDev:
    """Device-resident CSR (crp_csr_dev_create)."""

    def __init__(self, nrow, ncol, rowptr, colidx, val):
        lib = L.load()
        self._lib = lib
        rp = np.ascontiguousarray(rowptr, dtype=np.int32)
        ci = np.ascontiguousarray(colidx, dtype=np.int32)
        va = np.ascontiguousarray(val, dtype=np.float64)
        if ci.size == 0:
            ci, va = np.zeros(1, np.int32), np.zeros(1, np.float64)
        self.handle = C.c_void_p()
        L.check(lib.crp_csr_dev_create(nrow, ncol, rp.ctypes.data_as(L.c_int_p), ci.ctypes.data_as(L.c_int_p),
                                       va.ctypes.data_as(L.c_dbl_p), C.byref(self.handle)), "crp_csr_dev_create")
        self.nrow, self.ncol = nrow, ncol

    @property
    def nnz(self):
        return int(self._lib.crp_csr_dev_nnz(self.handle))

    def row_part_comm_size(self, rblk_ptr, x_displs):
        """csr_mat_row_part_comm_size (src/spmat_part.c:38-64) evaluated on the device-resident CSR -> (sizes, total)."""
        rb = np.ascontiguousarray(rblk_ptr, dtype=np.int32)
        xd = np.ascontiguousarray(x_displs, dtype=np.int32)
        nblk = rb.size - 1
        sizes = np.zeros(nblk, dtype=np.int32)
        tot = C.c_int()
        L.check(self._lib.crp_csr_dev_row_part_comm_size(self.handle, nblk, rb.ctypes.data_as(L.c_int_p), xd.ctypes.data_as(L.c_int_p),
                                                         sizes.ctypes.data_as(L.c_int_p), C.byref(tot)), "crp_csr_dev_row_part_comm_size")
        return sizes, tot.value

    def free(self):
        if self.handle:
            self._lib.crp_csr_dev_destroy(C.byref(self.handle))
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _stream(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def spmm_csr(A, B0, C_out, n=None, layout=0, B1=None, variant=0, stream=None):
    """C_out := A * B (crp_spmm_csr_f64). B0 / B1 / C_out are 2-D float64 cuda tensors;
    layout 1 operands are passed as (n, ld) tensors holding the column-major data."""
    lib = L.load()
    if layout == 0:
        n = C_out.shape[1] if n is None else n
    else:
        n = C_out.shape[0] if n is None else n
    b1p, ld1 = (B1.data_ptr(), B1.stride(0)) if B1 is not None else (None, 0)
    L.check(lib.crp_spmm_csr_f64(A.handle, layout, n, B0.data_ptr() if B0 is not None else None,
                                 B0.stride(0) if B0 is not None else 0, b1p, ld1, C_out.data_ptr(),
                                 C_out.stride(0), variant, _stream(C_out) if stream is None else stream),
            "crp_spmm_csr_f64")


def gather_rows(ridx, src, dst, layout=0, scatter=False, stream=None):
    lib = L.load()
    fn = lib.crp_scatter_rows_f64 if scatter else lib.crp_gather_rows_f64
    if layout == 0:
        n = (src if scatter else dst).shape[1]
    else:
        n = (src if scatter else dst).shape[0]
    L.check(fn(layout, ridx.numel(), n, ridx.data_ptr(), src.data_ptr(), src.stride(0), dst.data_ptr(),
               dst.stride(0), _stream(dst) if stream is None else stream), "crp_gather/scatter_rows_f64")


def transpose(src, dst, stream=None):
    lib = L.load()
    L.check(lib.crp_transpose_f64(src.shape[0], src.shape[1], src.data_ptr(), src.stride(0), dst.data_ptr(),
                                  dst.stride(0), _stream(dst) if stream is None else stream), "crp_transpose_f64")


def device_info(dev=0):
    lib = L.load()
    name = C.create_string_buffer(256)
    cu, mem = C.c_int(), C.c_size_t()
    L.check(lib.crp_hip_device_info(dev, name, C.byref(cu), C.byref(mem)), "crp_hip_device_info")
    return name.value.decode(), cu.value, mem.value


def panel_format_host(rowptr, colidx, val, R):
    """Host-only view of the row-panel format (crp_panel_format_host) as numpy arrays."""
    lib = L.load()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    if ci.size == 0:
        ci, va = np.zeros(1, np.int32), np.zeros(1, np.float64)
    npanel, ent, nord = C.c_int(), C.c_longlong(), C.c_int()
    pptr, pcol, pmask, pval, pord = L.c_int_p(), L.c_int_p(), C.POINTER(C.c_uint)(), L.c_dbl_p(), L.c_int_p()
    L.check(lib.crp_panel_format_host(rp.size - 1, rp.ctypes.data_as(L.c_int_p), ci.ctypes.data_as(L.c_int_p),
                                      va.ctypes.data_as(L.c_dbl_p), R, C.byref(npanel), C.byref(pptr), C.byref(pcol),
                                      C.byref(pmask), C.byref(pval), C.byref(ent), C.byref(pord), C.byref(nord)),
            "crp_panel_format_host")
    P = npanel.value
    pp = np.ctypeslib.as_array(pptr, (P + 1,)).copy()
    tot = int(pp[P])
    out = dict(R=R, npanel=P, pptr=pp, real_entries=ent.value,
               porder=np.ctypeslib.as_array(pord, (max(nord.value, 1),))[:nord.value].copy(),
               pcol=np.ctypeslib.as_array(pcol, (max(tot, 1),))[:tot].copy(),
               pmask4=np.ctypeslib.as_array(pmask, (tot // 4 + 2,)).copy(),
               pval=np.ctypeslib.as_array(pval, (max(tot * R, 1),))[:tot * R].copy().reshape(tot, R))
    for p in (pptr, pcol, pmask, pval, pord):
        L.c_free(C.cast(p, C.c_void_p))
    return out


def team_format_host(rowptr, colidx, val):
    """crp_team_format_host -> dict(nteam, lattice, tpanel, tptr, tcol, tmask, torder)."""
    lib = L.load()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    if ci.size == 0:
        ci, va = np.zeros(1, np.int32), np.zeros(1)
    nteam, lat = C.c_int(), C.c_int()
    tp, tt, tc, to = L.c_int_p(), L.c_int_p(), L.c_int_p(), L.c_int_p()
    tm = C.POINTER(C.c_uint)()
    L.check(lib.crp_team_format_host(rp.size - 1, rp.ctypes.data_as(L.c_int_p), ci.ctypes.data_as(L.c_int_p),
                                     va.ctypes.data_as(L.c_dbl_p), C.byref(nteam), C.byref(lat), C.byref(tp), C.byref(tt),
                                     C.byref(tc), C.byref(tm), C.byref(to)), "crp_team_format_host")
    nt = nteam.value

    def take(ptr, cnt, dt):
        out = np.ctypeslib.as_array(ptr, (max(cnt, 1),))[:cnt].astype(dt).copy()
        L.c_free(C.cast(ptr, C.c_void_p))
        return out
    tptr = take(tt, nt + 1, np.int32)
    tot = int(tptr[-1]) if nt else 0
    return dict(nteam=nt, lattice=bool(lat.value), tpanel=take(tp, 4 * nt, np.int32).reshape(nt, 4), tptr=tptr,
                tcol=take(tc, tot, np.int32), tmask=take(tm, tot, np.uint32), torder=take(to, nt, np.int32))


def team2_format_host(rowptr, colidx, val):
    """crp_team2_format_host -> dict(nteam, waves W = 8, lattice, tpanel[nteam, 8], tinfo[nteam, 4], tpro[nteam, 3, 8, 2],
    trec (uint32 words), tvoff (units of 4 values), tval (the value streams), torder, vmap, tgrid[8, entries per XCD])."""
    lib = L.load()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    nnz = int(rp[-1])
    if ci.size == 0:
        ci, va = np.zeros(1, np.int32), np.zeros(1)
    nteam, lat = C.c_int(), C.c_int()
    tp, ti, tpr, to = L.c_int_p(), L.c_int_p(), L.c_int_p(), L.c_int_p()
    tr, vm = C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)()
    tv = C.POINTER(C.c_longlong)()
    tval = L.c_dbl_p()
    nrw, nve = C.c_longlong(), C.c_longlong()
    L.check(lib.crp_team2_format_host(rp.size - 1, rp.ctypes.data_as(L.c_int_p), ci.ctypes.data_as(L.c_int_p),
                                      va.ctypes.data_as(L.c_dbl_p), C.byref(nteam), C.byref(lat), C.byref(tp), C.byref(ti),
                                      C.byref(tpr), C.byref(tr), C.byref(nrw), C.byref(tv), C.byref(tval), C.byref(nve),
                                      C.byref(to), C.byref(vm)), "crp_team2_format_host")
    nt = nteam.value
    P, W = 1, 8
    tg, ng = L.c_int_p(), C.c_int()
    L.check(lib.crp_team2_format_host_grid(C.byref(tg), C.byref(ng)), "crp_team2_format_host_grid")

    def take(ptr, cnt, dt):
        out = np.ctypeslib.as_array(ptr, (max(cnt, 1),))[:cnt].astype(dt).copy()
        L.c_free(C.cast(ptr, C.c_void_p))
        return out
    return dict(nteam=nt, waves=W, panels_per_wave=P, compact=bool(lib.crp_team2_format_host_compact()), lattice=bool(lat.value), tpanel=take(tp, W * P * nt, np.int32).reshape(nt, W * P),
                tinfo=take(ti, 4 * nt, np.int32).reshape(nt, 4), tpro=take(tpr, 6 * W * nt, np.int32).reshape(nt, 3, W, 2),
                trec=take(tr, nrw.value, np.uint32), tvoff=take(tv, W * nt + 1, np.int64),
                tval=take(tval, nve.value, np.float64), torder=take(to, nt, np.int32),
                vmap=take(vm, nnz, np.uint32), tgrid=take(tg, ng.value, np.int32).reshape(8, -1))


def team2r_format_host(rowptr, colidx, val, G=4):
    """crp_team2r_format_host -> dict(G, nteam, lattice, tpanel[nteam, 8], tinfo[nteam, 2], trec[rounds, 8, 16] (uint32), tvoff (units
    of 16 bytes), tval (the streams as float64 words; view as uint16 for the offsets), tgrid[8, -1], vmap, rounds, steps, slots_filled, nnz)."""
    lib = L.load()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    nnz = int(rp[-1])
    if ci.size == 0:
        ci, va = np.zeros(1, np.int32), np.zeros(1)
    nteam, lat, ng = C.c_int(), C.c_int(), C.c_int()
    tp, ti, tg = L.c_int_p(), L.c_int_p(), L.c_int_p()
    tr, vm, te = C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)()
    tv = C.POINTER(C.c_longlong)()
    tval = L.c_dbl_p()
    nrw, nwd = C.c_longlong(), C.c_longlong()
    stats = (C.c_longlong * 4)()
    L.check(lib.crp_team2r_format_host(rp.size - 1, rp.ctypes.data_as(L.c_int_p), ci.ctypes.data_as(L.c_int_p), va.ctypes.data_as(L.c_dbl_p),
                                       int(G), C.byref(nteam), C.byref(lat), C.byref(tp), C.byref(ti), C.byref(tr), C.byref(nrw), C.byref(tv),
                                       C.byref(tval), C.byref(nwd), C.byref(tg), C.byref(ng), C.byref(vm), stats, C.byref(te)), "crp_team2r_format_host")
    nt = nteam.value

    def take(ptr, cnt, dt):
        out = np.ctypeslib.as_array(ptr, (max(cnt, 1),))[:cnt].astype(dt).copy()
        L.c_free(C.cast(ptr, C.c_void_p))
        return out
    return dict(G=int(G), rowdma=2, nteam=nt, lattice=bool(lat.value), tpanel=take(tp, 8 * nt, np.int32).reshape(nt, 8),
                tinfo=take(ti, 2 * nt, np.int32).reshape(nt, 2), trec=take(tr, nrw.value, np.uint32).reshape(-1, 8, 16),
                tvoff=take(tv, 8 * nt + 1, np.int64), tval=take(tval, nwd.value, np.float64),
                tgrid=take(tg, ng.value, np.int32).reshape(8, -1), vmap=take(vm, nnz, np.uint32),
                tent=take(te, 256 * ng.value, np.uint32).reshape(-1, 8, 32), rounds=int(stats[0]), steps=int(stats[1]), slots_filled=int(stats[2]), nnz=int(stats[3]))


def locality_order_host(rowptr, colidx, ncol=None, nparts=8):
    """crp_locality_order_host -> (perm, info dict or None when the matrix does not qualify)."""
    lib = L.load()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    if ci.size == 0:
        ci = np.zeros(1, np.int32)
    m = rp.size - 1
    perm = np.zeros(max(m, 1), dtype=np.int32)
    info = np.zeros(4)
    rc = lib.crp_locality_order_host(m, m if ncol is None else int(ncol), rp.ctypes.data_as(L.c_int_p), ci.ctypes.data_as(L.c_int_p),
                                     int(nparts), perm.ctypes.data_as(L.c_int_p), info.ctypes.data_as(L.c_dbl_p))
    if rc < 0:
        raise RuntimeError("crp_locality_order_host failed: %d" % rc)
    if rc == 1:
        return perm[:m], None
    return perm[:m], dict(groups=int(info[0]), parts=int(info[1]), mean_dist_before=info[2], mean_dist_after=info[3])


def spmm_csr_f32(A, B0, C_out, n=None, B1=None, variant=0, stream=None):
    """crp_spmm_csr_f32: C := A * B with values, B and C in fp32 (row-major float32 torch tensors on the device)."""
    lib = L.load()
    if n is None:
        n = C_out.shape[1]
    b1p, ldb1 = (B1.data_ptr(), B1.stride(0)) if B1 is not None else (None, 0)
    L.check(lib.crp_spmm_csr_f32(A.handle, n, B0.data_ptr(), B0.stride(0), b1p, ldb1, C_out.data_ptr(), C_out.stride(0), variant,
                                 _stream(C_out) if stream is None else stream), "crp_spmm_csr_f32")
