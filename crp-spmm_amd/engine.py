"""Host-side mirror of the reference's engine API over the C ABI.

``RpSpmm`` ~ rp_spmm_*  (/root/reference/src/rowpara_spmm.h:60-87)
``Para2dSpmm`` ~ para2d_spmm_* (/root/reference/src/para2d_spmm.h:42-75)

Same argument names, meaning and call protocol (init -> exec ... -> print_stat
-> free) as the reference; B and C are torch tensors (device-resident: the
zero-copy path) or numpy arrays (host pointers, staged like the reference API).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return a.ctypes.data_as(L.c_int_p)


def _dp(a):
    return a.ctypes.data_as(L.c_dbl_p)


def _ptr_ld(x, layout):
    """(address, leading dimension, keepalive) of a 2-D operand."""
    try:
        import torch
        if isinstance(x, torch.Tensor):
            if x.dtype != torch.float64 or x.dim() != 2:
                raise TypeError("B and C must be 2-D float64")
            if x.stride(1) != 1:
                raise ValueError("operand must be contiguous along its fast dimension")
            return x.data_ptr(), x.stride(0), x
    except ImportError:
        pass
    if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim != 2 or x.strides[1] != 8:
        raise TypeError("B and C must be 2-D float64 torch tensors or numpy arrays, contiguous along the fast dimension")
    return x.ctypes.data, x.strides[0] // 8, x


def _current_stream(x):
    try:
        import torch
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return torch.cuda.current_stream(x.device).cuda_stream
    except ImportError:
        pass
    return None


class RpSpmm:
    """1D row-parallel SpMM engine: C := A * B with A's row block on this rank.

    Arguments as rp_spmm_init (src/rowpara_spmm.h:49-64): ``A_rowptr`` is this
    rank's slice of the GLOBAL row pointer (global nnz offsets), ``B_row_displs``
    has nproc + 1 entries, ``comm`` is a TorchComm / SelfComm and must outlive
    the engine."""

    def __init__(self, A_srow, A_nrow, A_rowptr, A_colidx, A_val, B_row_displs, glb_n, comm, plan_only=False):
        lib = L.load()
        self._lib = lib
        self.comm = comm
        rp, ci, va, bd = _i32(A_rowptr), _i32(A_colidx), _f64(A_val), _i32(B_row_displs)
        if ci.size == 0:
            ci, va = np.zeros(1, np.int32), np.zeros(1, np.float64)
        self.handle = C.c_void_p()
        fn = lib.crp_rp_spmm_init_plan_only if plan_only else lib.crp_rp_spmm_init
        fn(A_srow, A_nrow, _ip(rp), _ip(ci), _dp(va), _ip(bd), glb_n, comm.ptr, C.byref(self.handle))
        self.glb_n = glb_n
        self.A_nrow = A_nrow
        self.loc_B_nrow = int(bd[comm.rank + 1] - bd[comm.rank])
        self._owned = True

    @classmethod
    def _wrap(cls, handle, comm, lib):
        self = cls.__new__(cls)
        self._lib, self.comm, self.handle, self._owned = lib, comm, C.c_void_p(handle), False
        v = self.plan_view()
        self.glb_n, self.A_nrow = v.glb_n, v.A_nrow
        return self

    def exec(self, BC_layout, B, C_out, stream=None):
        """rp_spmm_exec (src/rowpara_spmm.h:69-81); ldB / ldC come from the strides.
        For layout 1 pass the operands as (n, ld) arrays holding the column-major data."""
        bp, ldb, _kb = _ptr_ld(B, BC_layout)
        cp, ldc, _kc = _ptr_ld(C_out, BC_layout)
        # the kernels index the operands by the plan's sizes: a wrong shape would be an out-of-bounds device access
        kb = getattr(self, "loc_B_nrow", None)
        for name, x, rows in (("B", B, kb), ("C", C_out, self.A_nrow)):
            if rows is None:
                continue
            want = (rows, self.glb_n) if BC_layout == 0 else (self.glb_n, rows)
            got = tuple(x.shape)
            if (BC_layout == 0 and (got[0] < want[0] or got[1] != want[1])) or \
               (BC_layout == 1 and (got[0] != want[0] or got[1] < want[1])):
                raise ValueError("%s has shape %s, the engine needs %s (layout %d)" % (name, got, want, BC_layout))
        if stream is None:
            stream = _current_stream(C_out)
        self._lib.crp_rp_spmm_exec_ex(self.handle, BC_layout, bp, ldb, cp, ldc, stream)

    def print_stat(self):
        self._lib.crp_rp_spmm_print_stat(self.handle)

    def clear_stat(self):
        self._lib.crp_rp_spmm_clear_stat(self.handle)

    def set_timing(self, on):
        self._lib.crp_rp_spmm_set_timing(self.handle, int(bool(on)))

    def kernel_info(self):
        """-> dict(variant, variant_name, reordered, lattice) of the local SpMM (crp_rp_spmm_kernel_info)."""
        v, ro, la = C.c_int(), C.c_int(), C.c_int()
        self._lib.crp_rp_spmm_kernel_info(self.handle, C.byref(v), C.byref(ro), C.byref(la))
        name = self._lib.crp_spmm_variant_name(v.value)
        return dict(variant=v.value, variant_name=name.decode() if name else None, reordered=bool(ro.value), lattice=bool(la.value))

    def set_variant(self, variant):
        self._lib.crp_rp_spmm_set_variant(self.handle, int(variant))

    def overlap_rows(self):
        """(interior rows, boundary rows) of the exchange / compute overlap split; (0, 0) when off."""
        a, b = C.c_int(), C.c_int()
        self._lib.crp_rp_spmm_overlap_rows(self.handle, C.byref(a), C.byref(b))
        return a.value, b.value

    def update_values(self, A_val):
        va = _f64(A_val)
        self._lib.crp_rp_spmm_update_values(self.handle, _dp(va))

    def alg_bytes(self):
        return int(self._lib.crp_rp_spmm_alg_bytes(self.handle))

    def nnz(self):
        return int(self._lib.crp_rp_spmm_nnz(self.handle))

    def plan_view(self):
        v = L.RpPlanView()
        self._lib.crp_rp_spmm_get_plan(self.handle, C.byref(v))
        return v

    def plan(self):
        """The struct's plan fields (src/rowpara_spmm.h:8-40) as numpy copies."""
        v = self.plan_view()
        P, nnz = v.nproc, self.nnz()

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, (n,)).astype(dt).copy() if n > 0 else np.zeros(0, dt)
        d = {k: getattr(v, k) for k in ("nproc", "my_rank", "glb_n", "A_nrow", "rB_nrow", "rB_self_src_offset",
                                        "rB_self_dst_offset", "rB_self_nrow", "rB_p2p", "rB_reidx", "rB_recv_size",
                                        "n_exec", "t_init", "t_pack", "t_a2a", "t_unpack", "t_spmm", "t_exec")}
        d["A_rowptr"] = arr(v.A_rowptr, v.A_nrow + 1, np.int32)
        d["A_colidx"] = arr(v.A_colidx, nnz, np.int32)
        d["A_val"] = arr(v.A_val, nnz, np.float64)
        d["rB_self_src_ridxs"] = arr(v.rB_self_src_ridxs, v.rB_self_nrow, np.int32)
        d["rB_scnts"] = arr(v.rB_scnts, P, np.int64)
        d["rB_sdispls"] = arr(v.rB_sdispls, P + 1, np.int64)
        d["rB_rcnts"] = arr(v.rB_rcnts, P, np.int64)
        d["rB_rdispls"] = arr(v.rB_rdispls, P + 1, np.int64)
        n = max(v.glb_n, 1)
        d["rB_sridxs"] = arr(v.rB_sridxs, int(d["rB_sdispls"][P]) // n if v.glb_n else 0, np.int32)
        d["rB_rridxs"] = arr(v.rB_rridxs, int(d["rB_rdispls"][P]) // n if v.glb_n else 0, np.int32)
        p = self._lib.crp_rp_spmm_dev_colidx_host(self.handle)
        d["dev_colidx"] = arr(p, nnz, np.int32)
        return d

    def free(self):
        if getattr(self, "handle", None) is not None and self.handle and self._owned:
            self._lib.crp_rp_spmm_free(C.byref(self.handle))
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Para2dSpmm:
    """2D (pm x pn) engine, arguments as para2d_spmm_init (src/para2d_spmm.h:22-47)."""

    def __init__(self, comm, pm, pn, A0_rowptr, B_rowptr, AC_rowptr, BC_colptr, A_rowptr, A_colidx, A_val,
                 plan_only=False):
        lib = L.load()
        self._lib, self.comm = lib, comm
        a0, br, ac, bc = _i32(A0_rowptr), _i32(B_rowptr), _i32(AC_rowptr), _i32(BC_colptr)
        rp, ci, va = _i32(A_rowptr), _i32(A_colidx), _f64(A_val)
        if ci.size == 0:
            ci, va = np.zeros(1, np.int32), np.zeros(1, np.float64)
        self.handle = C.c_void_p()
        fn = lib.crp_para2d_spmm_init_plan_only if plan_only else lib.crp_para2d_spmm_init
        fn(comm.ptr, pm, pn, _ip(a0), _ip(br), _ip(ac), _ip(bc), _ip(rp), _ip(ci), _dp(va),
                                 C.byref(self.handle))
        self.pm, self.pn = pm, pn
        self.pi, self.pj = comm.rank // pn, comm.rank % pn
        self.rp = RpSpmm._wrap(lib.crp_para2d_spmm_rp(self.handle), comm, lib)

    def exec(self, BC_layout, B, C_out, stream=None):
        bp, ldb, _kb = _ptr_ld(B, BC_layout)
        cp, ldc, _kc = _ptr_ld(C_out, BC_layout)
        if stream is None:
            stream = _current_stream(C_out)
        self._lib.crp_para2d_spmm_exec_ex(self.handle, BC_layout, bp, ldb, cp, ldc, stream)

    def print_stat(self):
        self._lib.crp_para2d_spmm_print_stat(self.handle)

    def clear_stat(self):
        self._lib.crp_para2d_spmm_clear_stat(self.handle)

    @property
    def rA_cost(self):
        return int(self._lib.crp_para2d_spmm_rA_cost(self.handle))

    @property
    def replicated_on_device(self):
        """True when init all-gathered the panel's column indices and values between device buffers."""
        return bool(self._lib.crp_para2d_spmm_replicated_on_device(self.handle))

    @property
    def value_uploads(self):
        """Times the panel's values crossed PCIe towards the device: 0 when the engine's matrices were filled from the device
        all-gather (crp_para2d_spmm_value_uploads)."""
        return int(self._lib.crp_para2d_spmm_value_uploads(self.handle))

    @property
    def t_ag_A(self):
        return float(self._lib.crp_para2d_spmm_t_ag_A(self.handle))

    def free(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self._lib.crp_para2d_spmm_free(C.byref(self.handle))
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MatRedist:
    """Generic dense 2D-block redistribution, arguments as mat_redist_engine_init
    (/root/reference/src/mat_redist.h:53-74) with the communicator in place of MPI_Comm / MPI_Datatype.
    dev_type: 0 host, 1 device staged through the host, 2 device to device."""

    def __init__(self, src_srow, src_scol, src_nrow, src_ncol, req_srow, req_scol, req_nrow, req_ncol, comm,
                 dt_size=8, dev_type=0):
        lib = L.load()
        self._lib, self.comm = lib, comm
        self.handle = C.c_void_p()
        lib.crp_mat_redist_init(src_srow, src_scol, src_nrow, src_ncol, req_srow, req_scol, req_nrow, req_ncol,
                                comm.ptr, dt_size, dev_type, C.byref(self.handle), None)
        if not self.handle:
            raise ValueError("mat_redist_engine_init rejected the arguments (invalid dev_type?)")
        self.dt_size, self.dev_type = dt_size, dev_type
        self.req_shape = (req_nrow, req_ncol)

    def exec(self, src_blk, dst_blk):
        """src_blk / dst_blk: 2-D row-major numpy arrays (dev_type 0) or cuda tensors (1, 2)."""
        sp, sld, _a = _ptr_ld_any(src_blk)
        dp, dld, _b = _ptr_ld_any(dst_blk)
        self._lib.crp_mat_redist_exec(self.handle, sp, sld, dp, dld)

    def view(self):
        v = L.MatRedistView()
        self._lib.crp_mat_redist_get_view(self.handle, C.byref(v))

        def arr(p, n):
            return np.ctypeslib.as_array(p, (n,)).copy() if n > 0 else np.zeros(0, np.int32)
        d = {k: getattr(v, k) for k in ("nproc", "rank", "n_proc_send", "n_proc_recv", "send_cnt", "recv_cnt", "hd_trans_ms")}
        d["send_ranks"], d["send_sizes"] = arr(v.send_ranks, v.n_proc_send), arr(v.send_sizes, v.n_proc_send)
        d["send_displs"], d["sblk_sizes"] = arr(v.send_displs, v.n_proc_send + 1), arr(v.sblk_sizes, 4 * v.n_proc_send)
        d["recv_ranks"], d["recv_sizes"] = arr(v.recv_ranks, v.n_proc_recv), arr(v.recv_sizes, v.n_proc_recv)
        d["recv_displs"], d["rblk_sizes"] = arr(v.recv_displs, v.n_proc_recv + 1), arr(v.rblk_sizes, 4 * v.n_proc_recv)
        return d

    def free(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self._lib.crp_mat_redist_free(C.byref(self.handle))
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CrpspmmEngine:
    """The older all-in-one engine, arguments as crpspmm_engine_init / _exec
    (/root/reference/deprecated/src/crpspmm.h:89-122) with the communicator in place of MPI_Comm:
    A in any 1D row distribution (src_A_rowptr holds GLOBAL nonzero offsets), B and C in arbitrary
    2D blocks, all host (numpy) arrays, A's values passed on every exec."""

    def __init__(self, m, n, k, src_A_srow, src_A_nrow, src_A_rowptr, src_A_colidx, src_B_srow, src_B_nrow, src_B_scol,
                 src_B_ncol, dst_C_srow, dst_C_nrow, dst_C_scol, dst_C_ncol, comm, plan_only=False):
        lib = L.load()
        self._lib, self.comm = lib, comm
        self.handle = C.c_void_p()
        self._rowptr, self._colidx = _i32(src_A_rowptr), _i32(src_A_colidx)
        if self._colidx.size == 0:
            self._colidx = np.zeros(1, np.int32)
        fn = lib.crp_crpspmm_init_plan_only if plan_only else lib.crp_crpspmm_init
        fn(m, n, k, src_A_srow, src_A_nrow, _ip(self._rowptr), _ip(self._colidx), src_B_srow, src_B_nrow, src_B_scol,
           src_B_ncol, dst_C_srow, dst_C_nrow, dst_C_scol, dst_C_ncol, comm.ptr, C.byref(self.handle))
        self.dst_shape = (dst_C_nrow, dst_C_ncol)

    def exec(self, src_A_val, src_B, dst_C):
        """src_B / dst_C: 2-D row-major float64 numpy arrays (the caller's blocks)."""
        val = _f64(src_A_val)
        if val.size == 0:
            val = np.zeros(1, np.float64)
        bp, ldb, _a = _ptr_ld_any(src_B)
        cp, ldc, _b = _ptr_ld_any(dst_C)
        self._lib.crp_crpspmm_exec(self.handle, _ip(self._rowptr), _ip(self._colidx), _dp(val), bp, ldb, cp, ldc)

    def view(self):
        v = L.CrpspmmView()
        self._lib.crp_crpspmm_get_view(self.handle, C.byref(v))
        d = {k: getattr(v, k) for k, _t in L.CrpspmmView._fields_ if not k.startswith(("loc_A_rowptr", "loc_A_colidx",
                                                                                     "loc_A_val", "red_B", "loc_C"))}

        def arr(p, n):
            return np.ctypeslib.as_array(p, (n,)).copy() if n > 0 else np.zeros(0)
        d["loc_A_rowptr"] = arr(v.loc_A_rowptr, v.loc_A_nrow + 1)
        d["loc_A_colidx"] = arr(v.loc_A_colidx, v.loc_A_nnz)
        d["loc_A_val"] = arr(v.loc_A_val, v.loc_A_nnz)
        d["red_B"] = arr(v.red_B, (v.rd_B_erow - v.rd_B_srow) * v.loc_B_ncol).reshape(v.rd_B_erow - v.rd_B_srow, v.loc_B_ncol)
        return d

    def print_stat(self):
        self._lib.crp_crpspmm_print_stat(self.handle)

    def clear_stat(self):
        self._lib.crp_crpspmm_clear_stat(self.handle)

    def free(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self._lib.crp_crpspmm_free(C.byref(self.handle))
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr_ld_any(x):
    """(address, leading dimension in elements, keepalive) of a 2-D row-major operand of any dtype."""
    try:
        import torch
        if isinstance(x, torch.Tensor):
            assert x.dim() == 2 and (x.shape[1] <= 1 or x.stride(1) == 1)
            return x.data_ptr(), x.stride(0) if x.shape[0] > 1 else max(x.shape[1], 1), x
    except ImportError:
        pass
    assert isinstance(x, np.ndarray) and x.ndim == 2 and (x.shape[1] <= 1 or x.strides[1] == x.itemsize)
    ld = x.strides[0] // x.itemsize if x.shape[0] > 1 else max(x.shape[1], 1)
    return x.ctypes.data, ld, x
