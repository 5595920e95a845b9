"""ctypes binding of the in-tree C-ABI library ``lib/libcrpspmm_hip.so``.

The product path has no CPU fallback: if the library is missing the import
fails loudly (``CrpLibraryError``) and every device entry point raises on a
non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRPSPMM_LIB_PATH") or os.path.join(_HERE, "lib", "libcrpspmm_hip.so")   # override: A/B builds


class CrpLibraryError(RuntimeError):
    pass


class CrpHipError(RuntimeError):
    pass


c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)
c_ll_p = C.POINTER(C.c_longlong)
c_sz_p = C.POINTER(C.c_size_t)
c_u64_p = C.POINTER(C.c_uint64)

# ---- crp_comm.h -----------------------------------------------------------------
A2A_FN = C.CFUNCTYPE(None, C.c_void_p, c_int_p, c_int_p, C.c_int)
A2AV_FN = C.CFUNCTYPE(None, C.c_void_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p)
AGV_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, c_sz_p, c_sz_p)
BARRIER_FN = C.CFUNCTYPE(None, C.c_void_p)
RED_F64_FN = C.CFUNCTYPE(None, C.c_void_p, c_dbl_p, c_dbl_p, C.c_int, C.c_int)
RED_U64_FN = C.CFUNCTYPE(None, C.c_void_p, c_u64_p, c_u64_p, C.c_int, C.c_int)
A2AV_DEV_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, c_ll_p, c_ll_p, C.c_void_p, c_ll_p, c_ll_p, C.c_void_p)
A2AV_BYTES_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, c_sz_p, c_sz_p, C.c_void_p, c_sz_p, c_sz_p)


class CrpComm(C.Structure):
    pass


AGV_DEV_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                         C.c_void_p)
SPLIT_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_int)   # returns crp_comm_t*
FREE_FN = C.CFUNCTYPE(None, C.POINTER(CrpComm))
CrpComm._fields_ = [
    ("ctx", C.c_void_p), ("nproc", C.c_int), ("rank", C.c_int),
    ("alltoall_i32", A2A_FN), ("alltoallv_i32", A2AV_FN), ("allgatherv_bytes", AGV_FN),
    ("barrier", BARRIER_FN), ("reduce_f64", RED_F64_FN), ("reduce_u64", RED_U64_FN),
    ("alltoallv_dev_f64", A2AV_DEV_FN), ("alltoallv_bytes", A2AV_BYTES_FN), ("split", SPLIT_FN), ("free", FREE_FN),
    ("allgatherv_dev", AGV_DEV_FN),
]


# ---- crp_engine.h ---------------------------------------------------------------
class RpPlanView(C.Structure):
    _fields_ = [
        ("nproc", C.c_int), ("my_rank", C.c_int), ("glb_n", C.c_int), ("A_nrow", C.c_int), ("rB_nrow", C.c_int),
        ("rB_self_src_offset", C.c_int), ("rB_self_dst_offset", C.c_int), ("rB_self_nrow", C.c_int),
        ("rB_p2p", C.c_int), ("rB_reidx", C.c_int),
        ("A_rowptr", c_int_p), ("A_colidx", c_int_p), ("A_val", c_dbl_p),
        ("rB_self_src_ridxs", c_int_p),
        ("rB_scnts", c_ll_p), ("rB_sdispls", c_ll_p), ("rB_sridxs", c_int_p),
        ("rB_rcnts", c_ll_p), ("rB_rdispls", c_ll_p), ("rB_rridxs", c_int_p),
        ("rB_recv_size", C.c_size_t), ("n_exec", C.c_int),
        ("t_init", C.c_double), ("t_pack", C.c_double), ("t_a2a", C.c_double), ("t_unpack", C.c_double),
        ("t_spmm", C.c_double), ("t_exec", C.c_double),
    ]


class MatRedistView(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("nproc", "rank", "src_srow", "src_scol", "src_nrow", "src_ncol", "req_srow",
                                       "req_scol", "req_nrow", "req_ncol", "n_proc_send", "n_proc_recv", "send_cnt",
                                       "recv_cnt")] + \
               [(k, c_int_p) for k in ("send_ranks", "send_sizes", "send_displs", "sblk_sizes", "recv_ranks",
                                       "recv_sizes", "recv_displs", "rblk_sizes")] + \
               [("dt_size", C.c_size_t), ("dev_type", C.c_int), ("hd_trans_ms", C.c_double)]


class CrpspmmView(C.Structure):
    """crp_crpspmm_view_t (include/crp_engine.h)."""
    _fields_ = [(k, C.c_int) for k in ("np_glb", "rank_glb", "np_row", "np_col", "rank_row", "rank_col", "glb_m", "glb_n",
                                       "glb_k", "loc_A_srow", "loc_A_erow", "loc_A_nrow", "loc_A_nnz", "loc_A_nnz_s",
                                       "rd_B_srow", "rd_B_erow", "loc_B_scol", "loc_B_ecol", "loc_B_ncol", "loc_B_srow",
                                       "loc_B_erow", "loc_B_nrow", "a2a_B_finegrain")] + \
               [("loc_A_rowptr", c_int_p), ("loc_A_colidx", c_int_p), ("loc_A_val", c_dbl_p), ("red_B", c_dbl_p),
                ("loc_C", c_dbl_p), ("n_exec", C.c_int)] + \
               [(k, C.c_double) for k in ("t_init", "t_exec", "t_rd_A", "t_agv_A", "t_rd_B", "t_a2a_B", "t_spmm", "t_rd_C",
                                          "t_exec_nr")] + \
               [(k, C.c_size_t) for k in ("nelem_A_rd", "nelem_A_agv", "nelem_B_rd", "nelem_B_a2av", "nelem_B_a2av_min")]


_V = C.c_void_p
_I = C.c_int
_LL = C.c_longlong

# name -> (restype, argtypes); every symbol declared in include/crpspmm_hip.h,
# include/crp_comm.h, include/crp_engine.h, include/utils.h, include/spmat_part.h,
# include/mmio_utils.h, include/crp_rccl.h.  tests/test_host.py (the ABI test there) checks that the library
# exports every function the headers declare.
SIGNATURES = {
    # crpspmm_hip.h
    "crp_hip_version": (C.c_char_p, []),
    "crp_hip_device_count": (_I, [c_int_p]),
    "crp_hip_set_device": (_I, [_I]),
    "crp_hip_get_device": (_I, [c_int_p]),
    "crp_hip_device_info": (_I, [_I, C.c_char_p, c_int_p, c_sz_p]),
    "crp_hip_device_bus_id": (_I, [C.c_char_p, C.c_size_t]),
    "crp_dev_malloc": (_I, [C.POINTER(_V), C.c_size_t]),
    "crp_dev_free": (_I, [_V]),
    "crp_dev_memset": (_I, [_V, _I, C.c_size_t, _V]),
    "crp_dev_memcpy": (_I, [_V, _V, C.c_size_t, _I, _V]),
    "crp_dev_memcpy2d": (_I, [_V, C.c_size_t, _V, C.c_size_t, C.c_size_t, C.c_size_t, _I, _V]),
    "crp_host_malloc": (_I, [C.POINTER(_V), C.c_size_t]),
    "crp_host_free": (_I, [_V]),
    "crp_dev_ptr_is_device": (_I, [_V, c_int_p]),
    "crp_stream_create": (_I, [C.POINTER(_V)]),
    "crp_stream_destroy": (_I, [_V]),
    "crp_stream_sync": (_I, [_V]),
    "crp_event_create": (_I, [C.POINTER(_V)]),
    "crp_event_destroy": (_I, [_V]),
    "crp_event_record": (_I, [_V, _V]),
    "crp_event_sync": (_I, [_V]),
    "crp_stream_wait_event": (_I, [_V, _V]),
    "crp_event_elapsed_ms": (_I, [_V, _V, C.POINTER(C.c_float)]),
    "crp_csr_dev_create": (_I, [_I, _I, c_int_p, c_int_p, c_dbl_p, C.POINTER(_V)]),
    "crp_csr_dev_create_dv": (_I, [_I, _I, c_int_p, c_int_p, c_dbl_p, _V, c_int_p, C.POINTER(_V)]),
    "crp_csr_dev_destroy": (_I, [C.POINTER(_V)]),
    "crp_csr_dev_update_values": (_I, [_V, _V, _V]),
    "crp_csr_dev_set_rowmap": (_I, [_V, c_int_p, _I]),
    "crp_csr_dev_nrow": (_I, [_V]),
    "crp_csr_dev_nnz": (_LL, [_V]),
    "crp_csr_dev_bytes": (_LL, [_V]),
    "crp_csr_dev_row_part_comm_size": (_I, [_V, _I, c_int_p, c_int_p, c_int_p, c_int_p]),
    "crp_csr_dev_auto_variant": (_I, [_V]),
    "crp_csr_dev_reordered": (_I, [_V]),
    "crp_csr_dev_resolved_variant": (_I, [_V, _I]),
    "crp_csr_dev_last_variant": (_I, [_V]),
    "crp_csr_dev_lattice": (_I, [_V]),
    "crp_panel_format_host": (_I, [_I, c_int_p, c_int_p, c_dbl_p, _I, c_int_p, C.POINTER(c_int_p), C.POINTER(c_int_p),
                                   C.POINTER(C.POINTER(C.c_uint)), C.POINTER(c_dbl_p), C.POINTER(_LL),
                                   C.POINTER(c_int_p), c_int_p]),
    "crp_team_format_host": (_I, [_I, c_int_p, c_int_p, c_dbl_p, c_int_p, c_int_p, C.POINTER(c_int_p), C.POINTER(c_int_p),
                                  C.POINTER(c_int_p), C.POINTER(C.POINTER(C.c_uint)), C.POINTER(c_int_p)]),
    "crp_probe_copy": (_I, [C.c_longlong, _V, _V, _I, _V, _V]),
    "crp_probe_stamp": (_I, [_V, _V]),
    "crp_stream_create_cu_mask": (_I, [C.POINTER(_V), _I, C.POINTER(C.c_uint)]),
    "crp_team2_format_host_grid": (_I, [C.POINTER(c_int_p), c_int_p]),
    "crp_team2_format_host_compact": (_I, []),
    "crp_team2_format_host": (_I, [_I, c_int_p, c_int_p, c_dbl_p, c_int_p, c_int_p, C.POINTER(c_int_p), C.POINTER(c_int_p),
                                   C.POINTER(c_int_p), C.POINTER(C.POINTER(C.c_uint)), C.POINTER(_LL),
                                   C.POINTER(C.POINTER(_LL)), C.POINTER(c_dbl_p), C.POINTER(_LL), C.POINTER(c_int_p),
                                   C.POINTER(C.POINTER(C.c_uint))]),
    "crp_team2r_format_host": (_I, [_I, c_int_p, c_int_p, c_dbl_p, _I, c_int_p, c_int_p, C.POINTER(c_int_p), C.POINTER(c_int_p),
                                    C.POINTER(C.POINTER(C.c_uint)), C.POINTER(_LL), C.POINTER(C.POINTER(_LL)), C.POINTER(c_dbl_p),
                                    C.POINTER(_LL), C.POINTER(c_int_p), c_int_p, C.POINTER(C.POINTER(C.c_uint)), C.POINTER(_LL),
                                    C.POINTER(C.POINTER(C.c_uint))]),
    "crp_locality_order_host": (_I, [_I, _I, c_int_p, c_int_p, _I, c_int_p, c_dbl_p]),
    "crp_spmm_csr_f32": (_I, [_V, _I, _V, _LL, _V, _LL, _V, _LL, _I, _V]),
    "crp_spmm_csr_f64": (_I, [_V, _I, _I, _V, _LL, _V, _LL, _V, _LL, _I, _V]),
    "crp_spmm_variant_name": (C.c_char_p, [_I]),
    "crp_spmm_variant_count": (_I, []),
    "crp_gather_rows_f64": (_I, [_I, _I, _I, _V, _V, _LL, _V, _LL, _V]),
    "crp_scatter_rows_f64": (_I, [_I, _I, _I, _V, _V, _LL, _V, _LL, _V]),
    "crp_transpose_f64": (_I, [_I, _I, _V, _LL, _V, _LL, _V]),
    # crp_comm.h
    "crp_comm_self": (C.POINTER(CrpComm), []),
    # crp_engine.h
    "crp_rp_spmm_init": (None, [_I, _I, c_int_p, c_int_p, c_dbl_p, c_int_p, _I, C.POINTER(CrpComm), C.POINTER(_V)]),
    "crp_rp_spmm_init_dv": (None, [_I, _I, c_int_p, c_int_p, c_dbl_p, _V, c_int_p, _I, C.POINTER(CrpComm), C.POINTER(_V)]),
    "crp_rp_spmm_values_from_device": (_I, [_V]),
    "crp_rp_spmm_init_plan_only": (None, [_I, _I, c_int_p, c_int_p, c_dbl_p, c_int_p, _I, C.POINTER(CrpComm),
                                          C.POINTER(_V)]),
    "crp_rp_spmm_free": (None, [C.POINTER(_V)]),
    "crp_rp_spmm_exec": (None, [_V, _I, _V, _I, _V, _I]),
    "crp_rp_spmm_exec_ex": (None, [_V, _I, _V, _LL, _V, _LL, _V]),
    "crp_rp_spmm_print_stat": (None, [_V]),
    "crp_rp_spmm_clear_stat": (None, [_V]),
    "crp_rp_spmm_get_plan": (None, [_V, C.POINTER(RpPlanView)]),
    "crp_rp_spmm_overlap_rows": (None, [_V, c_int_p, c_int_p]),
    "crp_rp_spmm_set_timing": (None, [_V, _I]),
    # crp_rccl.h
    "crp_rccl_get_unique_id": (_I, [_V]),
    "crp_rccl_create": (_I, [_V, _I, _I, C.POINTER(_V)]),
    "crp_rccl_destroy": (_I, [C.POINTER(_V)]),
    "crp_rccl_nranks": (_I, [_V]),
    "crp_rccl_create_seconds": (C.c_double, [_V]),
    "crp_rccl_is_blocking": (_I, [_V]),
    "crp_rccl_issue_seconds": (C.c_double, [_V, C.POINTER(C.c_longlong)]),
    "crp_rp_spmm_exchange_host_seconds": (C.c_double, [_V]),
    "crp_rccl_rank": (_I, [_V]),
    "crp_rccl_alltoallv_f64": (_I, [_V, _V, c_ll_p, c_ll_p, _V, c_ll_p, c_ll_p, _V]),
    "crp_rccl_allgatherv": (_I, [_V, _V, C.c_size_t, _V, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _V]),
    "crp_rccl_alltoallv_bytes": (_I, [_V, _V, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _V, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t), _V]),
    "crp_rccl_comm_alltoallv_dev_f64": (None, [_V, _V, c_ll_p, c_ll_p, _V, c_ll_p, c_ll_p, _V]),
    "crp_rccl_comm_allgatherv_dev": (None, [_V, _V, C.c_size_t, _V, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _V]),
    "crp_para2d_spmm_replicated_on_device": (_I, [_V]),
    "crp_para2d_spmm_value_uploads": (_I, [_V]),
    "crp_rp_spmm_set_variant": (None, [_V, _I]),
    "crp_rp_spmm_kernel_info": (None, [_V, c_int_p, c_int_p, c_int_p]),
    "crp_rp_spmm_alg_bytes": (_LL, [_V]),
    "crp_rp_spmm_update_values": (None, [_V, c_dbl_p]),
    "crp_rp_spmm_nnz": (_LL, [_V]),
    "crp_rp_spmm_dev_colidx_host": (c_int_p, [_V]),
    "crp_para2d_spmm_init": (None, [C.POINTER(CrpComm), _I, _I, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p,
                                    c_dbl_p, C.POINTER(_V)]),
    "crp_para2d_spmm_init_plan_only": (None, [C.POINTER(CrpComm), _I, _I, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p,
                                              c_int_p, c_dbl_p, C.POINTER(_V)]),
    "crp_para2d_spmm_free": (None, [C.POINTER(_V)]),
    "crp_para2d_spmm_exec": (None, [_V, _I, _V, _I, _V, _I]),
    "crp_para2d_spmm_exec_ex": (None, [_V, _I, _V, _LL, _V, _LL, _V]),
    "crp_para2d_spmm_print_stat": (None, [_V]),
    "crp_para2d_spmm_clear_stat": (None, [_V]),
    "crp_para2d_spmm_rp": (_V, [_V]),
    "crp_para2d_spmm_rA_cost": (C.c_size_t, [_V]),
    "crp_para2d_spmm_t_ag_A": (C.c_double, [_V]),
    "crp_mat_redist_init": (None, [_I] * 8 + [C.POINTER(CrpComm), C.c_size_t, _I, C.POINTER(_V), c_sz_p]),
    "crp_mat_redist_attach_workbuf": (None, [_V, _V, _V]),
    "crp_mat_redist_exec": (None, [_V, _V, _I, _V, _I]),
    "crp_mat_redist_free": (None, [C.POINTER(_V)]),
    "crp_mat_redist_get_view": (None, [_V, C.POINTER(MatRedistView)]),
    "crp_crpspmm_init": (None, [_I, _I, _I, _I, _I, c_int_p, c_int_p] + [_I] * 8 + [C.POINTER(CrpComm), C.POINTER(_V)]),
    "crp_crpspmm_init_plan_only": (None, [_I, _I, _I, _I, _I, c_int_p, c_int_p] + [_I] * 8 +
                                   [C.POINTER(CrpComm), C.POINTER(_V)]),
    "crp_crpspmm_exec": (None, [_V, c_int_p, c_int_p, c_dbl_p, _V, _I, _V, _I]),
    "crp_crpspmm_free": (None, [C.POINTER(_V)]),
    "crp_crpspmm_print_stat": (None, [_V]),
    "crp_crpspmm_clear_stat": (None, [_V]),
    "crp_crpspmm_get_view": (None, [_V, C.POINTER(CrpspmmView)]),
    "crp_spmm_part2d_amortized": (None, [_I, _I, _I, _I, c_int_p, c_int_p, c_int_p, _I, c_int_p, c_int_p, c_sz_p,
                                         C.POINTER(c_int_p), C.POINTER(c_int_p), C.POINTER(c_int_p),
                                         C.POINTER(c_int_p)]),
    "crp_spmm_part2d_timed": (None, [_I, _I, _I, _I, c_int_p, c_int_p, c_int_p, _I, c_dbl_p, c_int_p, c_int_p, c_dbl_p,
                                     C.POINTER(c_int_p), C.POINTER(c_int_p), C.POINTER(c_int_p), C.POINTER(c_int_p)]),
    "crp_csr_cache_write": (_I, [C.c_char_p, _I, _I, c_int_p, c_int_p, c_dbl_p]),
    "crp_csr_cache_read": (_I, [C.c_char_p, c_int_p, c_int_p, C.POINTER(c_int_p), C.POINTER(c_int_p), C.POINTER(c_dbl_p)]),
    "crp_crpspmm_plan_grid": (None, [_I, _I, _I, _I, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    # dev_type.h
    "is_dev_type_valid": (_I, [_I]),
    "dev_type_malloc": (_V, [C.c_size_t, _I]),
    "dev_type_free": (None, [_V, _I]),
    "dev_type_realloc": (None, [c_sz_p, C.c_size_t, _I, C.POINTER(_V)]),
    "dev_type_memset": (None, [_V, _I, C.c_size_t, _I]),
    "dev_type_memcpy": (None, [_V, _V, C.c_size_t, _I, _I]),
    "dev_type_copy_matrix": (None, [C.c_size_t, _I, _I, _V, _I, _V, _I, _I]),
    # utils.h
    "get_wtime_sec": (C.c_double, []),
    "calc_block_spos_size": (None, [_I, _I, _I, c_int_p, c_int_p]),
    "malloc_aligned": (_V, [C.c_size_t, C.c_size_t]),
    "free_aligned": (None, [_V]),
    "calc_2norm": (C.c_double, [_I, c_dbl_p]),
    "calc_err_2norm": (None, [_I, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]),
    "copy_matrix": (None, [C.c_size_t, _I, _I, _V, _I, _V, _I, _I]),
    "print_matrix": (None, [_I, _I, _V, _I, _I, _I, C.c_char_p, C.c_char_p]),
    "dump_binary": (None, [C.c_char_p, _V, C.c_size_t]),
    # spmat_part.h
    "csr_mat_row_partition": (None, [_I, c_int_p, _I, c_int_p]),
    "prime_factorization": (_I, [_I, C.POINTER(c_int_p)]),
    "csr_mat_row_part_comm_size": (None, [_I, _I, c_int_p, c_int_p, _I, c_int_p, c_int_p, c_int_p, c_int_p]),
    "calc_spmm_part2d_from_1d": (None, [_I, _I, _I, _I, c_int_p, c_int_p, c_int_p, _I, c_int_p, c_int_p, c_sz_p,
                                        C.POINTER(c_int_p), C.POINTER(c_int_p), C.POINTER(c_int_p),
                                        C.POINTER(c_int_p), _I]),
    # mmio_utils.h
    "mm_read_sparse_RPI": (_I, [C.c_char_p, _I, c_int_p, c_int_p, c_int_p, C.POINTER(c_int_p), C.POINTER(c_int_p),
                                C.POINTER(c_dbl_p)]),
    "coo2csr": (None, [_I, _I, _I, c_int_p, c_int_p, c_dbl_p, C.POINTER(c_int_p), C.POINTER(c_int_p),
                       C.POINTER(c_dbl_p)]),
}

_lib = None


def load():
    """Load libcrpspmm_hip.so (once); raises CrpLibraryError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrpLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C crp-spmm_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    # torch ships its own copies of the ROCm runtime libraries (torch/lib/libamdhip64.so, librccl.so, ...).  Whoever
    # is loaded first decides which copy the process uses; when this library came first and torch second, the process
    # ended up with two HIP runtimes and aborted in their exit handlers ("free(): invalid pointer").  torch is this
    # package's plumbing anyway (device memory, streams, process groups), so it goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # e.g. libamdhip64 missing
        raise CrpLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("CRPSPMM_LIB_LENIENT"):      # A/B runs against an older build (CRPSPMM_LIB_PATH): skip what it lacks
                continue
            raise CrpLibraryError("%s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise CrpHipError("%s failed with code %d" % (what, rc))


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.free.restype = None


def c_free(ptr):
    _libc.free(ptr)
