"""Worker for tests/test_gpu_dist.py::test_rccl_backend_single_rank: ONE rank on the one GPU with the
real device transport (backend "cpu:gloo,cuda:nccl" == RCCL).  RCCL refuses two ranks on one device,
so this is as far as the RCCL path can be exercised on a 1-GPU box: the device all-to-all of the
communicator table on an engine-owned (non-torch) stream with raw hipMalloc buffers, communicator
split / barrier / reductions, and the row-parallel engine end to end over TorchComm."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import oracle as orc
    import crp_spmm_amd
    from crp_spmm_amd import comm as crp_comm, engine, gen

    lib = crp_spmm_amd.load()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    crp_comm.init_process_group(device=0)
    assert crp_comm.exchange_mode() == "nccl" and dist.get_backend() in ("nccl", "undefined", "cpu:gloo,cuda:nccl")
    world = crp_comm.TorchComm()
    assert (world.nproc, world.rank) == (1, 0)
    # device payloads go through the library's own RCCL communicator (include/crp_rccl.h), not through Python
    assert world.device_ranks() == 1 and world.struct.ctx
    assert C.cast(world.struct.alltoallv_dev_f64, C.c_void_p).value == C.cast(lib.crp_rccl_comm_alltoallv_dev_f64, C.c_void_p).value
    ctx = world.struct.ctx

    # ---- device all-to-all on a stream torch does not know, between raw device buffers
    stream = C.c_void_p()
    assert lib.crp_stream_create(C.byref(stream)) == 0
    nel = 3 * 1000 * 256
    src_h = np.arange(nel, dtype=np.float64) * 0.5 - 7.0
    send, recv = C.c_void_p(), C.c_void_p()
    assert lib.crp_dev_malloc(C.byref(send), nel * 8) == 0 and lib.crp_dev_malloc(C.byref(recv), nel * 8) == 0
    assert lib.crp_dev_memcpy(send, src_h.ctypes.data, nel * 8, 0, stream) == 0
    assert lib.crp_dev_memset(recv, 0, nel * 8, stream) == 0
    LL = C.c_longlong
    scn, sds = (LL * 1)(nel), (LL * 2)(0, nel)
    for rep in range(3):
        world.struct.alltoallv_dev_f64(ctx, C.cast(send, C.POINTER(C.c_double)), scn, sds,
                                       C.cast(recv, C.POINTER(C.c_double)), scn, sds, stream)
    out_h = np.zeros(nel)
    assert lib.crp_dev_memcpy(out_h.ctypes.data, recv, nel * 8, 1, stream) == 0      # ordered after the exchange on `stream`
    assert lib.crp_stream_sync(stream) == 0
    assert np.array_equal(out_h, src_h)
    # empty exchange (a rank with nothing to send or receive still enters the collective)
    z = (LL * 1)(0)
    zd = (LL * 2)(0, 0)
    world.struct.alltoallv_dev_f64(ctx, C.cast(send, C.POINTER(C.c_double)), z, zd,
                                   C.cast(recv, C.POINTER(C.c_double)), z, zd, stream)
    assert lib.crp_stream_sync(stream) == 0

    # ---- all-gather between device buffers (the A panel replication): at one rank the own piece lands at its displacement
    assert lib.crp_dev_memset(recv, 0, nel * 8, stream) == 0
    SZ = C.c_size_t
    half = nel * 4
    rb, rd = (SZ * 1)(half), (SZ * 1)(64)
    world.struct.allgatherv_dev(ctx, send, half, recv, rb, rd, stream)
    assert lib.crp_dev_memcpy(out_h.ctypes.data, recv, nel * 8, 1, stream) == 0
    assert lib.crp_stream_sync(stream) == 0
    assert np.array_equal(out_h[8:8 + nel // 2], src_h[:nel // 2]) and not out_h[:8].any()

    # ---- split / barrier / reductions on the mixed backend
    sub = world.split(0, 0)
    assert (sub.nproc, sub.rank) == (1, 0) and sub.device_ranks() == 1
    world.struct.barrier(None)
    a, b = (C.c_double * 2)(1.5, -2.0), (C.c_double * 2)()
    world.struct.reduce_f64(None, a, b, 2, 1)
    assert list(b) == [1.5, -2.0]
    dist.barrier()
    t = torch.tensor([3.25], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == 3.25

    # ---- the engine over TorchComm (bench.py's N > 1 code path, at one rank)
    m = k = 3000
    n = 64
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, 50, 700), seed=4)
    B = orc.fill_B(0, k, 0, n)
    e = engine.RpSpmm(0, m, rp, ci, va, [0, k], n, world)
    e.set_timing(False)
    Bd = torch.from_numpy(B).to(dev)
    Cd = torch.full((m, n), float("nan"), dtype=torch.float64, device=dev)
    for _ in range(3):
        e.exec(0, Bd, Cd)
    torch.cuda.synchronize()
    assert orc.rel_fro_err(orc.spmm_csr(rp, ci, va, B), Cd.cpu().numpy()) <= 1e-12
    e.free()
    sub.free()
    lib.crp_dev_free(send)
    lib.crp_dev_free(recv)
    lib.crp_stream_destroy(stream)
    print("GPU_NCCL_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
