"""Communicators for the engines: ``crp_comm_t`` (include/crp_comm.h) filled in
from Python.

``TorchComm`` maps the table onto ``torch.distributed``: host control-plane
collectives (the MPI_Alltoall / Alltoallv / Allgatherv / Reduce calls of
/root/reference/src/rowpara_spmm.c:154-162,439-442 and src/para2d_spmm.c:41-83)
run on CPU tensors (gloo), the per-multiply B exchange
(src/rowpara_spmm.c:275-309) runs on device tensors over the ``nccl`` backend,
which is RCCL over xGMI on MI355X.  Initialise the default group with
``backend="cpu:gloo,cuda:nccl"`` (``init_process_group`` below does).
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L

_live = {}        # address of CrpComm struct -> TorchComm (keeps callbacks alive)
_retired = []     # freed communicators: a callback object must not be destroyed while it runs


def exchange_mode():
    """How device payloads travel: "nccl" (RCCL over xGMI, one GPU per rank) or "host"
    (device -> pinned host -> gloo -> device).  "host" is chosen automatically when the ranks of
    this node outnumber its GPUs (RCCL refuses two ranks on one device) -- the rehearsal mode used
    to test the N > 1 GPU path on a single-GPU box -- or with CRPSPMM_EXCHANGE=host."""
    env = os.environ.get("CRPSPMM_EXCHANGE", "").lower()
    if env in ("host", "nccl"):
        return env
    if not torch.cuda.is_available():
        return "host"
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return "host" if local_world > torch.cuda.device_count() else "nccl"


def init_process_group(device=None):
    """One process per GPU; reads RANK / WORLD_SIZE / MASTER_* from the env."""
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if exchange_mode() == "nccl":
        kw = {}
        if device is not None:
            kw["device_id"] = torch.device("cuda", device)
        dist.init_process_group(backend="cpu:gloo,cuda:nccl", **kw)
    else:
        dist.init_process_group(backend="gloo")


def _np_from_ptr(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    ctype = {np.int32: C.c_int, np.float64: C.c_double, np.int64: C.c_longlong, np.uint64: C.c_uint64,
             np.uint8: C.c_ubyte}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,))


class _CudaView:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {"shape": (nelem,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class TorchComm:
    """A crp_comm_t backed by a torch.distributed process group."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (call comm.init_process_group())")
        self.group = group
        self.nproc = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._glb_ranks = dist.get_process_group_ranks(group) if group is not None else list(range(self.nproc))
        s = L.CrpComm()
        s.ctx = None
        s.nproc = self.nproc
        s.rank = self.rank
        # keep the CFUNCTYPE objects referenced from self
        self._cbs = (L.A2A_FN(self._alltoall_i32), L.A2AV_FN(self._alltoallv_i32), L.AGV_FN(self._allgatherv_bytes),
                     L.BARRIER_FN(self._barrier), L.RED_F64_FN(self._reduce_f64), L.RED_U64_FN(self._reduce_u64),
                     L.A2AV_DEV_FN(self._alltoallv_dev_f64), L.A2AV_BYTES_FN(self._alltoallv_bytes),
                     L.SPLIT_FN(self._split), L.FREE_FN(self._free))
        (s.alltoall_i32, s.alltoallv_i32, s.allgatherv_bytes, s.barrier, s.reduce_f64, s.reduce_u64,
         s.alltoallv_dev_f64, s.alltoallv_bytes, s.split, s.free) = self._cbs
        self.struct = s
        self.ptr = C.pointer(s)
        _live[C.addressof(s)] = self

    # ---- host control plane ---------------------------------------------------
    def _alltoall_i32(self, ctx, send, recv, count):
        n = self.nproc * count
        src = torch.from_numpy(_np_from_ptr(send, n, np.int32).copy())
        dst = torch.empty(n, dtype=torch.int32)
        dist.all_to_all_single(dst, src, group=self.group)
        _np_from_ptr(recv, n, np.int32)[:] = dst.numpy()

    def _alltoallv_i32(self, ctx, send, scnts, sdispls, recv, rcnts, rdispls):
        P = self.nproc
        sc = [int(scnts[i]) for i in range(P)]
        sd = [int(sdispls[i]) for i in range(P)]
        rc = [int(rcnts[i]) for i in range(P)]
        rd = [int(rdispls[i]) for i in range(P)]
        sbuf = _np_from_ptr(send, max((sd[i] + sc[i] for i in range(P)), default=0), np.int32)
        src = torch.from_numpy(np.concatenate([sbuf[sd[i]:sd[i] + sc[i]] for i in range(P)]).astype(np.int32))
        dst = torch.empty(sum(rc), dtype=torch.int32)
        dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        out = _np_from_ptr(recv, max((rd[i] + rc[i] for i in range(P)), default=0), np.int32)
        off = 0
        d = dst.numpy()
        for i in range(P):
            out[rd[i]:rd[i] + rc[i]] = d[off:off + rc[i]]
            off += rc[i]

    def _allgatherv_bytes(self, ctx, send, sbytes, recv, rbytes, rdispls):
        P = self.nproc
        rb = [int(rbytes[i]) for i in range(P)]
        rd = [int(rdispls[i]) for i in range(P)]
        mx = max(rb) if rb else 0
        mine = torch.zeros(max(mx, 1), dtype=torch.uint8)
        if sbytes:
            mine[:sbytes] = torch.from_numpy(_np_from_ptr(send, sbytes, np.uint8).copy())
        parts = [torch.empty(max(mx, 1), dtype=torch.uint8) for _ in range(P)]
        dist.all_gather(parts, mine, group=self.group)
        out = _np_from_ptr(recv, max((rd[i] + rb[i] for i in range(P)), default=0), np.uint8)
        for i in range(P):
            out[rd[i]:rd[i] + rb[i]] = parts[i].numpy()[:rb[i]]

    def _barrier(self, ctx):
        t = torch.zeros(1, dtype=torch.int32)
        dist.all_reduce(t, group=self.group)

    def _reduce(self, inp, out, count, op, dtype):
        t = torch.from_numpy(_np_from_ptr(inp, count, dtype).astype(np.float64 if dtype == np.float64 else np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 0 else dist.ReduceOp.SUM, group=self.group)
        _np_from_ptr(out, count, dtype)[:] = t.numpy().astype(dtype)

    def _reduce_f64(self, ctx, inp, out, count, op):
        self._reduce(inp, out, count, op, np.float64)

    def _reduce_u64(self, ctx, inp, out, count, op):
        self._reduce(inp, out, count, op, np.uint64)

    # ---- device payload -------------------------------------------------------
    def _alltoallv_dev_f64(self, ctx, send, scnts, sdispls, recv, rcnts, rdispls, stream):
        P = self.nproc
        sc = [int(scnts[i]) for i in range(P)]
        rc = [int(rcnts[i]) for i in range(P)]
        ns, nr = int(sdispls[P]), int(rdispls[P])
        if ns == 0 and nr == 0:
            # nothing to move for this rank, but the collective must still be entered
            pass
        if torch.cuda.is_available() and exchange_mode() == "host":
            # staged exchange: the same collective on host copies (gloo), ordered on `stream`
            dev = torch.device("cuda", torch.cuda.current_device())
            ext = torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.current_stream()
            with torch.cuda.stream(ext):
                src_d = torch.as_tensor(_CudaView(send, ns), device=dev) if ns else torch.empty(0, dtype=torch.float64, device=dev)
                src = src_d.cpu()                      # synchronises with `stream`
                dst = torch.empty(nr, dtype=torch.float64)
                dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
                if nr:
                    torch.as_tensor(_CudaView(recv, nr), device=dev).copy_(dst)
        elif torch.cuda.is_available():
            dev = torch.device("cuda", torch.cuda.current_device())
            src = torch.as_tensor(_CudaView(send, ns), device=dev) if ns else torch.empty(0, dtype=torch.float64, device=dev)
            dst = torch.as_tensor(_CudaView(recv, nr), device=dev) if nr else torch.empty(0, dtype=torch.float64, device=dev)
            ext = torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.current_stream()
            with torch.cuda.stream(ext):
                dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        else:
            src = torch.from_numpy(_np_from_ptr(send, ns, np.float64))
            dst = torch.from_numpy(_np_from_ptr(recv, nr, np.float64))
            dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.group)

    def _alltoallv_bytes(self, ctx, send, scnts, sdispls, recv, rcnts, rdispls):
        P = self.nproc
        sc = [int(scnts[i]) for i in range(P)]
        sd = [int(sdispls[i]) for i in range(P)]
        rc = [int(rcnts[i]) for i in range(P)]
        rd = [int(rdispls[i]) for i in range(P)]
        sbuf = _np_from_ptr(send, max((sd[i] + sc[i] for i in range(P)), default=0), np.uint8)
        parts = [sbuf[sd[i]:sd[i] + sc[i]] for i in range(P)]
        src = torch.from_numpy(np.concatenate(parts).astype(np.uint8) if parts else np.zeros(0, np.uint8))
        dst = torch.empty(sum(rc), dtype=torch.uint8)
        dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        out = _np_from_ptr(recv, max((rd[i] + rc[i] for i in range(P)), default=0), np.uint8)
        off, d = 0, dst.numpy()
        for i in range(P):
            out[rd[i]:rd[i] + rc[i]] = d[off:off + rc[i]]
            off += rc[i]

    # ---- split / free ---------------------------------------------------------
    def _split(self, ctx, color, key):
        # ctypes callbacks may only return simple types: hand back the struct's address
        return C.addressof(self.split(color, key).struct)

    def split(self, color, key):
        """MPI_Comm_split semantics: ranks with the same color form a group, ordered by (key, rank)."""
        mine = torch.tensor([color, key, self.rank], dtype=torch.int64)
        allv = [torch.empty(3, dtype=torch.int64) for _ in range(self.nproc)]
        dist.all_gather(allv, mine, group=self.group)
        table = [tuple(int(x) for x in t) for t in allv]
        members = sorted((t for t in table if t[0] == color), key=lambda t: (t[1], t[2]))
        ranks = [self._glb_ranks[t[2]] for t in members]
        # only the members enter new_group (use_local_synchronization), so splitting a
        # sub-communicator works as well as splitting the world group
        g = dist.new_group(ranks=ranks, use_local_synchronization=True)
        return TorchComm(g)

    def _free(self, ptr):
        c = _live.pop(C.addressof(ptr.contents), None)
        if c is not None:
            _retired.append(c)

    def free(self):
        c = _live.pop(C.addressof(self.struct), None)
        if c is not None:
            _retired.append(c)


class SelfComm:
    """The library's own single-rank communicator (crp_comm_self)."""

    def __init__(self):
        self.ptr = L.load().crp_comm_self()
        self.nproc, self.rank = 1, 0

    def free(self):
        if self.ptr:
            self.ptr.contents.free(self.ptr)
            self.ptr = None
