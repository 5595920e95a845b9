"""Communicators for the engines: ``crp_comm_t`` (include/crp_comm.h) filled in
from Python.

``TorchComm`` maps the table onto ``torch.distributed``: host control-plane
collectives (the MPI_Alltoall / Alltoallv / Allgatherv / Reduce calls of
/root/reference/src/rowpara_spmm.c:154-162,439-442 and src/para2d_spmm.c:41-83)
run on CPU tensors (gloo).  Device payloads -- the per-multiply B exchange
(src/rowpara_spmm.c:275-309) and the replication of an A row panel
(src/para2d_spmm.c:56-86) -- do not pass through Python at all when every rank
has a GPU of its own: the communicator then carries a native RCCL handle
(include/crp_rccl.h, unique id broadcast over gloo) and its device members point
at the library's own functions.  The Python implementations of the device
members below serve the CPU tests (gloo on host buffers) and the "host"
rehearsal mode (several ranks on one GPU: device -> host -> gloo -> device).
"""
import ctypes as C
import functools
import os
import sys
import traceback

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


def _fatal_on_error(fn):
    """ctypes swallows an exception raised inside a callback ("Exception ignored on calling ctypes callback") and
    the C caller carries on with whatever its buffers held: a failed collective would become a silently wrong
    product.  Every callback of the communicator table is wrapped with this: print, then abort the process."""
    @functools.wraps(fn)
    def wrapper(*a, **k):
        try:
            return fn(*a, **k)
        except BaseException:          # noqa: B902 -- nothing may escape into ctypes
            traceback.print_exc()
            sys.stderr.write("[FATAL] communicator callback %s failed: aborting\n" % fn.__name__)
            sys.stderr.flush()
            os.abort()
    return wrapper


def _np_from_ptr(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    ctype = {np.int32: C.c_int, np.float64: C.c_double, np.int64: C.c_longlong, np.uint64: C.c_uint64,
             np.uint8: C.c_ubyte}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,))


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
        # device payloads: a native RCCL communicator (include/crp_rccl.h) when every rank has a GPU of its own --
        # the engines then call straight into the library for the per-multiply B exchange and the replication of A,
        # no Python in the hot loop.  Its unique id travels over this group's gloo side.
        self._rccl = None
        if torch.cuda.is_available() and exchange_mode() == "nccl":
            lib = L.load()
            # byte 128 = "rank 0 has an id": a failure there must reach the peers through this broadcast, not leave
            # them waiting in it (the MPI facade does the same, csrc/mpi_facade.cpp)
            idbuf = torch.zeros(129, dtype=torch.uint8)
            if self.rank == 0:
                raw = (C.c_ubyte * 128)()
                if lib.crp_rccl_get_unique_id(raw) == 0:
                    idbuf = torch.tensor(list(raw) + [1], dtype=torch.uint8)
            dist.broadcast(idbuf, src=self._glb_ranks[0], group=self.group)
            h = C.c_void_p()
            rc = -1
            if int(idbuf[128]) == 1:
                raw = (C.c_ubyte * 128)(*[int(v) for v in idbuf[:128]])
                # (non-blocking creation under a deadline, csrc/crp_rccl.cpp: a peer that never arrives makes this
                #  return an error instead of blocking inside ncclCommInitRank)
                rc = lib.crp_rccl_create(raw, self.nproc, self.rank, C.byref(h))
            # every rank of the group must take the same transport: agree on the outcome over the control plane
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok[0]) == 1:
                self._rccl = h
                s.ctx = h
                s.alltoallv_dev_f64 = C.cast(lib.crp_rccl_comm_alltoallv_dev_f64, L.A2AV_DEV_FN)
                s.allgatherv_dev = C.cast(lib.crp_rccl_comm_allgatherv_dev, L.AGV_DEV_FN)
            else:
                # (loud, but not fatal: the payloads then travel device -> host -> gloo -> device, the rehearsal path)
                print("[crp_spmm_amd] RCCL communicator of %d ranks could not be created (rank %d: code %d); device payloads "
                      "are staged through the host" % (self.nproc, self.rank, rc), file=sys.stderr, flush=True)
                if rc == 0:
                    lib.crp_rccl_destroy(C.byref(h))
                self._staged_fallback = True
                self._agv_dev_cb = L.AGV_DEV_FN(self._allgatherv_dev_staged)
                s.allgatherv_dev = self._agv_dev_cb
        elif torch.cuda.is_available():
            # rehearsal mode (several ranks on one GPU): the device all-gather staged through the host, so that the
            # engines' device-replication branch runs on a 1-GPU box as well
            self._agv_dev_cb = L.AGV_DEV_FN(self._allgatherv_dev_staged)
            s.allgatherv_dev = self._agv_dev_cb
        self.struct = s
        self.ptr = C.pointer(s)
        _live[C.addressof(s)] = self

    @_fatal_on_error
    def _allgatherv_dev_staged(self, ctx, send, sbytes, recv, rbytes, rdispls, stream):
        P = self.nproc
        lib = L.load()
        rb = [int(rbytes[i]) for i in range(P)]
        rd = [int(rdispls[i]) for i in range(P)]
        mine = np.zeros(max(sbytes, 1), dtype=np.uint8)
        if sbytes:
            L.check(lib.crp_dev_memcpy(mine.ctypes.data, send, sbytes, 1, stream), "staged all-gather: device -> host")
        L.check(lib.crp_stream_sync(stream), "stream sync")
        tot = max((rd[i] + rb[i] for i in range(P)), default=0)
        out = np.zeros(max(tot, 1), dtype=np.uint8)
        szt = C.c_size_t * P
        self._allgatherv_bytes(None, mine.ctypes.data, sbytes, out.ctypes.data, szt(*rb), szt(*rd))
        for i in range(P):
            if rb[i]:
                L.check(lib.crp_dev_memcpy(int(recv) + rd[i], out[rd[i]:].ctypes.data, rb[i], 0, stream), "staged all-gather: host -> device")
        L.check(lib.crp_stream_sync(stream), "stream sync")

    def device_ranks(self):
        """Ranks of the native RCCL communicator behind the device collectives, or None (host-staged / CPU)."""
        return int(L.load().crp_rccl_nranks(self._rccl)) if self._rccl else None

    def device_create_seconds(self):
        """Wall time the creation of that communicator took on this rank, or None."""
        return float(L.load().crp_rccl_create_seconds(self._rccl)) if self._rccl else None

    def device_issue(self):
        """(host seconds inside the device collectives' calls so far, calls, blocking communicator?) or None: what issuing the
        grouped sends / receives costs the host per exec (csrc/crp_rccl.cpp)."""
        if not self._rccl:
            return None
        lib = L.load()
        calls = C.c_longlong(0)
        sec = float(lib.crp_rccl_issue_seconds(self._rccl, C.byref(calls)))
        return {"host_s": sec, "calls": int(calls.value), "blocking": bool(lib.crp_rccl_is_blocking(self._rccl))}

    # ---- host control plane ---------------------------------------------------
    @_fatal_on_error
    def _alltoall_i32(self, ctx, send, recv, count):
        n = self.nproc * count
        src = torch.from_numpy(_np_from_ptr(send, n, np.int32).copy())
        dst = torch.empty(n, dtype=torch.int32)
        dist.all_to_all_single(dst, src, group=self.group)
        _np_from_ptr(recv, n, np.int32)[:] = dst.numpy()

    @_fatal_on_error
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

    @_fatal_on_error
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

    @_fatal_on_error
    def _barrier(self, ctx):
        t = torch.zeros(1, dtype=torch.int32)
        dist.all_reduce(t, group=self.group)

    def _reduce(self, inp, out, count, op, dtype):
        t = torch.from_numpy(_np_from_ptr(inp, count, dtype).astype(np.float64 if dtype == np.float64 else np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 0 else dist.ReduceOp.SUM, group=self.group)
        _np_from_ptr(out, count, dtype)[:] = t.numpy().astype(dtype)

    @_fatal_on_error
    def _reduce_f64(self, ctx, inp, out, count, op):
        self._reduce(inp, out, count, op, np.float64)

    @_fatal_on_error
    def _reduce_u64(self, ctx, inp, out, count, op):
        self._reduce(inp, out, count, op, np.uint64)

    # ---- device payload -------------------------------------------------------
    @_fatal_on_error
    def _alltoallv_dev_f64(self, ctx, send, scnts, sdispls, recv, rcnts, rdispls, stream):
        """Python form of the device all-to-all: CPU tests (host buffers) and the "host" rehearsal mode (device ->
        host -> gloo -> device).  With a GPU per rank the table points at the library's RCCL function instead.
        Honours arbitrary displacements, like the contract of crp_comm.h says."""
        P = self.nproc
        sc = [int(scnts[i]) for i in range(P)]
        sd = [int(sdispls[i]) for i in range(P)]
        rc = [int(rcnts[i]) for i in range(P)]
        rd = [int(rdispls[i]) for i in range(P)]
        ns = max((sd[i] + sc[i] for i in range(P)), default=0)
        nr = max((rd[i] + rc[i] for i in range(P)), default=0)
        on_dev = torch.cuda.is_available()
        lib = L.load() if on_dev else None
        if on_dev:
            src_h = np.zeros(max(ns, 1))
            if ns:
                L.check(lib.crp_dev_memcpy(src_h.ctypes.data, send, ns * 8, 1, stream), "staged exchange: device -> host")
            L.check(lib.crp_stream_sync(stream), "stream sync")
        else:
            src_h = _np_from_ptr(send, ns, np.float64)
        parts = [src_h[sd[i]:sd[i] + sc[i]] for i in range(P)]
        src = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0))
        dst = torch.empty(sum(rc), dtype=torch.float64)
        dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        d = dst.numpy()
        if on_dev:
            off = 0
            for i in range(P):
                if rc[i]:
                    L.check(lib.crp_dev_memcpy(int(C.cast(recv, C.c_void_p).value) + rd[i] * 8, d[off:].ctypes.data, rc[i] * 8, 0, stream),
                            "staged exchange: host -> device")
                off += rc[i]
            L.check(lib.crp_stream_sync(stream), "stream sync")      # `d` must outlive the copies
        else:
            out = _np_from_ptr(recv, nr, np.float64)
            off = 0
            for i in range(P):
                out[rd[i]:rd[i] + rc[i]] = d[off:off + rc[i]]
                off += rc[i]

    @_fatal_on_error
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
    @_fatal_on_error
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

    @_fatal_on_error
    def _free(self, ptr):
        c = _live.pop(C.addressof(ptr.contents), None)
        if c is not None:
            _retired.append(c)
            c._drop_rccl()

    def free(self):
        c = _live.pop(C.addressof(self.struct), None)
        if c is not None:
            _retired.append(c)
        self._drop_rccl()

    def _drop_rccl(self):
        if self._rccl:
            L.load().crp_rccl_destroy(C.byref(self._rccl))
            self._rccl = None


class SelfComm:
    """The library's own single-rank communicator (crp_comm_self)."""

    def __init__(self):
        self.ptr = L.load().crp_comm_self()
        self.nproc, self.rank = 1, 0

    def free(self):
        if self.ptr:
            self.ptr.contents.free(self.ptr)
            self.ptr = None
