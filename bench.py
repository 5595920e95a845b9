#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: SpMM GFLOP/s + achieved HBM GB/s, pwtk n=256.

One "step" = one C := A * B of the hot path (rp_spmm_exec / para2d_spmm_exec,
/root/reference/src/rowpara_spmm.c:212-422) with A, B and C resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 256] [--matrix pwtk]

N = 1 : BASELINE configs[1] -- pwtk (seeded stand-in, gen.banded_fem(217918): no
        SuiteSparse files and no network in the containers) x n = 256, fp64,
        1 MI355X, rp_spmm HIP kernel.
N > 1 : the same matrix and n, 2D grid chosen by the planner for an A that is
        multiplied (steps + warmup) times -- crp_spmm_part2d_amortized, the
        reference's cost terms with its "rA = times A is reused" applied
        consistently (--grid reference: the reference rule with rA = 1) --, one
        rank per GPU over torch.distributed (control plane gloo, B exchange
        nccl == RCCL); strong scaling (total work fixed).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def build_matrix(name):
    from crp_spmm_amd import gen
    if name == "pwtk":
        rp, ci, va = gen.banded_fem(217918)
        return "pwtk-standin banded_fem(217918, seed 20261004)", 217918, 217918, rp, ci, va
    if name == "pwtk_shell":
        # irregular pwtk-class stand-in: jittered shell mesh, 6 unknowns per node, far seam band (gen.shell_fem)
        rp, ci, va = gen.shell_fem()
        return "pwtk-class shell_fem(160 x 227 nodes x 6 dof, jittered; seed 20261005)", 217918, 217918, rp, ci, va
    if name == "pwtk_shell_rcm":
        # diagnostic: the same matrix with its rows permuted on the host by reverse Cuthill-McKee on the graph of
        # row groups with identical column sets (what the locality reordering at create does on the device side)
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        rp, ci, va = gen.shell_fem()
        m = len(rp) - 1
        rows = np.repeat(np.arange(m), np.diff(rp))
        h = np.zeros(m, dtype=np.uint64)
        np.add.at(h, rows, ci.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        new = np.ones(m, dtype=bool)
        new[1:] = (h[1:] != h[:-1]) | (np.diff(rp)[1:] != np.diff(rp)[:-1])
        sn = np.cumsum(new) - 1
        ns = int(sn[-1]) + 1
        S = sp.csr_matrix((np.ones(m), (np.arange(m), sn)), shape=(m, ns))
        A = sp.csr_matrix((np.ones(len(ci), dtype=np.int8), ci, rp), shape=(m, m))
        Q = (S.T @ A @ S)
        Q = ((Q + Q.T) != 0).astype(np.int8).tocsr()
        perm = np.asarray(reverse_cuthill_mckee(Q, symmetric_mode=True))
        order = np.argsort(np.argsort(perm)[sn], kind="stable")          # rows grouped by supernode in perm order
        Ap = sp.csr_matrix((va, ci, rp), shape=(m, m))[order]
        Ap.sort_indices()
        return "pwtk-class shell_fem, rows RCM-ordered on the host (diagnostic)", m, m, Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32), Ap.data
    if name == "pwtk_l2":
        # diagnostic only: same row structure, every column folded into the first 1024 rows of B
        # (2 MiB at n = 256) so that B is always L2-resident
        rp, ci, va = gen.banded_fem(217918)
        return "pwtk-standin with columns mod 1024 (L2-resident B; diagnostic)", 217918, 217918, rp, (ci % 1024).astype(np.int32), va
    if name == "small":
        rp, ci, va = gen.banded_fem(20000, offsets=(1, 2, 3, 4, 5, 6, 100, 101, 3000))
        return "banded_fem(20000) smoke-size", 20000, 20000, rp, ci, va
    # stand-ins for the other BASELINE configs at sizes one GPU builds in seconds (parity / side numbers only;
    # the bench line of record is pwtk n=256)
    if name == "kkt":
        rp, ci, va = gen.kkt3d(96)
        m = len(rp) - 1
        return "nlpkkt-standin kkt3d(96)", m, m, rp, ci, va
    if name == "fem3d":
        rp, ci, va = gen.fem3d(56)
        m = len(rp) - 1
        return "Queen-standin fem3d(56, dof 3)", m, m, rp, ci, va
    if name == "er":
        rp, ci, va = gen.erdos_renyi(1 << 20, 1 << 20, 32, seed=1)
        return "Erdos-Renyi 2^20 x 2^20, 32 nnz/row", 1 << 20, 1 << 20, rp, ci, va
    raise SystemExit("unknown --matrix %s" % name)


def cpu_baseline(rp, ci, va, k, n, budget_s=12.0):
    """The oracle's OpenMP restatement (kind 'port') timed on this box's host cores."""
    import oracle
    oracle.lib()
    B = oracle.fill_B(0, k, 0, n)
    m = len(rp) - 1
    cores = os.cpu_count() or 1
    os.environ["OMP_NUM_THREADS"] = str(cores)
    oracle.spmm_csr(rp, ci, va, B, fast=True)            # warm-up
    reps, t0 = 0, time.time()
    while True:
        oracle.spmm_csr(rp, ci, va, B, fast=True)
        reps += 1
        if time.time() - t0 > budget_s or reps >= 50:
            break
    dt = (time.time() - t0) / reps
    return {"value": 2.0 * len(ci) * n / dt / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": "full workload (%d rows, %d nnz, n=%d), %d reps of the oracle's OpenMP CSR loop, %.3f s each"
                      % (m, len(ci), n, reps, dt)}


def measured_traffic(args, world):
    """HBM bytes per launch from the committed rocprofv3 PMC pass (profiles/r01_traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction applied) -- only for the configuration it was
    measured on; null otherwise."""
    try:
        if world != 1 or args.matrix != "pwtk" or args.n != 256 or args.variant != 0:
            return None
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return json.load(f)["traffic_bytes_per_launch"] / 1e9 * 1e9
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--matrix", default="pwtk")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--grid", default="amortized", choices=("amortized", "reference"),
                    help="N > 1: planner rule for the process grid (see module docstring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--sweep-variants", action="store_true", help="also time kernel variants 1..3 (stderr)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import crp_spmm_amd
    from crp_spmm_amd import comm as crp_comm, engine, planner
    lib = crp_spmm_amd.load()            # fails loudly when the HIP library is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)"
                         % (args.gpus, world))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    distributed = world > 1
    if distributed:
        crp_comm.init_process_group(device=dev.index)
        comm = crp_comm.TorchComm()
    else:
        comm = crp_comm.SelfComm()

    label, m, k, rp, ci, va = build_matrix(args.matrix)
    n, nnz = args.n, int(rp[-1])
    flops = 2.0 * nnz * n

    # ---- partition (planner runs on every rank: deterministic, same answer everywhere)
    rb = planner.csr_mat_row_partition(rp, world)
    if distributed:
        if args.grid == "reference":
            pl = planner.calc_spmm_part2d_from_1d(world, m, n, k, rb, rp, ci, rA=1)
        else:
            pl = planner.spmm_part2d_amortized(world, m, n, k, rb, rp, ci, max(1, args.steps + args.warmup))
        pm, pn = pl["pm"], pl["pn"]
        a0, br, ac, bc = pl["A0_rowptr"], pl["B_rowptr"], pl["AC_rowptr"], pl["BC_colptr"]
        s, e_ = int(a0[rank]), int(a0[rank + 1])
        eng = engine.Para2dSpmm(comm, pm, pn, a0, br, ac, bc, rp[s:e_ + 1], ci[rp[s]:rp[e_]], va[rp[s]:rp[e_]])
        rp_eng = eng.rp
        pi, pj = rank // pn, rank % pn
        b_r0, b_r1, c_r0, c_r1 = int(br[pi]), int(br[pi + 1]), int(ac[pi]), int(ac[pi + 1])
        col0, col1 = int(bc[pj]), int(bc[pj + 1])
    else:
        pm = pn = 1
        eng = engine.RpSpmm(0, m, rp, ci, va, [0, k], n, comm)
        rp_eng = eng
        b_r0, b_r1, c_r0, c_r1, col0, col1 = 0, k, 0, m, 0, n
    rp_eng.set_timing(False)
    rp_eng.set_variant(args.variant)
    n_loc = col1 - col0

    # ---- operands resident in HBM: B = fill_B(0.19, 0.24) (examples/test_utils.c:121-154)
    ii = torch.arange(b_r0, b_r1, dtype=torch.float64, device=dev)[:, None]
    jj = torch.arange(col0, col1, dtype=torch.float64, device=dev)[None, :]
    B = (ii * 0.19 + jj * 0.24).contiguous()
    Cmat = torch.empty((c_r1 - c_r0, n_loc), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.exec(0, B, Cmat, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # ---- correctness guard on the local block (closed form for fill_B); not timed
    if args.check:
        rows = np.repeat(np.arange(m), np.diff(rp))
        s1 = np.bincount(rows, weights=va * ci, minlength=m)[c_r0:c_r1]
        s0 = np.bincount(rows, weights=va, minlength=m)[c_r0:c_r1]
        expect = 0.19 * s1[:, None] + 0.24 * np.arange(col0, col1)[None, :] * s0[:, None]
        step()
        torch.cuda.synchronize()
        got = Cmat.cpu().numpy()
        err = np.linalg.norm(got - expect) / max(np.linalg.norm(expect), 1e-300)
        if not err <= 1e-12:
            raise SystemExit("rank %d: result check failed, rel. Frobenius error %.3e" % (rank, err))

    if args.sweep_variants and not distributed:
        for v in range(1, lib.crp_spmm_variant_count()):
            rp_eng.set_variant(v)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print("variant %d (%s): %.4f ms/step wall" % (v, lib.crp_spmm_variant_name(v).decode(), dt * 1e3),
                  file=sys.stderr)
        rp_eng.set_variant(args.variant)
        step()
        torch.cuda.synchronize()

    # ---- spin the clocks up: after the host-side check the GPU has idled and a handful of warm-up
    #      steps (a few ms) is not enough for it to leave the low-power state; untimed
    if distributed and pm > 1:
        # a step holds a collective (the B exchange): every rank must run the same number of them
        for _ in range(300):
            step()
        torch.cuda.synchronize()
    else:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.25:
            for _ in range(20):
                step()
            torch.cuda.synchronize()

    # ---- timed region: K steps between barrier + synchronize; HIP events per step on the launch stream
    ev = [(C.c_void_p(), C.c_void_p()) for _ in range(args.steps)]
    for a, b in ev:
        lib.crp_event_create(C.byref(a))
        lib.crp_event_create(C.byref(b))
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        lib.crp_event_record(a, stream)
        step()
        lib.crp_event_record(b, stream)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    ms = C.c_float()
    per_step = []
    for a, b in ev:
        lib.crp_event_elapsed_ms(a, b, C.byref(ms))
        per_step.append(ms.value)
        lib.crp_event_destroy(a)
        lib.crp_event_destroy(b)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    ms_per_step = elapsed / args.steps * 1e3
    value = flops * args.steps / elapsed / 1e9
    kern_ms = float(np.mean(per_step))
    alg_bytes = rp_eng.alg_bytes()            # this rank's compulsory bytes per launch (DESIGN.md "bytes per unit")
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": "SpMM GFLOP/s (pwtk n=%d, fp64)" % n if args.matrix == "pwtk" else "SpMM GFLOP/s (%s n=%d)" % (args.matrix, n),
        "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s x n=%d, rp_spmm/para2d_spmm exec, operands resident in HBM" % (label, n),
                   "rows": m, "nnz": nnz, "n": n, "grid": "%dx%d" % (pm, pn), "kernel_variant": args.variant,
                   "achieved_hbm_GBs_alg": alg_bytes * (world if distributed else 1) / (ms_per_step * 1e-3) / 1e9
                   if not distributed else None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, world),
                     "kernel": "spmm_rm_f64 (rank 0 launch: %d algorithmic bytes, %.4f ms avg by HIP events)"
                               % (alg_bytes, kern_ms)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rp, ci, va, k, n)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    eng.free()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
