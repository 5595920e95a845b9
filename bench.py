#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: SpMM GFLOP/s + achieved HBM GB/s, pwtk n=256.

One "step" = one C := A * B of the hot path (rp_spmm_exec / para2d_spmm_exec,
/root/reference/src/rowpara_spmm.c:212-422) with A, B and C resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 256] [--matrix pwtk | --mtx FILE]

N = 1 : BASELINE configs[1] -- pwtk x n = 256, fp64, 1 MI355X, rp_spmm HIP kernel.  The matrix is the real
        pwtk.mtx when --mtx names it or $CRPSPMM_MTX_DIR holds it (read through the library's own Matrix-Market
        ingest, examples/mmio_utils.c:11-190 restated in csrc/mmio_utils.cpp); otherwise -- no SuiteSparse files
        and no network in the containers -- the seeded stand-in gen.banded_fem(217918), and the JSON says so.
        The default run also times the IRREGULAR pwtk-class stand-in (gen.shell_fem: jittered shell mesh, 6
        unknowns per node) with the same protocol and reports it under config.also (--no-also skips it).
N > 1 : the same matrix and n, 2D grid chosen by the planner for an A that is multiplied (steps + warmup)
        times -- crp_spmm_part2d_amortized, the reference's cost terms with its "rA = times A is reused"
        applied consistently (--grid reference: the reference rule with rA = 1) --, one rank per GPU over
        torch.distributed (control plane gloo, device payloads over RCCL); strong scaling (total work fixed).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

# kernel symbol behind every variant name (what rocprofv3 --kernel-trace shows for it)
KERNEL_SYMBOL = {"csr-rowgroup": "crp::spmm_rm_f64_kernel<LPR,VW,NV>", "rowpanel-R4": "crp::spmm_panel_f64_kernel<4,...>",
                 "rowpanel-R8": "crp::spmm_panel_f64_kernel<8,...>",
                 "team2-R8": "crp::spmm_team2_kernel<double,NV,HAS_B1,COMPACT>", "team2r-R8": "crp::spmm_team2r_kernel<G,HAS_B1,2>"}


def find_mtx(args):
    """--mtx FILE, else $CRPSPMM_MTX_DIR/{pwtk.mtx,pwtk/pwtk.mtx} for --matrix pwtk; None when there is no file."""
    if args.mtx:
        return args.mtx
    d = os.environ.get("CRPSPMM_MTX_DIR")
    if d and args.matrix == "pwtk":
        for cand in (os.path.join(d, "pwtk.mtx"), os.path.join(d, "pwtk", "pwtk.mtx")):
            if os.path.exists(cand):
                return cand
    return None


def build_matrix(name, mtx=None):
    """-> (label, data, m, k, rowptr, colidx, val); data = "real" for a Matrix-Market file, else "synthetic".
    A name of the form  base@l2  is the DIAGNOSTIC variant of `base` with every column folded into the first 1024 rows of B (2 MiB
    at n = 256: B is always L2-resident) -- same rows, same teams' sizes, no B traffic beyond L2: the design's ceiling."""
    if mtx is None and name.endswith("@l2"):
        label, data, m, k, rp, ci, va = build_matrix(name[:-3], None)
        return label + " with columns mod 1024 (L2-resident B; diagnostic)", data, m, k, rp, (ci % 1024).astype(np.int32), va
    from crp_spmm_amd import gen
    if mtx is not None:
        from crp_spmm_amd import mmio
        t0 = time.time()
        m, k, rp, ci, va = mmio.read_mtx_csr(mtx, verbose=False)
        dt = time.time() - t0
        rows = np.repeat(np.arange(m, dtype=np.int64), np.diff(rp))
        bw = int(np.abs(ci.astype(np.int64) - rows).max()) if ci.size else 0
        # (the line examples/test_utils.c:43-47 prints)
        print("A size = %d * %d, nnz = %d, nnz/row = %d, bandwidth = %d" % (m, k, rp[-1], rp[-1] // max(m, 1), bw), file=sys.stderr)
        return ("%s (Matrix-Market file, ingest %.2f s, bandwidth %d)" % (os.path.basename(mtx), dt, bw), "real", m, k, rp, ci, va)
    if name == "pwtk":
        rp, ci, va = gen.banded_fem(217918)
        return "pwtk stand-in banded_fem(217918, seed 20261004)", "synthetic", 217918, 217918, rp, ci, va
    if name == "pwtk_shell":
        # irregular pwtk-class stand-in: jittered shell mesh, 6 unknowns per node, far seam band (gen.shell_fem)
        rp, ci, va = gen.shell_fem()
        return "pwtk-class stand-in shell_fem(160 x 227 nodes x 6 dof, jittered; seed 20261005)", "synthetic", 217918, 217918, rp, ci, va
    if name == "pwtk_l2":
        # diagnostic only: same row structure, every column folded into the first 1024 rows of B
        # (2 MiB at n = 256) so that B is always L2-resident
        rp, ci, va = gen.banded_fem(217918)
        return "pwtk stand-in with columns mod 1024 (L2-resident B; diagnostic)", "synthetic", 217918, 217918, rp, (ci % 1024).astype(np.int32), va
    if name == "pwtk_mall":
        # diagnostic only: the pwtk stand-in's structure at 48,000 rows -- A, B and C together (232 MB at n = 256) fit
        # the 256 MiB Infinity Cache, so back-to-back launches never reach HBM
        rp, ci, va = gen.banded_fem(48000)
        return "pwtk stand-in structure, 48000 rows (Infinity-Cache-resident; diagnostic)", "synthetic", 48000, 48000, rp, ci, va
    if name == "small":
        rp, ci, va = gen.banded_fem(20000, offsets=(1, 2, 3, 4, 5, 6, 100, 101, 3000))
        return "banded_fem(20000) smoke-size", "synthetic", 20000, 20000, rp, ci, va
    # stand-ins for the other BASELINE configs (parity / side numbers; the bench line of record is pwtk n=256)
    if name == "kkt":
        rp, ci, va = gen.kkt3d(96)
        m = len(rp) - 1
        return "nlpkkt stand-in kkt3d(96)", "synthetic", m, m, rp, ci, va
    if name == "kkt240":
        rp, ci, va = kkt240_cached()
        m = len(rp) - 1
        return "nlpkkt240-size stand-in kkt3d(241): %d rows" % m, "synthetic", m, m, rp, ci, va
    if name == "kkt240d":
        rp, ci, va = big_cached("kkt3d_241_c13", lambda: gen.kkt3d_big(241, coupling=gen._OFF13))
        m = len(rp) - 1
        return "nlpkkt240-size stand-in kkt3d(241), 13-point coupling: %d rows, %.1f nnz per row" % (m, rp[-1] / m), "synthetic", m, m, rp, ci, va
    if name == "fem3d_queen":
        rp, ci, va = big_cached("fem3d_111", lambda: gen.fem3d_big(111))
        m = len(rp) - 1
        return "Queen_4147-size stand-in fem3d(111, dof 3): %d rows" % m, "synthetic", m, m, rp, ci, va
    if name == "er16m":
        rp, ci, va = big_cached("er_2p24_32", lambda: gen.erdos_renyi(1 << 24, 1 << 24, 32, seed=1))
        return "Erdos-Renyi 2^24 x 2^24, 32 nnz/row", "synthetic", 1 << 24, 1 << 24, rp, ci, va
    if name == "fem3d":
        rp, ci, va = gen.fem3d(56)
        m = len(rp) - 1
        return "Queen stand-in fem3d(56, dof 3)", "synthetic", m, m, rp, ci, va
    if name == "er":
        rp, ci, va = gen.erdos_renyi(1 << 20, 1 << 20, 32, seed=1)
        return "Erdos-Renyi 2^20 x 2^20, 32 nnz/row", "synthetic", 1 << 20, 1 << 20, rp, ci, va
    raise SystemExit("unknown --matrix %s" % name)


def big_cached(tag, make):
    """A large generated matrix through the library's binary CSR cache (generated once per box)."""
    from crp_spmm_amd import mmio
    path = os.path.join(os.environ.get("CRPSPMM_CACHE_DIR", tempfile.gettempdir()), "crpspmm_%s.csrbin" % tag)
    got = mmio.csr_cache_read(path)
    if got is not None:
        return got[2], got[3], got[4]
    rp, ci, va = make()
    m = len(rp) - 1
    mmio.csr_cache_write(path, m, int(ci.max()) + 1 if ci.size else m, rp, ci, va)
    return rp, ci, va


def kkt240_cached():
    """nlpkkt240-size stand-in (kkt3d(241): 27,995,042 rows) through the library's binary CSR cache: generated once
    per box (minutes of numpy), afterwards read back in seconds."""
    from crp_spmm_amd import gen, mmio
    path = os.path.join(os.environ.get("CRPSPMM_CACHE_DIR", tempfile.gettempdir()), "crpspmm_kkt3d_241.csrbin")
    got = mmio.csr_cache_read(path)
    if got is not None:
        return got[2], got[3], got[4]
    rp, ci, va = gen.kkt3d_big(241)
    m = len(rp) - 1
    mmio.csr_cache_write(path, m, m, rp, ci, va)
    return rp, ci, va


def cpu_baseline(rp, ci, va, k, n, budget_s=20.0):
    """The reference's CPU path on this box's host cores: mkl_sparse_d_create_csr + mkl_sparse_d_mm +
    mkl_sparse_destroy PER CALL, as /root/reference/src/rowpara_spmm.c:398-408 does (kind "reference", engine "mkl"; run in a fresh
    process with MKL_THREADING_LAYER=GNU and all cores, oracle/mkl_baseline.py).  When libmkl_rt does not load:
    the oracle's OpenMP restatement (kind "port")."""
    cores = os.cpu_count() or 1
    m = len(rp) - 1
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "a.npz")
        np.savez(path, rp=rp, ci=ci, va=va, k=k, n=n)
        # thread counts swept (the 256-thread run of round 2 was slower than the survey's 8-vCPU probe: NUMA / affinity
        # bound); the best one is reported, the sweep is named in "sample"
        sweep = sorted({c for c in (cores, cores // 2, cores // 4, 64, 32, 16) if 1 <= c <= cores}, reverse=True)
        best, tried = None, []
        for thr in sweep:
            env = dict(os.environ, MKL_THREADING_LAYER="GNU", OMP_NUM_THREADS=str(thr), MKL_NUM_THREADS=str(thr),
                       OMP_PLACES="cores", OMP_PROC_BIND="close")
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "mkl_baseline.py"), path, str(max(2.0, budget_s / len(sweep)))],
                                   capture_output=True, text=True, env=env, timeout=240)
                if r.returncode == 0 and r.stdout.strip():
                    res = json.loads(r.stdout.strip().splitlines()[-1])
                    res["cores"] = thr
                    tried.append("%d: %.1f" % (thr, res["value"]))
                    if best is None or res["value"] > best["value"]:
                        best = res
            except Exception:
                pass
        if best is not None:
            best["sample"] += "; thread counts swept (threads: GFLOP/s) %s of %d hardware threads, best reported" % (", ".join(tried), cores)
            return best
    import oracle
    oracle.lib()
    B = oracle.fill_B(0, k, 0, n)
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


def measured_traffic(matrix, data, n, world, kernel_name, dtype="f64"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r04_traffic.json: 2 x FETCH_SIZE +
    WRITE_SIZE, gfx950 correction applied; tools/prof_pmc.sh) -- a pointer to that run, not a measurement of this one:
    returned only for the configuration, dtype and kernel it was taken on, with its source; (None, None) otherwise."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_traffic.json")) as f:
            t = json.load(f)
        for e in t["entries"]:
            if world == 1 and data == "synthetic" and e["matrix"] == matrix and e["n"] == n and e["kernel"] == kernel_name and \
               e.get("dtype", "f64") == dtype:
                return float(e["traffic_bytes_per_launch"]), "profiles/r04_traffic.json: %s" % e["source"]
    except Exception:
        pass
    return None, None


def measure_f32(args, matrix, mtx, steps, lib, torch, dev):
    """--dtype f32 (BASELINE configs[3]): the fp32 value path through the device-level C ABI (crp_spmm_csr_f32), one GPU.
    Checked against the closed form of the fp64 product at relative Frobenius error 1e-5."""
    from crp_spmm_amd import hip
    label, data, m, k, rp, ci, va = build_matrix(matrix, mtx)
    n, nnz = args.n, int(rp[-1])
    A = hip.CsrDev(m, k, rp, ci, va)
    ii = torch.arange(0, k, dtype=torch.float64, device=dev)[:, None]
    jj = torch.arange(0, n, dtype=torch.float64, device=dev)[None, :]
    B = ((ii * 0.19 + jj * 0.24) / float(k)).to(torch.float32).contiguous()       # fill_B scaled to O(1): fp32 has 24 bits
    Cmat = torch.empty((m, n), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        hip.spmm_csr_f32(A, B, Cmat, n=n, variant=args.variant if args.variant in (0, 1, 5) else 0, stream=stream)
    step()
    torch.cuda.synchronize()
    err = None
    if args.check:
        rows = np.repeat(np.arange(m), np.diff(rp))
        s1 = np.bincount(rows, weights=va * ci, minlength=m)
        s0 = np.bincount(rows, weights=va, minlength=m)
        sel = np.arange(0, m, max(1, m // 20000))
        expect = (0.19 * s1[sel, None] + 0.24 * np.arange(n)[None, :] * s0[sel, None]) / float(k)
        got = Cmat[torch.from_numpy(sel).to(dev)].cpu().numpy().astype(np.float64)
        err = float(np.linalg.norm(got - expect) / max(np.linalg.norm(expect), 1e-300))
        if not err <= 1e-5:
            raise SystemExit("fp32 result check failed, rel. Frobenius error %.3e" % err)
    for _ in range(args.warmup):
        step()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.25:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    ev = [(C.c_void_p(), C.c_void_p()) for _ in range(steps)]
    for a, b in ev:
        lib.crp_event_create(C.byref(a))
        lib.crp_event_create(C.byref(b))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        lib.crp_event_record(a, stream)
        step()
        lib.crp_event_record(b, stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = C.c_float()
    per_step = []
    for a, b in ev:
        lib.crp_event_elapsed_ms(a, b, C.byref(ms))
        per_step.append(ms.value)
        lib.crp_event_destroy(a)
        lib.crp_event_destroy(b)
    kern_ms = float(np.mean(per_step))
    from crp_spmm_amd import gen
    alg = gen.alg_bytes(m, int(np.unique(ci).size), n, nnz, vb=4)       # A once, needed B rows once, C once, fp32 values
    rv = int(lib.crp_csr_dev_last_variant(A.handle))          # what the products above actually launched
    vname = lib.crp_spmm_variant_name(rv if rv in (1, 5) else 1).decode()
    ki = {"variant": rv, "variant_name": vname, "reordered": bool(lib.crp_csr_dev_reordered(A.handle)), "lattice": bool(lib.crp_csr_dev_lattice(A.handle))}
    free_b, total_b = torch.cuda.mem_get_info()
    A.free()
    f32_traffic = measured_traffic(matrix, data, n, 1, vname, dtype="f32")
    return {"hbm_in_use_GB": (total_b - free_b) / 1e9, "label": label, "data": data, "rows": m, "nnz": nnz, "n": n, "grid": "1x1",
            "value": 2.0 * nnz * n * steps / elapsed / 1e9, "ms_per_step": elapsed / steps * 1e3, "kern_ms": kern_ms, "alg_bytes": alg,
            "achieved": alg / (kern_ms * 1e-3) / 1e9, "kernel_info": ki,
            "kernel": {"team2-R8": "crp::spmm_team2_kernel<float,NV,HAS_B1>", "csr-rowgroup": "crp::spmm_rm_f32_kernel<LPR,VW>"}.get(vname, vname),
            "traffic": f32_traffic[0], "traffic_source": f32_traffic[1], "first_exec_s": None, "check_rel_err": err,
            "step_ms_min": float(np.min(per_step)), "step_ms_max": float(np.max(per_step))}


def measure(args, matrix, mtx, steps, lib, torch, dist, comm, dev, world, rank, with_cpu):
    from crp_spmm_amd import engine, planner
    distributed = world > 1
    t_stage = time.time()

    def stage(msg):
        # progress on stderr (large workloads take minutes before the first timed step)
        nonlocal t_stage
        if rank == 0:
            print("[bench %7.1f s] %s" % (time.time() - t_stage, msg), file=sys.stderr)
            sys.stderr.flush()
    stage("building matrix %s" % matrix)
    label, data, m, k, rp, ci, va = build_matrix(matrix, mtx)
    n, nnz = args.n, int(rp[-1])
    stage("matrix ready: %d rows, %d nnz; engine init" % (m, nnz))
    flops = 2.0 * nnz * n

    # ---- partition (planner runs on every rank: deterministic, same answer everywhere)
    rb = planner.csr_mat_row_partition(rp, world)
    if distributed:
        if args.grid == "reference":
            pl = planner.calc_spmm_part2d_from_1d(world, m, n, k, rb, rp, ci, rA=1)
        elif args.grid == "timed":
            pl = planner.spmm_part2d_timed(world, m, n, k, rb, rp, ci, max(1, steps + args.warmup))
        else:
            pl = planner.spmm_part2d_amortized(world, m, n, k, rb, rp, ci, max(1, steps + args.warmup))
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
    del ii, jj
    Cmat = torch.empty((c_r1 - c_r0, n_loc), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.exec(0, B, Cmat, stream=stream)

    stage("operands resident; first exec (builds the kernel's formats)")
    t_first = time.perf_counter()
    step()                                   # (builds the kernel's formats on first use)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    stage("first exec done in %.1f s; device memory in use %.1f GB" % (t_first, torch.cuda.memory_allocated() / 1e9))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # ---- correctness guard on the local block (closed form for fill_B); not timed.  Above 2^28 result elements
    #      only sampled rows are compared (the reference's own check gives up at 2^31: examples/test_utils.c:3-19)
    err = None
    if args.check:
        rows = np.repeat(np.arange(m), np.diff(rp))
        s1 = np.bincount(rows, weights=va * ci, minlength=m)[c_r0:c_r1]
        s0 = np.bincount(rows, weights=va, minlength=m)[c_r0:c_r1]
        del rows
        step()
        torch.cuda.synchronize()
        if (c_r1 - c_r0) * n_loc <= (1 << 28):
            sel = np.arange(c_r1 - c_r0)
            got = Cmat.cpu().numpy()
        else:
            sel = np.unique(np.concatenate([np.arange(0, c_r1 - c_r0, 1009), np.arange(min(4096, c_r1 - c_r0)),
                                            np.arange(max(0, c_r1 - c_r0 - 4096), c_r1 - c_r0)]))
            got = Cmat[torch.from_numpy(sel).to(dev)].cpu().numpy()
        expect = 0.19 * s1[sel, None] + 0.24 * np.arange(col0, col1)[None, :] * s0[sel, None]
        err = float(np.linalg.norm(got - expect) / max(np.linalg.norm(expect), 1e-300))
        del got, expect
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

    stage("checked; timing %d steps" % steps)
    # ---- timed region: K steps between barrier + synchronize; HIP events per step on the launch stream
    ev = [(C.c_void_p(), C.c_void_p()) for _ in range(steps)]
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

    ms_per_step = elapsed / steps * 1e3
    kern_ms = float(np.mean(per_step))
    alg_bytes = rp_eng.alg_bytes()            # this rank's compulsory bytes per launch (DESIGN.md "bytes per unit")
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    ki = rp_eng.kernel_info()
    kname = KERNEL_SYMBOL.get(ki["variant_name"], str(ki["variant_name"]))
    if ki["variant_name"] == "rowpanel-R8" and 24 <= n // pn <= 32 and (n // pn) % 2 == 0:
        kname = "crp::spmm_narrow_f64_kernel<NP,HAS_B1,OFF32> (row-panel format, four entries per instruction)"
    traffic, tsrc = measured_traffic(matrix, data, n, world, ki["variant_name"])
    free_b, total_b = torch.cuda.mem_get_info()
    res = {
        "hbm_in_use_GB": (total_b - free_b) / 1e9,
        "label": label, "data": data, "rows": m, "nnz": nnz, "n": n, "grid": "%dx%d" % (pm, pn),
        "value": flops * steps / elapsed / 1e9, "ms_per_step": ms_per_step, "kern_ms": kern_ms, "alg_bytes": alg_bytes,
        "achieved": achieved, "kernel_info": ki, "kernel": kname, "traffic": traffic, "traffic_source": tsrc,
        "first_exec_s": t_first, "check_rel_err": err,
    }
    res["step_ms_min"], res["step_ms_max"] = float(np.min(per_step)), float(np.max(per_step))
    res["host_pointer_exec_ms"] = None
    if not distributed and args.host_exec and (c_r1 - c_r0) * n_loc <= (1 << 28):
        # the reference's own calling convention: B and C are HOST arrays (src/rowpara_spmm.h:69-81); the engine stages
        # them over PCIe.  One warm call, then the mean of three -- reported beside the device-resident rate, never as `value`.
        Bh = B.cpu().numpy()
        Ch = np.empty((c_r1 - c_r0, n_loc))
        eng.exec(0, Bh, Ch)
        t_h = time.perf_counter()
        for _ in range(3):
            eng.exec(0, Bh, Ch)
        res["host_pointer_exec_ms"] = (time.perf_counter() - t_h) / 3 * 1e3
        del Bh, Ch
    if with_cpu:
        res["cpu_baseline"] = cpu_baseline(rp, ci, va, k, n)
    if distributed:
        # every native RCCL communicator this rank holds (world, grid row, grid column): ranks and creation time
        from crp_spmm_amd import comm as crp_comm
        res["rccl"] = [{"ranks": c.device_ranks(), "create_s": c.device_create_seconds(), "issue": c.device_issue()} for c in list(crp_comm._live.values())
                       if hasattr(c, "device_ranks")]
    eng.free()
    del B, Cmat
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--matrix", default="pwtk")
    ap.add_argument("--mtx", default=None, help="Matrix-Market file to multiply instead of a generated matrix")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--dtype", default="f64", choices=("f64", "f32"), help="f32: the fp32 value path (one GPU, device-level API)")
    ap.add_argument("--grid", default="amortized", choices=("amortized", "reference", "timed"),
                    help="N > 1: planner rule for the process grid (see module docstring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the irregular pwtk-class stand-in of the default run")
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--host-exec", type=int, default=1, help="also time rp_spmm_exec with host B / C (config.host_pointer_exec_ms)")
    ap.add_argument("--sweep-variants", action="store_true", help="also time the other kernel variants (stderr)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import crp_spmm_amd
    from crp_spmm_amd import comm as crp_comm
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
    rccl_ranks = None
    rccl_info = None
    if distributed:
        crp_comm.init_process_group(device=dev.index)
        comm = crp_comm.TorchComm()
        rccl_ranks = getattr(comm, "device_ranks", lambda: None)()
    else:
        comm = crp_comm.SelfComm()

    mtx = find_mtx(args)
    if args.dtype == "f32":
        if world != 1:
            raise SystemExit("--dtype f32 is a one-GPU measurement")
        main_res = measure_f32(args, args.matrix, mtx, args.steps, lib, torch, dev)
        args.no_also = True
    else:
        main_res = measure(args, args.matrix, mtx, args.steps, lib, torch, dist, comm, dev, world, rank,
                           with_cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline))
    rccl_info = main_res.get("rccl")
    also = None
    if world == 1 and mtx is None and args.matrix == "pwtk" and args.variant == 0 and not args.no_also:
        r2 = measure(args, "pwtk_shell", None, min(args.steps, 100), lib, torch, dist, comm, dev, world, rank, with_cpu=False)
        also = {"workload": "%s x n=%d" % (r2["label"], r2["n"]), "nnz": r2["nnz"], "ms_per_step": r2["ms_per_step"],
                "GFLOP/s": r2["value"], "kernel": r2["kernel"], "roofline_frac": r2["achieved"] / HBM_PEAK_GBS,
                "kernel_ms": r2["kern_ms"], "alg_bytes": r2["alg_bytes"], "locality_order": r2["kernel_info"]["reordered"],
                "lattice_detected": r2["kernel_info"]["lattice"], "traffic": r2["traffic"], "traffic_source": r2["traffic_source"]}

    r = main_res
    what = "pwtk" if (args.matrix == "pwtk" and r["data"] == "real") else \
           ("pwtk stand-in" if args.matrix == "pwtk" else (os.path.basename(mtx) if mtx else args.matrix))
    out = {
        "metric": "SpMM GFLOP/s (%s n=%d, %s)" % (what, r["n"], "fp32" if args.dtype == "f32" else "fp64"),
        "value": r["value"], "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": r["data"],
        "config": {"workload": "%s x n=%d, rp_spmm/para2d_spmm exec, operands resident in HBM" % (r["label"], r["n"]),
                   "rows": r["rows"], "nnz": r["nnz"], "n": r["n"], "grid": r["grid"], "kernel_variant": args.variant,
                   "kernel_variant_resolved": r["kernel_info"]["variant_name"],
                   "locality_order": r["kernel_info"]["reordered"], "lattice_detected": r["kernel_info"]["lattice"],
                   "rccl_ranks": rccl_ranks, "rccl": rccl_info, "first_exec_s": r["first_exec_s"],
                   "step_ms_min": r.get("step_ms_min"), "step_ms_max": r.get("step_ms_max"),
                   "host_pointer_exec_ms": r.get("host_pointer_exec_ms"), "hbm_in_use_GB": r["hbm_in_use_GB"], "check_rel_err": r["check_rel_err"],
                   "achieved_hbm_GBs_alg": r["alg_bytes"] / (r["ms_per_step"] * 1e-3) / 1e9 if not distributed else None,
                   "also": also},
        "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": r["achieved"] / HBM_PEAK_GBS, "traffic": r["traffic"], "traffic_source": r["traffic_source"],
                     "kernel": "%s (rank 0 launch: %d algorithmic bytes, %.4f ms avg by HIP events)"
                               % (r["kernel"], r["alg_bytes"], r["kern_ms"])},
    }
    if "cpu_baseline" in r:
        out["cpu_baseline"] = r["cpu_baseline"]
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
