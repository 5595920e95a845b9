"""Worker for tests/test_gpu_dist.py: N ranks sharing ONE GPU (rehearsal of the N > 1 device
path: RCCL refuses two ranks on one device, so device payloads are staged through the host and
gloo -- every kernel, plan and buffer is the real one)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import oracle as orc
    from crp_spmm_amd import comm as crp_comm, engine, gen, planner

    # one-GPU rehearsal (CRPSPMM_EXCHANGE=host: every rank on device 0, payloads staged through the host), or -- on a node
    # with a GPU per rank, tests/test_gpu_dist.py::test_engines_native_rccl_multi_gpu -- the native RCCL transport
    native = os.environ.get("CRPSPMM_EXPECT_NATIVE_RCCL") == "1"
    idev = int(os.environ.get("LOCAL_RANK", "0")) if native else 0
    torch.cuda.set_device(idev)
    dev = torch.device("cuda", idev)
    crp_comm.init_process_group(device=idev if native else None)
    assert crp_comm.exchange_mode() == ("nccl" if native else "host")
    world = crp_comm.TorchComm()
    if native:
        assert world.device_ranks() == world.nproc, "the native RCCL communicator did not come up"
    P, me = world.nproc, world.rank
    m = k = 6000
    rp, ci, va = gen.banded_fem(m, offsets=(1, 2, 3, 4, 50, 51, 1400), seed=5)
    for n in (24, 256):
        B = orc.fill_B(0, k, 0, n)
        C_ref = orc.spmm_csr(rp, ci, va, B)
        rb = planner.csr_mat_row_partition(rp, P)
        # ---- 1D engine, layout 0 and 1, device operands
        s, e = int(rb[me]), int(rb[me + 1])
        eng = engine.RpSpmm(s, e - s, rp[s:e + 1], ci[rp[s]:rp[e]], va[rp[s]:rp[e]], rb, n, world)
        Bd = torch.from_numpy(B[s:e].copy()).to(dev)
        Cd = torch.full((e - s, n), float("nan"), dtype=torch.float64, device=dev)
        eng.exec(0, Bd, Cd)
        eng.exec(0, Bd, Cd)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(C_ref[s:e], Cd.cpu().numpy()) <= 1e-12, (me, n, "1D rm")
        # exchange / compute overlap: rows with no column owned by a peer were split off (when there
        # are enough of them to pay for a launch: middle ranks of this matrix have none)
        lo_c = np.minimum.reduceat(ci[rp[s]:rp[e]], rp[s:e] - rp[s])
        hi_c = np.maximum.reduceat(ci[rp[s]:rp[e]], rp[s:e] - rp[s])
        exp_int = int(np.count_nonzero((lo_c >= s) & (hi_c < e)))
        exp = (exp_int, e - s - exp_int) if (exp_int >= (e - s) // 16 and exp_int < e - s) else (0, 0)
        n_int, n_bnd = eng.overlap_rows()
        assert (n_int, n_bnd) == exp, (me, n_int, n_bnd, exp)
        if P == 2:
            assert n_int > 0 and n_bnd > 0
        C_seq = Cd.cpu().numpy().copy()             # timing on: exchange, then the two products, in sequence
        eng.set_timing(False)                       # asynchronous path: exchange beside the interior rows
        Cd.fill_(float("nan"))
        for _ in range(3):
            eng.exec(0, Bd, Cd)
        torch.cuda.synchronize()
        assert np.array_equal(Cd.cpu().numpy(), C_seq), (me, n, "overlapped exec differs from the sequential one")
        eng.update_values(va[rp[s]:rp[e]] * 2.0)
        eng.exec(0, Bd, Cd)
        torch.cuda.synchronize()
        assert orc.rel_fro_err(2.0 * C_ref[s:e], Cd.cpu().numpy()) <= 1e-12, (me, n, "overlap + update_values")
        eng.update_values(va[rp[s]:rp[e]])
        eng.set_timing(True)
        Ch = np.full((n, e - s), np.nan)
        eng.exec(1, np.ascontiguousarray(B[s:e].T), Ch)            # host pointers, column-major
        assert orc.rel_fro_err(C_ref[s:e], Ch.T) <= 1e-12, (me, n, "1D cm host")
        eng.print_stat()
        eng.free()
        # ---- 2D engine on every grid of P ranks
        for pn in [d for d in range(1, P + 1) if P % d == 0]:
            pm = P // pn
            ac = np.array([rb[i * pn] for i in range(pm + 1)], dtype=np.int32)
            a0 = np.zeros(P + 1, dtype=np.int32)
            for i in range(pm):
                a0[i * pn:(i + 1) * pn + 1] = planner.csr_mat_row_partition(rp[ac[i]:ac[i + 1] + 1] - rp[ac[i]], pn) + ac[i]
            bc = planner.even_displs(n, pn)
            pi, pj = me // pn, me % pn
            s0, e0 = int(a0[me]), int(a0[me + 1])
            e2 = engine.Para2dSpmm(world, pm, pn, a0, ac, ac, bc, rp[s0:e0 + 1], ci[rp[s0]:rp[e0]], va[rp[s0]:rp[e0]])
            # the panel was replicated between device buffers (staged all-gather in this rehearsal mode, RCCL with a
            # GPU per rank) whenever a grid row has more than one rank and something to gather
            assert e2.replicated_on_device == (pn > 1), (pm, pn, e2.replicated_on_device)
            # the values gathered between devices fill the engine's matrices: no second upload (src/para2d_spmm.c:56-86,111-125)
            assert e2.value_uploads == (0 if pn > 1 else 1), (pm, pn, e2.value_uploads)
            e2.rp.set_timing(False)
            Bl = torch.from_numpy(np.ascontiguousarray(B[ac[pi]:ac[pi + 1], bc[pj]:bc[pj + 1]])).to(dev)
            Cl = torch.full((int(ac[pi + 1] - ac[pi]), int(bc[pj + 1] - bc[pj])), float("nan"), dtype=torch.float64, device=dev)
            for _ in range(3):
                e2.exec(0, Bl, Cl)
            torch.cuda.synchronize()
            assert orc.rel_fro_err(C_ref[ac[pi]:ac[pi + 1], bc[pj]:bc[pj + 1]], Cl.cpu().numpy()) <= 1e-12, (me, n, pm, pn)
            e2.print_stat()
            e2.free()
            dist.barrier()
    # ---- a stride-lattice matrix: the local matrices (whole, or the interior / boundary row subsets
    #      of the overlap split) get the team schedule, with remote columns inside the teams
    nx, ny, nz = 300, 8, 6
    ml = nx * ny * nz
    rpl, cil, val = gen.banded_fem(ml, offsets=(1, 2, 3, nx, nx + 1, nx * ny, nx * ny + 1), seed=7)
    rbl = planner.csr_mat_row_partition(rpl, P)
    s, e = int(rbl[me]), int(rbl[me + 1])
    for n in (256, 96):
        Bl = orc.fill_B(0, ml, 0, n)
        Cl_ref = orc.spmm_csr(rpl[s:e + 1] - rpl[s], cil[rpl[s]:rpl[e]], val[rpl[s]:rpl[e]], Bl, fast=True)
        eng = engine.RpSpmm(s, e - s, rpl[s:e + 1], cil[rpl[s]:rpl[e]], val[rpl[s]:rpl[e]], rbl, n, world)
        Bd = torch.from_numpy(Bl[s:e].copy()).to(dev)
        Cd = torch.full((e - s, n), float("nan"), dtype=torch.float64, device=dev)
        for timing in (True, False):
            eng.set_timing(timing)
            for variant in (0, 3, 5, 1):
                eng.set_variant(variant)
                Cd.fill_(float("nan"))
                eng.exec(0, Bd, Cd)
                torch.cuda.synchronize()
                assert orc.rel_fro_err(Cl_ref, Cd.cpu().numpy()) <= 1e-12, (me, n, timing, variant, "lattice")
        eng.free()
        dist.barrier()
    # ---- mat_redist with device-resident blocks (dev_type 1: staged through pinned host memory;
    #      2: device to device through the communicator's device all-to-all) against the fixture
    #      produced by the reference's own engine
    import json
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "mat_redist_P%d.json" % P)))
    M, N = fx["M"], fx["N"]
    G = np.arange(M, dtype=np.float64)[:, None] * 4096.0 + np.arange(N, dtype=np.float64)[None, :]
    for si, sc in enumerate(fx["scenarios"]):
        r = sc[me]
        exp = fx["expected"][str(si)][str(me)]
        for dev_type in (1, 2):
            e = engine.MatRedist(*r, world, dt_size=8, dev_type=dev_type)
            src = torch.zeros((max(r[2], 1), r[3] + 1), dtype=torch.float64, device=dev)
            if r[2] and r[3]:
                src[:r[2], :r[3]] = torch.from_numpy(G[r[0]:r[0] + r[2], r[1]:r[1] + r[3]].copy()).to(dev)
            dst = torch.full((max(r[6], 1), r[7] + 2), -1.0, dtype=torch.float64, device=dev)
            e.exec(src, dst)
            torch.cuda.synchronize()
            got = dst.cpu().numpy()
            assert list(got[:r[6], :r[7]].reshape(-1)) == [float(x) for x in exp["dst"]], (si, me, dev_type)
            assert (got[:r[6], r[7]:] == -1).all()
            e.free()
    dist.barrier()
    # ---- the older all-in-one engine: A in nnz-balanced row blocks, B / C on a P x 1 or 1 x P grid
    #      of the CALLER's choosing, C gathered whole on rank 0 in the second pass
    for (mm, nn, offs) in ((3000, 24, (1, 2, 3, 30)), (1200, 512, (1, 599))):
        rp2, ci2, va2 = gen.banded_fem(mm, offsets=offs, seed=2)
        B2 = orc.fill_B(0, mm, 0, nn)
        C2 = orc.spmm_csr(rp2, ci2, va2, B2)
        rb2 = planner.csr_mat_row_partition(rp2, P)
        s, e = int(rb2[me]), int(rb2[me + 1])
        for gather in (False, True):
            bs, bn = planner.calc_block_spos_size(mm, P, me)            # B: row blocks
            cs, cn = planner.calc_block_spos_size(nn, P, me)            # C: column blocks
            dst = (0, mm if me == 0 else 0, 0, nn if me == 0 else 0) if gather else (0, mm, cs, cn)
            ce = engine.CrpspmmEngine(mm, nn, mm, s, e - s, rp2[s:e + 1], ci2[rp2[s]:rp2[e]], bs, bn, 0, nn, *dst, world)
            Cout = np.full((max(dst[1], 1), max(dst[3], 1)), np.nan)
            for trial in range(2):
                ce.exec(va2[rp2[s]:rp2[e]] * (trial + 1), np.ascontiguousarray(B2[bs:bs + bn]), Cout)
                want = C2[dst[0]:dst[0] + dst[1], dst[2]:dst[2] + dst[3]] * (trial + 1)
                if want.size:
                    assert orc.rel_fro_err(want, Cout[:dst[1], :dst[3]]) <= 1e-12, (me, mm, nn, gather, trial)
            ce.print_stat()
            ce.free()
            dist.barrier()
    if me == 0:
        print("GPU_DIST_WORKER_OK world=%d" % P)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
