#!/usr/bin/env python3
"""One-GPU proxy for the 8-GPU target of BASELINE.json, per GRID (extends tools/width_sweep.py, which covers 1 x P only).

On a pm x pn grid (reference mapping: src/para2d_spmm.h:37) a GPU multiplies the rows of ONE of pm contiguous row blocks by n / pn
columns; the B rows its columns name outside its own block arrive by the per-exec exchange (src/rowpara_spmm.c:275-309).  Both
parts can be measured on one GPU: the local product -- one row block as a matrix of its own, all
columns addressable, n / pn columns wide, through the device-level C ABI the engine calls -- and the exchange volume (distinct
columns outside the block x n / pn x 8 bytes, what rp_spmm_init's plan would request: src/rowpara_spmm.c:70-118).  Printed per grid:
T_local, the halo bytes, and the speed-up bounds T(1 GPU) / T_local (exchange fully hidden behind the interior rows' product,
which the engine overlaps) and T(1 GPU) / (T_local + halo bytes / link rate) (exchange not hidden at all; --link-gbs, default
the 153 GB/s of one xGMI link: a block's halo comes from its two neighbours over two links, so this is the pessimistic end).
Row blocks are the planner's nnz-balanced contiguous blocks; per grid the block with the largest halo is measured.

usage: grid_proxy.py [--matrix kkt240] [--n 256] [--gpus 8] [--steps 10] [--out FILE.jsonl]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="kkt240")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--link-gbs", type=float, default=153.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import crp_spmm_amd
    import bench
    from crp_spmm_amd import hip
    lib = crp_spmm_amd.load()
    dev = torch.device("cuda", 0)
    t0 = time.time()
    label, data, m, k, rp, ci, va = bench.build_matrix(a.matrix, None)
    nnz = int(rp[-1])
    print("[grid_proxy %6.1f s] %s: %d rows, %d nnz" % (time.time() - t0, label, m, nnz), file=sys.stderr, flush=True)
    stream = torch.cuda.current_stream().cuda_stream
    grids = [(pm, a.gpus // pm) for pm in range(1, a.gpus + 1) if a.gpus % pm == 0 and a.n % (a.gpus // pm) == 0]
    grids = [(1, 1)] + grids
    lines = []
    for pm, pn in grids:
        # nnz-balanced contiguous row blocks, as csr_mat_row_partition makes them (src/spmat_part.c:12-36); per grid the block
        # with the LARGEST halo is measured (the exec time of a grid is its slowest rank's)
        cuts = [int(np.searchsorted(rp, nnz * b // pm)) for b in range(pm + 1)]
        cuts[0], cuts[-1] = 0, m
        best = None
        for b in range(pm):
            c0, c1 = cuts[b], cuts[b + 1]
            fl = np.zeros(k, dtype=np.bool_)
            fl[ci[int(rp[c0]):int(rp[c1])]] = True
            h = int(fl.sum()) - (int(fl[c0:c1].sum()) if m == k else 0)
            if best is None or h > best[0]:
                best = (h, b)
        blk = best[1]
        r0, r1 = cuts[blk], cuts[blk + 1]
        e0, e1 = int(rp[r0]), int(rp[r1])
        rpl = (rp[r0:r1 + 1] - rp[r0]).astype(np.int32)
        cil, val = ci[e0:e1], va[e0:e1]
        flags = np.zeros(k, dtype=np.bool_)
        flags[cil] = True
        halo = int(flags.sum()) - int(flags[r0:r1].sum()) if m == k else int(flags.sum())
        del flags
        nl = a.n // pn
        ml = r1 - r0
        A = hip.CsrDev(ml, k, rpl, cil, val)
        ii = torch.arange(0, k, dtype=torch.float64, device=dev)[:, None]
        jj = torch.arange(0, nl, dtype=torch.float64, device=dev)[None, :]
        B = (ii * 0.19 + jj * 0.24).contiguous()
        del ii, jj
        Cm = torch.empty((ml, nl), dtype=torch.float64, device=dev)
        tf = time.perf_counter()
        hip.spmm_csr(A, B, Cm, n=nl, variant=0, stream=stream)
        torch.cuda.synchronize()
        first = time.perf_counter() - tf
        # spot check against the closed form of fill_B (examples/test_utils.c:121-154)
        sel = np.unique(np.concatenate([np.arange(0, ml, 4099), np.arange(min(1024, ml)), np.arange(max(0, ml - 1024), ml)]))
        s1 = np.array([np.dot(val[rpl[i]:rpl[i + 1]], cil[rpl[i]:rpl[i + 1]].astype(np.float64)) for i in sel])
        s0 = np.array([val[rpl[i]:rpl[i + 1]].sum() for i in sel])
        got = Cm[torch.from_numpy(sel).to(dev)].cpu().numpy()
        expect = 0.19 * s1[:, None] + 0.24 * np.arange(nl)[None, :] * s0[:, None]
        err = float(np.linalg.norm(got - expect) / max(np.linalg.norm(expect), 1e-300))
        assert err <= 1e-12, (pm, pn, err)
        for _ in range(2):
            hip.spmm_csr(A, B, Cm, n=nl, variant=0, stream=stream)
        torch.cuda.synchronize()
        ev = [(C.c_void_p(), C.c_void_p()) for _ in range(a.steps)]
        for x, y in ev:
            lib.crp_event_create(C.byref(x))
            lib.crp_event_create(C.byref(y))
        for x, y in ev:
            lib.crp_event_record(x, stream)
            hip.spmm_csr(A, B, Cm, n=nl, variant=0, stream=stream)
            lib.crp_event_record(y, stream)
        torch.cuda.synchronize()
        ms = C.c_float()
        per = []
        for x, y in ev:
            lib.crp_event_elapsed_ms(x, y, C.byref(ms))
            per.append(ms.value)
            lib.crp_event_destroy(x)
            lib.crp_event_destroy(y)
        rv = int(lib.crp_csr_dev_last_variant(A.handle))
        line = {"grid": "%d x %d" % (pm, pn), "block": blk, "block_rows": ml, "block_nnz": e1 - e0, "n_local": nl, "kernel_variant": lib.crp_spmm_variant_name(rv).decode(),
                "T_local_ms": float(np.mean(per)), "ms_min": float(np.min(per)), "ms_max": float(np.max(per)), "halo_rows": halo,
                "halo_MB": halo * nl * 8 / 1e6, "exchange_ms_one_link": halo * nl * 8 / (a.link_gbs * 1e9) * 1e3, "first_product_s": first, "check_rel_err": err}
        lines.append(line)
        print(json.dumps(line), flush=True)
        A.free()
        del B, Cm, A
        torch.cuda.empty_cache()
    t1 = lines[0]["T_local_ms"]
    summary = {"summary": "speed-up bounds over 1 GPU (T = %.3f ms) for %d GPUs: exchange hidden / not hidden at %.0f GB/s" % (t1, a.gpus, a.link_gbs),
               "bounds": {l["grid"]: [round(t1 / l["T_local_ms"], 2), round(t1 / (l["T_local_ms"] + l["exchange_ms_one_link"]), 2)] for l in lines[1:]}}
    print(json.dumps(summary), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            for l in lines + [summary]:
                f.write(json.dumps(l) + "\n")


if __name__ == "__main__":
    main()
