#!/usr/bin/env python3
"""One-GPU proxy for the multi-GPU target of BASELINE.json (>= 6 x at 8 GPUs over 1 GPU on nlpkkt240, n = 256).

Under the planner's 1 x P grids (A replicated once, no B exchange: crp_spmm_part2d_amortized, the reference's cost terms of
/root/reference/src/spmat_part.c:113-159 with rA applied) every GPU multiplies ALL rows by n / P columns.  T(n) / T(n / P),
measured here on one GPU for n / P = 128, 64, 32, is therefore a hard upper bound on the speed-up of that grid.  The matrix
is built once, the device matrix is created once (formats are built by the first product that needs them; their build
times are printed with CRPSPMM_TIMING=1), then every width is timed with HIP events through the device-level C ABI
(crp_spmm_csr_f64: the call rp_spmm_exec makes for device operands).

usage: width_sweep.py [--matrix kkt240] [--widths 256 128 64 32] [--steps 20] [--variant 0] [--out FILE.jsonl]
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
    ap.add_argument("--widths", type=int, nargs="+", default=[256, 128, 64, 32])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = what the library picks)")
    a = ap.parse_args()
    import torch
    import crp_spmm_amd
    import bench
    from crp_spmm_amd import gen, hip
    lib = crp_spmm_amd.load()
    dev = torch.device("cuda", 0)
    t0 = time.time()
    label, data, m, k, rp, ci, va = bench.build_matrix(a.matrix, None)
    nnz = int(rp[-1])
    print("[width_sweep %6.1f s] %s: %d rows, %d nnz" % (time.time() - t0, label, m, nnz), file=sys.stderr, flush=True)
    A = hip.CsrDev(m, k, rp, ci, va)
    print("[width_sweep %6.1f s] device matrix created" % (time.time() - t0), file=sys.stderr, flush=True)
    rows = np.repeat(np.arange(m), np.diff(rp))
    s1 = np.bincount(rows, weights=va * ci, minlength=m)
    s0 = np.bincount(rows, weights=va, minlength=m)
    del rows
    sel = np.unique(np.concatenate([np.arange(0, m, 1009), np.arange(min(4096, m)), np.arange(max(0, m - 4096), m)]))
    kcols = int(np.unique(ci).size) if nnz < (1 << 28) else k
    lines = []
    stream = torch.cuda.current_stream().cuda_stream
    for n in a.widths:
        ii = torch.arange(0, k, dtype=torch.float64, device=dev)[:, None]
        jj = torch.arange(0, n, dtype=torch.float64, device=dev)[None, :]
        B = (ii * 0.19 + jj * 0.24).contiguous()
        del ii, jj
        Cm = torch.empty((m, n), dtype=torch.float64, device=dev)
        tf = time.perf_counter()
        hip.spmm_csr(A, B, Cm, n=n, variant=a.variant, stream=stream)
        torch.cuda.synchronize()
        first = time.perf_counter() - tf
        got = Cm[torch.from_numpy(sel).to(dev)].cpu().numpy()
        expect = 0.19 * s1[sel, None] + 0.24 * np.arange(n)[None, :] * s0[sel, None]
        err = float(np.linalg.norm(got - expect) / max(np.linalg.norm(expect), 1e-300))
        assert err <= 1e-12, (n, err)
        for _ in range(3):
            hip.spmm_csr(A, B, Cm, n=n, variant=a.variant, stream=stream)
        torch.cuda.synchronize()
        ev = [(C.c_void_p(), C.c_void_p()) for _ in range(a.steps)]
        for x, y in ev:
            lib.crp_event_create(C.byref(x))
            lib.crp_event_create(C.byref(y))
        for x, y in ev:
            lib.crp_event_record(x, stream)
            hip.spmm_csr(A, B, Cm, n=n, variant=a.variant, stream=stream)
            lib.crp_event_record(y, stream)
        torch.cuda.synchronize()
        ms = C.c_float()
        per = []
        for x, y in ev:
            lib.crp_event_elapsed_ms(x, y, C.byref(ms))
            per.append(ms.value)
            lib.crp_event_destroy(x)
            lib.crp_event_destroy(y)
        kern_ms = float(np.mean(per))
        alg = gen.alg_bytes(m, kcols, n, nnz)
        rv = int(lib.crp_csr_dev_last_variant(A.handle))
        free_b, total_b = torch.cuda.mem_get_info()
        line = {"matrix": label, "rows": m, "nnz": nnz, "n": n, "kernel_variant": lib.crp_spmm_variant_name(rv).decode(),
                "ms": kern_ms, "ms_min": float(np.min(per)), "ms_max": float(np.max(per)), "GFLOP/s": 2.0 * nnz * n / (kern_ms * 1e-3) / 1e9,
                "alg_bytes": alg, "roofline_frac": alg / (kern_ms * 1e-3) / 8e12, "first_product_s": first,
                "hbm_in_use_GB": (total_b - free_b) / 1e9, "check_rel_err": err}
        lines.append(line)
        print(json.dumps(line), flush=True)
        del B, Cm
        torch.cuda.empty_cache()
    t = {l["n"]: l["ms"] for l in lines}
    nmax = max(t)
    summary = {"summary": "T(%d) / T(n): upper bound on the speed-up of a 1 x (%d / n) grid with A replicated" % (nmax, nmax),
               "ratios": {"T(%d)/T(%d)" % (nmax, n): t[nmax] / t[n] for n in sorted(t) if n != nmax}}
    print(json.dumps(summary), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            for l in lines + [summary]:
                f.write(json.dumps(l) + "\n")
    A.free()


if __name__ == "__main__":
    main()
