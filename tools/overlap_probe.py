#!/usr/bin/env python3
"""Does a transport's copy kernel make progress beside the team2 product?  ("overlap may be illusory")

rp_spmm_exec launches the B exchange on a second stream beside the interior rows' product (csrc/rp_engine.cpp; the
reference overlaps nothing: /root/reference/src/rowpara_spmm.c:275-309 then :388-408).  The team kernel fills every CU
with two 512-thread workgroups (148 KiB of LDS, all 512 VGPRs of every SIMD), so a small copy kernel -- the shape of
RCCL's send/recv kernels -- may have to wait for workgroups to exit.  One GPU is enough to measure it:

  stream A: stamp | K back-to-back products of the pwtk stand-in, n = 256 | stamp
  stream B: (after a delay) a copy kernel of `--blocks` workgroups moving `--mb` MB, stamping its own start and end

with the 100 MHz device wall clock.  Reported: the copy alone, the copy beside the product (start delay after launch,
duration), and the product's slow-down; then the same with the product's stream restricted by a CU mask that leaves
`--reserve` CUs per XCD to stream B (hipExtStreamCreateWithCUMask).
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
    ap.add_argument("--blocks", type=int, default=64)
    ap.add_argument("--mb", type=float, default=64.0)
    ap.add_argument("--products", type=int, default=40)
    ap.add_argument("--reserve", type=int, default=1, help="CUs per XCD kept free of the product in the masked run")
    ap.add_argument("--matrix", default="pwtk")
    ap.add_argument("--n", type=int, default=256)
    a = ap.parse_args()
    import torch
    import crp_spmm_amd
    import bench
    from crp_spmm_amd import hip
    lib = crp_spmm_amd.load()
    dev = torch.device("cuda", 0)
    _, _, m, k, rp, ci, va = bench.build_matrix(a.matrix, None)
    A = hip.CsrDev(m, k, rp, ci, va)
    B = torch.rand((k, a.n), dtype=torch.float64, device=dev)
    Cm = torch.empty((m, a.n), dtype=torch.float64, device=dev)
    nbytes = int(a.mb * 1e6) // 16 * 16
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    stamps = torch.zeros(8, dtype=torch.int64, device=dev)        # [copy start, copy end, product start, product end, launch]

    def mk_stream(mask=None):
        s = C.c_void_p()
        if mask is None:
            assert lib.crp_stream_create(C.byref(s)) == 0
        else:
            arr = (C.c_uint * len(mask))(*mask)
            assert lib.crp_stream_create_cu_mask(C.byref(s), len(mask), arr) == 0
        return s

    def product(stream):
        hip.spmm_csr(A, B, Cm, n=a.n, variant=0, stream=stream)

    def run(sa, sb, with_product, with_copy):
        stamps.zero_()
        stamps[0] = torch.iinfo(torch.int64).max
        torch.cuda.synchronize()
        base = stamps.data_ptr()
        if with_product:
            lib.crp_probe_stamp(C.c_void_p(base + 16), sa)
            for _ in range(a.products):
                product(sa)
            lib.crp_probe_stamp(C.c_void_p(base + 24), sa)
        if with_copy:
            if with_product:
                time.sleep(0.002)                                 # let the products occupy the chip first
            lib.crp_probe_stamp(C.c_void_p(base + 32), sb)        # when stream B reached the launch
            lib.crp_probe_copy(nbytes, src.data_ptr(), dst.data_ptr(), a.blocks, C.c_void_p(base), sb)
        lib.crp_stream_sync(sa)
        lib.crp_stream_sync(sb)
        torch.cuda.synchronize()
        v = stamps.cpu().numpy().astype(np.int64)
        us = lambda x, y: (int(x) - int(y)) / 100.0
        out = {}
        if with_copy:
            out["copy_us"] = us(v[1], v[0])
            out["copy_start_after_launch_us"] = us(v[0], v[4])
            out["copy_GBs"] = nbytes / (us(v[1], v[0]) * 1e-6) / 1e9
        if with_product:
            out["product_ms_each"] = us(v[3], v[2]) / 1e3 / a.products
            if with_copy:
                out["copy_start_after_products_start_us"] = us(v[0], v[2])
                out["copy_end_before_products_end_us"] = us(v[3], v[1])
        return out

    sa, sb = mk_stream(), mk_stream()
    product(sa)                                                    # formats, clocks
    lib.crp_stream_sync(sa)
    for _ in range(3):
        run(sa, sb, True, True)
    res = {"config": {"matrix": a.matrix, "n": a.n, "copy_blocks": a.blocks, "copy_MB": nbytes / 1e6, "products": a.products}}
    res["copy_alone"] = run(sa, sb, False, True)
    res["product_alone"] = run(sa, sb, True, False)
    res["both_unreserved"] = run(sa, sb, True, True)
    # the product's stream without `reserve` CUs of every XCD.  CU numbering of the mask: bit i = CU i of the device list;
    # the XCDs interleave (CU i sits on XCD i % 8), so CUs 0 .. 8 * reserve - 1 are `reserve` per XCD
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    mask = [0xFFFFFFFF] * words
    for cu in range(8 * a.reserve):
        mask[cu // 32] &= ~(1 << (cu % 32))
    try:
        sm = mk_stream(mask)
        product(sm)
        lib.crp_stream_sync(sm)
        res["product_alone_masked"] = run(sm, sb, True, False)
        res["both_reserved"] = run(sm, sb, True, True)
        res["config"]["reserve_cus_per_xcd"] = a.reserve
    except AssertionError:
        res["both_reserved"] = "hipExtStreamCreateWithCUMask failed"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
