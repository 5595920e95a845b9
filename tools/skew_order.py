"""Experiment: processing orders for the pwtk stand-in's R = 8 panels that co-schedule the panels
sharing B rows (teeth of the +-1200 / +-36000 bands) with a time skew.  Writes int32 permutations
for CRPSPMM_PANEL_ORDER_FILE.  usage: skew_order.py OUT A B [mode]"""
import sys

import numpy as np

NP_, P1, P2 = 27240, 150, 4500          # panels, panels per 1200-row tooth, per 36000-row block
CHUNK = 3408                            # order positions per XCD (ceil(ceil(27240/4)/8)*4)


def main():
    out, A, B = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "iblock"
    U = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    p = np.arange(NP_)
    j, i, t = p // P2, (p % P2) // P1, p % P1
    if mode == "iblock":        # XCD = block of consecutive i (all j), sweep t with skewed teeth
        lin = (i * 7 + j) * P1 + t
        rank = np.argsort(np.argsort(lin, kind="stable"), kind="stable")
        xcd = rank // CHUNK
        teeth = 22.7
    elif mode == "tslab":       # XCD = slab of t (all teeth)
        lin = t * 10000 + j * 100 + i
        rank = np.argsort(np.argsort(lin, kind="stable"), kind="stable")
        xcd = rank // CHUNK
        teeth = 181.6
    elif mode == "ionly":       # one stride known (1200): teeth in natural order, 22.7 consecutive teeth per XCD
        lin = p
        rank = p
        xcd = rank // CHUNK
        teeth = 22.7
    elif mode == "jonly":       # one stride known (36000): 7 teeth of 4500 panels, XCD = slab of t
        t = p % P2
        i = 0 * p
        lin = t * 10 + j
        rank = np.argsort(np.argsort(lin, kind="stable"), kind="stable")
        xcd = rank // CHUNK
        teeth = 6.05
    elif mode == "iblock2":     # like iblock but two i-blocks per XCD processed one after the other
        lin = (i * 7 + j) * P1 + t
        rank = np.argsort(np.argsort(lin, kind="stable"), kind="stable")
        xcd = rank // CHUNK
        half = (rank % CHUNK) // (CHUNK // 2)
        teeth = 11.35
        key = A * i + B * j + teeth * t + 1e7 * half
        order = np.lexsort((p, key, xcd)).astype(np.int32)
        order.tofile(out)
        return
    elif mode.startswith("ibT"):    # ibT<k>: 8/k i-blocks x k t-slabs
        k = int(mode[3:])
        nb = 8 // k
        tslab = np.minimum(t * k // P1, k - 1)
        lin = ((tslab * 30 + i) * 7 + j) * P1 + t          # slab-major, then i-major teeth
        rank = np.argsort(np.argsort(lin, kind="stable"), kind="stable")
        xcd = rank // CHUNK
        teeth = 181.6 / nb
    else:
        raise SystemExit("mode?")
    key = A * i + B * j + teeth * (t // U) * U
    if mode == "iblock" and len(sys.argv) > 6:
        inner = sys.argv[6]
        if inner == "ij":        # inside one t: i-major
            tie = i * 100 + j
        elif inner == "2x2":     # inside one t: 2x2 blocks of (i, j)
            tie = (i // 2) * 10000 + (j // 2) * 100 + (j % 2) * 2 + (i % 2)
        else:
            tie = p
        order = np.lexsort((tie, key, xcd)).astype(np.int32)
        order.tofile(out)
        print(out, inner)
        return
    order = np.lexsort((p, key, xcd)).astype(np.int32)
    assert np.array_equal(np.sort(order), p)
    order.tofile(out)
    print(out, mode, A, B, "xcd sizes", np.bincount(xcd))


if __name__ == "__main__":
    main()
