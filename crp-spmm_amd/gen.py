"""Seeded synthetic matrices standing in for the SuiteSparse inputs of
BASELINE.json (no .mtx files and no network in the build or GPU containers;
SURVEY.md section 8d).  Everything is a pure function of its arguments: values
come from a counter-based hash (splitmix64), not from RNG state."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(key, seed):
    h = _mix(key.astype(np.uint64) ^ _mix(np.uint64(seed) + np.zeros(1, dtype=np.uint64)))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


PWTK_OFFSETS = tuple(range(1, 15)) + tuple(range(1200, 1206)) + tuple(range(36000, 36006))


def banded_fem(m, offsets=PWTK_OFFSETS, seed=20261004):
    """Symmetric banded FEM-like matrix: entries at |i - j| in `offsets` plus the
    diagonal; off-diagonal values U(-1, 1) (symmetric), diagonal 30 + U(0, 1).
    banded_fem(217918) is the pwtk stand-in: 217,918 rows, 11,102,984 nnz (~51/row).
    -> (rowptr int32, colidx int32, val float64), columns ascending per row."""
    offs = np.array(sorted([-d for d in offsets] + [0] + list(offsets)), dtype=np.int64)
    rows = np.arange(m, dtype=np.int64)
    rowptr = np.zeros(m + 1, dtype=np.int64)
    cols_parts, vals_parts = [], []
    chunk = 1 << 16
    for r0 in range(0, m, chunk):
        r = rows[r0:r0 + chunk]
        cc = r[:, None] + offs[None, :]
        ok = (cc >= 0) & (cc < m)
        rowptr[r0 + 1:r0 + 1 + r.size] = ok.sum(axis=1)
        ri = np.broadcast_to(r[:, None], cc.shape)[ok]
        ci = cc[ok]
        lo, d = np.minimum(ri, ci), np.abs(ri - ci)
        u = _u01(lo * np.int64(65536 * 4) + d, seed)
        v = np.where(d == 0, 30.0 + u, 2.0 * u - 1.0)
        cols_parts.append(ci.astype(np.int32))
        vals_parts.append(v)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    return rowptr, np.concatenate(cols_parts), np.concatenate(vals_parts)


def erdos_renyi(m, k, deg, seed=1):
    """deg nonzeros per row, columns i.i.d. uniform (duplicates kept), ascending per row."""
    idx = np.arange(m * deg, dtype=np.uint64)
    cols = (_mix(idx ^ _mix(np.uint64(seed) + np.zeros(1, dtype=np.uint64))) % np.uint64(k)).astype(np.int32)
    cols = np.sort(cols.reshape(m, deg), axis=1).reshape(-1)
    vals = 2.0 * _u01(idx + np.uint64(1 << 40), seed) - 1.0
    rowptr = (np.arange(m + 1, dtype=np.int64) * deg).astype(np.int32)
    return rowptr, cols, vals


def _stencil_csr(shape_rows, blocks, seed):
    """Assemble a CSR from stencil blocks.  blocks: list of (row_lo, row_hi, g, dof_r, col_base, dof_c,
    offsets) -- rows [row_lo, row_hi) are the dof_r unknowns of the g^3 grid nodes in node-major
    order; each couples to the dof_c unknowns (starting at column col_base) of the neighbour nodes
    at the (dx, dy, dz) offsets that stay inside the grid.  Values: hash of (row, col), diagonal
    boosted; columns ascending per row."""
    rows_all, cols_all = [], []
    for (row_lo, g, dof_r, col_base, dof_c, offsets) in blocks:
        nn = g * g * g
        node = np.arange(nn, dtype=np.int64)
        x, y, z = node % g, (node // g) % g, node // (g * g)
        for (dx, dy, dz) in offsets:
            ok = (x + dx >= 0) & (x + dx < g) & (y + dy >= 0) & (y + dy < g) & (z + dz >= 0) & (z + dz < g)
            src = node[ok]
            dst = src + dx + g * dy + g * g * dz
            for a in range(dof_r):
                for b in range(dof_c):
                    rows_all.append(row_lo + src * dof_r + a)
                    cols_all.append(col_base + dst * dof_c + b)
    rows = np.concatenate(rows_all)
    cols = np.concatenate(cols_all)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowptr = np.zeros(shape_rows + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
    u = _u01(lo * np.int64(1 << 26) + hi, seed)                      # symmetric values
    vals = np.where(rows == cols, 40.0 + u, 2.0 * u - 1.0)
    return rowptr.astype(np.int32), cols.astype(np.int32), vals


_OFF7 = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
_OFF27 = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]


def kkt3d(g, seed=3):
    """nlpkkt-like stand-in (BASELINE configs[2]; nlpkkt240 itself is 2*240^3 + ... rows): the KKT matrix
    [[H, J^T], [J, 0]] of a PDE-constrained problem on a g^3 grid -- H a 27-point stencil on the
    N = g^3 state unknowns, J a 7-point stencil coupling them to N multipliers; 2N rows, ~20 nnz/row,
    symmetric, zero (2,2) block."""
    N = g * g * g
    return _stencil_csr(2 * N, [(0, g, 1, 0, 1, _OFF27), (0, g, 1, N, 1, _OFF7), (N, g, 1, 0, 1, _OFF7)], seed)


_OFF13 = _OFF7 + [(1, 1, 0), (-1, -1, 0), (1, -1, 0), (-1, 1, 0), (0, 1, 1), (0, -1, -1)]


def kkt3d_big(g, seed=3, chunk=1 << 21, coupling=None):
    """The same matrix as kkt3d(g), built without a global sort (kkt3d(241), the nlpkkt240-size stand-in, has 574 M
    nonzeros): the stencil offsets are taken in ascending linear order, so every row's columns come out ascending and
    the CSR is written chunk by chunk.  -> (rowptr int32, colidx int32, val float64); identical to kkt3d(g)."""
    N = g * g * g
    assert (27 + 2 * (7 if coupling is None else len(coupling))) * N < 2 ** 31, "nnz must fit the int32 row pointer of the reference's CSR"

    def offs(pattern):
        o = sorted(pattern, key=lambda d: d[0] + g * d[1] + g * g * d[2])
        return np.array(o, dtype=np.int64)
    # coupling = the stencil of J (default 7 points: 20.4 nonzeros per row; _OFF13: 26.5, the density of nlpkkt240 -- SURVEY 8d)
    o27, o7 = offs(_OFF27), offs(_OFF7 if coupling is None else coupling)
    lin27 = o27[:, 0] + g * o27[:, 1] + g * g * o27[:, 2]
    lin7 = o7[:, 0] + g * o7[:, 1] + g * g * o7[:, 2]
    rowptr = np.zeros(2 * N + 1, dtype=np.int64)
    cols_parts, vals_parts = [], []

    def block(row0, nodes, parts):
        """rows row0 + (nodes - nodes[0]); parts = [(offset triples, linear offsets, column base)]"""
        x, y, z = nodes % g, (nodes // g) % g, nodes // (g * g)
        cs, oks = [], []
        for (o3, lin, base) in parts:
            ok = ((x[:, None] + o3[None, :, 0] >= 0) & (x[:, None] + o3[None, :, 0] < g) &
                  (y[:, None] + o3[None, :, 1] >= 0) & (y[:, None] + o3[None, :, 1] < g) &
                  (z[:, None] + o3[None, :, 2] >= 0) & (z[:, None] + o3[None, :, 2] < g))
            cs.append(nodes[:, None] + lin[None, :] + base)
            oks.append(ok)
        cc, ok = np.concatenate(cs, axis=1), np.concatenate(oks, axis=1)
        rowptr[row0 + 1:row0 + 1 + nodes.size] = ok.sum(axis=1)
        rows = np.broadcast_to((row0 + np.arange(nodes.size, dtype=np.int64))[:, None], cc.shape)[ok]
        cols = cc[ok]
        lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
        u = _u01(lo * np.int64(1 << 26) + hi, seed)
        cols_parts.append(cols.astype(np.int32))
        vals_parts.append(np.where(rows == cols, 40.0 + u, 2.0 * u - 1.0))

    for n0 in range(0, N, chunk):
        nodes = np.arange(n0, min(N, n0 + chunk), dtype=np.int64)
        block(n0, nodes, [(o27, lin27, 0), (o7, lin7, N)])                 # [H  J^T]
    for n0 in range(0, N, chunk):
        nodes = np.arange(n0, min(N, n0 + chunk), dtype=np.int64)
        block(N + n0, nodes, [(o7, lin7, 0)])                              # [J  0]
    return np.cumsum(rowptr).astype(np.int32), np.concatenate(cols_parts), np.concatenate(vals_parts)


def fem3d_big(g, dof=3, seed=5, chunk=1 << 18):
    """The same matrix as fem3d(g, dof), written chunk by chunk without a global sort (fem3d_big(111) is the Queen_4147-size
    stand-in: 4,102,893 rows, ~327 M nonzeros): neighbour offsets in ascending linear order and unknowns in ascending
    order give ascending columns.  -> (rowptr int32, colidx int32, val float64); identical to fem3d(g, dof)."""
    N = g * g * g
    o = np.array(sorted(_OFF27, key=lambda d: d[0] + g * d[1] + g * g * d[2]), dtype=np.int64)
    lin = o[:, 0] + g * o[:, 1] + g * g * o[:, 2]
    assert 27 * dof * dof * N < 2 ** 31, "nnz must fit the int32 row pointer of the reference's CSR"
    rowptr = np.zeros(dof * N + 1, dtype=np.int64)
    cols_parts, vals_parts = [], []
    b = np.arange(dof, dtype=np.int64)
    for n0 in range(0, N, chunk):
        nodes = np.arange(n0, min(N, n0 + chunk), dtype=np.int64)
        x, y, z = nodes % g, (nodes // g) % g, nodes // (g * g)
        ok = ((x[:, None] + o[None, :, 0] >= 0) & (x[:, None] + o[None, :, 0] < g) & (y[:, None] + o[None, :, 1] >= 0) &
              (y[:, None] + o[None, :, 1] < g) & (z[:, None] + o[None, :, 2] >= 0) & (z[:, None] + o[None, :, 2] < g))     # [node, offset]
        nb = nodes[:, None] + lin[None, :]
        cnt = ok.sum(axis=1) * dof                                        # nonzeros of every row of the node
        rowptr[n0 * dof + 1:(n0 + nodes.size) * dof + 1] = np.repeat(cnt, dof)
        # [node, a, offset, b] -> rows node*dof + a, cols nb*dof + b
        okf = np.broadcast_to(ok[:, None, :, None], (nodes.size, dof, 27, dof))
        rows = np.broadcast_to((nodes[:, None, None, None] * dof + b[None, :, None, None]), okf.shape)[okf]
        cols = np.broadcast_to((nb[:, None, :, None] * dof + b[None, None, None, :]), okf.shape)[okf]
        lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
        u = _u01(lo * np.int64(1 << 26) + hi, seed)
        cols_parts.append(cols.astype(np.int32))
        vals_parts.append(np.where(rows == cols, 40.0 + u, 2.0 * u - 1.0))
    return np.cumsum(rowptr).astype(np.int32), np.concatenate(cols_parts), np.concatenate(vals_parts)


def fem3d(g, dof=3, seed=5):
    """Queen_4147-like stand-in (BASELINE configs[3]): 3D solid mechanics, `dof` unknowns per node of a
    g^3 grid, every node coupled to its 27 neighbours: dof*g^3 rows, up to 27*dof (= 81) nnz/row."""
    return _stencil_csr(dof * g * g * g, [(0, g, dof, 0, dof, _OFF27)], seed)


def random_csr(m, k, max_deg, seed=7, empty_every=0):
    """General test matrix: row i has (hash % (max_deg + 1)) distinct sorted columns;
    every `empty_every`-th row is empty."""
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, max_deg + 1, size=m)
    if empty_every:
        deg[::empty_every] = 0
    deg = np.minimum(deg, k)
    rowptr = np.zeros(m + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    cols = np.empty(int(rowptr[-1]), dtype=np.int32)
    for i in range(m):
        if deg[i]:
            cols[rowptr[i]:rowptr[i + 1]] = np.sort(rng.choice(k, size=deg[i], replace=False))
    vals = rng.uniform(-1.0, 1.0, size=cols.size)
    return rowptr, cols, vals


def alg_bytes(m, k, n, nnz, vb=8):
    """compulsory HBM bytes of one SpMM: A once, B once, C written once."""
    return (vb + 4) * nnz + 4 * (m + 1) + vb * k * n + vb * m * n


def shell_fem(nc=160, nl=227, dof=6, m=217918, seed=20261005, jitter=64, p_drop=0.12, p_extra=0.05, seam=3, seam_to=197):
    """Irregular pwtk-class stand-in (the real pwtk, /root/reference/README.md:62-63, is a shell-element model of a
    pressurised wind tunnel: 217,918 rows, ~53 nnz/row, bandwidth 189,331): a closed tube surface of nc x nl
    quadrilateral shell elements' nodes, `dof` unknowns per node, every node coupled to its (up to) 8 surface
    neighbours through dense dof x dof blocks.  Nothing is Toeplitz: node numbers are shuffled inside windows of
    `jitter` nodes of the ring-by-ring order, a fraction p_drop of the diagonal couplings is missing (triangulated
    patches), a fraction p_extra of the nodes has a stiffener coupling two nodes along the ring, and the first
    `seam` rings are tied to rings seam_to.. (the far band that gives pwtk its bandwidth: 197 rings = 189,1xx rows).  Rows beyond m are cut (the
    last node keeps fewer unknowns).  Symmetric pattern and values, diagonal blocks boosted; columns ascending.
    -> (rowptr int32, colidx int32, val float64)"""
    import scipy.sparse as sp
    nn = nc * nl
    node = np.arange(nn, dtype=np.int64)
    c, l = node % nc, node // nc
    # window-local shuffle of the numbering (deterministic: sort by hash inside each window)
    win = node // jitter
    key = win.astype(np.float64) + 0.999 * _u01(node + np.int64(7919), seed)
    perm = np.empty(nn, dtype=np.int64)
    perm[np.argsort(key, kind="stable")] = node          # perm[old] = new number
    edges_a, edges_b = [], []

    def add(a, b, keep):
        edges_a.append(a[keep])
        edges_b.append(b[keep])
    for dc, dl in ((1, 0), (0, 1), (1, 1), (-1, 1)):
        ok = (l + dl < nl)
        nb = ((c + dc) % nc) + (l + dl) * nc
        if dc != 0 and dl != 0:
            ok = ok & (_u01(node * np.int64(4) + np.int64(dc + 2), seed + 1) >= p_drop)
        add(node, np.where(ok, nb, 0), ok)
    ex = _u01(node + np.int64(1 << 30), seed + 2) < p_extra
    add(node, ((c + 2) % nc) + l * nc, ex)
    if seam > 0:
        s = l < seam
        add(node, c + (np.minimum(seam_to + l, nl - 1)) * nc, s & (nl > 2 * seam))
    a = np.concatenate(edges_a)
    b = np.concatenate(edges_b)
    a, b = perm[a], perm[b]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    und = np.unique(lo * np.int64(nn) + hi)
    lo, hi = und // nn, und % nn
    nr = np.concatenate([lo, hi, np.arange(nn, dtype=np.int64)])       # node-level pattern, both triangles + diagonal
    ncol = np.concatenate([hi, lo, np.arange(nn, dtype=np.int64)])
    P = sp.csr_matrix((np.ones(nr.size, dtype=np.int8), (nr, ncol)), shape=(nn, nn))
    P.sort_indices()
    # expand every node entry to a dof x dof block
    deg = np.diff(P.indptr).astype(np.int64)
    row_node = np.repeat(np.arange(nn, dtype=np.int64), deg)
    nnz_n = P.indices.size
    rows = (row_node[:, None, None] * dof + np.arange(dof)[None, :, None]) + np.zeros((1, 1, dof), dtype=np.int64)
    cols = (P.indices.astype(np.int64)[:, None, None] * dof + np.arange(dof)[None, None, :]) + np.zeros((1, dof, 1), dtype=np.int64)
    rows, cols = rows.reshape(-1), cols.reshape(-1)
    keep = (rows < m) & (cols < m)
    rows, cols = rows[keep], cols[keep]
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
    u = _u01(lo * np.int64(1 << 20) + (hi - lo) + (hi % 97) * np.int64(1 << 40), seed + 3)
    vals = np.where(rows == cols, 40.0 + u, 2.0 * u - 1.0)
    del nnz_n
    return rowptr.astype(np.int32), cols.astype(np.int32), vals
