"""Synthetic inputs of the BASELINE configurations - shared by bench.py, the tools and the tests.
Data generators only; nothing here touches the GPU."""
import os

import numpy as np


def rmat_graph(scale, n_undirected, seed=42, a=0.57, b=0.19, c=0.19):
    """R-MAT -> symmetrised, de-duplicated, no self loops, isolated nodes attached to a random node.
    Returns CSR by target (rowptr int64, col int64) with ascending neighbour order (utils.py:100)."""
    rng = np.random.default_rng(seed)
    N = 1 << scale
    src = np.zeros(n_undirected, dtype=np.int64)
    dst = np.zeros(n_undirected, dtype=np.int64)
    for bit in range(scale):
        r = rng.random(n_undirected, dtype=np.float32)
        sb = (r >= a + b).astype(np.int64)                       # quadrants c,d set the source bit
        db = (((r >= a) & (r < a + b)) | (r >= a + b + c)).astype(np.int64)   # quadrants b,d set the target bit
        src |= sb << bit
        dst |= db << bit
    perm = rng.permutation(N)                                    # break the bit-pattern locality of raw R-MAT ids
    src, dst = perm[src], perm[dst]
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * N + dst, dst * N + src]))
    row, col = key // N, key % N
    deg = np.bincount(row, minlength=N)
    iso = np.nonzero(deg == 0)[0]
    if len(iso):                                                 # reference needs d >= 1 (Q12)
        nb = rng.integers(0, N, len(iso))
        nb = np.where(nb == iso, (nb + 1) % N, nb)
        key = np.unique(np.concatenate([key, iso * N + nb, nb * N + iso]))
        row, col = key // N, key % N
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=N), out=rowptr[1:])
    return rowptr, col


_BLOCK = 1 << 16


def feature_rows(lo, hi, width, seed, relu=True):
    """Rows [lo, hi) of the synthetic (N, width) fp32 matrix `relu(randn)` (stands in for relu(gc1), models.py:65).
    Generated in 65536-row blocks keyed by (seed, block), so a rank materialises only its own rows and every
    partition of the rows sees the same values."""
    out = np.empty((hi - lo, width), dtype=np.float32)
    for blk in range(lo // _BLOCK, -(-hi // _BLOCK) if hi > lo else lo // _BLOCK):
        b0 = blk * _BLOCK
        rows = np.random.default_rng([seed, blk]).standard_normal((_BLOCK, width), dtype=np.float32)
        s, e = max(lo, b0), min(hi, b0 + _BLOCK)
        out[s - lo:e - lo] = rows[s - b0:e - b0]
    return np.maximum(out, 0) if relu else out


def feature_rows_by_id(ids, width, seed, relu=True):
    """The rows `ids` (ascending) of the same synthetic matrix as feature_rows(): what a rank RECEIVES for its halo rows."""
    ids = np.asarray(ids, dtype=np.int64)
    out = np.empty((len(ids), width), dtype=np.float32)
    blk = ids // _BLOCK
    starts = np.flatnonzero(np.r_[True, blk[1:] != blk[:-1]]) if len(ids) else np.zeros(0, np.int64)
    ends = np.r_[starts[1:], len(ids)]
    for s, e in zip(starts, ends):
        b = int(blk[s])
        rows = np.random.default_rng([seed, b]).standard_normal((_BLOCK, width), dtype=np.float32)
        out[s:e] = rows[ids[s:e] - b * _BLOCK]
    return np.maximum(out, 0) if relu else out


def molecule_batch(rng, n_graphs=12, return_sizes=False):
    """ZINC-like: trees of ~23 nodes + ring closures, max degree 4, symmetrised."""
    src, dst, off, sizes = [], [], 0, []
    for _ in range(n_graphs):
        n = int(rng.integers(12, 30))
        deg = np.zeros(n, int)
        for v in range(1, n):
            cand = [u for u in range(v) if deg[u] < 3]
            u = int(rng.choice(cand))
            src += [off + u, off + v]; dst += [off + v, off + u]; deg[u] += 1; deg[v] += 1
        for _ in range(int(rng.integers(1, 4))):
            u, v = rng.choice(n, 2, replace=False)
            if deg[u] < 4 and deg[v] < 4:
                src += [off + u, off + v]; dst += [off + v, off + u]; deg[u] += 1; deg[v] += 1
        off += n
        sizes.append(n)
    if return_sizes:
        return np.array([src, dst]), off, np.array(sizes)
    return np.array([src, dst]), off


def golden_csr(name):
    """CSR (rowptr, col int64) of a committed fixture graph (tests/golden/<name>.npz: Cora / Pubmed structure)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", name + ".npz"))
    return z["rowptr"].astype(np.int64), z["col"].astype(np.int64)
