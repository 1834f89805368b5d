#!/usr/bin/env python3
"""Halo volume of the 1-D node partition of the C4 graph (numpy only, no GPU): distinct remote source rows per rank at
2/4/8 ranks under three node orders - the generator's (random permutation of R-MAT ids), degree-descending and reverse
Cuthill-McKee (BFS-like) - with the edge-balanced cut and with a row-cost-weighted cut (mma_amd.sharded.partition_bounds).

    python tools/halo_report.py [--scale 20] [--edges 5000000]

What it shows: a locality relabel does NOT pay on this graph.  R-MAT with permuted ids has no community
structure to find, and a degree- or BFS-order concentrates the hubs in one range: the edge-balanced cut then hands one
rank most of the ROWS (own + halo rows per rank up to 2.3x the random order's), a row-weighted cut trades that for edge
imbalance.  The random order keeps every rank at ~N/P own rows and ~E/P edges; its halo (0.73x / 1.48x / 2.29x the own rows
at 2 / 4 / 8 ranks) is the price of a 1-D partition of a power-law graph."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mma_amd.sharded import halo_report  # noqa: E402
from tools.synth import rmat_graph  # noqa: E402


def relabel(rowptr, col, order):
    """CSR under new ids (order[new] = old), neighbours ascending."""
    N = len(rowptr) - 1
    inv = np.empty(N, dtype=np.int64); inv[order] = np.arange(N)
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    key = np.sort(inv[dst] * N + inv[col])
    rp = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(key // N, minlength=N), out=rp[1:])
    return rp, key % N


def orders(rowptr, col):
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    N = len(rowptr) - 1
    deg = np.diff(rowptr)
    yield "generator (random ids)", None
    yield "degree-descending", np.argsort(-deg, kind="stable")
    A = sp.csr_matrix((np.ones(len(col), dtype=np.int8), col, rowptr), shape=(N, N))
    yield "reverse Cuthill-McKee", np.asarray(reverse_cuthill_mckee(A, symmetric_mode=True), dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edges", type=int, default=5_000_000)
    a = ap.parse_args()
    rowptr, col = rmat_graph(a.scale, a.edges, seed=42)
    avg = len(col) / (len(rowptr) - 1)
    print("| node order | cut | ranks | own rows min..max | halo rows max | halo/own max | own+halo rows max | edges max | own-source edges |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, order in orders(rowptr, col):
        rp, c = (rowptr, col) if order is None else relabel(rowptr, col, order)
        for cut, rc in (("edges", 0.0), ("edges + %.0f/row" % avg, avg)):
            for w in (2, 4, 8):
                h = np.array(halo_report(rp, c, w, rc))
                print("| %s | %s | %d | %d..%d | %d | %.2f | %d | %d | %.0f %% |" % (
                    name, cut, w, h[:, 0].min(), h[:, 0].max(), h[:, 1].max(), (h[:, 1] / np.maximum(h[:, 0], 1)).max(),
                    (h[:, 0] + h[:, 1]).max(), h[:, 2].max(), 100.0 * h[:, 3].sum() / h[:, 2].sum()))


if __name__ == "__main__":
    main()
