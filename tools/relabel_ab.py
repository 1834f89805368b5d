#!/usr/bin/env python3
"""Measure-first (round-4 VERDICT item 6): does a hub-first labelling of the nodes make the fused kernels faster on one GPU?

K1 / K2b gather ~27.8 GB of x | Q rows per step at C4 out of 2.6 GB of distinct rows; on an R-MAT graph (a = 0.57) the top 5 % of the
nodes are more than half of all edge endpoints.  With degree-descending ids those rows (~130 MB of x | Q) are contiguous - inside the
256 MB Infinity Cache - instead of scattered over the 2.6 GB table.  A: the generator's order.  B: the same graph with node ids
relabelled by descending in-degree (targets AND sources: every table is permuted consistently; the edge order inside a target's list is
kept, so each target's sums are the same bits - checked - after un-permuting).  Prints K1 / K2b / step times of both.

    python tools/relabel_ab.py [--steps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import mma_amd
from mma_amd import functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--order", default="degree", choices=["degree", "random"])
args = ap.parse_args()
dev = torch.device("cuda:0")
H, C, names, p = 128, 16, ["sum", "mean", "max", "min"], 0.5
rowptr, col = bench.rmat_graph(20, 5_000_000, seed=42)
N, E = len(rowptr) - 1, int(rowptr[-1])
deg = np.diff(rowptr)
x_all = bench.feature_rows(0, N, H, 42)
cot_all = bench.feature_rows(0, N, C, 43, relu=False)


def relabel(order):
    """order[new] = old.  CSR with targets in the new order, sources renamed, the edge order inside every list kept."""
    rank = np.empty(N, dtype=np.int64); rank[order] = np.arange(N)
    d2 = deg[order]
    rp2 = np.concatenate([[0], np.cumsum(d2)])
    src_pos = np.repeat(rowptr[:-1][order], d2) + (np.arange(E) - np.repeat(rp2[:-1], d2))      # old position of every new position
    return rp2, rank[col[src_pos]]


def run(tag, rp, cl, order):
    graph = mma_amd.NCGraph(rp, cl, dev, H=H)
    layer = bench.make_layer(mma_amd, graph, H, C, names, p, dev)
    torch.manual_seed(1)
    with torch.no_grad():
        for q in layer.owned:
            q.uniform_(-0.09, 0.09)
    layer.drop_override = None
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))
    adj = mma_amd.graph.SpmmGraph(dst, cl, None, N, N, dev)
    x = torch.from_numpy(x_all[order]).to(dev).requires_grad_(True)
    cot = torch.from_numpy(cot_all[order]).to(dev)

    def step():
        x.grad = None
        for q in layer.owned:
            q.grad = None
        layer(x, adj).backward(cot)
    for _ in range(3):
        step()
    Fn.TIMER = t = bench.KernelTimer(); t.enabled = True
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    sp = t.summary(); Fn.TIMER = None
    ks = {k: round(v[1] / args.steps, 3) for k, v in sp.items()}
    print("%s: step %.3f ms  K1 %.3f  K2b %.3f  %s" % (tag, ms, ks.get("nc_fused_fwd", 0), ks.get("nc_fused_bwd", 0), ks), flush=True)
    # p = 0 forward of the aggregate for the bit-equality check (dropout bits are keyed by edge position, which the relabelling moves)
    layer.dropout = 0.0
    with torch.no_grad():
        out = layer(x, adj)
    return out.detach().cpu().numpy()


ident = np.arange(N)
oA = run("A generator order", rowptr, col, ident)
order = np.argsort(-deg, kind="stable") if args.order == "degree" else np.random.default_rng(0).permutation(N)
rp2, cl2 = relabel(order)
oB = run("B %s order" % args.order, rp2, cl2, order)
back = np.empty_like(oB); back[order] = oB
same = np.array_equal(back, oA)
print("outputs after un-permuting: %s (max |diff| %.3g of max %.3g)" % ("bit-equal" if same else "NOT bit-equal", np.abs(back - oA).max(), np.abs(oA).max()))
