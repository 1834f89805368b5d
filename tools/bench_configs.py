#!/usr/bin/env python3
"""Timings of the BASELINE.json parity configurations that are NOT the bench.py workload (they are launch-bound on any
GPU; reported for completeness, one JSON line each):
  C1  Cora structure, H=64, aggregators mean,mean2            (tests/golden/cora_h64.npz holds the CSR)
  C3  Pubmed structure, H=16, aggregators min,min2,min3,min4  (tests/golden/pubmed_h16.npz)
  C2  ZINC-like molecule batch (64 graphs), MMAConv 75->75, towers=5, edge_dim=50, min,max x identity,amplification,linear
  C2L the same layer on a 10 000-graph batch (the whole ZINC-subset training split in one batch)
Each line: wall ms per layer forward+backward (host-timed, synchronised), HIP-event time of the fused kernels, their
algorithmic bytes and the resulting GB/s."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mma_amd  # noqa: E402
from mma_amd import functional as Fn  # noqa: E402

DEV = "cuda:0"


def graph_replay_ms(step, n=50):
    """Capture `step` (layer forward+backward on static tensors) into a hipGraph and time replays."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def wall(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def nc_config(tag, npz, H, names, C, reps=50):
    z = np.load(os.path.join(ROOT, "tests", "golden", npz))
    rowptr, col = z["rowptr"].astype(np.int64), z["col"].astype(np.int64)
    N, E, K = len(rowptr) - 1, len(col), len(names)
    graph = mma_amd.NCGraph(rowptr, col, DEV)
    layer = bench.make_layer(mma_amd, graph, H, C, names, 0.5, DEV)
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, DEV)
    x = torch.relu(torch.randn(N, H, device=DEV)).requires_grad_(True)
    cot = torch.randn(N, C, device=DEV)

    def step():
        x.grad = None
        layer(x, adj).backward(cot)
    ms = wall(step, reps)
    Fn.TIMER = t = bench.KernelTimer(); t.enabled = True
    for _ in range(reps):
        step()
    spans = t.summary(); Fn.TIMER = None
    layer.graph_capturable = True
    gms = graph_replay_ms(step)
    ab = bench.algorithmic_bytes(N, E, H, K)
    k = {n: {"avg_us": tot / c * 1e3, **({"algorithmic_bytes": ab[n], "GBs": ab[n] / (tot / c * 1e-3) / 1e9} if n in ab else {})}
         for n, (c, tot) in spans.items()}
    print(json.dumps({"config": tag, "nodes": N, "edges": E, "H": H, "K": K, "ms_per_layer_fwd_bwd": ms,
                      "ms_per_layer_fwd_bwd_hipgraph": gms, "edges_per_s": E / ms * 1e3, "edges_per_s_hipgraph": E / gms * 1e3,
                      "kernels": k}), flush=True)


def gr_config(tag, n_graphs, reps=20):
    from test_gr_gpu import molecule_batch
    rng = np.random.default_rng(0)
    ei, N = molecule_batch(rng, n_graphs)
    E = ei.shape[1]
    T, F, K, S = 5, 75, 2, 3
    hist = np.bincount(np.bincount(ei[1], minlength=N), minlength=5)
    conv = mma_amd.MMAConv(75, 75, ["min", "max"], ["identity", "amplification", "linear"], torch.tensor(hist), edge_dim=50,
                           towers=5).to(DEV)
    x = torch.randn(N, 75, device=DEV, requires_grad=True)
    ea = torch.randn(E, 50, device=DEV)
    eig = torch.from_numpy(ei).to(DEV)
    cot = torch.randn(N, 75, device=DEV)

    def step():
        x.grad = None
        conv(x, eig, ea).backward(cot)
    ms = wall(step, reps)
    Fn.TIMER = t = bench.KernelTimer(); t.enabled = True
    for _ in range(reps):
        step()
    spans = t.summary(); Fn.TIMER = None
    conv.graph_capturable = True
    gms = graph_replay_ms(step, 20)
    D = T * F
    ab = {"gr_fused_fwd": 4 * (E * (2 + 2 * D) + N * (1 + D + T * K * S * F)),
          "gr_fused_bwd": 4 * (E * (2 + D) + N * (1 + T * K * S * F + 2 * D))}
    k = {n: {"avg_us": tot / c * 1e3, **({"algorithmic_bytes": ab[n], "GBs": ab[n] / (tot / c * 1e-3) / 1e9} if n in ab else {})}
         for n, (c, tot) in spans.items()}
    print(json.dumps({"config": tag, "graphs": n_graphs, "nodes": N, "edges": E, "towers": T, "F": F,
                      "ms_per_layer_fwd_bwd": ms, "ms_per_layer_fwd_bwd_hipgraph": gms, "edges_per_s": E / ms * 1e3,
                      "edges_per_s_hipgraph": E / gms * 1e3, "kernels": k}), flush=True)


if __name__ == "__main__":
    nc_config("C1 cora H=64 mean,mean2", "cora_h64.npz", 64, ["mean", "mean2"], 7)
    nc_config("C3 pubmed H=16 min,min2,min3,min4", "pubmed_h16.npz", 16, ["min", "min2", "min3", "min4"], 3)
    gr_config("C2 zinc-like batch 64", 64)
    gr_config("C2L zinc-like batch 10000", 10000, reps=5)
