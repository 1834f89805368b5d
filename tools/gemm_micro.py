#!/usr/bin/env python3
"""The dense products of the MMA layer step on their own, at the C4 shape (M = 2^20 rows, H = 128, K = 4: [P|Q] is 1024 wide) and at the
C5 shard shape (H = 256, K = 8: 4096 wide), as the layer calls them (mma_amd.dense), interleaved rounds in ONE process
(cdna_hip_programming.md rule 24), random operands (rule 25).  Per product: median / min ms, the HBM bytes it has to move, the 16-bit
MFMA work of its three piece products, and both roofline fractions (HBM 8.0 TB/s; fp16 MFMA 2.5 PFLOP/s dense).

    python tools/gemm_micro.py [--shape c4|c5|both] [--rounds 7] [--json]        (on the GPU box)
A/B of a build switch: run it once per environment (e.g. MMA_DX_ACC=atomic) in the SAME gpurun call - the switches are read once
per process."""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mma_amd import dense  # noqa: E402

DEV = "cuda:0"
HBM, MFMA16 = 8.0e12, 2.5e15


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def products(M, H, K):
    """name -> (callable, HBM bytes, fp32-equivalent flops).  Operands as the step has them: x = relu(randn), weights U(+-1/sqrt(H)),
    the gradient block with row sizes spread over a few binades (K2b's output looks like that)."""
    g = torch.Generator(device=DEV).manual_seed(0)
    W2 = 2 * K * H
    x = torch.relu(torch.randn(M, H, device=DEV, generator=g))
    W = (torch.rand(H, W2, device=DEV, generator=g) * 2 - 1) / H ** 0.5
    gpq = torch.randn(M, W2, device=DEV, generator=g) * torch.exp(torch.randn(M, 1, device=DEV, generator=g))
    out = torch.empty(M, W2, device=DEV)
    gx = torch.zeros(M, H, device=DEV)
    rm = dense.row_absmax(gpq)
    xm = dense.row_absmax(x)
    Wt = W.t()
    flops = 2.0 * M * H * W2
    fwd = lambda: dense.mm_into(x, W, out)
    if dense.f16x2_n128_ok(M, W2, H):
        dx = lambda: dense.gemm_f16x2_n128(gpq, rm, Wt, gx, accumulate=True)
    else:
        dx = lambda: dense.rows_mm_add_(gx, gpq, Wt)
    tn = lambda: dense.xt_g(x, gpq, xm, rm)
    return {
        "forward [P|Q] = x W (M,%d)x(%d,%d)" % (H, H, W2): (fwd, 4.0 * M * (H + W2), flops),
        "dL/dx += g W^T (M,%d)x(%d,%d)" % (W2, W2, H): (dx, 4.0 * M * (W2 + 2 * H), flops),           # read g, read + write gx
        "weight gradient x^T g (%d,M)x(M,%d)" % (H, W2): (tn, 4.0 * M * (H + W2), flops),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="both", choices=["c4", "c5", "both"])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    res = {}
    for tag, (M, H, K) in (("c4", (1 << 20, 128, 4)), ("c5", (1 << 20, 256, 8))):
        if args.shape not in (tag, "both"):
            continue
        prods = products(M, H, K)
        reps = args.reps if tag == "c4" else max(2, args.reps // 4)
        for fn, _, _ in prods.values():
            fn(); fn()
        torch.cuda.synchronize()
        times = {n: [] for n in prods}
        for _ in range(args.rounds):
            for n, (fn, _, _) in prods.items():
                times[n].append(timed(fn, reps))
        for n, (fn, nbytes, flops) in prods.items():
            med, mn = statistics.median(times[n]), min(times[n])
            res["%s %s" % (tag.upper(), n)] = {"median_ms": med, "min_ms": mn, "hbm_bytes": nbytes, "flops_fp32": flops, "mfma16_flops": 3 * flops,
                                                "hbm_frac": nbytes / (med * 1e-3) / HBM, "mfma_frac": 3 * flops / (med * 1e-3) / MFMA16}
        del prods
        torch.cuda.empty_cache()
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("MMA_")}
    if args.json:
        print(json.dumps({"env": env, "products": res}))
        return
    print("switches: %s" % (env or "defaults"))
    print("| product | median ms | min ms | HBM bytes | of 8.0 TB/s | 3-product fp16 MFMA TFLOP | of 2.5 PFLOP/s |")
    print("|---|---|---|---|---|---|---|")
    for n, r in res.items():
        print("| %s | %.3f | %.3f | %.2f GB | %.2f | %.2f | %.2f |" % (n, r["median_ms"], r["min_ms"], r["hbm_bytes"] / 1e9, r["hbm_frac"],
                                                                      r["mfma16_flops"] / 1e12, r["mfma_frac"]))


if __name__ == "__main__":
    main()
