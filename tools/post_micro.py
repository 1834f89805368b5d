#!/usr/bin/env python3
"""K13 / K14 / K15 (MMAConv's factored post-NN) and K16 (the 75 -> 75 skinny Linear) on their own at the C2L shape (204 552 nodes,
5 towers, K*F = 152, O = 15, scalers identity / amplification / linear), interleaved rounds in ONE process, random operands.  Per kernel:
median ms of the bf16-piece form and of the exact-fp32 form (MMA_POST_EXACT is read per call), algorithmic bytes, HBM fraction.

    python tools/post_micro.py [--rounds 7] [--n 204552]                 (on the GPU box)
MMA_LIB_OVERRIDE=<.so> runs a measurement build of the library."""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mma_amd import _lib, functional as Fn  # noqa: E402
from mma_amd._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--n", type=int, default=204552)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--towers", type=int, default=5, help="towers T (the same bytes with --towers 1 --n 1022760: one contiguous 608-byte row per node)")
args = ap.parse_args()

N, T, KF, O = args.n, args.towers, 152, 15
scalers = tuple(Fn.GR_SCALER[s] for s in ("identity", "amplification", "linear"))
S = len(scalers)
g = torch.Generator(device=DEV).manual_seed(0)
deg = torch.randint(1, 5, (N,), device=DEV, generator=g)
rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), torch.cumsum(deg, 0)]).to(torch.int32)
agg = torch.randn(N, T, KF, device=DEV, generator=g)
Wo = torch.randn(T, O, S * KF, device=DEV, generator=g) / (S * KF) ** 0.5
gy = torch.randn(N, T * O, device=DEV, generator=g)
KFp = int(_lib.lib().mma_tower_post_kfp(KF))
Wb = torch.empty((T, S * 16, KFp + 16), device=DEV)
Wa = torch.empty((T, KFp, S * 16), device=DEV)
call("mma_tower_post_weights", ptr(Wo), T, O, S, KF, ptr(Wa), ptr(Wb), stream_ptr())
pre = torch.empty((N, 8), device=DEV)
codes = Fn.host_codes(scalers)
call("mma_tower_post_pre", ptr(rowptr), ptr(pre), N, S, codes, 1.3, 2.2, stream_ptr())
y = torch.empty(N, T * O, device=DEV)
gagg = torch.empty_like(agg)
kfp16 = -(-KF // 16) * 16
n_chunks = int(_lib.lib().mma_tower_post_gw_chunks(N, T))
part = torch.empty((n_chunks, T * S * 16 * kfp16), device=DEV)
# K16: 75 -> 75
K16 = 75
x16 = torch.randn(N, K16, device=DEV, generator=g)
W16 = torch.randn(K16, K16, device=DEV, generator=g) / K16 ** 0.5
b16 = torch.randn(K16, device=DEV, generator=g)
kfp_16 = int(_lib.lib().mma_tower_post_kfp(K16))
R16 = -(-K16 // 16) * 16
Wa16, Wb16 = torch.empty(kfp_16, R16, device=DEV), torch.empty(R16, kfp_16 + 16, device=DEV)
call("mma_skinny_linear_weights", ptr(W16), K16, K16, ptr(Wa16), ptr(Wb16), stream_ptr())
y16, gx16 = torch.empty(N, K16, device=DEV), torch.empty(N, K16, device=DEV)

gw16, gb16 = torch.empty(K16, K16, device=DEV), torch.empty(K16, device=DEV)


def skinny_gw():
    # the workspace size follows MMA_SKINNY_GW_WAVES (read per call): ask per call
    n_part = int(_lib.query("mma_skinny_linear_gw_part", N, K16, K16))
    part16 = torch.empty(n_part, device=DEV)
    call("mma_skinny_linear_gw", ptr(y16), K16, ptr(x16), K16, ptr(part16), n_part, ptr(gw16), ptr(gb16), N, K16, K16, stream_ptr())


kernels = {
    "K13 tower_post_fwd": (lambda: call("mma_tower_post_fwd", ptr(agg), T * KF, ptr(pre), ptr(Wa), ptr(y), T * O, N, T, KF, S, O, codes, 1.3, 2.2, stream_ptr()),
                           4 * N * (T * KF + T * O + 8)),
    "K14 tower_post_bwd": (lambda: call("mma_tower_post_bwd", ptr(gy), T * O, ptr(pre), ptr(Wb), ptr(gagg), T * KF, None, 0, N, T, KF, S, O, codes, 1.3, 2.2,
                                        stream_ptr()), 4 * N * (T * KF + T * O + 8)),
    "K15 tower_post_gw": (lambda: call("mma_tower_post_gw", ptr(gy), T * O, ptr(agg), T * KF, ptr(pre), ptr(part), n_chunks, N, T, KF, S, O, codes, 1.3, 2.2,
                                       stream_ptr()), 4 * N * (T * KF + T * O + 8) + 4 * part.numel()),
    "K16 skinny fwd 75->75": (lambda: call("mma_skinny_linear_fwd", ptr(x16), K16, ptr(Wa16), ptr(b16), None, 0, ptr(y16), K16, N, K16, K16, stream_ptr()), 4 * N * 2 * K16),
    "K16 skinny bwd 75->75": (lambda: call("mma_skinny_linear_bwd_dx", ptr(x16), K16, ptr(Wb16), ptr(gx16), K16, N, K16, K16, stream_ptr()), 4 * N * 2 * K16),
    "K16 skinny gw+gb 75->75": (lambda: skinny_gw(), 4 * N * 2 * K16),
}


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {(k, m): [] for k in kernels for m in ("x3", "exact")}
for rnd in range(args.rounds + 1):
    for k, (fn, _) in kernels.items():
        for m in ("x3", "exact"):
            os.environ["MMA_POST_EXACT"] = "1" if m == "exact" else "0"
            t = timed(fn, args.reps)
            if rnd:
                res[(k, m)].append(t)
os.environ["MMA_POST_EXACT"] = "0"
for k, (fn, nbytes) in kernels.items():
    a, b = statistics.median(res[(k, "x3")]), statistics.median(res[(k, "exact")])
    print("%-24s bf16 pieces %.4f ms (%.2f of 8 TB/s)   exact fp32 %.4f ms (%.2f)   %.3f GB" % (k, a, nbytes / a / 8e9, b, nbytes / b / 8e9, nbytes / 1e9), flush=True)
