#!/usr/bin/env python3
"""The three split-precision GEMMs of the C4 step (M = 2^20 rows, H = 128, K = 4 masks) on random and on ZERO-filled operands, same
binary, same launches: the difference is what the chip's power management takes (MI355X_MICROARCH.md "DVFS give-back": the clock
a kernel holds depends on how much its operands toggle), i.e. how far each kernel is from its own schedule bound on real data.
    python tools/gemm_power.py        (on the GPU box; prints a markdown table)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mma_amd import dense  # noqa: E402

DEV = "cuda:0"
N, H, K = 1 << 20, 128, 4


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    rows = []
    for name, mk in (("random (N(0,1))", lambda *s: torch.randn(*s, device=DEV)), ("zeros", lambda *s: torch.zeros(*s, device=DEV))):
        x, W, g, Wt = mk(N, H), mk(H, 2 * K * H), mk(N, 2 * K * H), mk(2 * K * H, H)
        out = torch.zeros(N, H, device=DEV)
        rm = g.abs().amax(1)                    # in the step K2a / K2b produce this bound
        dx = ((lambda: dense.gemm_f16x2_n128(g, rm, Wt, out, accumulate=True)) if dense.f16x2_n128_ok(N, 2 * K * H, H)
              else (lambda: dense.gemm_bf16x3(g, Wt, out=out, accumulate=True)))
        xm = x.abs().amax(1)                    # in the step the forward GEMM leaves this
        three = dense.USE_F16X2 and dense.USE_F16X2_TN
        tn = (lambda: dense.gemm_f16x2_tn(x, g, xm, rm)) if three else (lambda: dense.gemm_bf16x3_tn(x, g))
        for _ in range(2):                      # second pass: clocks settled
            r = (timed(lambda: dense.gemm_bf16x3(x, W)), timed(dx), timed(tn), timed(lambda: dense.gemm_bf16x3_tn(x, g)))
        rows.append((name, r))
    print("| operands | forward x [Wtop|Wbot] (M,128)x(128,1024) | dL/dx += g [Wtop|Wbot]^T (M,1024)x(1024,128) | weight gradient x^T g (128,M)x(M,1024) "
          "| the same, six-product bf16x3 kernel |")
    print("|---|---|---|---|---|")
    for name, r in rows:
        print("| %s | %.3f ms | %.3f ms | %.3f ms | %.3f ms |" % ((name,) + r))
    rows = [(n, r[:3]) for n, r in rows]
    base = 2.0 * N * H * 2 * K * H                # fp32 flops of each product
    mf = (3, 3 if dense.f16x2_n128_ok(N, 2 * K * H, H) else 6, 3 if three else 6) if dense.USE_F16X2 else (6, 6, 6)     # piece products per fp32 product
    rate = lambda ts: " / ".join("%.2f" % (base * m / 1e15 / (t * 1e-3)) for t, m in zip(ts, mf))
    print("\nEach is 275 GFLOP of fp32 work = %s TFLOP of 16-bit MFMA (%s piece products per fp32 product); on random data that is %s "
          "PFLOP/s, on zeros %s PFLOP/s (dense bf16/fp16 peak 2.5 PFLOP/s at 2.4 GHz)." % (
              " / ".join("%.2f" % (base * m / 1e12) for m in mf), " / ".join(str(m) for m in mf), rate(rows[0][1]), rate(rows[1][1])))


if __name__ == "__main__":
    main()
