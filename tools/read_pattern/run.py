#!/usr/bin/env python3
"""Read-only micro-benchmark of the access patterns K13-K15 could use on the aggregate tensor of the graph-regression layer (agg (N, T*KF)
fp32, C2L: 204 552 x 760; a workgroup = one tower's 608-byte slice of 256 node rows, tower fastest): what the memory system delivers for
the pattern alone, at full occupancy, no LDS, no arithmetic.  DESIGN.md 3 (K13) quotes it.

    python tools/read_pattern/run.py            (on the GPU box; compiles rdpat.hip with hipcc on first use)"""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librdpat.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "rdpat.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", SO, os.path.join(HERE, "rdpat.hip")])
lib = ctypes.CDLL(SO)
lib.rd_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
N, T, KF = 204552, 5, 152
agg = torch.randn(N, T * KF, device="cuda")
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


names = {0: "K13's loads: 16 rows x 64 B per instruction, 128 B of every row per k-step", 1: "whole 608-byte rows per lane group, node tile by node tile",
         2: "the fp32 kernel's loads: 8 rows x 128 B per instruction", 3: "one row per load instruction (608 contiguous bytes)"}
print("| tiles per wave | pattern | ms | TB/s |\n|---|---|---|---|")
for tpw in (1, 4):
    for v in (0, 1, 2, 3):
        ms = t(lambda: lib.rd_launch(agg.data_ptr(), out.data_ptr(), N, T, KF, v, tpw, 0, st))
        print("| %d | %s | %.4f | %.2f |" % (tpw, names[v], ms, N * T * KF * 4 / ms / 1e9), flush=True)
print("\n`agg.sum()` (torch): %.4f ms" % t(lambda: agg.sum()))
