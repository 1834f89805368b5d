"""The binding stubs printed in INTEGRATION.md are executed as written: the ctypes code a maintainer of the reference would
paste must keep working against the built library (argument order, types, conventions)."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stubs():
    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", src, re.S) if "ctypes" in b or "def aggregate" in b]
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                     # the stub opens the library by its repo-relative path
    try:
        for b in blocks:
            exec(b, ns)
    finally:
        os.chdir(cwd)
    return ns


def test_nc_stub_matches_the_layer_kernels():
    import mma_amd
    from mma_amd import functional as Fn
    ns = _stubs()
    rng = np.random.default_rng(0)
    N, H, K = 400, 32, 3
    deg = rng.integers(1, 9, N)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(N, d, replace=False)) for d in deg]).astype(np.int64)
    graph = mma_amd.NCGraph(rowptr, col, DEV, chunk=1 << 20)            # no hub chunks, as the sketch assumes
    x = torch.relu(torch.randn(N, H, device=DEV))
    Ws = [torch.randn(2 * H, H, device=DEV) * 0.2 for _ in range(K)]
    kinds, acts = [Fn.KIND["sum"], Fn.KIND["max"], Fn.KIND["mean"]], [Fn.ACT_SIGMOID] * K
    # the stub wants one work item per wavefront: (node, ebeg, eend, slot) rows, longest first
    order = np.argsort(-deg, kind="stable")
    items = torch.from_numpy(np.stack([order, rowptr[order], rowptr[order + 1], np.full(N, -1)], 1).astype(np.int32)).to(DEV)
    got = ns["learnable_all"](x, Ws, graph.rowptr, graph.col, items, kinds, acts, 0, 0)
    P = x @ torch.cat([w[:H] for w in Ws], 1)
    Q = x @ torch.cat([w[H:] for w in Ws], 1)
    want = Fn.nc_fused_aggregate(x, P, Q, graph, kinds, acts)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)


def test_gr_stub_matches_aggregate():
    from test_gr_gpu import make_conv
    ns = _stubs()
    rng = np.random.default_rng(0)
    N, E, T, F = 50, 300, 2, 8
    idx = torch.from_numpy(rng.integers(0, N - 5, E)).to(DEV)
    x = torch.randn(E, T, F, device=DEV)
    aggs, scal = ["min", "max", "mean"], ["identity", "amplification"]
    conv = make_conv(aggs, scal, towers=T, F=F)
    want = conv.aggregate(x, idx, N)
    got = ns["aggregate"](x, idx, N, aggs, scal, conv.avg_deg["log"], conv.avg_deg["lin"])
    assert torch.equal(got, want)
