"""SURVEY 8 f-2 / round-2 VERDICT item 6: the NC graph plan built ON THE DEVICE (NCGraph.from_device_csr, SpmmGraph.from_device_csr:
K6 radix sorts + scans) equals the host numpy plan bit for bit - CSR, transposed CSR (utils.py:97-100 ordering: ascending neighbour
id), work items, hub lists, the halo / own split of a shard - on the Cora and Pubmed structures, a hub-heavy random graph with halo
sources, and the C4 R-MAT graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FIELDS = ["rowptr", "col", "items", "hubs", "t_rowptr", "t_col", "t_eid", "t_items", "t_hubs", "inv_deg"]
SCALARS = ["N", "E", "n_src", "chunk", "edge_base", "n_slots", "t_n_slots", "n_wave_items", "t_n_wave_items", "max_degree"]


def _same(a, b, what):
    for f in SCALARS:
        assert getattr(a, f) == getattr(b, f), "%s: %s %r != %r" % (what, f, getattr(a, f), getattr(b, f))
    for f in FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), "%s: %s differs" % (what, f)
    assert (a.t_parts is None) == (b.t_parts is None)
    if a.t_parts is not None:
        for (ia, na, ha), (ib, nb, hb) in zip(a.t_parts, b.t_parts):
            assert na == nb and torch.equal(ia, ib) and torch.equal(ha, hb), what + ": halo / own parts differ"


def _both(rowptr, col, what, **kw):
    import mma_amd
    host = mma_amd.NCGraph(rowptr, col, DEV, **kw)
    dev = mma_amd.NCGraph.from_device_csr(torch.from_numpy(np.asarray(rowptr)).to(DEV), torch.from_numpy(np.asarray(col)).to(DEV), **kw)
    _same(host, dev, what)
    return host, dev


@pytest.mark.parametrize("fixture,H", [("cora_h64", 64), ("pubmed_h16", 16), ("citeseer_h128", 128)])
def test_device_plan_equals_the_numpy_plan_on_the_planetoid_structures(fixture, H):
    from tools.synth import golden_csr
    rowptr, col = golden_csr(fixture)
    _both(rowptr, col, fixture, H=H)
    _both(rowptr, col, fixture + " chunk 8", chunk=8, group_below=3, t_group_below=5)


def test_device_plan_with_hubs_empty_rows_and_halo_sources():
    rng = np.random.default_rng(5)
    N, S = 700, 1100                                      # 400 source-only (halo) rows behind the targets
    deg = rng.integers(0, 9, N)
    deg[[3, 250, 699]] = [900, 0, 333]
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.integers(0, S, d)) for d in deg]).astype(np.int64)
    _both(rowptr, col, "halo", n_src=S, chunk=64, group_below=4, t_group_below=6, edge_base=12345)
    _both(np.zeros(6, np.int64), np.zeros(0, np.int64), "no edges", chunk=16)


def test_device_plan_equals_the_numpy_plan_on_c4_and_is_timed():
    import time
    import mma_amd
    from tools.synth import rmat_graph
    rowptr, col = rmat_graph(20, 5_000_000, seed=42)
    t0 = time.perf_counter()
    host = mma_amd.NCGraph(rowptr, col, DEV)
    torch.cuda.synchronize(); t_host = time.perf_counter() - t0
    rp, cl = torch.from_numpy(rowptr).to(DEV), torch.from_numpy(col).to(DEV)
    mma_amd.NCGraph.from_device_csr(rp, cl)               # warm-up (allocator, rocPRIM)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev = mma_amd.NCGraph.from_device_csr(rp, cl)
    torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    _same(host, dev, "C4")
    print("C4 plan: numpy %.2f s, device %.3f s" % (t_host, t_dev))
    assert t_dev < t_host


def test_spmm_plan_on_the_device_equals_the_host_plan():
    import mma_amd
    from tools.synth import golden_csr
    rowptr, col = golden_csr("cora_h64")
    N = len(rowptr) - 1
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    host = mma_amd.graph.SpmmGraph(dst, col, None, N, N, DEV)
    dev = mma_amd.graph.SpmmGraph.from_device_csr(torch.from_numpy(rowptr).to(DEV), torch.from_numpy(col).to(DEV))
    for f in ["rowptr", "col", "t_rowptr", "t_col", "items", "hubs", "t_items", "t_hubs"]:
        assert torch.equal(getattr(host, f), getattr(dev, f)), f
    for f in ["n_rows", "n_cols", "n_slots", "t_n_slots", "n_wave_items", "t_n_wave_items"]:
        assert getattr(host, f) == getattr(dev, f), f
    assert host.val is None and dev.val is None
