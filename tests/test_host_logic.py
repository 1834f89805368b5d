"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports every symbol include/mma_amd.h declares
(no compute calls), argument checks fire on the host, and the host-side graph plan is right."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from mma_amd import _lib
from mma_amd.graph import make_items, transpose_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mma_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mma_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert "mma_nc_fused_fwd" in names and "mma_nc_fused_bwd" in names and len(names) >= 8
    for n in names:
        assert hasattr(L, n), "libmma_amd.so does not export " + n
    for n in _lib.PROTOTYPES:
        assert n in names, "binding %s is not declared in the header" % n
    assert _lib.lib().mma_abi_version() == _lib.ABI_VERSION


def test_bindings_are_generated_from_the_header_and_the_op_library_registers_every_launcher():
    """mma_amd/_abi.py and csrc/torch_ops.cpp are generated from include/mma_amd.h (tools/gen_bindings.py --check), and the
    built op library registers one `torch.ops.mma_amd.<entry point>` per launcher of the header."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _lib.binding() == "torch", "libmma_amd_torch.so not built/loaded"
    for name in _lib.PROTOTYPES:
        op = getattr(torch.ops.mma_amd, name)
        n_args = len(op.default._schema.arguments)
        assert n_args == len(_lib.PROTOTYPES[name]) - 1, name           # every parameter but the stream


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_argument_checks_run_on_host_in_both_bindings(binding, monkeypatch):
    monkeypatch.setattr(_lib, "_ops", None if binding == "torch" else False)
    if binding == "torch":
        monkeypatch.delenv("MMA_BINDING", raising=False)
    with pytest.raises(_lib.MMALibraryError, match="pitch"):
        _lib.call("mma_csr_spmm", None, None, None, None, 2, 5, 1, None, None, 2, 5, 4, None)
    with pytest.raises(TypeError):
        _lib.call("mma_csr_spmm", None)


def test_argument_checks_run_on_host_without_gpu():
    # a bad shape must be refused before any launch (works with no GPU present)
    with pytest.raises(_lib.MMALibraryError, match="K="):
        _lib.call("mma_nc_fused_fwd", None, 4, None, 4, None, 4, None, None, None, 0, 0, None, 0, None, 0, None, None, 4, None, None, 4, None, 0,
                  10, 10, 4, 99, None, None, 0, 0, 0, None, 0, None, None, None)
    with pytest.raises(_lib.MMALibraryError, match="pitch"):
        _lib.call("mma_csr_spmm", None, None, None, None, 2, 5, 1, None, None, 2, 5, 4, None)


def test_tower_post_fits_is_the_kernels_own_limit():
    """the Python gates (mma_conv.py `factored`, dense._skinny_ok) ask the library which shapes K13 / K14 / K16 take instead
    of restating its limits: KF <= 512, S <= 5 and weights + wave tiles inside 160 KB of LDS in both layouts."""
    from mma_amd import dense
    assert dense.tower_post_fits(152, 3)            # ZINC: 2 aggregators x 76 columns, 3 scalers
    assert dense.tower_post_fits(75, 5)             # the 75 -> 75 Linear layers (S = ceil(75/16))
    assert not dense.tower_post_fits(608, 3)        # 4 aggregators at F_in = 150: K*Fw > 512
    assert not dense.tower_post_fits(512, 5)        # Linear(512 -> 80): 160 KB of weights alone
    assert dense.tower_post_fits(512, 3) and not dense.tower_post_fits(0, 1) and not dense.tower_post_fits(8, 6)
    KFp, tiles = 512, 4 * 64 * 34
    for S in range(1, 6):                           # the answer IS the two LDS sums of tower_post.hip
        fits = 4 * (KFp * S * 16 + tiles) <= 160 * 1024 and 4 * (S * 16 * (KFp + 16) + tiles) <= 160 * 1024
        assert dense.tower_post_fits(512, S) == fits, S


def test_host_code_lists_are_length_checked_by_the_torch_ops():
    """a `*_host` list shorter than the count the call states used to feed uninitialised stack bytes to the C ABI."""
    if _lib.binding() != "torch":
        pytest.skip("torch-op binding not built")
    with pytest.raises(_lib.MMALibraryError, match="scaler_host has 1 entries, the call says 3"):
        _lib.call("mma_tower_post_pre", None, None, 10, 3, [0], 1.0, 1.0, None)


def test_cpu_tensors_are_refused():
    import mma_amd
    from mma_amd import functional as Fn
    x = torch.zeros(4, 4)
    with pytest.raises(_lib.MMALibraryError, match="GPU only"):
        Fn.nc_fused_aggregate(x, torch.zeros(4, 4), torch.zeros(4, 4), None, [0], [0])


def test_make_items_chunks_and_hubs():
    rowptr = np.array([0, 0, 3, 13, 14, 14, 30])     # degrees 0,3,10,1,0,16
    items, hubs, n_slots = make_items(rowptr, chunk=4)
    # every edge covered exactly once, in order, by its node's items
    cover = np.zeros(30, dtype=int)
    for node, b, e, slot in items:
        assert rowptr[node] <= b <= e <= rowptr[node + 1] and e - b <= 4
        cover[b:e] += 1
    assert (cover == 1).all()
    whole = items[items[:, 3] < 0]
    assert sorted(whole[:, 0].tolist()) == [0, 1, 3, 4]           # degree <= chunk (incl. degree 0): one item, no slot
    assert hubs[:, 0].tolist() == [2, 5] and n_slots == 3 + 4
    for node, sb, se, _ in hubs:
        mine = items[(items[:, 0] == node)]
        mine = mine[np.argsort(mine[:, 1])]
        assert mine[:, 3].tolist() == list(range(sb, se))          # slots of a hub are consecutive, in edge order
    assert (np.diff(items[:, 2] - items[:, 1]) <= 0).all()        # longest items first


def test_transpose_csr_roundtrip():
    rng = np.random.default_rng(0)
    N, S = 50, 60
    deg = rng.integers(0, 7, N)
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    col = rng.integers(0, S, rowptr[-1])
    t_rowptr, t_col, t_eid = transpose_csr(rowptr, col, S)
    dst = np.repeat(np.arange(N), deg)
    assert (col[t_eid] == np.repeat(np.arange(S), np.diff(t_rowptr))).all()
    assert (dst[t_eid] == t_col).all()
    for s in range(S):   # stable: forward positions ascend within a source
        seg = t_eid[t_rowptr[s]:t_rowptr[s + 1]]
        assert (np.diff(seg) > 0).all()


def test_integration_md_stub_matches_binding():
    """The ctypes stub shown to the reference's maintainers must have the ABI's current prototype."""
    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"L\.mma_nc_fused_fwd\.argtypes = \[(.*?)\]", src, re.S)
    toks = [t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip()]
    names = {"P": _lib._P, "I64": _lib._I64, "I32": _lib._I32, "U32": _lib._U32, "U64": _lib._U64}
    assert [names[t] for t in toks] == _lib.PROTOTYPES["mma_nc_fused_fwd"]
    assert "ABI version %d" % _lib.ABI_VERSION in src


@pytest.mark.parametrize("towers,F,edge_dim", [(5, 75, 50), (2, 8, None), (1, 6, 3)])
def test_weight_plan_table_reproduces_the_torch_plumbing(towers, F, edge_dim):
    """K18's block table (mma_conv._WeightPlan) applied in numpy = the matrices MMAConv.forward builds with stack / slice / pad / cat:
    [Wi;Wj], the bias pair, We, Wx, Wo, bp - zero padding included.  Host logic only: no kernel runs."""
    import ctypes
    import mma_amd
    torch.manual_seed(1)
    conv = mma_amd.MMAConv(F, F * towers, ["min", "max"], ["identity", "amplification", "linear"], torch.tensor([0, 10, 30, 50, 10]),
                           edge_dim=edge_dim, towers=towers)
    lins = [seq[0].active_linear() for seq in conv.pre_nns["max"]]
    if lins[0].weight.is_cuda:
        pytest.skip("the mask Linears sit on a GPU here (G2): the table check reads host memory")
    plan = conv._packed_weights(lins, edge_dim is not None, torch.device("cpu"), plan_only=True)
    outs = [np.full(shp, np.nan, np.float32) for shp in plan.outs]
    for a, lda, rows, cols, oi, off, ldb, b_rows, b_cols, flags in plan.fwd.tolist():
        flat = outs[oi].reshape(-1)
        src = np.ctypeslib.as_array(ctypes.cast(a, ctypes.POINTER(ctypes.c_float)), shape=(max(rows, 1) * lda,)) if rows else None
        for r in range(b_rows):
            for c in range(b_cols):
                flat[off + r * ldb + c] = src[r * lda + c] if (r < rows and c < cols) else 0.0
    T, Fw, Fo = towers, conv.fused_width(), conv.F_out
    KS = len(conv.aggregators) * len(conv.scalers)
    TF = T * Fw
    Wall = torch.stack([l.weight for l in lins]).detach()

    def rows_(lo, hi):
        return conv._pad_dim(Wall[:, :, lo:hi], 1, Fw).reshape(TF, hi - lo).numpy()
    b = conv._pad_dim(torch.stack([l.bias for l in lins]).detach(), 1, Fw).reshape(TF).numpy()
    Wp = torch.stack([seq[0].weight for seq in conv.post_nns]).detach()
    want = [np.concatenate([rows_(0, F), rows_(F, 2 * F)]), np.concatenate([b, np.zeros_like(b)]),
            rows_(2 * F, 3 * F) if edge_dim is not None else None, Wp[:, :, :F].reshape(T * Fo, F).numpy(),
            conv._pad_dim(Wp[:, :, F:].reshape(T, Fo, KS, F), 3, Fw).reshape(T, Fo, KS * Fw).numpy(),
            torch.cat([seq[0].bias for seq in conv.post_nns]).detach().numpy()]
    for got, w in zip(outs, want):
        if w is not None:
            assert got.shape == w.shape and np.array_equal(got, w)
    # the way back: every element of every Parameter's gradient is written exactly once
    hits = np.zeros(plan.g_off[-1], np.int32)
    n_acc_floats = plan.g_off[plan.n_acc]
    base = plan.acc.data_ptr() if plan.n_acc else 0
    for a, lda, rows, cols, oi, off, ldb, b_rows, b_cols, flags in plan.bwd.tolist():
        start = (a - base) // 4 if flags & 1 else n_acc_floats + a
        for r in range(rows):
            hits[start + r * lda:start + r * lda + cols] += 1
    assert hits.min() == 1 and hits.max() == 1
