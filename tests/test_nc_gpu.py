"""GPU parity tests for the node-classification hot path: the HIP kernels (through the C ABI and the drop-in
`mma_amd.MMA` module) against (a) the golden vectors captured from the reference and (b) the CPU oracle on
seeded random graphs with the edge cases the domain has (hubs split into chunks, degree-0/1 nodes, widths that
are not multiples of 4 or of the wave, K = 1..8, dropout).  Bar: fp32 within 1e-5 (see golden_util.check_close);
selection codes (max/min) and dropout keep bits bit-exact."""
import numpy as np
import pytest
import torch

from golden_util import CASES, Golden, check_close, nc_truth
from golden.inputs import ALL_MASK_NAMES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_module(gold, act, p, aggs, chunk=512):
    import mma_amd
    x, masks, weight, bias, cot = gold.torch_inputs(DEV)
    P = lambda t: torch.nn.Parameter(t.clone())
    mp = {n: P(masks[n]) for n in ALL_MASK_NAMES}
    w, b = P(weight), P(bias)
    mod = mma_amd.MMA(gold.add_all, act, 2, gold.H, gold.C, w, b, *[mp[n] for n in ALL_MASK_NAMES], p, list(aggs), DEV,
                      chunk=chunk)
    with torch.no_grad():   # the ctor re-initialises, like the reference: restore the fixture values
        for n in ALL_MASK_NAMES:
            mp[n].copy_(masks[n])
        w.copy_(weight); b.copy_(bias)
    return mod, mp, w, b, x, cot


def keep_tensor(gold, aggs, p):
    if p == 0:
        return None
    return torch.from_numpy(np.stack([gold.keep(a, p) for a in aggs]).astype(np.uint8)).to(DEV)


def adj_of(gold):
    z = gold.z
    idx = torch.from_numpy(np.stack([z["adj_row"], z["adj_col"]]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(z["adj_val"]), (gold.N, gold.N)).to(DEV)


@pytest.fixture(scope="module", params=CASES)
def gold(request):
    return Golden(request.param)


def test_golden_single_aggregators(gold):
    from mma_amd import functional as Fn
    for key in gold.single_keys():
        _, act, p, (agg,) = gold.parse(key)
        mod, mp, w, b, x, cot = build_module(gold, act, p, [agg])
        if p > 0:
            mod.drop_override = Fn.DropoutSpec(p, keep=keep_tensor(gold, [agg], p))
        with torch.no_grad():
            m = getattr(mod, "learnable_" + agg)(x, None)
        check_close(m, gold.z[key], gold.rows, gold.z[key + "/stats"], what=key)


@pytest.mark.parametrize("chunk,shared_bwd", [(512, False), (3, False), (512, True), (3, True)])
def test_golden_forward_and_grads(gold, chunk, shared_bwd, monkeypatch):
    from mma_amd import functional as Fn
    if chunk == 3 and gold.N > 3000:
        pytest.skip("small-chunk (hub path) variant only on the smaller graphs")
    monkeypatch.setattr(Fn, "SHARED_GRAD_BWD", shared_bwd)      # both K2b forms of the shared-gradient backward
    adj = adj_of(gold)
    for key in gold.set_keys():
        _, act, p, aggs = gold.parse(key)
        mod, mp, w, b, x, cot = build_module(gold, act, p, aggs, chunk=chunk)
        if p > 0:
            mod.drop_override = Fn.DropoutSpec(p, keep=keep_tensor(gold, aggs, p))
        x.requires_grad_(True)
        z = gold.z
        with torch.no_grad():
            ms = mod._aggregate_all(list(aggs), x)
        for a, m in zip(aggs, ms):
            check_close(m, z[key + "/m/" + a], gold.rows, z[key + "/m/" + a + "/stats"], what=key + "/m/" + a)
        out = mod(x, adj)
        tr = nc_truth(gold, key)          # the same quantities by the float64 oracle: sets the data-following slack
        check_close(out, z[key + "/out"], gold.rows, z[key + "/out/stats"], what=key + "/out", signed_sum=True, truth=tr["out"])
        grads = torch.autograd.grad((out * cot).sum(), [x, w, b] + [mp[a] for a in aggs])
        check_close(grads[0], z[key + "/gx"], gold.rows, None, what=key + "/gx", signed_sum=True, truth=tr["gx"])
        check_close(grads[1], z[key + "/gweight"], None, None, what=key + "/gweight", signed_sum=True, truth=tr["gweight"])
        check_close(grads[2], z[key + "/gbias"], None, None, what=key + "/gbias", signed_sum=True, truth=tr["gbias"])
        for a, g in zip(aggs, grads[3:]):
            check_close(g, z[key + "/gmask/" + a], None, None, what=key + "/gmask/" + a, signed_sum=True, truth=tr["gmask/" + a])


# ---- seeded random graphs vs the CPU oracle ------------------------------------------------------------
def random_graph(rng, N, avg_deg, hub_deg=0, isolated=2):
    deg = rng.poisson(avg_deg, N)
    deg[:isolated] = 0                      # degree-0 nodes (extension Q12: s = 0, mean divides by 1)
    deg[isolated:isolated + 2] = 1
    if hub_deg:
        deg[-1] = hub_deg
        deg[-2] = hub_deg // 2 + 1
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(N, size=d, replace=d > N)) for d in deg]).astype(np.int64) if rowptr[-1] else np.zeros(0, np.int64)
    return rowptr, col


CONFIGS = [
    # N, H, avg_deg, hub, names, activation, p, chunk
    (300, 128, 6, 700, ["sum", "mean", "max", "min"], "new_sigmoid", 0.0, 256),
    (300, 128, 6, 700, ["sum", "mean", "max", "min"], "new_sigmoid", 0.5, 256),
    (257, 64, 3, 0, ["mean", "mean2"], "new_sigmoid", 0.5, 512),
    (500, 16, 4, 200, ["min", "min2", "min3", "min4"], "new_sigmoid", 0.25, 64),
    (120, 75, 5, 90, ["sum3", "max2", "softmax"], "sigmoid", 0.5, 32),          # H % 4 != 0 -> scalar path
    (90, 6, 2, 0, ["mean3"], "new_sigmoid", 0.0, 512),
    (64, 260, 3, 70, ["sum", "mean", "max", "min", "sum2", "mean2", "max2", "min2"], "new_sigmoid", 0.5, 16),  # K=8, H>256
    (200, 32, 5, 100, ["sum", "mean", "max", "min", "sum2"], "sigmoid", 0.5, 48),   # K=5 -> slices 4+1
    (150, 20, 4, 0, ["sum", "mean", "max", "min", "softmin", "mean4", "max4"], "new_sigmoid", 0.0, 512),  # K=7
    # round 5: probabilities that are no multiple of 1/256 (16-bit thresholds, kernels' HASH16 form): vector + hub, scalar path, K=8, slices 4+1
    (300, 128, 6, 700, ["sum", "mean", "max", "min"], "new_sigmoid", 0.6, 256),
    (120, 75, 5, 90, ["sum3", "max2", "softmax"], "sigmoid", 0.3, 32),
    (64, 260, 3, 70, ["sum", "mean", "max", "min", "sum2", "mean2", "max2", "min2"], "new_sigmoid", 0.1, 16),
    (200, 32, 5, 100, ["sum", "mean", "max", "min", "sum2"], "sigmoid", 0.9373, 48),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "N%d_H%d_K%d_p%g_c%d" % (c[0], c[1], len(c[4]), c[6], c[7]))
def test_random_graph_vs_oracle(cfg):
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import nc_oracle as O
    from oracle.dropout_rng import keep_mask16, threshold16
    N, H, avg_deg, hub, names, act, p_asked, chunk = cfg
    thr = threshold16(p_asked)
    p = thr / 65536.0                      # the probability the kernels apply (|p - p_asked| <= 2^-17); the oracle divides by 1 - p
    rng = np.random.default_rng(1234 + N + H)
    rowptr, col = random_graph(rng, N, avg_deg, hub)
    E = int(rowptr[-1])
    K = len(names)
    x = torch.from_numpy(np.maximum(rng.standard_normal((N, H)), 0).astype(np.float32))
    Ws = {n: torch.from_numpy(((rng.random((2 * H, H)) * 2 - 1) / np.sqrt(H)).astype(np.float32)) for n in names}
    cot = torch.from_numpy(rng.standard_normal((K, N, H)).astype(np.float32))
    seed = 0x1234567890ABCDEF

    # oracle (CPU), fed the keep mask the kernel's RNG produces
    keep = keep_mask16(seed, thr, K, E, H) if p > 0 else None

    def oracle(dtype):
        xo = x.to(dtype).requires_grad_(True)
        Wo = {n: Ws[n].to(dtype).requires_grad_(True) for n in names}
        mo = torch.stack([O.aggregate(n, xo, Wo[n], rowptr, col, act, p, None if keep is None else keep[k])
                          for k, n in enumerate(names)])
        return mo, torch.autograd.grad((mo * cot.to(dtype)).sum(), [xo] + [Wo[n] for n in names])
    mo, go = oracle(torch.float32)
    _, g64 = oracle(torch.float64)          # exact value of the same formulas: sets the slack of the signed sums

    # HIP
    graph = mma_amd.NCGraph(rowptr, col, DEV, chunk=chunk)
    if hub:
        assert graph.n_slots > 0 and graph.t_n_slots >= 0
    xg = x.to(DEV).requires_grad_(True)
    Wg = {n: Ws[n].to(DEV).requires_grad_(True) for n in names}
    kinds = [Fn.KIND[O.AGGREGATORS[n][0]] for n in names]
    acts = [Fn.ACT_RAW if O.uses_raw_logits(n, act) else Fn.ACT_SIGMOID for n in names]
    P = xg @ torch.cat([Wg[n][:H] for n in names], 1)
    Q = xg @ torch.cat([Wg[n][H:] for n in names], 1)
    spec = Fn.DropoutSpec(p_asked, seed=seed)
    assert spec.thr == thr and abs(spec.p_applied - p_asked) <= 8e-6
    mg = Fn.nc_fused_aggregate(xg, P, Q, graph, kinds, acts, spec)
    gg = torch.autograd.grad((mg * cot.to(DEV)).sum(), [xg] + [Wg[n] for n in names])

    rows = np.arange(N)
    for k, n in enumerate(names):
        check_close(mg[k], mo[k].detach().numpy(), rows, None, what="m/" + n)
    check_close(gg[0], go[0].numpy(), rows, None, what="gx", signed_sum=True, truth=g64[0].numpy())
    for n, a, b, t in zip(names, gg[1:], go[1:], g64[1:]):
        check_close(a, b.numpy(), None, None, what="gW/" + n, signed_sum=True, truth=t.numpy())

    # explicit-mask mode must agree bit-for-bit with hash mode given the same bits
    if p > 0:
        with torch.no_grad():
            me = Fn.nc_fused_aggregate(xg.detach(), P.detach(), Q.detach(), graph, kinds, acts,
                                       Fn.DropoutSpec(p, keep=torch.from_numpy(keep).to(DEV)))
        assert torch.equal(me, mg.detach()), "hash-mode and explicit-mode dropout disagree"
    # determinism: no atomics anywhere, a second run is bitwise identical
    with torch.no_grad():
        m2 = Fn.nc_fused_aggregate(xg.detach(), P.detach(), Q.detach(), graph, kinds, acts, Fn.DropoutSpec(p, seed=seed))
    assert torch.equal(m2, mg.detach())


def test_selection_codes_bit_exact_on_ties():
    """max/min against x_i with exact ties: output and the 0.5/0.5 gradient split (torch.max backward) exact."""
    import mma_amd
    from mma_amd import functional as Fn
    # 3 nodes; node 0 <- {1}; raw logits with P=Q=0.5 so mask z == 1.0 exactly => s == x_1
    rowptr, col = np.array([0, 1, 1, 1]), np.array([1])
    H = 4
    x = torch.tensor([[1., 2., 3., 4.], [1., 5., 3., 0.], [0., 0., 0., 0.]], device=DEV)
    P = torch.full((3, 2 * H), 0.5, device=DEV)
    Q = torch.full((3, 2 * H), 0.5, device=DEV)
    graph = mma_amd.NCGraph(rowptr, col, DEV)
    xg = x.clone().requires_grad_(True)
    m = Fn.nc_fused_aggregate(xg, P, Q, graph, [Fn.KIND["max"], Fn.KIND["min"]], [Fn.ACT_RAW, Fn.ACT_RAW])
    assert torch.equal(m[0, 0], torch.tensor([1., 5., 3., 4.], device=DEV))
    assert torch.equal(m[1, 0], torch.tensor([1., 2., 3., 0.], device=DEV))
    g = torch.zeros_like(m); g[:, 0] = 1.0
    gx, = torch.autograd.grad((m * g).sum(), [xg])
    # d/dx_0: max: [tie .5, 0, tie .5, 1] ; min: [tie .5, 1, tie .5, 0]
    assert torch.equal(gx[0], torch.tensor([1.0, 1.0, 1.0, 1.0], device=DEV))
    # d/dx_1 (through s, mask == 1): max: [.5, 1, .5, 0], min: [.5, 0, .5, 1]
    assert torch.equal(gx[1], torch.tensor([1.0, 1.0, 1.0, 1.0], device=DEV))


def test_spmm_matches_torch_sparse():
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.graph import SpmmGraph
    rng = np.random.default_rng(5)
    N, C, K = 300, 7, 3
    nnz = 2000
    r, c = rng.integers(0, N, nnz), rng.integers(0, N, nnz)
    v = rng.standard_normal(nnz).astype(np.float32)
    adj = torch.sparse_coo_tensor(torch.tensor(np.stack([r, c])), torch.tensor(v), (N, N)).coalesce()
    B = torch.from_numpy(rng.standard_normal((K * N, C)).astype(np.float32)).requires_grad_(True)
    bias = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).requires_grad_(True)
    ref = torch.sparse.mm(torch.cat((adj,) * K, 1), B) + bias
    cot = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32))
    gref = torch.autograd.grad((ref * cot).sum(), [B, bias])
    sg = SpmmGraph.from_torch_sparse(adj.to(DEV))
    Bg = B.detach().to(DEV).requires_grad_(True); bg = bias.detach().to(DEV).requires_grad_(True)
    out = Fn.csr_spmm(Bg, bg, sg, K)
    gg = torch.autograd.grad((out * cot.to(DEV)).sum(), [Bg, bg])
    B64, b64 = B.detach().double().requires_grad_(True), bias.detach().double().requires_grad_(True)
    r64 = torch.sparse.mm(torch.cat((adj.double(),) * K, 1), B64) + b64
    g64 = torch.autograd.grad((r64 * cot.double()).sum(), [B64, b64])
    check_close(out, ref.detach().numpy(), None, None, what="spmm", signed_sum=True, truth=r64.detach().numpy())
    check_close(gg[0], gref[0].numpy(), None, None, what="spmm/gB", signed_sum=True, truth=g64[0].numpy())
    check_close(gg[1], gref[1].numpy(), None, None, what="spmm/gbias", signed_sum=True, truth=g64[1].numpy())


def test_unusable_aggregators_and_errors():
    gold = Golden("toy6_h8")
    with pytest.raises(KeyError):
        build_module(gold, "sigmoid", 0.0, ["mean", "bogus"])
    mod, *_ = build_module(gold, "sigmoid", 0.0, ["std"])
    x = gold.torch_inputs(DEV)[0]
    with pytest.raises(NotImplementedError):
        mod(x, adj_of(gold))


@pytest.mark.parametrize("N,H,names,edges", [
    (1, 4, ["sum"], []),                                   # single isolated node
    (5, 1, ["mean", "max"], [(0, 1), (1, 0), (4, 4)]),     # H = 1, a self loop, isolated nodes
    (7, 12, ["sum", "mean", "max", "min", "softmax", "softmin"], []),          # no edges at all, K = 6
    (3, 8, ["min"], [(0, 1), (0, 1), (0, 2), (1, 0)]),     # duplicate (multi-)edges
])
def test_degenerate_graphs(N, H, names, edges):
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import nc_oracle as O
    rng = np.random.default_rng(N * 10 + H)
    deg = np.zeros(N, dtype=np.int64)
    for i, _ in edges:
        deg[i] += 1
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    col = np.array([j for i in range(N) for (t, j) in edges if t == i], dtype=np.int64)
    x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
    Ws = {n: torch.from_numpy((rng.standard_normal((2 * H, H)) * 0.3).astype(np.float32)) for n in names}
    def oracle(dtype):
        xo = x.to(dtype).requires_grad_(True)
        mo = torch.stack([O.aggregate(n, xo, Ws[n].to(dtype), rowptr, col, "new_sigmoid") for n in names])
        return mo, torch.autograd.grad(mo.sum(), [xo])[0]
    mo, go = oracle(torch.float32)
    m64, g64 = oracle(torch.float64)
    graph = mma_amd.NCGraph(rowptr, col, DEV)
    xg = x.to(DEV).requires_grad_(True)
    kinds = [Fn.KIND[O.AGGREGATORS[n][0]] for n in names]
    acts = [Fn.ACT_RAW if O.uses_raw_logits(n, "new_sigmoid") else Fn.ACT_SIGMOID for n in names]
    P = xg @ torch.cat([Ws[n][:H] for n in names], 1).to(DEV)
    Q = xg @ torch.cat([Ws[n][H:] for n in names], 1).to(DEV)
    mg = Fn.nc_fused_aggregate(xg, P, Q, graph, kinds, acts)
    gg, = torch.autograd.grad(mg.sum(), [xg])
    check_close(mg.reshape(-1, H), mo.detach().reshape(-1, H).numpy(), None, None, what="degenerate m")
    check_close(gg, go.numpy(), None, None, what="degenerate gx", signed_sum=True, truth=g64.numpy())
    ms = Fn.nc_fused_aggregate(xg.detach(), P.detach(), Q.detach(), graph, kinds, acts, reduce_k=True)
    check_close(ms, mo.detach().sum(0).numpy(), None, None, what="degenerate msum", signed_sum=True, truth=m64.detach().sum(0).numpy())


@pytest.mark.parametrize("name,n_edges", [("softmax", 19), ("softmin", 19), ("softmax", 22), ("softmax", 16), ("softmax", 17)])
def test_degenerate_softmax_gradient_is_nan_where_the_reference_gradient_is(name, n_edges):
    """learnable_softmax / softmin return (e / e) * s with e = exp(+-s) (layers.py:676-682, 716-720).  Beyond the overflow of the
    value (|s| > ~104 on the small side, 88.7 on the large side) there is a band in which the value is still s but autograd's
    division backward forms g s / e - g s ((e / e) / e) = inf - inf: the reference's fp32 gradient is NaN (found by the
    generated cases; fp64 / exact arithmetic give g).  Node 2 gets s = -+5 per edge and column:
      16 edges -> |s| = 80: finite everywhere;  19 -> 95: 1/e overflows, NaN gradient whatever g (reproduced: selection code 3);
      22 -> 110: NaN value;  17 -> 85: e is a normal number and only g s / e overflows - that depends on the size of the
      incoming gradient, which the forward kernel cannot know: KNOWN DEVIATION, the HIP path returns the
      exact-arithmetic gradient g where the reference returns NaN."""
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import nc_oracle as O
    N, H = 3, 4
    sign = 1.0 if name == "softmax" else -1.0
    rowptr = np.array([0, 0, 1, 1 + n_edges])
    col = np.array([0] + [k % 2 for k in range(n_edges)], dtype=np.int64)
    x = torch.full((N, H), 5.0)
    W = torch.full((2 * H, H), -sign / 40.0)                 # z = -+1 on every edge and column; raw logits (new_sigmoid quirk, Q5)
    def oracle(dtype):
        xo = x.to(dtype).requires_grad_(True)
        m = O.aggregate(name, xo, W.to(dtype), rowptr, col, "new_sigmoid")
        return m.detach(), torch.autograd.grad(m.sum(), [xo])[0]
    mo, go = oracle(torch.float32)
    _, g64 = oracle(torch.float64)
    assert torch.isnan(mo).any() == (n_edges >= 22) and torch.isnan(go).any() == (n_edges >= 17) and torch.isfinite(g64).all()
    graph = mma_amd.NCGraph(rowptr, col, DEV)
    xg = x.to(DEV).requires_grad_(True)
    kinds, acts = [Fn.KIND[O.AGGREGATORS[name][0]]], [Fn.ACT_RAW]
    mg = Fn.nc_fused_aggregate(xg, xg @ W[:H].to(DEV), xg @ W[H:].to(DEV), graph, kinds, acts)
    gg, = torch.autograd.grad(mg.sum(), [xg])
    check_close(mg.reshape(-1, H), mo.reshape(-1, H).numpy(), None, None, what="softmax band m")
    if n_edges == 17:
        check_close(gg, g64.float().numpy(), None, None, what="softmax band gx (deviation: exact-arithmetic gradient)", signed_sum=True)
    else:
        check_close(gg, go.numpy(), None, None, what="softmax band gx", signed_sum=True)


@pytest.mark.parametrize("compound,scalers", [(False, None), (True, ["identity", "amplification", "attenuation", "linear", "inverse_linear"]),
                                              (True, ["amplification", "identity", "inverse_linear"])])
def test_true_degree_scalers_vs_oracle(compound, scalers):
    """strict_reference=False (BASELINE configs[4] "+ all scalers"): the layer with TRUE-degree scalers - the reference's
    three as PNA meant them, or the compounding five of mma_conv.py:181-196 - against the oracle's literal restatement
    (cat of the scaled blocks, mm with the stacked weight), forward and every gradient; the default stays the reference's
    degenerate factor."""
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.layers import _MASK_NAMES
    from oracle import nc_oracle as O
    rng = np.random.default_rng(99)
    N, H, C, names, act = 300, 32, 5, ["sum", "mean", "max", "min"], "new_sigmoid"
    rowptr, col = random_graph(rng, N, 5, hub_deg=150)
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    x = torch.from_numpy(np.maximum(rng.standard_normal((N, H)), 0).astype(np.float32))
    cot = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32))
    Ws, weight, bias = O.init_like_reference(H, C, names, 3)
    sc = scalers if scalers is not None else ["identity", "amplification", "attenuation"]

    def oracle(dtype):
        xo, wo, bo = x.to(dtype).requires_grad_(True), weight.to(dtype).requires_grad_(True), bias.to(dtype).requires_grad_(True)
        Wo = {n: Ws[n].to(dtype).requires_grad_(True) for n in names}
        out = O.mma_forward(names, xo, Wo, wo, bo, rowptr, col, dst, col, np.ones(len(col), np.float32), act,
                            true_degree_scalers=sc, compound=compound)
        return out.detach(), torch.autograd.grad((out * cot.to(dtype)).sum(), [xo, wo, bo] + [Wo[n] for n in names])
    want, gw = oracle(torch.float32)
    w64, g64 = oracle(torch.float64)

    P = lambda t: torch.nn.Parameter(t.clone().to(DEV))
    mp = {n: P(Ws[n]) if n in names else P(torch.zeros(2, 1)) for n in _MASK_NAMES}
    w, b = P(weight), P(bias)
    add_all = [col[rowptr[i]:rowptr[i + 1]] for i in range(N)]
    mod = mma_amd.MMA(add_all, act, 2, H, C, w, b, *[mp[n] for n in _MASK_NAMES], 0.0, names, DEV, chunk=64,
                      strict_reference=False, scalers=scalers, compound_scalers=compound)
    with torch.no_grad():
        for n in names:
            mp[n].copy_(Ws[n])
        w.copy_(weight); b.copy_(bias)
    adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = mod(xg, adj)
    gg = torch.autograd.grad((out * cot.to(DEV)).sum(), [xg, w, b] + [mp[n] for n in names])
    check_close(out, want.numpy(), None, None, what="true-degree out", signed_sum=True, truth=w64.numpy())
    for name, a, r, t in zip(["gx", "gweight", "gbias"] + ["gmask/" + n for n in names], gg, gw, g64):
        check_close(a, r.numpy(), None, None, what="true-degree " + name, signed_sum=True, truth=t.numpy())
    # the extension is opt-in: the default module on the same inputs gives the reference's (different) result
    ref = mma_amd.MMA(add_all, act, 2, H, C, w, b, *[mp[n] for n in _MASK_NAMES], 0.0, names, DEV, chunk=64)
    with torch.no_grad():
        for n in names:
            mp[n].copy_(Ws[n])
        w.copy_(weight); b.copy_(bias)
    assert not torch.allclose(ref(xg, adj), out)
    with pytest.raises(ValueError):
        mma_amd.MMA(add_all, act, 2, H, C, w, b, *[mp[n] for n in _MASK_NAMES], 0.0, names, DEV, scalers=["linear"])


@pytest.mark.parametrize("N,H,names,p,chunk,hub", [
    (400, 64, ["mean", "mean2"], 0.75, 16, 170),                                   # Cora's layer: hubs split into 11 chunks
    (300, 16, ["min", "min2", "min3", "min4"], 0.5, 64, 200),
    (257, 128, ["sum", "mean", "max", "min"], 0.5, 32, 300),
    (200, 32, ["sum", "mean", "max", "min", "sum2", "mean2", "max2", "softmax"], 0.0, 8, 90),      # K = 8
    (150, 64, ["max"], 0.25, 512, 0),                                              # no hubs: wave + grouped items only
])
def test_one_launch_form_equals_the_separate_launches(N, H, names, p, chunk, hub, monkeypatch):
    """Small graphs run K1 / K2b as ONE launch each (wave items + grouped items + the hub sums by the wavefront that stores the last
    chunk partial, mma_amd.h `sync`): bit for bit the results of the separate launches, forward and backward, repeatedly (the
    ticket counter must be back at zero after every call)."""
    import mma_amd
    from mma_amd import functional as Fn, graph as G
    from oracle import nc_oracle as O
    rng = np.random.default_rng(N + H + len(names))
    rowptr, col = random_graph(rng, N, 4, hub)
    graph = mma_amd.NCGraph(rowptr, col, DEV, chunk=chunk, group_below=4, t_group_below=4)
    assert (graph.n_slots > 0) == bool(hub) and 0 < graph.n_wave_items < graph.items.shape[0]
    K = len(names)
    x = torch.from_numpy(np.maximum(rng.standard_normal((N, H)), 0).astype(np.float32)).to(DEV)
    wcat = torch.from_numpy(((rng.random((H, 2 * K * H)) * 2 - 1) / np.sqrt(H)).astype(np.float32)).to(DEV)
    cot = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(DEV)
    kinds = [Fn.KIND[O.AGGREGATORS[n][0]] for n in names]
    acts = [Fn.ACT_RAW if O.uses_raw_logits(n, "new_sigmoid") else Fn.ACT_SIGMOID for n in names]

    def run():
        xg, wg = x.clone().requires_grad_(True), wcat.clone().requires_grad_(True)
        out = Fn.nc_local_layer(xg, wg, None, graph, kinds, acts, Fn.DropoutSpec(p, seed=77))
        return (out.detach(),) + torch.autograd.grad((out * cot).sum(), [xg, wg])
    monkeypatch.setattr(G, "ONE_LAUNCH", False)
    assert graph.sync(0) is None
    ref = run()
    monkeypatch.setattr(G, "ONE_LAUNCH", True)
    assert graph.sync(0) is not None
    for _ in range(3):
        got = run()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        assert int(graph._sync.abs().sum()) == 0


@pytest.mark.parametrize("C", [3, 7, 16, 64])
def test_spmm_one_launch_equals_the_two_launches(C, monkeypatch):
    """Round 4: long items (one per wavefront) and short items (one per lane group) of the K = 1 SpMM run in ONE launch; the same items
    walked the same way - bit-identical to the two launches (MMA_SPMM_ONE_LAUNCH=0), forward and transposed, with hub chunks."""
    from mma_amd import functional as Fn
    from mma_amd.graph import SpmmGraph
    rng = np.random.default_rng(C)
    N = 3000
    deg = np.minimum(rng.zipf(1.7, N), 1500)                     # most rows short, a few hubs beyond the 512-edge chunk
    row = np.repeat(np.arange(N), deg)
    col = rng.integers(0, N, row.size)
    adj = torch.sparse_coo_tensor(torch.tensor(np.stack([row, col])), torch.ones(row.size), (N, N)).coalesce()
    sg = SpmmGraph.from_torch_sparse(adj.to(DEV))
    assert 0 < sg.n_wave_items < sg.items.shape[0]
    B = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(DEV)
    cot = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32)).to(DEV)
    res = []
    for sw in ("1", "0"):
        monkeypatch.setenv("MMA_SPMM_ONE_LAUNCH", sw)
        Bg = B.clone().requires_grad_(True)
        out = Fn.csr_spmm(Bg, bias, sg, 1)
        out.backward(cot)
        res.append((out.detach().clone(), Bg.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ref = torch.sparse.mm(adj.to(DEV), B) + bias
    assert (res[0][0] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
