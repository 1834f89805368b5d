"""Hypothesis-generated CSR graphs through the fused kernels: empty rows, degree-1 nodes, hubs longer than a
work-item chunk, duplicate edges, widths that are not multiples of 4 or of the wave, every K from 1 to 6 - NC forward and
backward against the CPU oracle; and GR aggregate() on random target lists (empty targets, long segments, exact ties)
against the oracle's scatter, with the second (sequential-loop) restatement deciding the arg of min/max."""
import os

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from golden_util import check_close

N_EXAMPLES = int(os.environ.get("MMA_HYP_EXAMPLES", "60"))         # a longer one-off search: MMA_HYP_EXAMPLES=1000 MMA_HYP_RANDOM=1
DERANDOMIZE = not os.environ.get("MMA_HYP_RANDOM")
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NC_NAMES = ["sum", "mean", "max", "min", "sum2", "mean3", "max2", "min3", "softmax", "softmin"]


@st.composite
def nc_case(draw):
    N = draw(st.integers(1, 60))
    H = draw(st.sampled_from([1, 3, 4, 7, 8, 16, 20, 33, 64, 68]))
    names = draw(st.lists(st.sampled_from(NC_NAMES), min_size=1, max_size=6, unique=True))
    degs = draw(st.lists(st.one_of(st.just(0), st.just(1), st.integers(0, 6), st.integers(20, 90)), min_size=N, max_size=N))
    chunk = draw(st.sampled_from([3, 16, 512]))
    p = draw(st.sampled_from([0.0, 0.25, 0.5]))
    act = draw(st.sampled_from(["sigmoid", "new_sigmoid"]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    return N, H, names, degs, chunk, p, act, seed


@settings(max_examples=N_EXAMPLES, deadline=None, suppress_health_check=list(HealthCheck), derandomize=DERANDOMIZE)
@given(nc_case())
def test_nc_kernels_on_generated_graphs(case):
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import nc_oracle as O
    from oracle.dropout_rng import keep_mask
    N, H, names, degs, chunk, p, act, seed = case
    rng = np.random.default_rng(seed)
    rowptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    col = rng.integers(0, N, rowptr[-1]).astype(np.int64)               # duplicates and self loops allowed
    E, K = int(rowptr[-1]), len(names)
    x = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
    Ws = {n: torch.from_numpy((rng.standard_normal((2 * H, H)) * 0.3).astype(np.float32)) for n in names}
    cot = torch.from_numpy(rng.standard_normal((K, N, H)).astype(np.float32))
    thr = int(round(p * 256))
    keep = keep_mask(seed, thr, K, E, H) if p > 0 else None

    def oracle(dtype):
        xo = x.to(dtype).requires_grad_(True)
        mo = torch.stack([O.aggregate(n, xo, Ws[n].to(dtype), rowptr, col, act, p, None if keep is None else keep[k])
                          for k, n in enumerate(names)])
        return mo.detach(), torch.autograd.grad((mo * cot.to(dtype)).sum(), [xo])[0]
    mo, go = oracle(torch.float32)
    m64, g64 = oracle(torch.float64)
    if not torch.isfinite(mo).all() or not torch.isfinite(go).all():
        return        # the degenerate softmax's over/underflow bands: NaN values and NaN gradients are pinned in test_nc_gpu.py
    graph = mma_amd.NCGraph(rowptr, col, DEV, chunk=chunk)
    xg = x.to(DEV).requires_grad_(True)
    kinds = [Fn.KIND[O.AGGREGATORS[n][0]] for n in names]
    acts = [Fn.ACT_RAW if O.uses_raw_logits(n, act) else Fn.ACT_SIGMOID for n in names]
    P = xg @ torch.cat([Ws[n][:H] for n in names], 1).to(DEV)
    Q = xg @ torch.cat([Ws[n][H:] for n in names], 1).to(DEV)
    mg = Fn.nc_fused_aggregate(xg, P, Q, graph, kinds, acts, Fn.DropoutSpec(p, seed=seed))
    gg, = torch.autograd.grad((mg * cot.to(DEV)).sum(), [xg])
    check_close(mg.reshape(K * N, H), mo.reshape(K * N, H).numpy(), None, None, what="hyp m", signed_sum=True, truth=m64.reshape(K * N, H).numpy())
    check_close(gg, go.numpy(), None, None, what="hyp gx", signed_sum=True, truth=g64.numpy())
    with torch.no_grad():
        ms = Fn.nc_fused_aggregate(xg.detach(), P.detach(), Q.detach(), graph, kinds, acts, Fn.DropoutSpec(p, seed=seed), reduce_k=True)
    check_close(ms, mo.sum(0).numpy(), None, None, what="hyp msum", signed_sum=True, truth=m64.sum(0).numpy())


@st.composite
def gr_case(draw):
    N = draw(st.integers(1, 40))
    E = draw(st.integers(0, 300))
    T = draw(st.integers(1, 3))
    F = draw(st.sampled_from([1, 3, 4, 8, 20]))
    aggs = draw(st.lists(st.sampled_from(["sum", "mean", "min", "max", "var", "std"]), min_size=1, max_size=4, unique=True))
    scalers = draw(st.lists(st.sampled_from(["identity", "amplification", "attenuation", "linear", "inverse_linear"]), min_size=1, max_size=3))
    hub = draw(st.booleans())
    seed = draw(st.integers(0, 2 ** 31 - 1))
    return N, E, T, F, aggs, scalers, hub, seed


@settings(max_examples=N_EXAMPLES, deadline=None, suppress_health_check=list(HealthCheck), derandomize=DERANDOMIZE)
@given(gr_case())
def test_gr_aggregate_on_generated_targets(case):
    import mma_amd
    from oracle import gr_oracle as G
    N, E, T, F, aggs, scalers, hub, seed = case
    rng = np.random.default_rng(seed)
    index = rng.integers(0, N, E)
    if hub and E > 80:
        index[:75] = index[0]                                         # a segment above the 64-edge block-kernel limit
    vals = (rng.integers(-3, 4, (E, T, F)) * 0.5).astype(np.float32)    # exact ties everywhere
    conv = mma_amd.MMAConv(F * T, F * T, aggs, scalers, torch.tensor([0, 4, 9, 3, 1]), towers=T, divide_input=True).to(DEV)
    xi = torch.from_numpy(vals).requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((N, T, len(aggs) * len(scalers) * F)).astype(np.float32))
    want = G.aggregate(xi, torch.from_numpy(index), N, aggs, scalers, conv.avg_deg)
    xg = torch.from_numpy(vals).to(DEV).requires_grad_(True)
    got = conv.aggregate(xg, torch.from_numpy(index).to(DEV), N)
    check_close(got, want.detach().numpy(), None, None, what="hyp aggregate")
    if E:
        gw, = torch.autograd.grad((want * cot).sum(), [xi])
        gg, = torch.autograd.grad((got * cot.to(DEV)).sum(), [xg], retain_graph=True)
        check_close(gg, gw.numpy(), None, None, what="hyp aggregate grad", signed_sum=True)
    for red in ("min", "max"):                                        # arg = first extremal edge: the sequential loop decides
        if red in aggs and scalers[0] == "identity" and E:
            k = aggs.index(red)
            g1, = torch.autograd.grad(got[:, :, k * F:(k + 1) * F].sum(), [xg], retain_graph=True)
            _, arg = G.scatter_sequential(vals, index, N, red)
            onehot = np.zeros_like(vals)
            it = np.nditer(arg, flags=["multi_index"])
            for v in it:
                if int(v) >= 0:
                    onehot[(int(v),) + it.multi_index[1:]] = 1.0
            assert np.array_equal(g1.cpu().numpy(), onehot), red


@st.composite
def conv_case(draw):
    N = draw(st.integers(1, 30))
    E = draw(st.integers(0, 160))
    T = draw(st.integers(1, 5))
    F = draw(st.sampled_from([1, 3, 4, 6, 16, 19]))
    aggs = draw(st.lists(st.sampled_from(["sum", "mean", "min", "max"]), min_size=1, max_size=4, unique=True))
    scalers = draw(st.lists(st.sampled_from(["identity", "amplification", "attenuation", "linear", "inverse_linear"]), min_size=1, max_size=3,
                            unique=True))
    edge_dim = draw(st.sampled_from([None, 1, 3, 7]))
    divide_input = draw(st.booleans())
    p = draw(st.sampled_from([0.0, 0.5]))
    hub = draw(st.booleans())
    seed = draw(st.integers(0, 2 ** 31 - 1))
    n_types = draw(st.sampled_from([0, 0, 1, 4, 33]))            # > 0: the edge features are rows of an n_types-row table
    return N, E, T, F, aggs, scalers, edge_dim, divide_input, p, hub, seed, n_types


@settings(max_examples=max(N_EXAMPLES // 2, 10), deadline=None, suppress_health_check=list(HealthCheck), derandomize=DERANDOMIZE)
@given(conv_case())
def test_mmaconv_layer_on_generated_cases(case):
    """The whole MMAConv.forward (fused [U|V] / Z products, K3, post-NN, lin) and its backward against the oracle's literal
    message -> aggregate -> update on generated graphs: towers 1-5, widths that are not multiples of 4, with and without edge
    features, divide_input, hubs, isolated nodes, no edges; hash dropout replayed in the oracle from the same counter stream."""
    from mma_amd import functional as Fn
    from oracle import gr_oracle as G
    from oracle.dropout_rng import keep_mask
    from test_gr_gpu import conv_params, make_conv, to64
    N, E, T, F, aggs, scalers, edge_dim, divide_input, p, hub, seed, n_types = case
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, N, E), rng.integers(0, N, E)
    if hub and E:
        dst[: E // 2] = rng.integers(0, N)
    ei = np.stack([src, dst]).astype(np.int64)
    conv = make_conv(aggs, scalers, towers=T, F=F, edge_dim=edge_dim, divide_input=divide_input)
    x = rng.standard_normal((N, conv.in_channels)).astype(np.float32)
    ea = rng.standard_normal((E, edge_dim)).astype(np.float32) if edge_dim else None
    types = table = None
    if edge_dim and n_types:                                      # mma_amd.CategoricalEdges on the HIP side, table[types] for the oracle
        table = rng.standard_normal((n_types, edge_dim)).astype(np.float32)
        types = rng.integers(0, n_types, E)
        ea = table[types]
    cot = rng.standard_normal((N, conv.out_channels)).astype(np.float32)
    dseed = 0x5EED0000 + seed
    conv.drop_override = Fn.DropoutSpec(p, seed=dseed)
    keep = None
    if p > 0:
        Fw = conv.fused_width()
        keep = torch.from_numpy(keep_mask(dseed, int(p * 256), 1, E, T * Fw)[0].reshape(E, T, Fw)[:, :, :F].astype(np.float32))

    def oracle(dt):
        xo = torch.from_numpy(x).to(dt).requires_grad_(True)
        eo = torch.from_numpy(ea).to(dt).requires_grad_(True) if ea is not None else None
        prm = conv_params(conv) if dt == torch.float32 else to64(conv_params(conv))
        w = G.conv_forward(xo, torch.from_numpy(ei), eo, prm, aggs, scalers, conv.avg_deg, T, divide_input, keep, p)
        g = torch.autograd.grad((w * torch.from_numpy(cot).to(dt)).sum(), [xo] + ([eo] if eo is not None and E else []), allow_unused=True)
        return w.detach(), g
    want, gw = oracle(torch.float32)
    w64, g64 = oracle(torch.float64)
    xg = torch.from_numpy(x).to(DEV).requires_grad_(True)
    eg = torch.from_numpy(ea).to(DEV).requires_grad_(True) if ea is not None else None
    if table is not None:
        import mma_amd
        tg = torch.from_numpy(table).to(DEV).requires_grad_(True)
        got = conv(xg, torch.from_numpy(ei).to(DEV), mma_amd.CategoricalEdges(torch.from_numpy(types).to(DEV), tg))
        gg = torch.autograd.grad((got * torch.from_numpy(cot).to(DEV)).sum(), [xg] + ([tg] if E else []), allow_unused=True)
    else:
        got = conv(xg, torch.from_numpy(ei).to(DEV), eg)
        gg = torch.autograd.grad((got * torch.from_numpy(cot).to(DEV)).sum(), [xg] + ([eg] if eg is not None and E else []), allow_unused=True)
    check_close(got, want.numpy(), None, None, what="hyp conv out", signed_sum=True, truth=w64.numpy())
    check_close(gg[0], gw[0].numpy(), None, None, what="hyp conv gx", signed_sum=True, truth=g64[0].numpy())
    if len(gg) > 1 and gw[1] is not None and table is None:
        check_close(gg[1], gw[1].numpy(), None, None, what="hyp conv g(edge_attr)", signed_sum=True, truth=g64[1].numpy())
    if len(gg) > 1 and gw[1] is not None and table is not None:      # the table's gradient = the per-edge gradients summed by type
        gt = np.zeros_like(table); np.add.at(gt, types, gw[1].numpy())
        g64t = np.zeros(table.shape); np.add.at(g64t, types, g64[1].numpy())
        check_close(gg[1], gt, None, None, what="hyp conv g(edge table)", signed_sum=True, truth=g64t)


ALL_NC = ["sum", "sum2", "sum3", "sum4", "mean", "mean2", "mean3", "mean4", "max", "max2", "max3", "max4", "min", "min2", "min3", "min4"]


@st.composite
def layer_case(draw):
    N = draw(st.integers(1, 40))
    H = draw(st.sampled_from([1, 3, 4, 8, 16, 33]))
    C = draw(st.sampled_from([1, 2, 5, 16]))
    names = draw(st.lists(st.sampled_from(ALL_NC), min_size=1, max_size=11, unique=True))       # > 8 masks: several launch groups
    degs = draw(st.lists(st.one_of(st.just(0), st.integers(0, 5), st.integers(20, 70)), min_size=N, max_size=N))
    act = draw(st.sampled_from(["sigmoid", "new_sigmoid"]))
    chunk = draw(st.sampled_from([3, 16, 512]))
    strict = draw(st.booleans())
    scalers = None if strict else draw(st.lists(st.sampled_from(["identity", "amplification", "attenuation", "linear", "inverse_linear"]),
                                                min_size=1, max_size=5, unique=True))
    compound = False if strict else draw(st.booleans())
    n_adj = draw(st.integers(0, 120))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    return N, H, C, names, degs, act, chunk, strict, scalers, compound, n_adj, seed


@settings(max_examples=max(N_EXAMPLES // 2, 10), deadline=None, suppress_health_check=list(HealthCheck), derandomize=DERANDOMIZE)
@given(layer_case())
def test_mma_layer_on_generated_cases(case):
    """The drop-in layer mma_amd.MMA (layers.py:853-867: K masked aggregators -> cat with the scalers -> mm -> spmm with ANY sparse
    adjacency + bias) against the oracle's literal restatement, forward and every gradient: 1-11 masks of all four families (more
    than 8 = several launch groups), both activations, the reference's degenerate scalers or the true-degree ones (plain and
    compounding), an adjacency that is NOT the neighbour structure (random entries with values, empty rows)."""
    import mma_amd
    from mma_amd.layers import _MASK_NAMES
    from oracle import nc_oracle as O
    N, H, C, names, degs, act, chunk, strict, scalers, compound, n_adj, seed = case
    rng = np.random.default_rng(seed)
    rowptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    col = rng.integers(0, N, rowptr[-1]).astype(np.int64)
    ar, ac = np.sort(rng.integers(0, N, n_adj)), rng.integers(0, N, n_adj)
    av = rng.standard_normal(n_adj).astype(np.float32)
    x = torch.from_numpy(np.maximum(rng.standard_normal((N, H)), 0).astype(np.float32))
    cot = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32))
    Ws, weight, bias = O.init_like_reference(H, C, names, seed % 1000)
    kw = {} if strict else dict(true_degree_scalers=scalers, compound=compound)

    def oracle(dtype):
        xo, wo, bo = x.to(dtype).requires_grad_(True), weight.to(dtype).requires_grad_(True), bias.to(dtype).requires_grad_(True)
        Wo = {n: Ws[n].to(dtype).requires_grad_(True) for n in names}
        out = O.mma_forward(names, xo, Wo, wo, bo, rowptr, col, ar, ac, av, act, **kw)
        return out.detach(), torch.autograd.grad((out * cot.to(dtype)).sum(), [xo, wo, bo] + [Wo[n] for n in names], allow_unused=True)
    want, gw = oracle(torch.float32)
    w64, g64 = oracle(torch.float64)
    if not torch.isfinite(want).all() or not all(torch.isfinite(g).all() for g in gw if g is not None):
        return
    P = lambda t: torch.nn.Parameter(t.clone().to(DEV))
    mp = {n: P(Ws[n]) if n in names else P(torch.zeros(2, 1)) for n in _MASK_NAMES}
    w, b = P(weight), P(bias)
    add_all = [col[rowptr[i]:rowptr[i + 1]] for i in range(N)]
    mkw = {} if strict else dict(strict_reference=False, scalers=scalers, compound_scalers=compound)
    mod = mma_amd.MMA(add_all, act, 2, H, C, w, b, *[mp[n] for n in _MASK_NAMES], 0.0, names, DEV, chunk=chunk, **mkw)
    with torch.no_grad():
        for n in names:
            mp[n].copy_(Ws[n])
        w.copy_(weight); b.copy_(bias)
    adj = mma_amd.graph.SpmmGraph(ar, ac, av, N, N, DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = mod(xg, adj)
    gg = torch.autograd.grad((out * cot.to(DEV)).sum(), [xg, w, b] + [mp[n] for n in names], allow_unused=True)
    check_close(out, want.numpy(), None, None, what="hyp layer out", signed_sum=True, truth=w64.numpy())
    for name, a, r, t in zip(["gx", "gweight", "gbias"] + ["gmask/" + n for n in names], gg, gw, g64):
        if r is None:
            assert a is None or float(a.abs().max()) == 0.0, name
            continue
        check_close(a if a is not None else torch.zeros_like(r), r.numpy(), None, None, what="hyp layer " + name, signed_sum=True, truth=t.numpy())
