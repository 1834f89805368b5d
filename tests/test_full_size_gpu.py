"""Full-size checks at BASELINE configs[3] (C4: 2^20 nodes / ~10.9 M edges, H=128, K=4, p=0.5) through size-independent
properties: bitwise determinism (no atomics anywhere), K-sum output == sum of the per-mask outputs, and parity of a
sample of target rows - the 64 largest hubs (each split into hub chunks), 64 lowest-degree nodes and ~2000 random
ones - against the CPU oracle on the sub-problem those rows induce, forward AND backward (cotangent supported on the
sample only, so the oracle's gradient on the sub-problem is the full gradient)."""
import numpy as np
import pytest
import torch

from golden_util import check_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


FULL_SIZE = {
    # BASELINE configs[3]: 2^20 nodes / ~10.9 M directed edges, H=128, K=4
    "C4": dict(edges=5_000_000, H=128, names=["sum", "mean", "max", "min"], n_hubs=64, n_low=64, n_rand=2000, min_edges=10_000_000),
    # BASELINE configs[4] at its PER-GPU shard shape (8 M nodes / 128 M edges over 8 GPUs): 2^20 nodes / ~16.4 M directed edges,
    # H=256, K=8 [sum,mean,max,min,sum2,mean2,max2,min2] (SURVEY 8d "C5"); a smaller row sample keeps the float64 oracle pass
    # (E_sample x 2H x 8 B per mask) within a few GB of host memory
    "C5shard": dict(edges=8_000_000, H=256, names=["sum", "mean", "max", "min", "sum2", "mean2", "max2", "min2"], n_hubs=8,
                    n_low=32, n_rand=500, min_edges=16_000_000),
}


@pytest.mark.parametrize("cfg", ["C4", "C5shard"])
def test_full_size_properties_and_sample_parity(cfg):
    import bench
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import nc_oracle as O
    from oracle.dropout_rng import keep_mask
    c = FULL_SIZE[cfg]
    H, names, act, p, seed = c["H"], c["names"], "new_sigmoid", 0.5, 0xC4C4C4C4C4
    K = len(names)
    rowptr, col = bench.rmat_graph(20, c["edges"], seed=42)
    N, E = len(rowptr) - 1, int(rowptr[-1])
    deg = np.diff(rowptr)
    assert N == 1 << 20 and E > c["min_edges"] and deg.min() >= 1 and deg.max() > 10_000
    graph = mma_amd.NCGraph(rowptr, col, DEV)
    assert graph.n_slots > 1000          # hubs are really split
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(N, H, generator=g))
    Ws = {n: ((torch.rand(2 * H, H, generator=g) * 2 - 1) / np.sqrt(H)) for n in names}
    kinds = [Fn.KIND[O.AGGREGATORS[n][0]] for n in names]
    acts = [Fn.ACT_RAW if O.uses_raw_logits(n, act) else Fn.ACT_SIGMOID for n in names]

    rng = np.random.default_rng(3)
    order = np.argsort(-deg)
    sample = np.unique(np.concatenate([order[:c["n_hubs"]], order[-c["n_low"]:], rng.choice(N, c["n_rand"], replace=False)]))
    cot = torch.zeros(K, N, H)
    cot[:, sample] = torch.randn(K, len(sample), H, generator=g)

    xg = x.to(DEV).requires_grad_(True)
    Wg = {n: Ws[n].to(DEV).requires_grad_(True) for n in names}

    from mma_amd.dense import mm          # the layers' own product (layers.py::_aggregate): split-reduction weight gradient, not torch's

    def run(reduce_k):
        P = mm(xg, torch.cat([Wg[n][:H] for n in names], 1))
        Q = mm(xg, torch.cat([Wg[n][H:] for n in names], 1))
        return Fn.nc_fused_aggregate(xg, P, Q, graph, kinds, acts, Fn.DropoutSpec(p, seed=seed), reduce_k=reduce_k)

    m = run(False)
    grads = torch.autograd.grad((m * cot.to(DEV)).sum(), [xg] + [Wg[n] for n in names])
    m2 = run(False)
    grads2 = torch.autograd.grad((m2 * cot.to(DEV)).sum(), [xg] + [Wg[n] for n in names])
    assert torch.equal(m, m2) and torch.equal(grads[0], grads2[0]), "not bitwise deterministic"
    msum = run(True).detach()
    ref = m.detach().sum(0)
    assert (msum - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()      # same terms, other association of the sum over k

    # ---- oracle on the sub-problem induced by the sampled targets -------------------------------------
    seg = np.concatenate([np.arange(rowptr[i], rowptr[i + 1]) for i in sample])          # global edge ids, target-major
    sub_rowptr = np.concatenate([[0], np.cumsum(deg[sample])])
    src = col[seg]
    nodes = np.unique(np.concatenate([sample, src]))
    remap = np.full(N, -1, dtype=np.int64); remap[nodes] = np.arange(len(nodes))
    # sub-graph node order: every involved node; the sampled targets keep their edges, the others have degree 0
    n_sub = len(nodes)
    rp = np.zeros(n_sub + 1, dtype=np.int64)
    d_sub = np.zeros(n_sub, dtype=np.int64); d_sub[remap[sample]] = deg[sample]
    rp[1:] = np.cumsum(d_sub)
    # edges must be listed in the sub-graph's target order (ascending remapped id == ascending global id)
    cj = remap[src]
    keep = keep_mask(seed, int(p * 256), K, len(seg), H, edge_ids=seg)
    tgt = torch.from_numpy(remap[sample])
    cot_sub = torch.zeros(K, n_sub, H); cot_sub[:, tgt] = cot[:, sample]

    def oracle(dtype):
        """One aggregator at a time (the masks only meet in the sum over k of dL/dx): bounds the host memory of the
        float64 pass to one mask's autograd tape."""
        xo = x[nodes].to(dtype)
        ms, gx, gws = [], torch.zeros_like(xo), []
        for k, n in enumerate(names):
            xk, wk = xo.clone().requires_grad_(True), Ws[n].to(dtype).requires_grad_(True)
            mk = O.aggregate(n, xk, wk, rp, cj, act, p, keep[k])
            a, b = torch.autograd.grad((mk * cot_sub[k].to(dtype)).sum(), [xk, wk])
            ms.append(mk.detach()); gx += a; gws.append(b)
            del mk
        return torch.stack(ms), [gx] + gws
    mo, go = oracle(torch.float32)
    m64, g64 = oracle(torch.float64)        # exact value of the same formulas: sets the slack of the long signed sums
    # forward rows
    for k, n in enumerate(names):
        # raw-logit masks (max/min under "new_sigmoid") sum ~26 k SIGNED terms on the hubs: scale-relative bar there
        check_close(m[k][torch.from_numpy(sample).to(DEV)], mo[k][tgt].numpy(), None, None, what=cfg + " sample m/" + n,
                    signed_sum=O.uses_raw_logits(n, act), truth=m64[k][tgt].numpy())
    # backward: gradient w.r.t. x on every involved node (0 elsewhere), and w.r.t. the mask weights
    gx = grads[0].cpu()
    check_close(gx[torch.from_numpy(nodes)], go[0].numpy(), None, None, what=cfg + " sample gx", signed_sum=True, truth=g64[0].numpy())
    mask = torch.ones(N, dtype=torch.bool); mask[torch.from_numpy(nodes)] = False
    assert gx[mask].abs().max().item() == 0.0
    for n, a, b, t in zip(names, grads[1:], go[1:], g64[1:]):
        check_close(a, b.numpy(), None, None, what=cfg + " sample gW/" + n, signed_sum=True, truth=t.numpy())


def test_c2l_full_batch_properties_and_sample_parity():
    """The GR layer at full scale (the whole ZINC-subset split as ONE batch: 10 000 molecules, ~2e5 nodes / ~4.3e5 edges,
    MMAConv 75->75, towers=5, edge_dim=50, [min,max] x [identity,amplification,linear], always-on dropout): bitwise
    repeatable, and - molecules being independent components laid out one after the other - the rows of the first 40
    molecules must equal the CPU oracle run on those 40 molecules alone (same edge positions => same dropout bits),
    forward and backward."""
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import gr_oracle as G
    from oracle.dropout_rng import keep_mask
    from test_gr_gpu import conv_params, molecule_batch
    rng = np.random.default_rng(5)
    ei, N, sizes = molecule_batch(rng, 10000, return_sizes=True)
    E = ei.shape[1]
    T, F, p, seed = 5, 75, 0.5, 0x2121ABCD77
    hist = np.bincount(np.bincount(ei[1], minlength=N), minlength=5)
    torch.manual_seed(3)
    conv = mma_amd.MMAConv(75, 75, ["min", "max"], ["identity", "amplification", "linear"], torch.tensor(hist), edge_dim=50,
                           towers=T).to(DEV)
    conv.drop_override = Fn.DropoutSpec(p, seed=seed)
    x = rng.standard_normal((N, 75)).astype(np.float32)
    ea = rng.standard_normal((E, 50)).astype(np.float32)
    cot = rng.standard_normal((N, 75)).astype(np.float32)
    xg = torch.from_numpy(x).to(DEV).requires_grad_(True)
    eig, eag, cg = torch.from_numpy(ei).to(DEV), torch.from_numpy(ea).to(DEV), torch.from_numpy(cot).to(DEV)
    out = conv(xg, eig, eag)
    gx, = torch.autograd.grad((out * cg).sum(), [xg])
    out2 = conv(xg, eig, eag)
    gx2, = torch.autograd.grad((out2 * cg).sum(), [xg])
    assert torch.equal(out, out2) and torch.equal(gx, gx2)                 # no atomics, fixed reduction orders
    assert torch.isfinite(out).all() and torch.isfinite(gx).all()
    # the first 40 molecules as their own batch on the CPU oracle
    n_s = int(sizes[:40].sum())
    e_s = int((ei[1] < n_s).sum())
    assert (ei[:, :e_s] < n_s).all() and (ei[:, e_s:] >= n_s).all()       # components are laid out one after the other
    Fw = conv.fused_width()
    keep = torch.from_numpy(keep_mask(seed, int(p * 256), 1, e_s, T * Fw)[0].reshape(e_s, T, Fw)[:, :, :F].astype(np.float32))
    xo = torch.from_numpy(x[:n_s]).requires_grad_(True)
    want = G.conv_forward(xo, torch.from_numpy(ei[:, :e_s]), torch.from_numpy(ea[:e_s]), conv_params(conv), conv.aggregators,
                          conv.scalers, conv.avg_deg, T, False, keep, p)
    gw, = torch.autograd.grad((want * torch.from_numpy(cot[:n_s])).sum(), [xo])
    from test_gr_gpu import to64
    x64 = torch.from_numpy(x[:n_s]).double().requires_grad_(True)
    w64 = G.conv_forward(x64, torch.from_numpy(ei[:, :e_s]), torch.from_numpy(ea[:e_s]).double(), to64(conv_params(conv)),
                         conv.aggregators, conv.scalers, conv.avg_deg, T, False, keep, p)
    g64, = torch.autograd.grad((w64 * torch.from_numpy(cot[:n_s]).double()).sum(), [x64])
    check_close(out[:n_s], want.detach().numpy(), None, None, what="C2L sample out", signed_sum=True, truth=w64.detach().numpy())
    check_close(gx[:n_s], gw.numpy(), None, None, what="C2L sample gx", signed_sum=True, truth=g64.numpy())
    # the tall Linear layers ([U|V], Z) run zero-padded on the bf16x3 kernels at this size: the library-GEMM path must give the
    # same layer - output, dL/dx and EVERY parameter gradient (whole-batch sums, which the 40-molecule oracle cannot provide)
    from mma_amd import dense
    prm = [q for q in conv.parameters() if q.requires_grad]
    assert dense.X3_LINEAR and dense.linear_x3_ok(xg, torch.empty(760, 75))
    g_x3 = torch.autograd.grad((conv(xg, eig, eag) * cg).sum(), prm, allow_unused=True)
    dense.X3_LINEAR = False
    try:
        out_lib = conv(xg, eig, eag)
        g_lib = torch.autograd.grad((out_lib * cg).sum(), [xg] + prm, allow_unused=True)
    finally:
        dense.X3_LINEAR = True
    check_close(out, out_lib.detach().cpu().numpy(), None, None, what="C2L x3 vs library out", signed_sum=True)
    # backward: the two paths differ in the last bits of U, V, Z, so among the 4.7e8 (node, column, min|max) selections a few
    # dozen near-ties pick another edge and re-route that column's gradient (bit-exact arg parity is asserted where the inputs
    # are identical: test_gr_gpu.py).  Rows touched by such a flip are counted, everything else meets the usual bar.
    gl = g_lib[0]
    flipped = ((gx - gl).abs() > 1e-5 + 1e-5 * gl.abs()).any(1)
    assert int(flipped.sum()) <= 200, int(flipped.sum())
    keep_rows = (~flipped).nonzero().flatten()
    check_close(gx[keep_rows], gl[keep_rows].cpu().numpy(), None, None, what="C2L x3 vs library gx (rows without an arg flip)", signed_sum=True)
    n_cmp = 0
    for a, b in zip(g_x3, g_lib[1:]):
        assert (a is None) == (b is None)
        if a is not None:       # whole-batch sums over 2e5 nodes: every flip moves one term of size |x|*|g| (up to a few 1e-2) between two edges
            assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item() + 1e-3, ((a - b).abs().max().item(), b.abs().max().item())
            n_cmp += 1
    assert n_cmp >= 8


def test_c5_layer_with_all_true_degree_scalers_sample_parity():
    """BASELINE configs[4] "K=8 aggregators + all scalers" at the per-GPU shard shape, through the drop-in layer:
    mma_amd.MMA(strict_reference=False, scalers=<the five of mma_conv.py:181-196>, compound_scalers=True), H=256, K=8,
    hash dropout p=0.5, forward + dL/dx.  The layer output of a target needs the aggregates of all its neighbours, i.e. a
    2-hop neighbourhood: the sample is ~100 low-degree targets whose neighbours have <= 100 in-edges, and the oracle
    (literal cat of the S=5 scaled blocks, mm with the 5x stacked weight, spmm) runs on the sub-problem they induce."""
    import bench
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.scalers import TRUE_DEGREE_SCALERS
    from oracle import nc_oracle as O
    from oracle.dropout_rng import keep_mask
    H, C, act, p, seed = 256, 16, "new_sigmoid", 0.5, 0xC5C5C5C5
    names = FULL_SIZE["C5shard"]["names"]
    K = len(names)
    rowptr, col = bench.rmat_graph(20, FULL_SIZE["C5shard"]["edges"], seed=42)
    N = len(rowptr) - 1
    deg = np.diff(rowptr)
    rng = np.random.default_rng(8)
    maxnb = np.maximum.reduceat(deg[col], rowptr[:-1])                  # largest in-degree among each node's neighbours
    cand = np.nonzero((deg <= 8) & (maxnb <= 100))[0]
    tsel = np.sort(rng.choice(cand, 100, replace=False))                # targets whose OUTPUT is compared
    mid = np.unique(np.concatenate([col[rowptr[t]:rowptr[t + 1]] for t in tsel]))      # their neighbours: aggregates needed
    seg = np.concatenate([np.arange(rowptr[i], rowptr[i + 1]) for i in mid])
    nodes = np.unique(np.concatenate([tsel, mid, col[seg]]))
    remap = np.full(N, -1, dtype=np.int64); remap[nodes] = np.arange(len(nodes))
    n_sub = len(nodes)
    d_sub = np.zeros(n_sub, dtype=np.int64); d_sub[remap[mid]] = deg[mid]
    rp = np.concatenate([[0], np.cumsum(d_sub)])
    cj = remap[col[seg]]
    avg_d = {"log": float(np.log(deg.astype(np.float32) + 1).mean()), "lin": float(deg.astype(np.float32).mean())}

    g = torch.Generator().manual_seed(2)
    x = torch.relu(torch.randn(N, H, generator=g))
    Ws, weight, bias = O.init_like_reference(H, C, names, 5)
    cot = torch.zeros(N, C); cot[tsel] = torch.randn(len(tsel), C, generator=g)
    keep = keep_mask(seed, int(p * 256), K, len(seg), H, edge_ids=seg)
    # spmm rows of the sampled targets only (their in-edges, sources remapped)
    a_row = np.concatenate([np.full(deg[t], remap[t]) for t in tsel])
    a_col = remap[np.concatenate([col[rowptr[t]:rowptr[t + 1]] for t in tsel])]

    def oracle(dtype):
        xo = x[nodes].to(dtype).requires_grad_(True)
        Wo = {n: Ws[n].to(dtype) for n in names}
        # true degrees of the sub-problem rows: only `mid` rows matter (their factor multiplies their aggregates); rows with no
        # edges in the sub-problem get degree 1 from the clamp and are never read by the sampled targets
        out = O.mma_forward(names, xo, Wo, weight.to(dtype), bias.to(dtype), rp, cj, a_row, a_col, np.ones(len(a_row), np.float32),
                            act, p, {n: keep[k] for k, n in enumerate(names)}, true_degree_scalers=list(TRUE_DEGREE_SCALERS),
                            compound=True, avg_d=avg_d)
        gx, = torch.autograd.grad((out * cot[nodes].to(dtype)).sum(), [xo])
        return out.detach(), gx
    want, gw = oracle(torch.float32)
    w64, g64 = oracle(torch.float64)

    graph = mma_amd.NCGraph(rowptr, col, DEV)
    layer = bench.make_layer(mma_amd, graph, H, C, names, p, DEV, strict_reference=False, scalers=list(TRUE_DEGREE_SCALERS),
                             compound_scalers=True, avg_d=avg_d)
    with torch.no_grad():
        for n in names:
            getattr(layer, "mask_" + n).copy_(Ws[n])
        layer.weight.copy_(weight); layer.bias.copy_(bias)
    layer.drop_override = Fn.DropoutSpec(p, seed=seed)
    dst = np.repeat(np.arange(N, dtype=np.int64), deg)
    adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = layer(xg, adj)
    gx, = torch.autograd.grad((out * cot.to(DEV)).sum(), [xg])
    t_sub = remap[tsel]
    check_close(out[torch.from_numpy(tsel).to(DEV)], want[t_sub].numpy(), None, None, what="C5 layer out (S=5 true-degree scalers)",
                signed_sum=True, truth=w64[t_sub].numpy())
    check_close(gx[torch.from_numpy(nodes).to(DEV)], gw.numpy(), None, None, what="C5 layer gx", signed_sum=True, truth=g64.numpy())
    mask = torch.ones(N, dtype=torch.bool); mask[torch.from_numpy(nodes)] = False
    assert gx.cpu()[mask].abs().max().item() == 0.0
