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


def test_c4_full_size_properties_and_sample_parity():
    import bench
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import nc_oracle as O
    from oracle.dropout_rng import keep_mask
    H, names, act, p, seed = 128, ["sum", "mean", "max", "min"], "new_sigmoid", 0.5, 0xC4C4C4C4C4
    K = len(names)
    rowptr, col = bench.rmat_graph(20, 5_000_000, seed=42)
    N, E = len(rowptr) - 1, int(rowptr[-1])
    deg = np.diff(rowptr)
    assert N == 1 << 20 and E > 10_000_000 and deg.min() >= 1 and deg.max() > 10_000
    graph = mma_amd.NCGraph(rowptr, col, DEV)
    assert graph.n_slots > 1000          # hubs are really split
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(N, H, generator=g))
    Ws = {n: ((torch.rand(2 * H, H, generator=g) * 2 - 1) / np.sqrt(H)) for n in names}
    kinds = [Fn.KIND[O.AGGREGATORS[n][0]] for n in names]
    acts = [Fn.ACT_RAW if O.uses_raw_logits(n, act) else Fn.ACT_SIGMOID for n in names]

    rng = np.random.default_rng(3)
    order = np.argsort(-deg)
    sample = np.unique(np.concatenate([order[:64], order[-64:], rng.choice(N, 2000, replace=False)]))
    cot = torch.zeros(K, N, H)
    cot[:, sample] = torch.randn(K, len(sample), H, generator=g)

    xg = x.to(DEV).requires_grad_(True)
    Wg = {n: Ws[n].to(DEV).requires_grad_(True) for n in names}

    def run(reduce_k):
        P = xg @ torch.cat([Wg[n][:H] for n in names], 1)
        Q = xg @ torch.cat([Wg[n][H:] for n in names], 1)
        return Fn.nc_fused_aggregate(xg, P, Q, graph, kinds, acts, Fn.DropoutSpec(p, seed=seed), reduce_k=reduce_k)

    m = run(False)
    grads = torch.autograd.grad((m * cot.to(DEV)).sum(), [xg] + [Wg[n] for n in names])
    m2 = run(False)
    grads2 = torch.autograd.grad((m2 * cot.to(DEV)).sum(), [xg] + [Wg[n] for n in names])
    assert torch.equal(m, m2) and torch.equal(grads[0], grads2[0]), "not bitwise deterministic"
    msum = run(True).detach()
    ref = m.detach().sum(0)
    assert (msum - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()

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
    xo = x[nodes].clone().requires_grad_(True)
    Wo = {n: Ws[n].clone().requires_grad_(True) for n in names}
    mo = torch.stack([O.aggregate(n, xo, Wo[n], rp, cj, act, p, keep[k]) for k, n in enumerate(names)])
    tgt = torch.from_numpy(remap[sample])
    cot_sub = torch.zeros(K, n_sub, H); cot_sub[:, tgt] = cot[:, sample]
    go = torch.autograd.grad((mo * cot_sub).sum(), [xo] + [Wo[n] for n in names])
    # forward rows
    for k, n in enumerate(names):
        # raw-logit masks (max/min under "new_sigmoid") sum ~26 k SIGNED terms on the hubs: scale-relative bar there
        check_close(m[k][torch.from_numpy(sample).to(DEV)], mo[k][tgt].detach().numpy(), None, None, what="C4 sample m/" + n,
                    signed_sum=O.uses_raw_logits(n, act))
    # backward: gradient w.r.t. x on every involved node (0 elsewhere), and w.r.t. the mask weights
    gx = grads[0].cpu()
    check_close(gx[torch.from_numpy(nodes)], go[0].numpy(), None, None, what="C4 sample gx", signed_sum=True)
    mask = torch.ones(N, dtype=torch.bool); mask[torch.from_numpy(nodes)] = False
    assert gx[mask].abs().max().item() == 0.0
    for n, a, b in zip(names, grads[1:], go[1:]):
        check_close(a, b.numpy(), None, None, what="C4 sample gW/" + n, signed_sum=True)
