"""BASELINE configs[4] (C5: R-MAT scale 23, 8.4 M nodes / ~134 M directed edges, feat=256, K=8 aggregators + all S=5 true-degree scalers,
8 GPUs) as far as ONE GPU can take it: rank 0 of 8 with its REAL halo (1.05 M own rows + 2.42 M halo rows, 16.8 M edges, ~89 GiB) runs
forward + backward on the HIP path through mma_amd.sharded.ShardedMMA.  The four all-to-all-v of a step are replaced by a local stand-in
that hands the rank exactly what its peers would send when the loss touches only this rank's sample rows: the halo rows of x are the
true feature rows of those nodes, the gradient rows coming back are zero.  Asserted: bitwise repeatability of outputs and every gradient,
peak memory, the K-sum identity, and parity with oracle/nc_oracle.py on the sub-problems the sampled rows induce - the layer output
with the S=5 scalers, dL/dx of own AND halo rows, every parameter gradient (2-hop sample, as test_full_size_gpu.py::
test_c5_layer_with_all_true_degree_scalers_sample_parity), and the K aggregates of hubs / low-degree / random targets forward and backward
(as test_full_size_gpu.py::test_full_size_properties_and_sample_parity).  NOTHING here is a multi-GPU measurement (round-4 VERDICT item 4)."""
import numpy as np
import pytest
import torch

from golden_util import check_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RANK, WORLD = 0, 8


class _Done:
    def __init__(self, r):
        self.r = r

    def wait(self):
        return self.r


def test_c5_rank0_of_8_with_its_real_halo(monkeypatch):
    import bench
    import mma_amd
    from mma_amd import functional as Fn, sharded as S
    from mma_amd.dense import mm
    from mma_amd.scalers import TRUE_DEGREE_SCALERS
    from oracle import nc_oracle as O
    from oracle.dropout_rng import keep_mask
    from tools.synth import feature_rows, feature_rows_by_id

    P5 = bench.C5_PRESET
    H, C, names, act, p, seed = P5["hidden"], P5["nclass"], P5["aggregators"].split(","), "new_sigmoid", P5["dropout"], 0xC5C5C5C5C5
    K = len(names)
    rowptr, col = bench.rmat_graph(P5["scale"], P5["edges"], seed=42)
    N, E = len(rowptr) - 1, int(rowptr[-1])
    deg = np.diff(rowptr)
    assert N == 1 << 23 and E > 128_000_000 and deg.min() >= 1
    plan, e0 = S.HaloPlan.from_full_graph(rowptr, col, RANK, WORLD)
    lo, hi, n, n_halo = plan.lo, plan.hi, plan.n_own, plan.n_halo
    assert n > 1_000_000 and n_halo > 2 * n and int(plan.rowptr[-1]) > 16_000_000          # the halo is 2.3x the own rows
    n_send = int(plan.send_counts.sum())
    assert n_send != n_halo                                                               # the stand-in tells the directions apart by row count
    d_ = np.maximum(deg, 1).astype(np.float32)
    avg_d = {"log": float(np.log(d_ + 1).mean()), "lin": float(d_.mean())}
    del d_

    Ws, weight, bias = O.init_like_reference(H, C, names, 5)
    Pm = lambda t: torch.nn.Parameter(t.clone().to(DEV))
    masks = {a: Pm(Ws[a]) for a in names}
    torch.cuda.reset_peak_memory_stats()
    sh = S.ShardedMMA(plan, DEV, H, C, names, masks, Pm(weight), Pm(bias), p, edge_base=e0, strict_reference=False,
                      scalers=list(TRUE_DEGREE_SCALERS), compound_scalers=True, avg_d=avg_d)
    sh.drop_override = Fn.DropoutSpec(p, seed=seed)
    x_halo = torch.from_numpy(feature_rows_by_id(plan.halo_ids, H, 42)).to(DEV)
    tail_halo = torch.randn(n_halo, C, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    sent = {}

    def stand_in(send, send_counts, recv_counts, group=None, out=None):
        """What the peers would deliver: x rows of the halo nodes (true values), their tail rows (unknown to one rank: fixed random
        values - the sampled targets read own rows only), and ZERO gradient rows back (no peer's loss touches this rank's rows)."""
        rows, width = int(sum(recv_counts)), tuple(send.shape[1:])
        r = out if out is not None else torch.empty((rows,) + width, device=send.device)
        if rows == n_halo and width == (H,):
            r.copy_(x_halo)
        elif rows == n_halo and width == (C,):
            r.copy_(tail_halo)
        else:
            assert rows == n_send
            r.zero_()
            sent[width] = send.detach().clone()                  # what this rank sends its peers: gradients of ITS halo rows
        return _Done(r)
    monkeypatch.setattr(S, "all_to_all_rows_start", stand_in)

    # ---- 2-hop sample for the LAYER output: targets whose neighbours are all own rows (their aggregates are computed here) ----------
    e_hi = int(rowptr[hi])
    assert lo == 0 and e0 == 0                                         # rank 0: local target / edge ids are the global ones
    cg = col[:e_hi]
    all_own = np.minimum.reduceat((cg < hi).astype(np.int8), rowptr[:hi]) == 1
    maxnb = np.maximum.reduceat(deg[cg], rowptr[:hi])
    cand = np.nonzero((deg[:hi] <= 8) & all_own & (maxnb <= 100))[0]
    assert len(cand) > 1000
    rng = np.random.default_rng(8)
    tsel = np.sort(rng.choice(cand, 100, replace=False))
    mid = np.unique(np.concatenate([col[rowptr[t]:rowptr[t + 1]] for t in tsel]))          # own rows: aggregates needed
    seg = np.concatenate([np.arange(rowptr[i], rowptr[i + 1]) for i in mid])               # global edge positions, target-major
    nodes = np.unique(np.concatenate([tsel, mid, col[seg]]))                               # own AND halo nodes (global ids)
    assert (nodes >= hi).sum() > 100                                                       # the sub-problem really reads halo rows
    remap = np.full(N, -1, dtype=np.int64); remap[nodes] = np.arange(len(nodes))
    n_sub = len(nodes)
    d_sub = np.zeros(n_sub, dtype=np.int64); d_sub[remap[mid]] = deg[mid]
    rp = np.concatenate([[0], np.cumsum(d_sub)])
    cj = remap[col[seg]]
    keep = keep_mask(seed, int(p * 256), K, len(seg), H, edge_ids=seg)
    a_row = np.concatenate([np.full(deg[t], remap[t]) for t in tsel])
    a_col = remap[np.concatenate([col[rowptr[t]:rowptr[t + 1]] for t in tsel])]
    g = torch.Generator().manual_seed(2)
    cot_sel = torch.randn(len(tsel), C, generator=g)
    x_sub = torch.from_numpy(feature_rows_by_id(nodes, H, 42))

    def layer_oracle(dtype):
        xo = x_sub.to(dtype).requires_grad_(True)
        Wo = {a: Ws[a].to(dtype).requires_grad_(True) for a in names}
        wo, bo = weight.to(dtype).requires_grad_(True), bias.to(dtype).requires_grad_(True)
        out = O.mma_forward(names, xo, Wo, wo, bo, rp, cj, a_row, a_col, np.ones(len(a_row), np.float32), act, p,
                            {a: keep[k] for k, a in enumerate(names)}, true_degree_scalers=list(TRUE_DEGREE_SCALERS), compound=True, avg_d=avg_d)
        t_sub = torch.from_numpy(remap[tsel])
        gr = torch.autograd.grad((out[t_sub] * cot_sel.to(dtype)).sum(), [xo, wo, bo] + [Wo[a] for a in names])
        return out.detach()[t_sub], gr
    want, gw = layer_oracle(torch.float32)
    w64, g64 = layer_oracle(torch.float64)

    x = sh.feature_buffer()
    with torch.no_grad():
        x.copy_(torch.from_numpy(feature_rows(lo, hi, H, 42)).to(DEV))
    cot = torch.zeros(n, C, device=DEV)
    cot[torch.from_numpy(tsel - lo).to(DEV)] = cot_sel.to(DEV)

    def step():
        x.grad = None
        for prm in sh.owned:
            prm.grad = None
        out = sh(x)
        out.backward(cot)
        return [out.detach().clone(), x.grad.clone()] + [prm.grad.clone() for prm in sh.owned] + [sent[(H,)], sent[(C,)]]
    r1 = step()
    r2 = step()
    for a, b in zip(r1, r2):
        assert torch.equal(a, b), "the sharded step is not bitwise repeatable"
    out, gx, gweight, gbias = r1[0], r1[1], r1[2], r1[3]
    gmask = dict(zip(names, r1[4:4 + K]))
    gx_halo = r1[4 + K]                                                # (n_halo, H): dL/dx of the halo rows, on its way to their owners
    assert gx_halo.shape == (n_halo, H)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print("C5 rank %d of %d: own %d halo %d edges %d, peak %.1f GiB" % (RANK, WORLD, n, n_halo, int(plan.rowptr[-1]), peak))
    assert peak < 200.0

    check_close(out[torch.from_numpy(tsel - lo).to(DEV)], want.numpy(), None, None, what="C5 rank0 layer out (S=5 true-degree scalers)",
                signed_sum=True, truth=w64.numpy())
    own_nodes, halo_nodes = nodes[nodes < hi], nodes[nodes >= hi]
    check_close(gx[torch.from_numpy(own_nodes - lo).to(DEV)], gw[0][remap[own_nodes]].numpy(), None, None, what="C5 rank0 layer gx (own rows)",
                signed_sum=True, truth=g64[0][remap[own_nodes]].numpy())
    hpos = torch.from_numpy(np.searchsorted(plan.halo_ids, halo_nodes)).to(DEV)
    check_close(gx_halo[hpos], gw[0][remap[halo_nodes]].numpy(), None, None, what="C5 rank0 layer gx (halo rows, sent to their owners)",
                signed_sum=True, truth=g64[0][remap[halo_nodes]].numpy())
    untouched = torch.ones(n, dtype=torch.bool); untouched[torch.from_numpy(own_nodes - lo)] = False
    assert gx.cpu()[untouched].abs().max().item() == 0.0
    check_close(gweight, gw[1].numpy(), None, None, what="C5 rank0 gweight", signed_sum=True, truth=g64[1].numpy())
    check_close(gbias, gw[2].numpy(), None, None, what="C5 rank0 gbias", signed_sum=True, truth=g64[2].numpy())
    for k, a in enumerate(names):
        check_close(gmask[a], gw[3 + k].numpy(), None, None, what="C5 rank0 gmask/" + a, signed_sum=True, truth=g64[3 + k].numpy())
    del r1, r2, out, gx, gx_halo, cot, want, gw, w64, g64, keep
    torch.cuda.empty_cache()

    # ---- the K aggregates on the shard's own graph plan: K-sum identity, hubs / low-degree / random targets forward and backward -------
    graph = sh.graph
    assert graph.n_slots > 100                                        # hubs of the shard are split into chunks
    x_src = sh._x_src.detach().clone().requires_grad_(True)           # [own | halo], as the step left it
    Wg = {a: masks[a].detach().clone().requires_grad_(True) for a in names}
    kinds = [Fn.KIND[O.AGGREGATORS[a][0]] for a in names]
    acts = [Fn.ACT_RAW if O.uses_raw_logits(a, act) else Fn.ACT_SIGMOID for a in names]
    drop = Fn.DropoutSpec(p, seed=seed)

    def run(reduce_k):
        Pt = mm(x_src[:n], torch.cat([Wg[a][:H] for a in names], 1))
        Qt = mm(x_src, torch.cat([Wg[a][H:] for a in names], 1))
        return Fn.nc_fused_aggregate(x_src, Pt, Qt, graph, kinds, acts, drop, reduce_k=reduce_k)
    dloc = deg[lo:hi]
    order = np.argsort(-dloc)
    sample = np.unique(np.concatenate([order[:4], order[-32:], rng.choice(n, 300, replace=False)]))       # local target ids
    sd = torch.from_numpy(sample).to(DEV)
    cot_rows = torch.randn(K, len(sample), H, generator=g)
    cot_m = torch.zeros(K, n, H, device=DEV)
    cot_m[:, sd] = cot_rows.to(DEV)
    m = run(False)
    grads = torch.autograd.grad((m * cot_m).sum(), [x_src] + [Wg[a] for a in names])
    del cot_m
    with torch.no_grad():
        msum = run(True)
        ref = m.detach().sum(0)
        assert (msum - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()      # same terms, other association of the sum over k
        del msum, ref
    segl = np.concatenate([np.arange(plan.rowptr[i], plan.rowptr[i + 1]) for i in sample])      # local edge positions (+ e0 = global)
    src_l = plan.col[segl]                                                                      # local source ids: own, then halo
    nodes_l = np.unique(np.concatenate([sample, src_l]))
    remap_l = np.full(plan.n_src, -1, dtype=np.int64); remap_l[nodes_l] = np.arange(len(nodes_l))
    d_sub = np.zeros(len(nodes_l), dtype=np.int64); d_sub[remap_l[sample]] = dloc[sample]
    rp = np.concatenate([[0], np.cumsum(d_sub)])
    cj = remap_l[src_l]
    keep = keep_mask(seed, int(p * 256), K, len(segl), H, edge_ids=segl + e0)
    tgt = torch.from_numpy(remap_l[sample])
    cot_sub = torch.zeros(K, len(nodes_l), H); cot_sub[:, tgt] = cot_rows
    x_cpu = x_src.detach()[torch.from_numpy(nodes_l).to(DEV)].cpu()

    def agg_oracle(dtype):
        xo = x_cpu.to(dtype)
        ms, gxo, gws = [], torch.zeros_like(xo), []
        for k, a in enumerate(names):                                  # one mask at a time: bounds the float64 tape
            xk, wk = xo.clone().requires_grad_(True), Ws[a].to(dtype).requires_grad_(True)
            mk = O.aggregate(a, xk, wk, rp, cj, act, p, keep[k])
            ga, gb = torch.autograd.grad((mk * cot_sub[k].to(dtype)).sum(), [xk, wk])
            ms.append(mk.detach()); gxo += ga; gws.append(gb)
            del mk
        return torch.stack(ms), [gxo] + gws
    mo, go = agg_oracle(torch.float32)
    m64, g64 = agg_oracle(torch.float64)
    for k, a in enumerate(names):
        check_close(m[k][sd], mo[k][tgt].numpy(), None, None, what="C5 rank0 sample m/" + a, signed_sum=O.uses_raw_logits(a, act),
                    truth=m64[k][tgt].numpy())
    gxs = grads[0].cpu()
    check_close(gxs[torch.from_numpy(nodes_l)], go[0].numpy(), None, None, what="C5 rank0 sample gx", signed_sum=True, truth=g64[0].numpy())
    rest = torch.ones(plan.n_src, dtype=torch.bool); rest[torch.from_numpy(nodes_l)] = False
    assert gxs[rest].abs().max().item() == 0.0
    for a, ga, gb, gt in zip(names, grads[1:], go[1:], g64[1:]):
        check_close(ga, gb.numpy(), None, None, what="C5 rank0 sample gW/" + a, signed_sum=True, truth=gt.numpy())
