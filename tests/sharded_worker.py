"""Worker for the multi-process tests (launched by torch.distributed.run with WORLD_SIZE ranks).

  mode cpu : gloo, CPU only.  Checks the HaloPlan (index exchange), the all-to-all plumbing and the sharding algebra
             by emulating the sharded layer with the CPU oracle as the compute (test infrastructure) and comparing
             with the unsharded oracle, forward and backward.
  mode gpu : gloo rendezvous, every rank computes on cuda:0 with the HIP kernels (halo buffers staged through the
             host).  Checks mma_amd.sharded.ShardedMMA against the single-GPU mma_amd.MMA, forward and backward.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def graph(seed, N, avg):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg, N) + 1
    deg[N // 3] = 5 * N // 4            # a hub that needs rows from every rank
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(N, size=d, replace=d > N)) for d in deg]).astype(np.int64)
    return rowptr, col


def directed_graph(N, world):
    """All sources lie in the first half of the first rank's range: that rank reads no remote row (no halo of its own) while
    every other rank reads rows of it - the case in which a rank must still join the reverse exchange."""
    rng = np.random.default_rng(11)
    deg = rng.integers(1, 6, N)
    blk = N // world
    rows = []
    for i in range(N):
        rows.append(np.sort(rng.integers(0, blk // 2, deg[i])))      # every source sits well inside the first rank's range
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    return rowptr, np.concatenate(rows).astype(np.int64)


def hub_first_graph(N):
    """Node 0 holds ~70 % of all edges: an edge-balanced 3-way partition then has an EMPTY middle shard."""
    rng = np.random.default_rng(12)
    deg = rng.integers(1, 4, N)
    deg[0] = 5 * int(deg[1:].sum()) // 2
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.integers(0, N, d)) for d in deg]).astype(np.int64)
    return rowptr, col


def close(a, b, what, signed_sum=True):
    """Same bars as tests/golden_util.check_close: strict 1e-5 + 1e-5|ref| element-wise; long signed sums get
    atol = max(1e-5, 1e-6 * max|ref|)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    atol = max(1e-5, 1e-6 * b.abs().max().item()) if (signed_sum and b.numel()) else 1e-5
    bad = err > atol + 1e-5 * b.abs()
    assert not bad.any(), "%s: max err %.3g (atol %.3g), %d/%d outside" % (what, err.max().item(), atol, int(bad.sum()), err.numel())


def messy_graph(seed, N):
    """Duplicate edges, self-loops, isolated nodes at both ends, one node read by everybody."""
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 7, N)
    deg[:2] = 0
    deg[-3:] = 0
    rows = []
    for i in range(N):
        c = rng.integers(0, N, deg[i])
        if deg[i] >= 2:
            c[0] = i                        # self-loop
            c[1] = c[-1]                    # duplicate
        if deg[i] >= 3:
            c[2] = N - 1                    # an isolated node that every rank reads
        rows.append(np.sort(c))
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    return rowptr, (np.concatenate(rows) if rows else np.zeros(0)).astype(np.int64)


def run_cpu(rank, world):
    cases = [("hub", 157, graph(3, 157, 4)), ("directed", 96, directed_graph(96, world)), ("hub-first", 64, hub_first_graph(64)),
             ("messy", 131, messy_graph(5, 131)), ("fewer nodes than ranks", 3, (np.array([0, 2, 2, 3]), np.array([1, 2, 0]))),
             ("no edges", 9, (np.zeros(10, np.int64), np.zeros(0, np.int64)))]
    for tag, N, (rowptr, col) in cases:
        check_cpu(rank, world, tag, N, rowptr.astype(np.int64), col.astype(np.int64))
    # BASELINE configs[4] in miniature: K = 8 masks and the S = 5 compounding true-degree scalers (mma_conv.py:181-196), whose row
    # factor needs the GLOBAL degree means on every rank
    check_cpu(rank, world, "c5 miniature (K=8, S=5)", 211, *[a.astype(np.int64) for a in graph(9, 211, 5)],
              names=["sum", "mean", "max", "min", "sum2", "mean2", "max2", "min2"],
              scalers=["identity", "amplification", "attenuation", "linear", "inverse_linear"])


def true_degree_factor(deg, scalers, avg_log, avg_lin):
    """Row factor sum_s prod_{q<=s} f_q(d) of the compounding scalers (mma_conv.py:181-196), numpy restatement for this test."""
    d = np.maximum(deg, 1).astype(np.float64)
    lg = np.log(d + 1)
    f = {"identity": np.ones_like(d), "amplification": lg / avg_log, "attenuation": avg_log / lg, "linear": d / avg_lin,
         "inverse_linear": avg_lin / d}
    total, run = np.zeros_like(d), np.ones_like(d)
    for name in scalers:
        run = run * f[name]
        total = total + run
    return torch.from_numpy(total.astype(np.float32)).unsqueeze(1)


def check_cpu(rank, world, tag, N, rowptr, col, names=("sum", "mean", "max", "min"), scalers=None):
    from mma_amd.sharded import HaloPlan, all_to_all_rows, partition_bounds
    from oracle import nc_oracle as O
    H, C, names, act = 12, 5, list(names), "new_sigmoid"
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(N, H, generator=g))
    Ws, weight, bias = O.init_like_reference(H, C, names, 1)
    cot = torch.randn(N, C, generator=g)
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    # unsharded oracle
    xf = x.clone().requires_grad_(True)
    deg_all = np.diff(rowptr)
    if scalers is None:
        out_f = O.mma_forward(names, xf, Ws, weight, bias, rowptr, col, dst, col, np.ones(len(col), np.float32), act)
    else:       # true-degree scalers: sum_k A ((R m_k) W) + b with the row factor R from the global degree means
        d1 = np.maximum(deg_all, 1).astype(np.float64)
        avg_log, avg_lin = float(np.log(d1 + 1).mean()), float(d1.mean())
        Rf = true_degree_factor(deg_all, scalers, avg_log, avg_lin)
        Sf = sum((O.aggregate(a, xf, Ws[a], rowptr, col, act) @ weight) * Rf for a in names)
        out_f = torch.zeros(N, C).index_add(0, torch.from_numpy(dst), Sf.index_select(0, torch.from_numpy(col))) + bias
    gx_f, = torch.autograd.grad((out_f * cot).sum(), [xf])

    bounds = partition_bounds(rowptr, world)
    assert bounds[0] == 0 and bounds[-1] == N and (np.diff(bounds) >= 0).all()
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    plan = HaloPlan(rowptr[lo:hi + 1] - e0, col[e0:e1], bounds, rank, world, "cpu")
    # (a) the plan delivers exactly the halo rows
    x_own = x[lo:hi].clone().requires_grad_(True)
    sidx = torch.from_numpy(plan.send_idx)

    class Exchange(torch.autograd.Function):       # CPU stand-in for mma_amd.sharded._HaloExchange (pack = index_select)
        @staticmethod
        def forward(ctx, t):
            ctx.n = t.shape[0]
            return all_to_all_rows(t.index_select(0, sidx), plan.send_counts, plan.recv_counts)

        @staticmethod
        def backward(ctx, gr):
            back = all_to_all_rows(gr.contiguous(), plan.recv_counts, plan.send_counts)
            return torch.zeros(ctx.n, gr.shape[1]).index_add_(0, sidx, back)

    x_halo = Exchange.apply(x_own)
    assert torch.equal(x_halo, x[torch.from_numpy(plan.halo_ids)]), "halo rows differ"
    # (b) the sharded layer algebra with the oracle as compute
    x_src = torch.cat([x_own, x_halo], 0)
    n = plan.n_own
    rp_pad = np.concatenate([plan.rowptr, np.full(plan.n_halo, plan.rowptr[-1])])      # halo rows: degree 0
    ms = [O.aggregate(a, x_src, Ws[a], rp_pad, plan.col, act)[:n] for a in names]
    if scalers is None:
        amp, att = O.scaler_factors(N)
        S = sum((m @ weight) * (1.0 + amp[:1] + att[:1]) for m in ms)
    else:       # a rank owns ALL in-edges of its targets: its local degrees are the true ones; the means are global
        S = sum((m @ weight) * true_degree_factor(np.diff(plan.rowptr), scalers, avg_log, avg_lin) for m in ms)
    S_src = torch.cat([S, Exchange.apply(S)], 0)
    dl = np.repeat(np.arange(n), np.diff(plan.rowptr))
    out = torch.zeros(n, C).index_add(0, torch.from_numpy(dl), S_src.index_select(0, torch.from_numpy(plan.col))) + bias
    close(out, out_f[lo:hi], "%s: sharded out (rank %d)" % (tag, rank))
    gx, = torch.autograd.grad((out * cot[lo:hi]).sum(), [x_own])
    close(gx, gx_f[lo:hi], "%s: sharded gx (rank %d)" % (tag, rank))


def run_gpu(rank, world, variant="hub"):
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.layers import _MASK_NAMES
    from mma_amd.sharded import ShardedMMA, partition_bounds
    dev = "cuda:%d" % torch.cuda.current_device()
    if variant == "sweep":        # the remaining graph shapes of the CPU test + the true-degree scalers, one launch
        for tag, n_, gr, kw in (("messy", 131, messy_graph(5, 131), {}), ("fewer nodes than ranks", 2, (np.array([0, 1, 2]), np.array([1, 0])), {}),
                                ("no edges", 9, (np.zeros(10, np.int64), np.zeros(0, np.int64)), {}),
                                ("true-degree scalers", 157, graph(3, 157, 4),
                                 dict(strict_reference=False, scalers=["identity", "amplification", "linear"], compound_scalers=True)),
                                ("c5 miniature: K=8 masks, S=5 true-degree scalers", 211, graph(9, 211, 5),
                                 dict(H=32, names=["sum", "mean", "max", "min", "sum2", "mean2", "max2", "min2"], strict_reference=False,
                                      scalers=["identity", "amplification", "attenuation", "linear", "inverse_linear"], compound_scalers=True)),
                                ("tall shards: three-product GEMMs with K2a/K2b row maxima", 150000 * world, big_graph(150000 * world), dict(H=128))):
            check_gpu(rank, world, tag, n_, gr[0].astype(np.int64), gr[1].astype(np.int64), kw)
        return
    N, H, C, names, p = 400, 32, 6, ["sum", "mean", "max", "min"], 0.5
    if variant == "directed":
        rowptr, col = directed_graph(N, world)
    elif variant == "empty":
        rowptr, col = hub_first_graph(N)
        assert (np.diff(partition_bounds(rowptr, world)) == 0).any(), "variant 'empty' needs an empty shard"
    else:
        rowptr, col = graph(5, N, 5)
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(N, H, generator=g))
    cot = torch.randn(N, C, generator=g)
    sh = ShardedMMA.build(rowptr, col, rank, world, dev, H, C, names, p, seed=7, chunk=64)
    seed = 0xABCDEF1234
    sh.drop_override = Fn.DropoutSpec(p, seed=seed)
    if variant == "directed":
        assert (rank == 0) == (sh.plan.n_halo == 0) and (rank == 0 or sh.plan.send_counts.sum() >= 0)
        assert rank != 0 or sh.plan.send_counts.sum() > 0            # rank 0: no halo, but its rows are read by the others
    xo = x[sh.lo:sh.hi].to(dev).requires_grad_(True)
    out = sh(xo)
    out.backward(cot[sh.lo:sh.hi].to(dev))
    sh.allreduce_grads()
    # single-GPU layer with the same parameters
    PP = lambda t: torch.nn.Parameter(t.detach().clone())
    masks = {n_: PP(sh.masks[n_]) if n_ in names else torch.nn.Parameter(torch.zeros(2, 1, device=dev)) for n_ in _MASK_NAMES}
    w, b = PP(sh.weight), PP(sh.bias)
    add_all = [col[rowptr[i]:rowptr[i + 1]] for i in range(N)]
    ref = mma_amd.MMA(add_all, "new_sigmoid", 2, H, C, w, b, *[masks[n_] for n_ in _MASK_NAMES], p, names, dev, chunk=64)
    with torch.no_grad():
        for n_ in names:
            masks[n_].copy_(sh.masks[n_])
        w.copy_(sh.weight); b.copy_(sh.bias)
    ref.drop_override = Fn.DropoutSpec(p, seed=seed)
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, dev)
    xf = x.to(dev).requires_grad_(True)
    of = ref(xf, adj)
    of.backward(cot.to(dev))
    close(out, of[sh.lo:sh.hi], "out rank %d" % rank)
    close(xo.grad, xf.grad[sh.lo:sh.hi], "gx rank %d" % rank)
    close(sh.weight.grad, w.grad, "gweight")
    close(sh.bias.grad, b.grad, "gbias")
    for n_ in names:
        close(sh.masks[n_].grad, masks[n_].grad, "gmask " + n_)


def big_graph(N):
    """Random sources, degree 2-5, vectorised (the per-row Python loop of graph() is too slow at this size)."""
    rng = np.random.default_rng(21)
    deg = rng.integers(2, 6, N)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    return rowptr, rng.integers(0, N, rowptr[-1]).astype(np.int64)


def check_gpu(rank, world, tag, N, rowptr, col, kw):
    """ShardedMMA (HIP kernels, halo through gloo) against the single-GPU layer with the same parameters on one graph."""
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.layers import _MASK_NAMES
    from mma_amd.sharded import ShardedMMA
    dev = "cuda:%d" % torch.cuda.current_device()
    kw = dict(kw)
    H, C, names, p, seed = kw.pop("H", 16), 4, kw.pop("names", ["sum", "mean3", "max", "min2"]), 0.5, 0x1234ABCD77
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(N, H, generator=g))
    cot = torch.randn(N, C, generator=g)
    # The SAME work-item plan on both sides (chunks of 16 edges, items below 8 edges grouped): a segment then takes the same kernel
    # path and summation order in the shard and in the whole graph.  With different plans the sums differ in the last bit and a
    # max / min mask flips its selection on a near tie now and then - a legitimate discontinuity of the reference's max(x_i, s), but
    # one flipped element moves a whole row of dL/dx through the mask-weight GEMM (seen: 5 rows of 450 000 at 1 %).
    PLAN = dict(chunk=16, group_below=8, t_group_below=8)
    sh = ShardedMMA.build(rowptr, col, rank, world, dev, H, C, names, p, seed=7, **PLAN, **kw)
    # the device-built shard plan (HaloPlan(plan_device=...), NCGraph.from_device_csr) against the host numpy one, list by list
    from mma_amd.sharded import HaloPlan, partition_bounds
    import mma_amd.graph as G
    assert sh.plan.col_dev is not None, "ShardedMMA.build did not take the device plan path"
    bnd = partition_bounds(rowptr, world)
    e0_, e1_ = int(rowptr[sh.lo]), int(rowptr[sh.hi])
    hp = HaloPlan(rowptr[sh.lo:sh.hi + 1] - e0_, col[e0_:e1_], bnd, rank, world, "cpu" if dist.get_backend() == "gloo" else dev)
    assert np.array_equal(hp.halo_ids, sh.plan.halo_ids) and np.array_equal(hp.col, sh.plan.col_dev.cpu().numpy()), tag + ": halo ids / local columns"
    for f_ in ("send_idx", "send_counts", "recv_counts", "unpack_rows", "unpack_segptr", "unpack_pos"):
        assert np.array_equal(getattr(hp, f_), getattr(sh.plan, f_)), tag + ": " + f_
    gh = G.NCGraph(hp.rowptr, hp.col, dev, n_src=hp.n_src, edge_base=e0_, H=H, **PLAN)
    for f_ in ("rowptr", "col", "items", "hubs", "t_rowptr", "t_col", "t_eid", "t_items", "t_hubs"):
        assert torch.equal(getattr(gh, f_), getattr(sh.graph, f_)), tag + ": graph." + f_
    sh.drop_override = Fn.DropoutSpec(p, seed=seed)
    xo = x[sh.lo:sh.hi].to(dev).requires_grad_(True)
    out = sh(xo)
    out.backward(cot[sh.lo:sh.hi].to(dev))
    sh.allreduce_grads()
    PP = lambda t: torch.nn.Parameter(t.detach().clone())
    masks = {n_: PP(sh.masks[n_]) if n_ in names else torch.nn.Parameter(torch.zeros(2, 1, device=dev)) for n_ in _MASK_NAMES}
    w, b = PP(sh.weight), PP(sh.bias)
    ref = mma_amd.MMA(G.NCGraph(rowptr, col, dev, H=H, **PLAN), "new_sigmoid", 2, H, C, w, b, *[masks[n_] for n_ in _MASK_NAMES], p, names,
                      dev, **kw)
    with torch.no_grad():
        for n_ in names:
            masks[n_].copy_(sh.masks[n_])
        w.copy_(sh.weight); b.copy_(sh.bias)
    ref.drop_override = Fn.DropoutSpec(p, seed=seed)
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, dev)
    xf = x.to(dev).requires_grad_(True)
    of = ref(xf, adj)
    of.backward(cot.to(dev))
    close(out, of[sh.lo:sh.hi], "%s: out rank %d" % (tag, rank))
    close(xo.grad if xo.grad is not None else torch.zeros_like(xo), xf.grad[sh.lo:sh.hi], "%s: gx rank %d" % (tag, rank))
    # the same call with the features written into the head of the rank's source table (no per-call copy of the own rows): bit-equal
    xb = sh.feature_buffer()
    with torch.no_grad():
        xb.copy_(x[sh.lo:sh.hi].to(dev))
    out_b = sh(xb)
    out_b.backward(cot[sh.lo:sh.hi].to(dev))
    assert torch.equal(out_b, out), tag + ": feature_buffer() forward differs"
    assert xo.grad is None or torch.equal(xb.grad, xo.grad), tag + ": feature_buffer() gradient differs"
    for prm in sh.owned:            # the second backward accumulated onto the (already all-reduced) gradients: halve is wrong - recompute
        prm.grad = None
    sh(xo.detach().requires_grad_(True)).backward(cot[sh.lo:sh.hi].to(dev))
    sh.allreduce_grads()
    close(sh.weight.grad, w.grad, tag + ": gweight")
    close(sh.bias.grad, b.grad, tag + ": gbias")
    for n_ in names:
        close(sh.masks[n_].grad, masks[n_].grad, tag + ": gmask " + n_)


def run_grads(rank, world):
    """allreduce_grads: every bucket size from 'all in one' to 'one tensor per bucket', sum and average, tensors without grad."""
    from mma_amd.sharded import allreduce_grads
    shapes = [(7, 5), (1,), (33,), (4, 4, 4), (129,)]
    for bucket_bytes in (1 << 20, 4 * 64, 4):
        for average in (False, True):
            ps = [torch.nn.Parameter(torch.zeros(*s_)) for s_ in shapes] + [torch.nn.Parameter(torch.zeros(3))]     # last: no grad
            for i, p_ in enumerate(ps[:-1]):
                p_.grad = torch.full(p_.shape, float(rank + 1) * (i + 1))
            n = allreduce_grads(ps, average=average, bucket_bytes=bucket_bytes)
            tot = sum(range(1, world + 1)) / (world if average else 1)
            for i, p_ in enumerate(ps[:-1]):
                assert torch.equal(p_.grad, torch.full(p_.shape, tot * (i + 1))), (bucket_bytes, average, i, p_.grad.flatten()[:3])
            assert ps[-1].grad is None
            if bucket_bytes == 1 << 20:
                assert n == 1
            if bucket_bytes == 4:
                assert n == len(shapes)                                        # an oversize tensor still travels (alone)
            assert 1 <= n <= len(shapes)
    assert allreduce_grads([torch.nn.Parameter(torch.zeros(2))]) == 0          # nothing to send: no collective is entered


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode in ("nccl", "nccl_multi"):   # the real RCCL code path (device-side index exchange, all_to_all_single on GPU tensors)
        lr = int(os.environ.get("LOCAL_RANK", 0)) if mode == "nccl_multi" else 0      # nccl: world size 1 on cuda:0; nccl_multi: one GPU per rank
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        try:
            if mode == "cpu":
                run_cpu(rank, world)
            elif mode == "grads":
                run_grads(rank, world)
            elif mode == "nccl":
                run_gpu(rank, world, "hub")
                run_gpu(rank, world, "sweep")
            elif mode == "nccl_multi":          # async RCCL all-to-all-v with uneven and zero counts, comm stream vs compute stream
                for variant in ("hub", "directed", "sweep") + (("empty",) if world == 3 else ()):
                    run_gpu(rank, world, variant)
            else:
                run_gpu(rank, world, sys.argv[2] if len(sys.argv) > 2 else "hub")
        except BaseException:
            import traceback
            print("RANK %d FAILED:\n%s" % (rank, traceback.format_exc()[-1800:]), flush=True)    # the launcher truncates stderr
            raise
        dist.barrier()
        if rank == 0:
            print("SHARDED_%s_OK world=%d" % (mode.upper(), world), flush=True)
    finally:
        dist.destroy_process_group()
