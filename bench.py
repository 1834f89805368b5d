#!/usr/bin/env python3
"""bench.py - edges/sec (fwd+bwd) of the MMA layer on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[3], the config the 1/2/4/8-GPU metric is quoted on; SURVEY 8d "C4"):
synthetic R-MAT power-law graph, 2^20 nodes / ~10 M directed edges, hidden H=128, K=4 masks
[sum, mean, max, min], activation new_sigmoid, mask dropout p=0.5, nclass C=16, fp32.
One step = one forward + backward of the drop-in `mma_amd.MMA` layer (GEMM-pre, fused K-mask aggregate,
GEMM-post, K-stacked SpMM, and their backward) with inputs resident in HBM.  N>1: the same graph is
1-D node-sharded over the ranks (edge-balanced contiguous target ranges) with an RCCL all-to-all halo
exchange per direction (strong scaling: total work fixed).

Prints ONE JSON line (rank 0) with `roofline` (dominant fused kernel, HIP-event timed inside the timed
region, algorithmic bytes from DESIGN.md) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured streaming copy)


# ---- synthetic graph ---------------------------------------------------------------------
def rmat_graph(scale, n_undirected, seed=42, a=0.57, b=0.19, c=0.19):
    """R-MAT -> symmetrised, de-duplicated, no self loops, isolated nodes attached to a random node.
    Returns CSR by target (rowptr int64, col int64) with ascending neighbour order (utils.py:100)."""
    rng = np.random.default_rng(seed)
    N = 1 << scale
    src = np.zeros(n_undirected, dtype=np.int64)
    dst = np.zeros(n_undirected, dtype=np.int64)
    for bit in range(scale):
        r = rng.random(n_undirected, dtype=np.float32)
        sb = (r >= a + b).astype(np.int64)                       # quadrants c,d set the source bit
        db = (((r >= a) & (r < a + b)) | (r >= a + b + c)).astype(np.int64)   # quadrants b,d set the target bit
        src |= sb << bit
        dst |= db << bit
    perm = rng.permutation(N)                                    # break the bit-pattern locality of raw R-MAT ids
    src, dst = perm[src], perm[dst]
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * N + dst, dst * N + src]))
    row, col = key // N, key % N
    deg = np.bincount(row, minlength=N)
    iso = np.nonzero(deg == 0)[0]
    if len(iso):                                                 # reference needs d >= 1 (Q12)
        nb = rng.integers(0, N, len(iso))
        nb = np.where(nb == iso, (nb + 1) % N, nb)
        key = np.unique(np.concatenate([key, iso * N + nb, nb * N + iso]))
        row, col = key // N, key % N
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=N), out=rowptr[1:])
    return rowptr, col


class KernelTimer:
    """HIP events around the C-ABI calls, on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.spans = {}
        self.enabled = False

    def span(self, name):
        timer = self

        class _Ctx:
            def __enter__(self_):
                if timer.enabled:
                    self_.e0 = torch.cuda.Event(enable_timing=True); self_.e1 = torch.cuda.Event(enable_timing=True)
                    self_.e0.record()

            def __exit__(self_, *a):
                if timer.enabled:
                    self_.e1.record()
                    timer.spans.setdefault(name, []).append((self_.e0, self_.e1))
        return _Ctx()

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.spans.items()}


def algorithmic_bytes(N, E, H, K, n_sel=None):
    """Zero-reuse byte counts per call (DESIGN.md 'algorithmic bytes'; fwd = SURVEY 8d B_fwd).
    n_sel: number of max/min/softmax-type masks when K2b runs in the shared-gradient form, None for the gs form."""
    fwd = 4 * (E * (1 + (K + 1) * H) + N * (1 + (2 * K + 1) * H))
    per_node = 4 * N * (1 + (2 * K + 3) * H)                      # x, Q, gxs in; gQ, gx out
    if n_sel is None:                                             # per edge: t_col, t_eid + gs and P rows
        bwd = 4 * E * (2 + 2 * K * H) + per_node
    else:                                                         # per edge: t_col, t_eid + P row + packed [g | 1/d | codes] row
        bwd = E * (8 + 4 * K * H + 4 * H + 16 + n_sel * H) + per_node
    return {"nc_fused_fwd": fwd, "nc_fused_bwd": bwd}


PMC_KERNEL = {"nc_fused_fwd": "mma::nc_fwd_", "nc_fused_bwd": "mma::nc_bwd_k"}   # fwd: kernel + finalize; bwd: kernel only (nc_bwd_node is K2a)


def pmc_traffic(name, N, E, H, K):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes of THIS command
    (profiles/r*_pmc_traffic.json, made by tools/pmc_summary.py); None when no profile matches the workload."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != {"nodes": N, "edges": E, "hidden": H, "K": K}:
            continue
        # one C-ABI call = up to two launches of the kernel (items run one per wavefront / grouped) + the hub finalize
        parts = [v["traffic_bytes"] for k, v in d["kernels"].items() if PMC_KERNEL.get(name, "?") in k]
        if parts:
            return sum(parts)
    return None


def cpu_baseline(rowptr, col, H, names, activation, p, n_targets, seed):
    """The CPU oracle (oracle/nc_oracle.py, torch-CPU vectorised restatement of layers.py) on a bounded sample:
    the first n_targets target nodes with ALL their in-edges; fwd+bwd of the K aggregators + dense tail."""
    from oracle import nc_oracle as O
    # the box gives one GPU job a share of the host (16 cores), whatever os.cpu_count() says
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    e_hi = int(rowptr[n_targets])
    sub_col = col[:e_hi]
    nodes = np.unique(np.concatenate([np.arange(n_targets), sub_col]))      # targets first (they are 0..n_targets-1)
    remap = np.full(len(rowptr) - 1, -1, dtype=np.int64); remap[nodes] = np.arange(len(nodes))
    n_sub = len(nodes)
    rp = np.full(n_sub + 1, e_hi, dtype=np.int64); rp[:n_targets + 1] = rowptr[:n_targets + 1]
    cj = remap[sub_col]
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n_sub, H, generator=g)).requires_grad_(True)
    Ws = {n: ((torch.rand(2 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).requires_grad_(True) for n in names}
    keeps = {n: (torch.rand(e_hi, H, generator=g) >= p).float() for n in names} if p > 0 else None
    t0 = time.perf_counter()
    ms = [O.aggregate(n, x, Ws[n], rp, cj, activation, p, None if keeps is None else keeps[n]) for n in names]
    loss = sum(m[:n_targets].sum() for m in ms)
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": e_hi / dt, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": "first %d target nodes of the C4 graph with all their in-edges (%d edges, %d distinct rows), "
                      "fwd+bwd of the K=%d masked aggregators, torch-CPU vectorised oracle, %.1f s" % (
                          n_targets, e_hi, n_sub, len(names), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=20, help="R-MAT scale (2^scale nodes)")
    ap.add_argument("--edges", type=int, default=5_000_000, help="undirected R-MAT edges before symmetrisation")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--nclass", type=int, default=16)
    ap.add_argument("--aggregators", type=str, default="sum,mean,max,min")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--force-sharded", action="store_true", help="use the sharded (RCCL) path even at world size 1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' (halo staged through the host) lets "
                    "several ranks share one GPU to rehearse the N>1 path on a 1-GPU box")
    ap.add_argument("--cpu-sample", type=int, default=150000, help="target nodes in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    if args.backend == "gloo":
        local_rank = 0                      # rehearsal: all ranks on cuda:0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if "RANK" not in os.environ:      # --force-sharded without a launcher
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import mma_amd
    from mma_amd import functional as Fn

    names = args.aggregators.split(",")
    K, H, C = len(names), args.hidden, args.nclass
    rowptr, col = rmat_graph(args.scale, args.edges, seed=42)
    N, E = len(rowptr) - 1, int(rowptr[-1])

    timer = KernelTimer()
    Fn.TIMER = timer
    torch.manual_seed(42)
    gen = torch.Generator(device="cpu").manual_seed(42)
    x_full = torch.relu(torch.randn(N, H, generator=gen))
    cot_full = torch.randn(N, C, generator=gen)

    if not sharded:
        graph = mma_amd.NCGraph(rowptr, col, dev)
        layer = make_layer(mma_amd, graph, H, C, names, args.dropout, dev)
        dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr))
        adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, dev)
        x = x_full.to(dev).requires_grad_(True)
        cot = cot_full.to(dev)

        def step():
            x.grad = None
            for prm in layer.owned:
                prm.grad = None
            out = layer(x, adj)
            out.backward(cot)
        local_edges, n_local = E, N
    else:
        from mma_amd.sharded import ShardedMMA
        sh = ShardedMMA.build(rowptr, col, rank, world, dev, H, C, names, args.dropout)
        x = x_full[sh.lo:sh.hi].to(dev).requires_grad_(True)
        cot = cot_full[sh.lo:sh.hi].to(dev)

        def step():
            x.grad = None
            for prm in sh.owned:
                prm.grad = None
            out = sh(x)
            out.backward(cot)
            sh.allreduce_grads()
        local_edges, n_local = sh.local_edges, sh.hi - sh.lo
    del x_full, cot_full

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if sharded:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    spans = timer.summary()
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = E * args.steps / dt
        n_sel = sum(1 for a in names if a.rstrip("234") in ("max", "min", "softmax", "softmin")) if Fn.SHARED_GRAD_BWD else None
        ab = algorithmic_bytes(n_local, local_edges, H, K, n_sel)
        kernels = {}
        for name, (cnt, tot_ms) in spans.items():
            avg = tot_ms / args.steps           # per step (a sharded backward issues the call twice: halo / own sources)
            k = {"launches": cnt, "avg_ms": avg}
            if name in ab:
                k["algorithmic_bytes"] = ab[name]
                k["achieved_GBs"] = ab[name] / (avg * 1e-3) / 1e9
            kernels[name] = k
        dom = max((n for n in kernels if n in ab), key=lambda n: kernels[n]["avg_ms"])
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["achieved_GBs"] / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom, N, E, H, K) if not sharded else None}
        cpu = None
        if args.cpu_sample and not sharded:
            cpu = cpu_baseline(rowptr, col, H, names, "new_sigmoid", args.dropout, min(args.cpu_sample, N), 42)
        line = {
            "metric": "aggregated edges/sec (fwd+bwd) MultiMaskConv", "value": value, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4: synthetic R-MAT power-law graph, %d nodes / %d directed edges, feat=%d, "
                                   "K=%d masks [%s], nclass=%d, mask dropout p=%g, MMA layer fwd+bwd" % (
                                       N, E, H, K, ",".join(names), C, args.dropout),
                       "nodes": N, "edges": E, "hidden": H, "K": K, "nclass": C,
                       "parallelism": "1-D node shard x%d, RCCL all-to-all halo" % world if world > 1 else "single GPU"},
            "masked_edges_per_s": value * K,
            "roofline": roof, "kernels": kernels, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if sharded:
        dist.destroy_process_group()


def make_layer(mma_amd, graph, H, C, names, p, dev):
    """The drop-in MMA layer with externally owned Parameters, as models.py:17-60 creates them."""
    from mma_amd.layers import _MASK_NAMES
    P = lambda *s: torch.nn.Parameter(torch.empty(*s, device=dev))
    # only the masks in use get a full (2H,H) tensor; the reference allocates all 21 (models.py:21-41)
    masks = {n: P(2 * H, H) if n in names else P(2, 1) for n in _MASK_NAMES}
    w, b = P(H, C), P(C)
    layer = mma_amd.MMA(graph, "new_sigmoid", 2, H, C, w, b, *[masks[n] for n in _MASK_NAMES], p, names, dev)
    layer.owned = [w, b] + [masks[n] for n in names]
    return layer


if __name__ == "__main__":
    main()
