#!/usr/bin/env python3
"""bench.py - edges/sec (fwd+bwd) of the MMA message-passing hot path on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2l]

N > 1 without a launcher (RANK unset): bench.py starts N child ranks itself (`python -m torch.distributed.run ...
bench.py <same args>`) BEFORE anything touches the GPU, relays rank 0's JSON line and exits with the children's status.
Under a launcher (RANK/WORLD_SIZE set, the driver's documented form) it is one of the ranks.

Workload `c4` (default; BASELINE.json configs[3], the config the 1/2/4/8-GPU metric is quoted on; SURVEY 8d "C4"):
synthetic R-MAT power-law graph, 2^20 nodes / ~10.9 M directed edges, hidden H=128, K=4 masks [sum, mean, max, min],
activation new_sigmoid, mask dropout p=0.5, nclass C=16, fp32.  One step = one forward + backward of the drop-in
`mma_amd.MMA` layer (GEMM-pre, fused K-mask aggregate, GEMM-post, K-stacked SpMM, and their backward), inputs resident
in HBM.  N>1: the same graph is 1-D node-sharded over the ranks (edge-balanced contiguous target ranges) with an RCCL
all-to-all halo exchange per direction (strong scaling: total work fixed).  Rank 0 generates the graph once and shares
it through /dev/shm; every rank materialises only its own rows of the features.
Workload `c2l` (BASELINE configs[1] at ZINC-split scale): the drop-in `mma_amd.MMAConv` 75->75, towers=5, edge_dim=50,
[min,max] x [identity,amplification,linear] on a 10 000-molecule batch; `roofline` is the fused GR kernel.  N>1: molecule
batches are independent, so the ranks are data-parallel REPLICAS (own batch each, parameters broadcast from rank 0, gradients
averaged by one bucketed all-reduce per step; weak scaling).

Prints ONE compact strict-JSON line (rank 0, stdout, < 4 KB) with `roofline` (dominant fused kernel, HIP-event timed inside the
timed region) and `cpu_baseline` (the CPU oracle on a bounded sample).  The verbose record - `kernels` (bytes / flops / bound / frac of
every timed call), `plan_build`, `ranks` (N > 1) and `extra`, the other BASELINE configs that fit one GPU (C1, C3, C2, C2L: layer
fwd+bwd, eager and one-hipGraph replay; C5 at its per-GPU shard shape) - goes to stderr as one `{"detail": ...}` line and to
$MMA_BENCH_DETAIL (default gpurun_out/bench_detail.json); the compact line keeps `kernels_ms` and `extra_summary`.

Byte counts.  `roofline.achieved` = `algorithmic_bytes()` of THIS kernel / its HIP-event time: the bytes the kernel's design has to
move with zero credit for cache reuse of gathered rows.  For K1 that is SURVEY 8d's B_fwd.  For the backward it is
NOT 8d's B_bwd (which priced a re-gather + per-edge atomic scatter, 63.2 GB at C4): K2b walks the transposed CSR, writes every
source row exactly once and carries the node-level backward in its epilogue - 41.35 GB at C4.  `roofline.frac_rocprof` prices the
same bytes against the kernels' own durations in the committed rocprofv3 pass (no launch gaps; None when the recorded profile is of
another build).  `kernels[name]` carries bytes / flops / bound / frac for EVERY timed call: the dense products state their work at
the call site (mma_amd.functional._span).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from tools.synth import feature_rows, golden_csr, molecule_batch, rmat_graph  # noqa: E402,F401

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured streaming copy)
C4_DEFAULTS = dict(scale=20, edges=5_000_000, hidden=128, nclass=16, aggregators="sum,mean,max,min", dropout=0.5)
# BASELINE configs[4] (SURVEY 8d "C5"): R-MAT scale 23, 64 M undirected -> ~128 M directed edges, feat=256, K=8, S=5 true-degree scalers.
# P/Q/T of the whole graph are > 200 GB: it runs 1-D sharded over the 8 GPUs of a node (`--workload c5 --gpus 8`), never on one.
C5_PRESET = dict(scale=23, edges=64_000_000, hidden=256, nclass=16, aggregators="sum,mean,max,min,sum2,mean2,max2,min2", dropout=0.5,
                 true_degree_scalers=True)
C5_MIN_GPUS = 4


# ---- self-launch -----------------------------------------------------------------------------------------
def self_launch(argv, gpus):
    """`python bench.py --gpus N` with no launcher: become the launcher.  Runs before any torch.cuda call - this process
    never touches the GPU, it only starts N fresh children and waits (no exec of a GPU-initialised process)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // gpus)))
    return subprocess.run(cmd, env=env).returncode


class KernelTimer:
    """HIP events around the C-ABI calls, on the stream the kernels are launched on (torch's current stream).  A call site may state
    the bytes / flops its call has to do (mma_amd.functional._span): they are summed per name beside the times."""

    def __init__(self):
        self.spans = {}
        self.work = {}          # name -> [bytes, flops, mfma kind]
        self.enabled = False

    def span(self, name, nbytes=0, flops=0, mfma=None):
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
                    w = timer.work.setdefault(name, [0, 0, None])
                    w[0] += nbytes; w[1] += flops; w[2] = mfma or w[2]
        return _Ctx()

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.spans.items()}


def algorithmic_bytes(N, E, H, K, n_sel=None, fused_node_bwd=None):
    """Zero-reuse byte counts per call (DESIGN.md 'algorithmic bytes'; fwd = SURVEY 8d B_fwd).
    n_sel: number of max/min/softmax-type masks when K2b runs in the shared-gradient form, None for the gs form.
    fused_node_bwd (default: what mma_amd.functional does): the node-level backward (K2a) runs in K2b's per-source epilogue, so the
    call also moves K2a's streams: g, T and the code row in, gP out - and no gxs."""
    fwd = 4 * (E * (1 + (K + 1) * H) + N * (1 + (2 * K + 1) * H))
    per_node = 4 * N * (1 + (2 * K + 3) * H)                      # x, Q, gxs in; gQ, gx out
    if n_sel is None:                                             # per edge: t_col, t_eid + gs and P rows
        bwd = 4 * E * (2 + 2 * K * H) + per_node
    else:                                                         # per edge: t_col, t_eid + P row + g row + code row [1/d,0,0,0 | codes]
        bwd = E * (8 + 4 * K * H + 4 * H + 16 + n_sel * H) + per_node
        if fused_node_bwd is None:
            from mma_amd import functional as Fn
            fused_node_bwd = Fn.FUSE_NODE_BWD
        if fused_node_bwd:                                        # + g, T, code row in, gP out; - gxs in
            bwd += N * (4 * H + 4 * K * H + 16 + n_sel * H + 4 * K * H) - 4 * N * H
    return {"nc_fused_fwd": fwd, "nc_fused_bwd": bwd}


def gr_algorithmic_bytes(N, E, T, F, K, S, has_z=True):
    """SURVEY 8d, GR fused kernels: fwd 4[E(2 + TF + TF_Z) + N(1 + TF + TKSF)]; bwd reads the (N,T,K*S*F) gradient and
    the saved state and writes one message gradient per edge + dU per node."""
    D = T * F
    if has_z == "categorical":        # "0 + 1 byte/edge type if categorical table"
        fwd = 4 * (E * (2 + D) + N * (1 + D + T * K * S * F)) + E
    else:
        fwd = 4 * (E * (2 + D + (D if has_z else 0)) + N * (1 + D + T * K * S * F))
    bwd = 4 * (E * (2 + D) + N * (1 + T * K * S * F + 2 * D))
    return {"gr_fused_fwd": fwd, "gr_fused_bwd": bwd}


PMC_KERNEL = {"nc_fused_fwd": "mma::nc_fwd_", "nc_fused_bwd": "mma::nc_bwd_k",   # fwd: kernel + finalize; bwd: kernel only
              "gr_fused_fwd": "mma::gr_fwd", "gr_fused_bwd": "mma::gr_bwd"}


def pmc_traffic(name, workload):
    """(HBM bytes per launch of the dominant kernel, source) from the COMMITTED rocprofv3 PMC passes of this command
    (profiles/r*_pmc_traffic*.json, made by tools/make_profiles.py) - a recorded profile, not measured in this run.  A profile is
    used only if its `build` stamp (ABI version + SHA over the kernel sources, tools/build_stamp.py, written on the GPU box next to
    the passes) equals the running tree's: a traffic figure never outlives the kernel it was measured on.  (None, reason) when no
    profile matches the workload or the newest matching one is stale."""
    import glob
    from tools.build_stamp import build_stamp
    now = build_stamp()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        if d.get("build") != now:
            return None, "REFUSED %s: recorded on build %s, this tree is %s - re-run tools/profile_round.sh" % (
                os.path.relpath(f, ROOT), json.dumps(d.get("build")), json.dumps(now)), None
        # one C-ABI call = up to two launches of the kernel (items run one per wavefront / grouped) + the hub finalize
        hit = [v for k, v in d["kernels"].items() if PMC_KERNEL.get(name, "?") in k]
        if hit:
            ns = [v.get("rocprof_avg_ns") for v in hit]
            return sum(v["traffic_bytes"] for v in hit), os.path.relpath(f, ROOT), (sum(ns) / 1e6 if all(ns) else None)
    return None, None, None


def _traffic_source(traffic, src):
    if src is None:
        return None
    return ("recorded rocprofv3 PMC passes of this command, " + src) if traffic is not None else src


# ---- CPU baselines (the oracle is the checker/baseline, never the product path) ------------------------------
def _threads():
    # the box gives one GPU job a share of the host (16 cores), whatever os.cpu_count() says
    n = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(n)
    return n


def cpu_baseline(rowptr, col, H, names, activation, p, n_targets, seed, label):
    """Vectorised CPU oracle (oracle/nc_oracle.py, torch-CPU restatement of layers.py) on a bounded sample: the first
    n_targets target nodes with ALL their in-edges; fwd+bwd of the K masked aggregators."""
    from oracle import nc_oracle as O
    threads = _threads()
    e_hi = int(rowptr[n_targets])
    sub_col = np.asarray(col[:e_hi])
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
    return {"value": e_hi / dt, "unit": "edges/s", "cores": threads, "kind": "port", "form": "vectorised",
            "sample": "first %d target nodes of the %s graph with all their in-edges (%d edges, %d distinct rows), "
                      "fwd+bwd of the K=%d masked aggregators, torch-CPU vectorised oracle, %.1f s" % (
                          n_targets, label, e_hi, n_sub, len(names), dt)}


def cpu_loop_baseline(tag, fixture, H, names, p, max_nodes, seed=42):
    """Faithful-loop CPU oracle (oracle.nc_oracle.aggregate_loop: the per-node op sequence of layers.py:205-226) on the
    Cora / Pubmed structure - SURVEY 8d "CPU baseline (i)"; expected ~ the reference's own speed (1.6-2.0 k
    edges/s on 8 cores).  Bounded: the first max_nodes target nodes, fwd+bwd."""
    from oracle import nc_oracle as O
    threads = _threads()
    rowptr, col = golden_csr(fixture)
    N = len(rowptr) - 1
    n = min(N, max_nodes)
    e_hi = int(rowptr[n])
    add_all = [col[rowptr[i]:rowptr[i + 1]] for i in range(n)]
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(N, H, generator=g)).requires_grad_(True)
    Ws = {a: ((torch.rand(2 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).requires_grad_(True) for a in names}
    keeps = {a: (torch.rand(e_hi, H, generator=g) >= p).float() for a in names} if p > 0 else None
    t0 = time.perf_counter()
    ms = [O.aggregate_loop(a, x, Ws[a], add_all, "new_sigmoid", p, None if keeps is None else keeps[a]) for a in names]
    sum(m.sum() for m in ms).backward()
    dt = time.perf_counter() - t0
    return {"config": tag, "value": e_hi / dt, "unit": "edges/s", "cores": threads, "kind": "port", "form": "faithful per-node loop",
            "sample": "first %d of %d target nodes of the %s structure (%d edges), H=%d, aggregators %s, p=%g, fwd+bwd, %.1f s" % (
                n, N, fixture, e_hi, H, ",".join(names), p, dt)}


# ---- secondary configurations (N = 1 only; BASELINE configs that are not the headline) --------------------------
def graph_replay_ms(step, n=50):
    """Capture `step` (layer forward+backward on static tensors) into a hipGraph and time replays."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def wall_ms(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# dense matrix-core peaks (MI355X_MICROARCH.md): fp16 / bf16 MFMA ~2.5 PFLOP/s; fp32-input MFMA = the fp32 vector rate, 157.3 TFLOP/s
MFMA16_PEAK, MFMA32_PEAK = 2.5e15, 157.3e12
PIECE_PRODUCTS = {"f16x3": 3, "bf16x6": 6}


def _kernel_table(spans, steps, ab, work=None, step_ms=None):
    """Per C-ABI call name: launches, ms per STEP, and - for every call whose work is known - its roofline (round-3 VERDICT item 1a):
    `bytes` = algorithmic HBM bytes per step (the fused kernels: bench.py's formulas, `ab`; the dense ones: stated by their call
    sites, mma_amd.functional._span), `flops` = fp32-equivalent flops per step, `bound` = the larger of the two floors - bytes at
    8.0 TB/s ("hbm") or the matrix-core work at its dense peak ("mfma": a split-precision product runs 3 (fp16) or 6 (bf16) piece
    products per fp32 product at 2.5 PFLOP/s; the exact-fp32 MFMA runs at 157 TFLOP/s) - `frac` = that floor / measured time (also
    given per resource: hbm_frac, mfma_frac), `share` = its part of the step."""
    kernels = {}
    for name, (cnt, tot_ms) in spans.items():
        avg = tot_ms / steps            # per step (a sharded backward issues the call twice: halo / own sources)
        k = {"launches": cnt, "avg_ms": avg}
        nbytes, flops, mfma = 0, 0, None
        if name in ab:
            nbytes = ab[name]
        elif work and name in work:
            nbytes, flops, mfma = work[name][0] / steps, work[name][1] / steps, work[name][2]
        if nbytes:
            k["algorithmic_bytes"] = nbytes
            k["achieved_GBs"] = nbytes / (avg * 1e-3) / 1e9
            hbm_ms = nbytes / (HBM_PEAK_GBS * 1e9) * 1e3
            mf_ms = 0.0
            if flops and mfma:
                mf_ms = (flops * PIECE_PRODUCTS[mfma] / MFMA16_PEAK if mfma in PIECE_PRODUCTS else flops / MFMA32_PEAK) * 1e3
                k.update(flops=flops, mfma=mfma, mfma_frac=mf_ms / avg)
            k.update(bytes=nbytes, hbm_frac=hbm_ms / avg, bound="mfma" if mf_ms > hbm_ms else "hbm", frac=max(hbm_ms, mf_ms) / avg)
        if step_ms:
            k["share"] = avg / step_ms
        kernels[name] = k
    return kernels


def nc_config(tag, fixture, H, names, C, p, dev, reps=50, replay=True):
    """MMA layer fwd+bwd on a committed fixture graph (Cora / Pubmed structure): eager wall, per-kernel HIP-event times,
    one-hipGraph replay."""
    import mma_amd
    from mma_amd import functional as Fn
    rowptr, col = golden_csr(fixture)
    N, E, K = len(rowptr) - 1, len(col), len(names)
    graph = mma_amd.NCGraph(rowptr, col, dev, H=H)
    layer = make_layer(mma_amd, graph, H, C, names, p, dev)
    dst = np.repeat(np.arange(N), np.diff(rowptr))
    adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, dev)
    x = torch.relu(torch.randn(N, H, device=dev)).requires_grad_(True)
    cot = torch.randn(N, C, device=dev)

    def step():
        x.grad = None
        layer.zero_grad(set_to_none=True)        # train.py:73 optimizer.zero_grad(): gradients are written, not accumulated
        layer(x, adj).backward(cot)
    ms = wall_ms(step, reps)
    prev = Fn.TIMER
    Fn.TIMER = t = KernelTimer(); t.enabled = True
    for _ in range(reps):
        step()
    spans = t.summary(); Fn.TIMER = prev
    n_sel = sum(1 for a in names if a.rstrip("234") in ("max", "min", "softmax", "softmin")) if Fn.SHARED_GRAD_BWD else None
    r = {"config": tag, "nodes": N, "edges": E, "hidden": H, "K": K, "dropout": p, "ms_per_step_eager": ms, "edges_per_s_eager": E / ms * 1e3,
         "kernels": _kernel_table(spans, reps, algorithmic_bytes(N, E, H, K, n_sel), t.work, ms),
         "note": "launch-bound: the fused kernels move tens of MB (a few us at the HBM roofline)"}
    if replay:
        layer.graph_capturable = True
        gms = graph_replay_ms(step)
        r.update(ms_per_step_hipgraph=gms, edges_per_s_hipgraph=E / gms * 1e3)
        r["fused_calls_in_graph_us"] = fused_calls_in_graph(layer, graph, x.detach(), names, p)
    return r


def fused_calls_in_graph(layer, graph, x, names, p, reps=20):
    """The two fused C-ABI calls (mma_nc_fused_fwd; mma_nc_fused_bwd with the node-level backward in its epilogue) on their own: `reps`
    back-to-back calls of each captured into one hipGraph, replay time / reps = what ONE call costs inside a replayed step (kernel
    start to next kernel start).  The eager HIP-event spans of the `kernels` table also hold the host's launch gaps."""
    from mma_amd import functional as Fn
    kinds, acts = layer._codes(names)
    N, H = x.shape
    KH = len(kinds) * H
    masks = [getattr(layer, "mask_" + n).detach() for n in names]
    PQ = x @ Fn.mask_weights(masks)
    drop = Fn.DropoutSpec(p, seed=1234)
    g = torch.randn(N, H, device=x.device)
    gPQ = torch.empty((N, 2 * KH), device=x.device)
    gx = torch.empty((N, H), device=x.device)
    partial = torch.empty((graph.t_n_slots, (len(kinds) + 1) * H), device=x.device) if graph.t_n_slots else None
    state = {}

    def fwd():
        state["out"] = Fn.nc_fwd_launch(x, PQ[:, :KH], PQ[:, KH:], graph, kinds, acts, drop, True, True)

    def bwd():
        _, T, _, crow = state["out"]
        Fn.nc_bwd_edges_launch(x, PQ[:, :KH], PQ[:, KH:], None, g, crow, None, graph, kinds, acts, drop, gPQ[:, KH:], gx, partial, T=T,
                               gP=gPQ[:, :KH])
    fwd()
    if state["out"][3] is None or not Fn.FUSE_NODE_BWD:
        return None
    res = {}
    for name, fn in (("nc_fused_fwd", fwd), ("nc_fused_bwd", bwd)):
        def many():
            for _ in range(reps):
                fn()
        res[name] = graph_replay_ms(many, 20) / reps * 1e3
    res["note"] = "%d back-to-back calls in one hipGraph, replay / %d" % (reps, reps)
    return res


def nc_model_config(tag, fixture, nfeat, density, hidden, names, nclass, p, n_train, dev, reps=30):
    """BASELINE configs[0] / configs[2] at model level: one epoch of train.py:72-80 on the reference's models.MMAConv
    (GraphConvolution -> ReLU -> dropout -> MMA layer -> log_softmax; nll_loss on idx_train; backward; Adam, lr 0.01, weight
    decay 5e-4) on the committed graph structure with synthetic bag-of-words-like features of the dataset's width and
    density: eager with torch's loss / Adam, eager with the fused K10 / K11, and the fused step replayed as ONE hipGraph."""
    import scipy.sparse as sp
    import mma_amd
    from mma_amd.models import MMAConv
    from mma_amd.utils import sparse_mx_to_torch_sparse_tensor
    rowptr, col = golden_csr(fixture)
    N, E = len(rowptr) - 1, len(col)
    rng = np.random.default_rng(3)
    add_all = [col[rowptr[i]:rowptr[i + 1]] for i in range(N)]
    adj = sparse_mx_to_torch_sparse_tensor(sp.csr_matrix((np.ones(E, np.float32), col, rowptr), shape=(N, N))).to(dev)
    feats = sp.random(N, nfeat, density=density, format="csr", dtype=np.float32, random_state=3)
    x = torch.from_numpy(feats.toarray()).to(dev)
    y = torch.from_numpy(rng.integers(0, nclass, N)).to(dev)
    idx = torch.arange(0, n_train, device=dev)
    res = {"config": tag, "nodes": N, "edges": E, "nfeat": nfeat, "hidden": hidden, "K": len(names), "dropout": p}
    for key, fused, graphed in (("ms_per_epoch_eager_torch_loss_adam", False, False), ("ms_per_epoch_eager_fused", True, False),
                                ("ms_per_epoch_hipgraph_fused", True, True)):
        torch.manual_seed(42)
        model = MMAConv(add_all, "new_sigmoid", 2, nfeat, hidden, nclass, p, list(names), dev).to(dev)
        for q in model.parameters():
            if q.dim() == 2 and not torch.isfinite(q).all():
                torch.nn.init.uniform_(q, -0.1, 0.1)
        model.train()
        params = [q for q in model.parameters() if q.requires_grad]
        if fused:
            opt = mma_amd.FusedAdam(params, lr=0.01, weight_decay=5e-4)
            loss_fn = lambda: model.nll_loss(x, adj, idx, y)[0]
        else:
            opt = torch.optim.Adam(params, lr=0.01, weight_decay=5e-4)
            loss_fn = lambda: torch.nn.functional.nll_loss(model(x, adj)[idx], y[idx])
        if graphed:
            step = mma_amd.GraphedTrainStep(model, opt, loss_fn)
        else:
            def step():
                opt.zero_grad(set_to_none=False)
                loss = loss_fn(); loss.backward(); opt.step()
                return loss
        first = step().item()
        res[key] = wall_ms(step, reps)
        res.setdefault("loss_first", first)
    res["epochs_per_s_hipgraph"] = 1e3 / res["ms_per_epoch_hipgraph_fused"]
    return res


def gr_setup(n_graphs, dev, seed=0, categorical=False):
    import mma_amd
    rng = np.random.default_rng(seed)
    ei, N = molecule_batch(rng, n_graphs)
    E = ei.shape[1]
    # the degree histogram is a property of the training SET (mma.py:57-60), the same on every replica: batch 0's
    hi, hN = (ei, N) if seed == 0 else molecule_batch(np.random.default_rng(0), n_graphs)
    hist = np.bincount(np.bincount(hi[1], minlength=hN), minlength=5)
    conv = mma_amd.MMAConv(75, 75, ["min", "max"], ["identity", "amplification", "linear"], torch.tensor(hist), edge_dim=50,
                           towers=5).to(dev)
    x = torch.randn(N, 75, device=dev, requires_grad=True)
    ea = torch.randn(E, 50, device=dev)
    if categorical:     # ZINC's real edge features: 4 bond types through an Embedding(4, 50) (mma.py:88,103)
        ea = mma_amd.CategoricalEdges(torch.from_numpy(rng.integers(0, 4, E)).to(dev), torch.randn(4, 50, device=dev))
    eig = torch.from_numpy(ei).to(dev)
    cot = torch.randn(N, 75, device=dev)

    params = [q for q in conv.parameters() if q.requires_grad]

    def step():
        x.grad = None
        for q in params:                   # optimizer.zero_grad(): the backward then assigns instead of launching an add per tensor
            q.grad = None
        conv(x, eig, ea).backward(cot)

    def check():          # outside every timed region: the long-segment list of the layer's CSR was never found clobbered (K3 / K4 skip such a list and flag it)
        from mma_amd import functional as Fn
        Fn.gr_graph(eig, N).by_target.check()
    step.check = check
    return conv, step, N, E


def gr_config(tag, n_graphs, dev, reps=20, replay=True, categorical=False):
    """MMAConv (mma.py:92-95 shape) layer fwd+bwd on a ZINC-like batch of n_graphs molecules."""
    from mma_amd import functional as Fn
    conv, step, N, E = gr_setup(n_graphs, dev, categorical=categorical)
    ms = wall_ms(step, reps)
    prev = Fn.TIMER
    Fn.TIMER = t = KernelTimer(); t.enabled = True
    for _ in range(reps):
        step()
    spans = t.summary(); Fn.TIMER = prev
    step.check()
    r = {"config": tag, "graphs": n_graphs, "nodes": N, "edges": E, "towers": 5, "F": 75, "ms_per_step_eager": ms,
         "edges_per_s_eager": E / ms * 1e3,
         "kernels": _kernel_table(spans, reps, gr_algorithmic_bytes(N, E, 5, 75, 2, 1 if __import__("mma_amd").mma_conv.FACTOR_SCALERS else 3,
                                                                    "categorical" if categorical else True), t.work, ms)}
    if replay:
        conv.graph_capturable = True
        gms = graph_replay_ms(step, 20)
        r.update(ms_per_step_hipgraph=gms, edges_per_s_hipgraph=E / gms * 1e3)
    return r


def gr_model_config(tag, n_graphs, dev, reps=10, n_batches=3, graphed=False):
    """BASELINE configs[1] at model level: one TRAINING step of the reference's graph-regression Net (mma.py:63-127: atom / bond
    embeddings, 4 x (MMAConv 75->75, towers=5, edge_dim=50 + BatchNorm + ReLU), add-pooling, MLP) with its L1 loss (mma.py:156)
    and Adam - forward, fused loss, backward, fused optimizer step - on a FRESH batch every step (rotating pre-generated
    batches, so the device CSR build of each batch is inside the step, as in training), eager."""
    import mma_amd
    rng = np.random.default_rng(11)
    batches = []
    for _ in range(n_batches):
        ei, N, sizes = molecule_batch(rng, n_graphs, return_sizes=True)
        E = ei.shape[1]
        batches.append(dict(ei=torch.from_numpy(ei).to(dev), x=torch.from_numpy(rng.integers(0, 21, (N, 1))).to(dev),
                            ea=torch.from_numpy(rng.integers(0, 4, E)).to(dev),
                            batch=torch.from_numpy(np.repeat(np.arange(n_graphs), sizes)).to(dev),
                            y=torch.from_numpy(rng.standard_normal(n_graphs).astype(np.float32)).to(dev), N=N, E=E))
    ei0 = batches[0]["ei"].cpu().numpy()
    hist = np.bincount(np.bincount(ei0[1], minlength=batches[0]["N"]), minlength=5)
    torch.manual_seed(0)
    from mma_amd.net import Net
    net = Net(["min", "max"], ["identity", "amplification", "linear"], torch.tensor(hist)).to(dev)
    opt = mma_amd.FusedAdam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    state = {"i": 0}

    def step():
        b = batches[state["i"] % n_batches]
        state["i"] += 1
        b["ei"] = b["ei"].clone()                                  # a new tensor, as a data loader hands over: no cached graph plan
        opt.zero_grad(set_to_none=True)
        out = net(b["x"], b["ei"], b["ea"], b["batch"])
        loss = mma_amd.fused_l1_loss(out.squeeze(-1), b["y"])
        loss.backward()
        opt.step()
        return loss
    first = [step().item() for _ in range(3)]
    ms = wall_ms(step, reps)
    last = step().item()
    E_mean = float(np.mean([b["E"] for b in batches]))
    res = {"config": tag, "graphs_per_batch": n_graphs, "nodes": int(np.mean([b["N"] for b in batches])), "edges": int(E_mean),
           "conv_layers": 4, "ms_per_step_eager": ms, "graphs_per_s": n_graphs / ms * 1e3, "edges_per_s_eager": 4 * E_mean / ms * 1e3,
           "loss_first_steps": first, "loss_after": last,
           "note": "whole Net training step (4 MMAConv layer calls): edges/s counts E x 4 layer calls per step"}
    if graphed:
        # the same training step as ONE hipGraph over padded static-shape buffers (mma_amd.GraphedNetStep): a fresh batch is copied into
        # the bucket every step and its CSR is built inside the graph
        n_pad = -(-(max(b["N"] for b in batches) + 1) // 64) * 64
        e_pad = -(-max(b["E"] for b in batches) // 128) * 128
        torch.manual_seed(0)
        net2 = Net(["min", "max"], ["identity", "amplification", "linear"], torch.tensor(hist)).to(dev)
        opt2 = mma_amd.FusedAdam([q for q in net2.parameters() if q.requires_grad], lr=1e-3)
        gstep = mma_amd.GraphedNetStep(net2, opt2, n_graphs, n_pad, e_pad, dev)
        st2 = {"i": 0}

        def gs():
            b = batches[st2["i"] % n_batches]
            st2["i"] += 1
            return gstep(b["x"], b["ei"], b["ea"], b["batch"], b["y"])
        l0 = [gs().item() for _ in range(3)]
        gms = wall_ms(gs, max(reps, 30))
        res.update(ms_per_step_hipgraph=gms, graphs_per_s_hipgraph=n_graphs / gms * 1e3, edges_per_s_hipgraph=4 * E_mean / gms * 1e3,
                   bucket={"n_pad": n_pad, "e_pad": e_pad}, loss_first_steps_hipgraph=l0, loss_after_hipgraph=gs().item(),
                   speedup_over_eager=ms / gms)
    return res


def c5_shard_config(dev, reps=3):
    """BASELINE configs[4] ("8 M nodes / 128 M edges, feat=256, K=8 aggregators + all scalers, 8 GPUs") at its PER-GPU shard
    shape on this one GPU: R-MAT 2^20 nodes / ~16.4 M directed edges, H=256, K=8, the S=5 true-degree compounding scalers of
    mma_conv.py:181-196 (strict_reference=False), p=0.5; layer fwd+bwd."""
    import mma_amd
    from mma_amd import functional as Fn
    names = ["sum", "mean", "max", "min", "sum2", "mean2", "max2", "min2"]
    H, C = 256, 16
    rowptr, col = rmat_graph(20, 8_000_000, seed=42)
    N, E, K = len(rowptr) - 1, int(rowptr[-1]), len(names)
    graph = mma_amd.NCGraph(rowptr, col, dev)
    layer = make_layer(mma_amd, graph, H, C, names, 0.5, dev, strict_reference=False, compound_scalers=True,
                       scalers=["identity", "amplification", "attenuation", "linear", "inverse_linear"])
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr))
    adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, dev)
    x = torch.from_numpy(feature_rows(0, N, H, 42)).to(dev).requires_grad_(True)
    cot = torch.from_numpy(feature_rows(0, N, C, 43, relu=False)).to(dev)

    def step():
        x.grad = None
        for prm in layer.owned:
            prm.grad = None
        layer(x, adj).backward(cot)
    step()
    prev = Fn.TIMER
    Fn.TIMER = t = KernelTimer(); t.enabled = True
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    spans = t.summary(); Fn.TIMER = prev
    n_sel = 4 if Fn.SHARED_GRAD_BWD else None
    return {"config": "C5 per-GPU shard shape: R-MAT 2^20 nodes / %d directed edges, feat=256, K=8 [%s], S=5 true-degree scalers, p=0.5" % (
                E, ",".join(names)), "nodes": N, "edges": E, "hidden": H, "K": K, "ms_per_step": ms, "edges_per_s": E / ms * 1e3,
            "kernels": _kernel_table(spans, reps, algorithmic_bytes(N, E, H, K, n_sel), t.work, ms),
            "note": "one rank's share of configs[4] without the halo; the 8-GPU run itself is the driver's"}


def extra_configs(dev):
    out = {}
    for key, fn in (("C5shard", lambda: c5_shard_config(dev)), ("C1", lambda: nc_config("C1: Cora structure, H=64, mean,mean2, p=0.75", "cora_h64", 64, ["mean", "mean2"], 7, 0.75, dev)),
                    ("C3", lambda: nc_config("C3: Pubmed structure, H=16, min,min2,min3,min4, p=0.5", "pubmed_h16", 16,
                                             ["min", "min2", "min3", "min4"], 3, 0.5, dev)),
                    ("C1model", lambda: nc_model_config("C1 model: train.py epoch on Cora structure (1433 features, 1.3 % dense), hidden 64, mean,mean2, "
                                                        "dropout 0.75", "cora_h64", 1433, 0.0127, 64, ["mean", "mean2"], 7, 0.75, 140, dev)),
                    ("C3model", lambda: nc_model_config("C3 model: train.py epoch on Pubmed structure (500 features, 10 % dense), hidden 16, "
                                                        "min,min2,min3,min4, dropout 0.5", "pubmed_h16", 500, 0.10, 16, ["min", "min2", "min3", "min4"], 3,
                                                        0.5, 60, dev)),
                    ("C2", lambda: gr_config("C2: ZINC-like batch of 64 molecules, MMAConv T=5 F=75 min,max x id,amp,lin", 64, dev)),
                    ("C2net", lambda: gr_model_config("C2 model: Net (mma.py:63-127) training step, batch 64 (mma.py:52-54 hard-codes 64)", 64, dev, reps=20,
                                                      graphed=True)),
                    ("C2net128", lambda: gr_model_config("C2 model at batch 128 (not a reference setting; round-2 entry kept for comparison)", 128, dev,
                                                         reps=20)),
                    ("C2Lnet", lambda: gr_model_config("C2L model: the same training step on 10 000 molecules per batch", 10000, dev, reps=5)),
                    ("C2L", lambda: gr_config("C2L: the same layer on a 10 000-molecule batch", 10000, dev, reps=5)),
                    ("C2Lcat", lambda: gr_config("C2L with ZINC's real edge features: 4 bond types through an embedding table "
                                                 "(mma_amd.CategoricalEdges: the kernels read a 4-row table + 1 byte per edge)", 10000, dev, reps=5,
                                                 categorical=True))):
        try:
            out[key] = fn()
        except Exception as e:      # a secondary entry must never take the headline line down with it
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


# ---- shared graph for the ranks of one node -----------------------------------------------------------------
def shared_graph(args, rank, world, barrier):
    """Rank 0 generates the R-MAT graph once and shares the CSR through /dev/shm; the others map it read-only (8 ranks
    generating a scale-23 graph each would need 8x the host memory and time)."""
    if world == 1:
        return rmat_graph(args.scale, args.edges, seed=42)
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else __import__("tempfile").gettempdir()
    tag = "%s/mma_bench_%d_%s_s%d_e%d" % (base, os.getuid(), os.environ.get("MASTER_PORT", "0"), args.scale, args.edges)
    if rank == 0:
        rowptr, col = rmat_graph(args.scale, args.edges, seed=42)
        np.save(tag + "_rowptr.npy", rowptr)
        np.save(tag + "_col.npy", col)
    barrier()
    rowptr = np.load(tag + "_rowptr.npy")
    col = np.load(tag + "_col.npy", mmap_mode="r")
    return rowptr, col, tag


def device_identity(dev):
    """What tells two GPUs apart in the N > 1 line: index, name, and the uuid / PCI address where this torch exposes them."""
    pr = torch.cuda.get_device_properties(dev)
    d = {"index": dev.index, "name": pr.name}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(pr, k, None)
        if v is not None:
            d[k] = str(v)
    d["key"] = d.get("uuid") or "%s:%s:%s" % (d.get("pci_domain_id"), d.get("pci_bus_id"), d.get("pci_device_id"))
    return d


def verify_sharded(rank, world, dev, backend, H, C, names, p, extra_kw, fatal=True):
    """`--verify` (default for N > 1): the sharded layer against the unsharded one on a 2^14-node R-MAT, BEFORE anything is timed -
    the first run on real multi-GPU hardware must say by itself whether its numbers mean anything.  Every rank runs its shard
    (forward, backward, gradient all-reduce) with a fixed dropout seed (the hash is keyed by GLOBAL edge ids, so the shards draw the
    bits of the whole graph); rank 0 also runs mma_amd.MMA on the whole graph with the same parameters and compares: output rows,
    dL/dx rows, every parameter gradient.  Bars as tests/sharded_worker.py: 1e-5 + 1e-5 |ref| element-wise, long signed sums with
    atol = max(1e-5, 1e-6 max|ref|).  Returns a dict for the JSON line; on a mismatch: SystemExit(3) on every rank when `fatal` (--verify
    given), else the dict says ok = false and the run goes on."""
    import torch.distributed as dist
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.sharded import ShardedMMA
    from mma_amd import dense
    rowptr, col = rmat_graph(14, 80_000, seed=7)
    N, E = len(rowptr) - 1, int(rowptr[-1])
    # The SAME work-item plan and the SAME GEMM kernels on both sides (tests/sharded_worker.py::check_gpu): a segment then takes the same
    # summation order in the shard and in the whole graph and the logits are the same bits, so no max / min mask flips its selection on
    # a near tie - a legitimate discontinuity of the reference's max(x_i, s), but one flip moves whole rows of dL/dx.  (A shard of a
    # 2^14-node graph has fewer rows than the split-precision kernels' usual threshold.)
    PLAN = dict(chunk=16, group_below=8, t_group_below=8)
    min_rows, dense._MIN_ROWS_X3 = dense._MIN_ROWS_X3, 1
    try:
        sh = ShardedMMA.build(rowptr, col, rank, world, dev, H, C, names, p, seed=123, **PLAN, **extra_kw)
        drop = Fn.DropoutSpec(p, seed=0xC0FFEE)
        sh.drop_override = drop
        x_all = feature_rows(0, N, H, 5)
        cot_all = feature_rows(0, N, C, 6, relu=False)
        x = torch.from_numpy(x_all[sh.lo:sh.hi]).to(dev).requires_grad_(True)
        out = sh(x)
        out.backward(torch.from_numpy(cot_all[sh.lo:sh.hi]).to(dev))
        sh.allreduce_grads()
        mine = {"lo": sh.lo, "hi": sh.hi, "out": out.detach().cpu(), "gx": (x.grad if x.grad is not None else torch.zeros_like(x)).cpu()}
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        res = {"graph": "R-MAT 2^14 nodes / %d directed edges, %d ranks" % (E, world), "ok": True,
               "bars": "rows: every element within 1e-5 + 1e-5|ref| + max(1e-5, 1e-6 max|ref|) - all rows of `out`, all but <= 0.1 % of the rows "
                       "of dL/dx (a selection flip on a near tie is a discontinuity of the reference's own max(x_i, s)); parameter "
                       "gradients: ||got - ref|| <= 1e-4 ||ref||"}
        if rank == 0:
            graph = mma_amd.NCGraph(rowptr, col, dev, H=H, **PLAN)
            layer = make_layer(mma_amd, graph, H, C, names, p, dev, **extra_kw)
            with torch.no_grad():
                layer.weight.copy_(sh.weight); layer.bias.copy_(sh.bias)
                for n in names:
                    getattr(layer, "mask_" + n).copy_(sh.masks[n])
            layer.drop_override = drop
            dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr))
            adj = mma_amd.graph.SpmmGraph(dst, col, None, N, N, dev)
            xf = torch.from_numpy(x_all).to(dev).requires_grad_(True)
            ref = layer(xf, adj)
            ref.backward(torch.from_numpy(cot_all).to(dev))
            order = sorted(parts, key=lambda q: q["lo"])
            worst = {}
            for what, a, b, allowed in (("out", torch.cat([q["out"] for q in order]), ref.detach().cpu(), 0.0),
                                        ("dL/dx", torch.cat([q["gx"] for q in order]), xf.grad.cpu(), 1e-3)):
                a, b = a.double(), b.double()
                err = (a - b).abs()
                atol = max(1e-5, 1e-6 * b.abs().max().item())
                rows_out = int((err > atol + 1e-5 * b.abs()).any(1).sum())
                ok = a.shape == b.shape and rows_out <= int(allowed * b.shape[0])
                worst[what] = {"max_err": err.max().item(), "max_ref": b.abs().max().item(), "rows_outside": rows_out, "rows": b.shape[0], "ok": ok}
                res["ok"] = res["ok"] and ok
            pg = [("dL/dweight", sh.weight.grad, layer.weight.grad), ("dL/dbias", sh.bias.grad, layer.bias.grad)]
            pg += [("dL/dmask_" + n, sh.masks[n].grad, getattr(layer, "mask_" + n).grad) for n in names]
            for what, a, b in pg:
                a, b = a.double().cpu(), b.double().cpu()
                rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
                worst[what] = {"rel_err": rel, "ok": rel <= 1e-4}
                res["ok"] = res["ok"] and rel <= 1e-4
            res["checks"] = worst
    finally:
        dense._MIN_ROWS_X3 = min_rows
    flag = torch.tensor([1.0 if res["ok"] else 0.0])
    if backend == "nccl":
        flag = flag.to(dev)
    dist.broadcast(flag, 0)
    if flag.item() != 1.0:
        if rank == 0:
            sys.stderr.write("bench.py --verify: the sharded layer does NOT match the unsharded one%s\n" % (
                " - nothing was timed" if fatal else " - timing goes on, the line carries verify.ok = false (pass --verify to make this fatal)"))
            if fatal:
                print(json.dumps({"verify": res}), flush=True)
        dist.barrier()
        if fatal:
            sys.exit(3)
        res["ok"] = False
    return res



# ---- output: ONE compact strict-JSON metric line on stdout; the verbose record elsewhere -------------------------------
COMPACT_LIMIT = 4096


def _finite(o):
    """json.dumps(allow_nan=False) raises on NaN / Infinity: replace them with None so the line always strict-parses."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return _finite(o.item())
    return o


def _clip(s, n=200):
    return s if s is None or len(s) <= n else s[:n - 3] + "..."


def _sig(o, digits=6):
    """Floats to 6 significant digits: the compact line is for parsing, the detail record keeps full precision."""
    if isinstance(o, float):
        return float("%.*g" % (digits, o))
    if isinstance(o, dict):
        return {k: _sig(v, digits) for k, v in o.items()}
    if isinstance(o, list):
        return [_sig(v, digits) for v in o]
    return o


def emit(line, detail):
    """rank 0: the verbose record (`kernels`, `extra`, `plan_build`, `ranks`, long notes) goes to stderr as ONE `{"detail": ...}` line and to
    $MMA_BENCH_DETAIL (default gpurun_out/bench_detail.json); stdout gets exactly ONE line: the compact metric record, strict JSON
    (no NaN / Infinity), < 4096 bytes (round-4 VERDICT item 1: a 25 KB line did not parse at the driver)."""
    detail = _finite(dict(line, **detail))
    blob = json.dumps({"detail": detail}, allow_nan=False)
    sys.stderr.write(blob + "\n"); sys.stderr.flush()
    path = os.environ.get("MMA_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(blob + "\n")
        line["detail"] = os.path.relpath(path, ROOT) + " (+ stderr)"
    except OSError:
        line["detail"] = "stderr"
    out = json.dumps(_sig(_finite(line)), allow_nan=False)
    if len(out) >= COMPACT_LIMIT:           # never let an optional field take the metric line down
        for k in ("extra_summary", "kernels_ms", "plan_build"):
            line.pop(k, None)
            out = json.dumps(_sig(_finite(line)), allow_nan=False)
            if len(out) < COMPACT_LIMIT:
                break
    assert len(out) < COMPACT_LIMIT, len(out)
    print(out, flush=True)


def _compact_roofline(roof):
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes", "frac_rocprof")
    r = {k: roof.get(k) for k in keep}
    r["traffic_source"] = _clip(r["traffic_source"], 160)
    return r


def _compact_cpu(cpu):
    if cpu is None:
        return None
    c = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "form", "sample") if k in cpu}
    c["sample"] = _clip(c.get("sample"))
    if cpu.get("loop"):
        c["loop"] = [{"config": l["config"], "value": l["value"], "unit": l["unit"], "form": l["form"]} for l in cpu["loop"][:2]]
    return c


def _extra_summary(extra):
    """A few figures per secondary config for the compact line (the tables are in the detail record)."""
    if not extra:
        return None
    out = {}
    for k, v in extra.items():
        if "error" in v:
            out[k] = {"error": _clip(v["error"], 80)}
            continue
        out[k] = {a: v[b] for a, b in (("ms", "ms_per_step"), ("ms", "ms_per_step_eager"), ("ms_graph", "ms_per_step_hipgraph"),
                                       ("ms", "ms_per_epoch_eager_fused"), ("ms_graph", "ms_per_epoch_hipgraph_fused")) if b in v}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=["c4", "c5", "c2l"], help="c4: MMA layer on the R-MAT graph (headline); "
                    "c5: BASELINE configs[4] (scale 23, 128 M directed edges, feat 256, K=8, S=5 true-degree scalers; needs --gpus >= 4); "
                    "c2l: MMAConv on a 10 000-molecule ZINC-like batch (roofline = the fused GR kernel)")
    ap.add_argument("--scale", type=int, default=C4_DEFAULTS["scale"], help="R-MAT scale (2^scale nodes)")
    ap.add_argument("--edges", type=int, default=C4_DEFAULTS["edges"], help="undirected R-MAT edges before symmetrisation")
    ap.add_argument("--hidden", type=int, default=C4_DEFAULTS["hidden"])
    ap.add_argument("--nclass", type=int, default=C4_DEFAULTS["nclass"])
    ap.add_argument("--aggregators", type=str, default=C4_DEFAULTS["aggregators"])
    ap.add_argument("--dropout", type=float, default=C4_DEFAULTS["dropout"])
    ap.add_argument("--true-degree-scalers", action="store_true", help="C5 option: the S=5 true-degree scalers of "
                    "mma_conv.py:181-196 on the NC aggregates (strict_reference=False) instead of the reference's degenerate three")
    ap.add_argument("--molecules", type=int, default=10000, help="c2l: molecules in the batch")
    ap.add_argument("--force-sharded", action="store_true", help="use the sharded (RCCL) path even at world size 1")
    ap.add_argument("--verify", dest="verify", action="store_true", default=None, help="before timing, check the sharded layer against the "
                    "unsharded one on a 2^14-node R-MAT (rank 0 computes both; forward, dL/dx, parameter gradients) and exit non-zero on "
                    "mismatch: the default whenever the sharded path runs on more than one rank")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' (halo staged through the host) lets "
                    "several ranks share one GPU to rehearse the N>1 path on a 1-GPU box")
    ap.add_argument("--cpu-sample", type=int, default=150000, help="target nodes in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations (C1, C3, C2, C2L) and the loop baseline")
    ap.add_argument("--rendezvous-only", action="store_true", help="(tests) ranks meet, all-reduce their rank and leave: "
                    "checks the launch path without a GPU")
    args = ap.parse_args()
    if os.environ.get("MMA_LIB_OVERRIDE"):
        sys.exit("bench.py: MMA_LIB_OVERRIDE is set (%s): a measurement / ablation library is not the product - nothing is timed on it" %
                 os.environ["MMA_LIB_OVERRIDE"])
    if args.workload == "c5":
        for k, v in C5_PRESET.items():
            setattr(args, k, v)
        if args.gpus < C5_MIN_GPUS:
            sys.exit("--workload c5 is BASELINE configs[4]: 8 M nodes / 128 M edges, feat 256, K=8 - its P, Q, T tables alone are > 200 GB, "
                     "so it runs sharded: `python bench.py --workload c5 --gpus 8` (at least %d GPUs).  Its per-GPU shard shape on one "
                     "GPU is `extra.C5shard` of the default run; one rank of eight WITH its halo is tools/shard_sim.py --c5." % C5_MIN_GPUS)

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    import torch.distributed as dist

    if args.rendezvous_only:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "world": world, "rank_sum": t.item()}), flush=True)
        dist.destroy_process_group()
        return

    if args.backend == "gloo":
        local_rank = 0                      # rehearsal: all ranks on cuda:0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        if "RANK" not in os.environ:      # --force-sharded without a launcher
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "c2l":       # molecule batches are independent: N > 1 = data-parallel replicas, no halo
        run_c2l(args, dev, rank, world, barrier)
        if sharded:
            dist.destroy_process_group()
        return

    import mma_amd
    from mma_amd import functional as Fn

    names = args.aggregators.split(",")
    K, H, C = len(names), args.hidden, args.nclass
    g = shared_graph(args, rank, world, barrier)
    rowptr, col, shm_tag = g if len(g) == 3 else (g[0], g[1], None)
    N, E = len(rowptr) - 1, int(rowptr[-1])

    timer = KernelTimer()
    Fn.TIMER = timer
    torch.manual_seed(42)
    extra_kw = dict(strict_reference=False, scalers=["identity", "amplification", "attenuation", "linear", "inverse_linear"],
                    compound_scalers=True) if args.true_degree_scalers else {}

    plan_build = None
    if not sharded:
        # the graph plan (CSR, transposed CSR, work items, hub lists; the tail SpMM's plan): built on the DEVICE (K6 radix sorts + scans,
        # NCGraph.from_device_csr) - the host numpy builder of round 2 is timed beside it
        rp_d, cl_d = torch.from_numpy(np.ascontiguousarray(rowptr)).to(dev), torch.from_numpy(np.ascontiguousarray(col)).to(dev)
        mma_amd.NCGraph.from_device_csr(rp_d, cl_d, H=H)                       # warm-up (allocator, rocPRIM temporary sizes)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        graph = mma_amd.NCGraph.from_device_csr(rp_d, cl_d, H=H)
        adj = mma_amd.graph.SpmmGraph.from_device_csr(rp_d, cl_d)
        torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
        plan_build = {"device_s": t_dev, "what": "NCGraph + SpmmGraph plans of the whole graph from a device-resident CSR"}
        if args.cpu_sample:
            t0 = time.perf_counter()
            mma_amd.NCGraph(rowptr, col, dev, H=H)
            dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr))
            mma_amd.graph.SpmmGraph(dst, col, None, N, N, dev)
            torch.cuda.synchronize(); plan_build["numpy_s"] = time.perf_counter() - t0
        layer = make_layer(mma_amd, graph, H, C, names, args.dropout, dev, **extra_kw)
        x = torch.from_numpy(feature_rows(0, N, H, 42)).to(dev).requires_grad_(True)
        cot = torch.from_numpy(feature_rows(0, N, C, 43, relu=False)).to(dev)

        def step():
            x.grad = None
            for prm in layer.owned:
                prm.grad = None
            out = layer(x, adj)
            out.backward(cot)
        local_edges, n_local = E, N
    else:
        from mma_amd.sharded import EXCHANGE_LOG, ShardedMMA
        verify = None
        if args.verify or (args.verify is None and world > 1):
            # asked for (--verify): a mismatch ends the run (exit 3).  By default at N > 1: the check runs and its verdict travels in the
            # line, but a mismatch does not throw the measurement away - the first multi-GPU run happens where nobody can re-run it
            verify = verify_sharded(rank, world, dev, args.backend, H, C, names, args.dropout, extra_kw, fatal=bool(args.verify))
        sh = ShardedMMA.build(rowptr, col, rank, world, dev, H, C, names, args.dropout, **extra_kw)
        x = torch.from_numpy(feature_rows(sh.lo, sh.hi, H, 42)).to(dev).requires_grad_(True)
        cot = torch.from_numpy(feature_rows(sh.lo, sh.hi, C, 43, relu=False)).to(dev)

        def step():
            x.grad = None
            for prm in sh.owned:
                prm.grad = None
            out = sh(x)
            out.backward(cot)
            sh.allreduce_grads()
        local_edges, n_local = sh.local_edges, sh.hi - sh.lo
        barrier()
        if rank == 0 and shm_tag:
            for suf in ("_rowptr.npy", "_col.npy"):
                os.unlink(shm_tag + suf)

    for _ in range(args.warmup):
        step()
    barrier()
    timer.enabled = True
    if sharded:
        EXCHANGE_LOG.reset()             # byte / call counts only: the headline runs the product's own synchronisation (timed = False)
    if sharded:
        # this rank's own clock, before it waits for the others: two events on the compute stream, read AFTER the closing barrier, so the
        # timed window holds no extra host synchronisation (round-4 ADVICE)
        own_start, own_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        own_start.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if sharded:
        own_end.record()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    ranks = None
    dt_own = dt
    if sharded:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        # exchange_ms: a SEPARATE short instrumented pass (side-stream work.wait() + two events per exchange), never the headline's path
        n_inst = max(1, min(args.steps, 3))
        counts = (EXCHANGE_LOG.bytes_sent, EXCHANGE_LOG.bytes_received, EXCHANGE_LOG.calls)
        EXCHANGE_LOG.timed = True
        for _ in range(n_inst):
            step()
        barrier()
        dt_own = own_start.elapsed_time(own_end) * 1e-3
        EXCHANGE_LOG.timed = False
        exchange_ms_inst = EXCHANGE_LOG.exchange_ms() / n_inst
        EXCHANGE_LOG.bytes_sent, EXCHANGE_LOG.bytes_received, EXCHANGE_LOG.calls = counts

    spans = timer.summary()
    if sharded:
        # one record per rank, gathered with ONE all_gather_object: who ran where, on what, what it exchanged and where its time went
        # (round-3 VERDICT item 3: the first multi-GPU run must explain itself)
        sp_ms = {k: v[1] / args.steps for k, v in spans.items()}
        plan = sh.plan
        n_send = int(plan.send_counts.sum())
        me = {"rank": rank, "device": device_identity(dev), "own_rows": int(plan.n_own), "halo_rows": int(plan.n_halo), "rows_sent": n_send,
              "local_edges": int(sh.local_edges), "peers_sent_to": int((plan.send_counts > 0).sum()), "peers_received_from": int((plan.recv_counts > 0).sum()),
              "max_bytes_to_one_peer_fwd_x": int(plan.send_counts.max() if world > 1 else 0) * H * 4,
              "bytes_sent": EXCHANGE_LOG.bytes_sent // max(args.steps, 1), "bytes_received": EXCHANGE_LOG.bytes_received // max(args.steps, 1),
              "exchanges_per_step": EXCHANGE_LOG.calls // max(args.steps, 1),
              "ms_per_step": dt_own / args.steps * 1e3, "exchange_ms_instrumented": exchange_ms_inst,
              "halo_wait_ms": sp_ms.get("halo_wait", 0.0), "halo_pack_ms": sp_ms.get("halo_pack", 0.0), "halo_unpack_ms": sp_ms.get("halo_unpack", 0.0),
              "nc_fused_fwd_ms": sp_ms.get("nc_fused_fwd", 0.0), "nc_fused_bwd_ms": sp_ms.get("nc_fused_bwd", 0.0),
              "gemm_ms": sum(v for k, v in sp_ms.items() if k.startswith("gemm_x3") or k == "lib_mm"),
              "note": "bytes per step over all four exchanges (x rows and tail rows forward, their gradients back); exchange_ms_instrumented = "
                      "sum over a step's exchanges of (collective enqueued -> last byte received) from events on a side stream, taken in a "
                      "SEPARATE short pass after the timed steps (the headline runs the product's own work.wait() path); halo_wait_ms = "
                      "HIP-event time of the compute stream's waits in the timed steps (what was NOT hidden behind compute)"}
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        keys = {r["device"]["key"] for r in ranks}
        if args.backend == "nccl" and world > 1 and len(keys) != world:
            if rank == 0:
                sys.stderr.write("bench.py: %d ranks but only %d distinct devices (%s): not a multi-GPU measurement\n" % (world, len(keys), sorted(keys)))
            sys.exit(4)
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = E * args.steps / dt
        n_sel = sum(1 for a in names if a.rstrip("234") in ("max", "min", "softmax", "softmin")) if Fn.SHARED_GRAD_BWD else None
        ab = algorithmic_bytes(n_local, local_edges, H, K, n_sel)
        kernels = _kernel_table(spans, args.steps, ab, timer.work, ms_per_step)
        dom = max((n for n in kernels if n in ab), key=lambda n: kernels[n]["avg_ms"])
        traffic, src, rocprof_ms = (None, None, None) if sharded else pmc_traffic(dom, {"nodes": N, "edges": E, "hidden": H, "K": K})
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["achieved_GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": _traffic_source(traffic, src),
                # WHICH byte count (round-3 VERDICT item 7): the bytes THIS kernel has to move with zero credit for cache reuse of gathered
                # rows - bench.py::algorithmic_bytes, derived in DESIGN.md 3 (K1 = SURVEY 8d's B_fwd; K2b walks the transposed CSR and
                # writes every source row once, so it is NOT 8d's B_bwd, which assumed a re-gather + per-edge scatter: that formula gives
                # 63.2 GB at C4 against the 41.35 GB this kernel's design moves)
                "algorithmic_bytes": ab[dom], "bytes_definition": "bench.py::algorithmic_bytes (DESIGN.md 3), not SURVEY 8d's B_bwd",
                "duration": "HIP events around the C-ABI call (its 2-3 launches and the gaps between them), avg over the timed steps",
                "frac_rocprof": (ab[dom] / (rocprof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rocprof_ms else None,
                "frac_rocprof_note": "the same bytes / the kernels' own durations in the committed --kernel-trace --stats pass (no launch gaps)"}
        is_c4 = all(getattr(args, k) == v for k, v in C4_DEFAULTS.items()) and not args.true_degree_scalers
        is_c5 = all(getattr(args, k) == v for k, v in C5_PRESET.items())
        label = "C4" if is_c4 else ("C5" if is_c5 else "custom (not a BASELINE config)")
        cpu = None
        extra = None
        if not sharded:
            if args.cpu_sample:
                cpu = cpu_baseline(rowptr, col, H, names, "new_sigmoid", args.dropout, min(args.cpu_sample, N), 42,
                                   "%s R-MAT" % label.split()[0])
            if not args.no_extra:
                if cpu is not None:
                    cpu["loop"] = [cpu_loop_baseline("C1", "cora_h64", 64, ["mean", "mean2"], 0.75, 2708),
                                   cpu_loop_baseline("C3", "pubmed_h16", 16, ["min", "min2", "min3", "min4"], 0.5, 2000)]
                Fn.TIMER = None
                extra = extra_configs(dev)
        line = {
            "metric": "aggregated edges/sec (fwd+bwd) MultiMaskConv", "value": value, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic R-MAT power-law graph (scale %d), %d nodes / %d directed edges, feat=%d, "
                                   "K=%d masks [%s], nclass=%d, mask dropout p=%g, %sMMA layer fwd+bwd" % (
                                       label, args.scale, N, E, H, K, ",".join(names), C, args.dropout,
                                       "S=5 true-degree scalers, " if args.true_degree_scalers else ""),
                       "nodes": N, "edges": E, "hidden": H, "K": K, "nclass": C,
                       "parallelism": "1-D node shard x%d, RCCL all-to-all halo" % world if world > 1 else "single GPU"},
            "masked_edges_per_s": value * K,
            "roofline": _compact_roofline(roof), "cpu_baseline": _compact_cpu(cpu),
            "kernels_ms": {k: v["avg_ms"] for k, v in kernels.items()}, "plan_build": plan_build and {k: v for k, v in plan_build.items() if k != "what"},
            "extra_summary": _extra_summary(extra),
        }
        detail = {"roofline": roof, "kernels": kernels, "cpu_baseline": cpu, "plan_build": plan_build, "extra": extra}
        if sharded:
            per = [r["ms_per_step"] for r in ranks]
            line.update(ms_per_step_rank_max=max(per), ms_per_step_rank_mean=sum(per) / len(per), distinct_devices=len(keys),
                        verify={"ok": verify["ok"]} if verify else None,
                        multi_gpu_note=None if (args.backend == "nccl" and world > 1) else
                        "NOT a multi-GPU measurement: %s" % ("gloo rehearsal, all ranks on one GPU, halo staged through the host"
                                                             if args.backend == "gloo" else "one rank"))
            detail.update(ranks=ranks, verify=verify, timing="ms_per_step = max over ranks of (barrier .. K steps .. barrier) / K (the contract); "
                          "ms_per_step_rank_max / _mean: each rank's own compute-stream clock over its K steps (HIP events), before the closing barrier")
        emit(line, detail)
    if sharded:
        dist.destroy_process_group()


def run_c2l(args, dev, rank=0, world=1, barrier=None):
    """`--workload c2l`: MMAConv fwd+bwd on a molecule batch; roofline = the fused GR kernels (K3/K4).  With N > 1 ranks every
    rank takes its own batch of `--molecules` molecules (weak scaling) and the parameter gradients are averaged in one
    bucketed all-reduce per step (mma.py:150-160 trains on independent mini-batches: replicas, no halo)."""
    import torch.distributed as dist
    from mma_amd import functional as Fn
    from mma_amd.sharded import allreduce_grads
    conv, fwd_bwd, N, E = gr_setup(args.molecules, dev, seed=rank)
    params = [q for q in conv.parameters() if q.requires_grad]
    if world > 1:
        with torch.no_grad():                                   # replicas start from rank 0's parameters
            for q in params:
                if args.backend == "nccl":
                    dist.broadcast(q.data, 0)
                else:
                    q.data.copy_(_bcast_cpu(q, dist))

    def step():
        fwd_bwd()
        if world > 1:
            allreduce_grads(params, average=True)
    barrier = barrier or torch.cuda.synchronize
    timer = KernelTimer()
    Fn.TIMER = timer
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
    fwd_bwd.check()
    tot = torch.tensor([float(E), float(N)], dtype=torch.float64)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        tot = tot.to(t.device)
        dist.all_reduce(tot)
    if rank != 0:
        return
    E_all, N_all = int(tot[0].item()), int(tot[1].item())
    T, F, K, S = 5, 75, 2, 3
    from mma_amd import mma_conv as _mc
    # with the degree scalers factored into the post-NN (round 3) the fused kernels move the K UNSCALED aggregates: S = 1 in their bytes
    ab = gr_algorithmic_bytes(N, E, T, F, K, 1 if _mc.FACTOR_SCALERS else S)
    kernels = _kernel_table(timer.summary(), args.steps, ab, timer.work, dt / args.steps * 1e3)
    roofs = {}
    for n in ("gr_fused_fwd", "gr_fused_bwd"):
        traffic, src, rocprof_ms = pmc_traffic(n, {"workload": "c2l", "nodes": N, "edges": E}) if world == 1 else (None, None, None)
        roofs[n] = {"bound": "hbm", "kernel": n, "achieved": kernels[n]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kernels[n]["achieved_GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": _traffic_source(traffic, src), "algorithmic_bytes": ab[n],
                    "bytes_definition": "bench.py::gr_algorithmic_bytes = SURVEY 8d's GR formulas on the UNSCALED aggregates (S = 1)",
                    "frac_rocprof": (ab[n] / (rocprof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rocprof_ms else None}
    dom = max(roofs, key=lambda n: kernels[n]["avg_ms"])
    cpu = None
    if args.cpu_sample and world == 1:
        cpu = gr_cpu_baseline(200)
    line = {"metric": "aggregated edges/sec (fwd+bwd) MultiMaskConv", "value": E_all * args.steps / dt, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2L: ZINC-like batch of %d molecules per GPU (%d nodes / %d directed edges over all ranks), MMAConv "
                                   "75->75, towers=5, edge_dim=50, aggregators min,max, scalers identity,amplification,linear, "
                                   "dropout 0.5, layer fwd+bwd%s" % (args.molecules, N_all, E_all,
                                                                     ", gradients averaged over the replicas" if world > 1 else ""),
                       "nodes": N_all, "edges": E_all, "towers": T, "F": F, "K": K, "S": S,
                       "parallelism": "data-parallel replicas x%d, one bucketed all-reduce of the gradients per step" % world
                       if world > 1 else "single GPU"},
            "roofline": _compact_roofline(roofs[dom]), "roofline_other": _compact_roofline(roofs[[n for n in roofs if n != dom][0]]),
            "cpu_baseline": _compact_cpu(cpu), "kernels_ms": {k: v["avg_ms"] for k, v in kernels.items()}}
    emit(line, {"roofline": roofs[dom], "roofline_other": roofs[[n for n in roofs if n != dom][0]], "kernels": kernels, "cpu_baseline": cpu})

def _bcast_cpu(q, dist):
    """gloo rehearsal (all ranks on one GPU): broadcast through a host copy."""
    h = q.detach().cpu()
    dist.broadcast(h, 0)
    return h.to(q.device)


def gr_cpu_baseline(n_graphs):
    """The GR CPU oracle (oracle/gr_oracle.conv_forward, pinned to the reference module run over third-party stand-ins - see its header) fwd+bwd on a small batch."""
    from oracle import gr_oracle as G
    threads = _threads()
    rng = np.random.default_rng(1)
    ei, N = molecule_batch(rng, n_graphs)
    E = ei.shape[1]
    T, F = 5, 75
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g) * 0.1
    prm = {"enc_w": r(F, 50), "enc_b": r(F), "pre_w": [r(F, 3 * F) for _ in range(T)], "pre_b": [r(F) for _ in range(T)],
           "post_w": [r(15, 7 * F) for _ in range(T)], "post_b": [r(15) for _ in range(T)], "lin_w": r(75, 75), "lin_b": r(75)}
    x = torch.randn(N, F, generator=g).requires_grad_(True)
    ea = torch.randn(E, 50, generator=g)
    keep = (torch.rand(E, T, F, generator=g) >= 0.5).float()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 10.0:
        out = G.conv_forward(x, torch.from_numpy(ei), ea, prm, ["min", "max"], ["identity", "amplification", "linear"],
                             {"lin": 2.0, "log": 1.0}, T, False, keep, 0.5)
        out.sum().backward()
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": E * reps / dt, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": "%d-molecule batch (%d nodes / %d edges), MMAConv fwd+bwd on the torch-CPU GR oracle, %d passes in %.1f s" % (
                n_graphs, N, E, reps, dt)}


def make_layer(mma_amd, graph, H, C, names, p, dev, **kw):
    """The drop-in MMA layer with externally owned Parameters, as models.py:17-60 creates them."""
    from mma_amd.layers import _MASK_NAMES
    P = lambda *s: torch.nn.Parameter(torch.empty(*s, device=dev))
    # only the masks in use get a full (2H,H) tensor; the reference allocates all 21 (models.py:21-41)
    masks = {n: P(2 * H, H) if n in names else P(2, 1) for n in _MASK_NAMES}
    w, b = P(H, C), P(C)
    layer = mma_amd.MMA(graph, "new_sigmoid", 2, H, C, w, b, *[masks[n] for n in _MASK_NAMES], p, names, dev, **kw)
    layer.owned = [w, b] + [masks[n] for n in names]
    return layer


if __name__ == "__main__":
    main()
