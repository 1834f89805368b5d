"""autograd wrappers over the C ABI (include/mma_amd.h).  Tensors are plumbing: every FLOP and byte of the
hot path moves inside libmma_amd.so; torch only owns the memory, the stream and the autograd tape."""

import os

import torch

from . import _lib
from ._lib import call, host_codes, ptr, require_gpu, stream_ptr

# K2b form for the reduce_k (shared upstream gradient) backward.  True: dL/ds_k is rebuilt per edge from ONE packed row per
# target, [g (H floats) | 1/d | one byte per element and max/min-type mask = 2*dm/ds], written by K2a - 2.8 KB instead of
# 4 KB gathered per edge at K=4, H=128 (C4: 5.5 vs 6.7 ms with dropout, 5.2 vs 6.8 ms without; K2a 1.3 vs 1.8 ms).
# False: gather a materialised (N,K*H) gs.  Both forms are covered by the parity tests.
SHARED_GRAD_BWD = True
# Round 3: with the shared-gradient form K1 leaves the packed code rows itself (no aux rows written by K2a), and the node-level
# backward (K2a: gP = g dm/ds T, the direct term of dL/dx) runs in K2b's per-source epilogue - no K2a launch, no gxs round trip.
# False: K2a stays a launch of its own (reading the code rows).  Both forms are covered by the parity tests.
FUSE_NODE_BWD = os.environ.get("MMA_FUSE_NODE_BWD", "1") != "0"
TIMER = None   # bench.py installs an object with .span(name) -> context manager (HIP events around the calls)


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _span(name, nbytes=0, flops=0, mfma=None):
    """A timed region around one C-ABI call (bench.py installs TIMER).  The call site states what the call HAS to do - nbytes: the HBM
    bytes it must move with no credit for cache reuse (DESIGN.md 3 "algorithmic bytes"), flops: its fp32-equivalent multiply-adds x 2,
    mfma: how those run on the matrix cores ("f16x3": three fp16 piece products per fp32 product, "bf16x6": six bf16 ones, "f32":
    v_mfma_f32_16x16x4_f32 at the vector rate, None: no matrix work) - so that bench.py can put every kernel, not only the fused ones,
    against its roofline (round-3 VERDICT item 1a).  The fused NC / GR kernels are priced by bench.py's own formulas."""
    return TIMER.span(name, nbytes, flops, mfma) if TIMER is not None else _NoSpan()


KIND = {"sum": 0, "mean": 1, "max": 2, "min": 3, "softmax": 4, "softmin": 5}
ACT_SIGMOID, ACT_RAW = 0, 1
DROP_NONE, DROP_HASH, DROP_EXPLICIT = 0, 1, 2


class DropoutSpec:
    """Mask dropout of one layer call (reference: F.dropout(mask0, p), training=True always, layers.py:219; any p in [0, 1)).

    HASH mode: the kernels drop an element iff its 16-bit hash value < thr = round(65536 p) and scale survivors by 65536 / (65536 - thr):
    `p_applied` = thr / 65536 is within 2^-17 (7.7e-6) of p for every p (round 5; rounds 1-4 quantised to 1/256 and warned), exact for the
    README's 0.5 and 0.75.  `keep` (K,E,H) uint8 switches to an explicit mask (parity tests)."""

    def __init__(self, p=0.0, seed=None, keep=None, seed_tensor=None):
        self.p = float(p)
        if not (0.0 <= self.p < 1.0):
            raise ValueError("dropout probability has to be in [0, 1), but got %r" % p)
        self.keep = keep
        self.seed_tensor = seed_tensor      # (1,) int64 GPU tensor: the kernels read the seed from it (hipGraph-safe)
        if keep is not None:
            self.mode, self.thr = DROP_EXPLICIT, int(round(self.p * 65536))
            if self.thr / 65536.0 != self.p:
                raise ValueError("explicit keep masks need p = i/65536 so that 1/(1-p) is reproduced exactly")
        elif self.p == 0.0:
            self.mode, self.thr = DROP_NONE, 0
        else:
            self.mode, self.thr = DROP_HASH, min(65535, max(1, int(round(self.p * 65536))))
        self.p_applied = self.thr / 65536.0 if self.mode != DROP_NONE else 0.0
        if seed is None:
            # drawn from torch's CPU generator so torch.manual_seed() controls it; no GPU sync
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) if (self.mode == DROP_HASH and seed_tensor is None) else 0
        self.seed = seed & 0xFFFFFFFFFFFFFFFF

    def args(self):
        return self.mode, self.thr, self.seed, ptr(self.seed_tensor), ptr(self.keep)


def crow_floats(H, kinds):
    return _lib.query("mma_nc_crow_floats", H, len(kinds), host_codes(kinds))


def nc_fwd_launch(x_src, P, Q, graph, kinds, acts, drop, reduce_k, save, shared=None):
    """K1 on prepared operands -> (m or msum, T, sel, crow).  No autograd; shared by _NCFused and the sharded layer.
    shared (default: reduce_k and SHARED_GRAD_BWD): the backward will take the shared-gradient form, so the selection state is
    saved as the packed code rows `crow` (N, ldc) K2b gathers per edge instead of the (N,K*H) byte array `sel`."""
    K = len(kinds)
    S, H = x_src.shape
    N = graph.N
    shared = (reduce_k and SHARED_GRAD_BWD) if shared is None else shared
    assert x_src.dtype == torch.float32 and P.dtype == torch.float32 and Q.dtype == torch.float32
    assert S == graph.n_src and P.shape == (N, K * H) and Q.shape == (S, K * H) and 1 <= K <= 8
    assert x_src.is_contiguous() and P.stride(1) == 1 and Q.stride(1) == 1
    dev = x_src.device
    m = None if reduce_k else torch.empty((K, N, H), device=dev, dtype=torch.float32)
    msum = torch.empty((N, H), device=dev, dtype=torch.float32) if reduce_k else None
    T = torch.empty((N, K * H), device=dev, dtype=torch.float32) if save else None
    sel = torch.empty((N, K * H), device=dev, dtype=torch.uint8) if save and not shared else None
    crow = None
    if save and shared:
        ldc = crow_floats(H, kinds)
        # zeros: the padding of a row (and, for H % 4 != 0, the tail of a code word) is never written by the kernel
        crow = (torch.zeros if H % 4 else torch.empty)((N, ldc), device=dev, dtype=torch.float32)
    partial = torch.empty((graph.n_slots, 2 * K * H), device=dev, dtype=torch.float32) if graph.n_slots else None
    if drop.keep is not None:
        assert drop.keep.dtype == torch.uint8 and drop.keep.is_contiguous() and drop.keep.is_cuda and \
            tuple(drop.keep.shape) == (K, graph.E, H), "explicit keep mask must be a contiguous (K,E,H) uint8 GPU tensor"
    mode, thr, seed, seed_dev, keep = drop.args()
    with _span("nc_fused_fwd"):
        call("mma_nc_fused_fwd", ptr(x_src), x_src.stride(0), ptr(P), P.stride(0), ptr(Q), Q.stride(0),
             ptr(graph.rowptr), ptr(graph.col), ptr(graph.items), graph.items.shape[0], graph.n_wave_items,
             ptr(graph.hubs) if graph.n_slots else None, graph.hubs.shape[0], ptr(partial), graph.n_slots,
             ptr(m), ptr(msum), H, ptr(T), ptr(sel), K * H, ptr(crow), crow.stride(0) if crow is not None else 0,
             N, graph.E, H, K, host_codes(kinds), host_codes(acts),
             mode, thr, seed, seed_dev, graph.edge_base, keep, ptr(graph.sync(0)), stream_ptr())
    return (msum if reduce_k else m), T, sel, crow


def nc_bwd_node_launch(g, reduce_k, sel, crow, T, graph, kinds, H, shared, gP=None, row_max=None):
    """K2a -> (gs or None, gP (N,K*H), gxs (n_src,H) with zero halo rows).  The selection state is `crow` (shared-gradient form)
    or `sel`.  gP may be a (N,K*H) column block of a wider buffer (row pitch = its stride(0))."""
    K, N, S = len(kinds), graph.N, graph.n_src
    dev = g.device
    gs = None if shared else torch.empty((N, K * H), device=dev, dtype=torch.float32)
    if gP is None:
        gP = torch.empty((N, K * H), device=dev, dtype=torch.float32)
    gxs = torch.empty((S, H), device=dev, dtype=torch.float32)
    if S > N:  # halo rows are sources only: no target-side gradient
        gxs[N:].zero_()
    with _span("nc_bwd_node"):
        call("mma_nc_bwd_node", ptr(g), 0 if reduce_k else N * H, H, ptr(sel), ptr(T), K * H, ptr(graph.rowptr),
             ptr(gs), K * H, ptr(crow), crow.stride(0) if crow is not None else 0, ptr(gP), gP.stride(0), ptr(gxs), H, ptr(row_max),
             N, H, K, host_codes(kinds), stream_ptr())
    return gs, gP, gxs


def nc_bwd_edges_launch(x_src, P, Q, gs, g, crow, gxs, graph, kinds, acts, drop, gQ, gx, partial, part=None, row_max=None, T=None, gP=None):
    """K2b over the transposed CSR; `part` = (items, n_wave_items, hubs) restricts it to a subset of the sources.
    gs given: materialised dL/ds rows; else the shared-gradient form on (g, crow); with T and gP: K2a fused into the epilogue."""
    K = len(acts)
    S, H = x_src.shape
    items, n_wave, hubs = part if part is not None else (graph.t_items, graph.t_n_wave_items, graph.t_hubs)
    shared = gs is None
    mode, thr, seed, seed_dev, keep = drop.args()
    with _span("nc_fused_bwd"):
        call("mma_nc_fused_bwd", ptr(x_src), x_src.stride(0), ptr(P), P.stride(0), ptr(Q), Q.stride(0),
             ptr(gs), K * H, ptr(g) if shared else None, g.stride(0) if shared else 0,
             ptr(crow) if shared else None, crow.stride(0) if shared else 0, host_codes(kinds) if shared else None,
             ptr(gxs), H, ptr(T), K * H if T is not None else 0, ptr(gP), gP.stride(0) if gP is not None else 0, graph.N,
             ptr(graph.t_col), ptr(graph.t_eid), ptr(items), items.shape[0], n_wave,
             ptr(hubs) if hubs.shape[0] else None, hubs.shape[0], ptr(partial), graph.t_n_slots if hubs.shape[0] else 0,
             ptr(gQ), gQ.stride(0), ptr(gx), H, ptr(row_max), S, graph.E, H, K, host_codes(acts), mode, thr, seed, seed_dev, graph.edge_base,
             keep, ptr(graph.sync(1)) if part is None else None, stream_ptr())


class _NCFused(torch.autograd.Function):
    """m[k] = combine_k(x_i, sum_j drop(act_k(P_k[i] + Q_k[j])) * x_j)   (K1 forward, K2a + K2b backward).
    reduce_k: return sum_k m[k] (N,H) instead of (K,N,H) - all MMA.forward needs; backward then takes the
    shared-gradient form (one (N,H) upstream gradient for all masks)."""

    @staticmethod
    def forward(ctx, x_src, P, Q, graph, kinds, acts, drop, reduce_k):
        # x_src: (n_src,H) feature table; its first N rows are the targets (n_src > N only in the sharded path,
        # where the tail holds halo rows).  P: (N,K*H) = x_src[:N] @ [W_k[:H]..], Q: (n_src,K*H) = x_src @ [W_k[H:]..]
        require_gpu(x_src, P, Q)
        x_src = x_src.contiguous()
        if P.stride(1) != 1:
            P = P.contiguous()
        if Q.stride(1) != 1:
            Q = Q.contiguous()
        out, T, sel, crow = nc_fwd_launch(x_src, P, Q, graph, kinds, acts, drop, reduce_k, any(ctx.needs_input_grad[:3]))
        ctx.graph, ctx.kinds, ctx.acts, ctx.drop, ctx.reduce_k = graph, kinds, acts, drop, reduce_k
        ctx.save_for_backward(x_src, P, Q, T, sel, crow)
        return out

    @staticmethod
    def backward(ctx, g):
        graph, kinds, acts, drop, reduce_k = ctx.graph, ctx.kinds, ctx.acts, ctx.drop, ctx.reduce_k
        x_src, P, Q, T, sel, crow = ctx.saved_tensors
        shared = crow is not None
        K = len(kinds)
        H, S = x_src.shape[1], graph.n_src
        g = g.contiguous()
        dev = g.device
        gQ = torch.empty((S, K * H), device=dev, dtype=torch.float32)
        gx = torch.empty((S, H), device=dev, dtype=torch.float32)
        partial = (torch.empty((graph.t_n_slots, (K + 1) * H), device=dev, dtype=torch.float32)
                   if graph.t_n_slots else None)
        if shared and FUSE_NODE_BWD:
            gP = torch.empty((graph.N, K * H), device=dev, dtype=torch.float32)
            nc_bwd_edges_launch(x_src, P, Q, None, g, crow, None, graph, kinds, acts, drop, gQ, gx, partial, T=T, gP=gP)
        else:
            gs, gP, gxs = nc_bwd_node_launch(g, reduce_k, sel, crow, T, graph, kinds, H, shared)
            nc_bwd_edges_launch(x_src, P, Q, gs, g, crow, gxs, graph, kinds, acts, drop, gQ, gx, partial)
        return gx, gP, gQ, None, None, None, None, None


class _NCLocalLayer(torch.autograd.Function):
    """sum_k m_k from (x, Wtop, Wbot) in one autograd node: owns its GEMMs, so P and Q (and their gradients) are column
    blocks of ONE (N,2*K*H) buffer - one forward GEMM, one dL/dx GEMM over the concatenated reduction, one dW GEMM."""

    @staticmethod
    def forward(ctx, x, wtop, wbot, graph, kinds, acts, drop):
        from .dense import mm_into
        require_gpu(x)
        x = x.contiguous()
        N, H = x.shape
        K = len(kinds)
        KH = K * H
        PQ = torch.empty((N, 2 * KH), device=x.device, dtype=torch.float32)
        ctx.cat_given = wbot is None                                        # wtop IS [Wtop | Wbot] (H, 2*K*H): mask_weights()
        wcat = wtop if ctx.cat_given else torch.cat([wtop, wbot], 1)        # (H, 2*K*H)
        need = any(ctx.needs_input_grad[:3])
        box = [] if need else None                                          # row maxima of x, when the forward GEMM forms them: the
        mm_into(x, wcat, PQ, row_max_box=box)                               # weight-gradient product's row scales (three-product TN form)
        msum, T, sel, crow = nc_fwd_launch(x, PQ[:, :KH], PQ[:, KH:], graph, kinds, acts, drop, True, need)
        ctx.graph, ctx.kinds, ctx.acts, ctx.drop = graph, kinds, acts, drop
        ctx.save_for_backward(x, PQ, T, sel, crow, wcat, box[0] if box else None)
        return msum

    @staticmethod
    def backward(ctx, g):
        from .dense import rows_mm_add_, xt_g
        graph, kinds, acts, drop = ctx.graph, ctx.kinds, ctx.acts, ctx.drop
        x, PQ, T, sel, crow, wcat, x_row_max = ctx.saved_tensors
        N, H = x.shape
        K = len(kinds)
        KH = K * H
        g = g.contiguous()
        from . import dense
        gPQ = torch.empty((N, 2 * KH), device=g.device, dtype=torch.float32)
        # the three-product dL/dx GEMM scales every row of [gP|gQ] by a power of two: K2a and K2b leave the row maxima here
        row_max = torch.zeros((N,), device=g.device, dtype=torch.float32) if dense.f16x2_n128_ok(N, 2 * KH, H) and K <= 8 else None
        gx = torch.empty((N, H), device=g.device, dtype=torch.float32)
        partial = (torch.empty((graph.t_n_slots, (K + 1) * H), device=g.device, dtype=torch.float32)
                   if graph.t_n_slots else None)
        shared = crow is not None
        if shared and FUSE_NODE_BWD:         # K2a inside K2b's epilogue: gP lands in the left half of [gP | gQ] from the same launch
            nc_bwd_edges_launch(x, PQ[:, :KH], PQ[:, KH:], None, g, crow, None, graph, kinds, acts, drop, gPQ[:, KH:], gx, partial,
                                row_max=row_max, T=T, gP=gPQ[:, :KH])
        else:
            gs, gP, gxs = nc_bwd_node_launch(g, True, sel, crow, T, graph, kinds, H, shared, gP=gPQ[:, :KH], row_max=row_max)
            nc_bwd_edges_launch(x, PQ[:, :KH], PQ[:, KH:], gs, g, crow, gxs, graph, kinds, acts, drop, gPQ[:, KH:], gx, partial,
                                row_max=row_max)
        dense.rows_mm_add_scaled_(gx, gPQ, wcat.t(), row_max)                # direct + through P and Q in one GEMM (C += A B)
        gw = xt_g(x, gPQ, x_row_max, row_max) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
        if ctx.cat_given:
            return gx, gw, None, None, None, None, None
        return gx, (gw[:, :KH] if gw is not None else None), (gw[:, KH:] if gw is not None else None), None, None, None, None


def nc_local_layer(x, wtop, wbot, graph, kinds, acts, drop=None):
    """sum_k m_k (graph.N, H) straight from the features and the concatenated mask weights (unsharded graphs).
    wbot=None: `wtop` is already [Wtop | Wbot] (H, 2*K*H), e.g. from mask_weights()."""
    assert graph.n_src == graph.N
    return _NCLocalLayer.apply(x, wtop, wbot, graph, tuple(kinds), tuple(acts), drop or DropoutSpec(0.0))


class DeviceSeeds:
    """Dropout seeds of a graph-capturable module: n device seeds (`.seeds`, what DropoutSpec(seed_tensor=) takes) advanced by ONE
    captured launch per step (mma_seed_advance: splitmix64 streams whose states are drawn once from torch's generator)."""

    def __init__(self, n, device):
        if torch.cuda.is_current_stream_capturing():
            raise _lib.MMALibraryError("mma_amd: run one warm-up step before capturing a graph (the dropout seed states are drawn on the first step)")
        self.n = n
        self.buf = torch.empty(2 * n, dtype=torch.int64, device=device)
        self.buf.random_()
        self.seeds = self.buf[:n]
        self.device = self.buf.device

    def advance(self):
        call("mma_seed_advance", ptr(self.buf), self.n, stream_ptr())
        return self.seeds


class SeedSlot:
    """Seed i of a DeviceSeeds shared by several modules (train_step.GraphedNetStep: the four MMAConv layers of the Net draw their
    seeds from ONE advance launch per step): slot 0 advances the whole set, the others read theirs."""

    def __init__(self, shared, i):
        self.shared, self.i, self.n, self.device = shared, i, 1, shared.device
        self.seeds = shared.seeds[i:i + 1]

    def advance(self):
        if self.i == 0:
            self.shared.advance()
        return self.seeds


class _MaskWeights(torch.autograd.Function):
    """[W_1[:H] .. W_K[:H] | W_1[H:] .. W_K[H:]] (H, 2*K*H) from the K caller-owned (2H,H) mask weights in one launch, and their
    gradients back in one: as slices + torch.cat the forward is 3 launches and autograd's backward ~5 per mask (zero-fill + slice copy
    per half, an add, the accumulation) - on Cora, where every kernel of the layer is a few microseconds, a fifth of the replay."""

    @staticmethod
    def forward(ctx, *masks):
        K = len(masks)
        H = masks[0].shape[1]
        ctx.shape = (K, H)
        # ONE concatenation kernel (the K top halves, then the K bottom halves, are contiguous row blocks of their masks); as
        # stack + permuted copy it was two launches
        return torch.cat([w[:H] for w in masks] + [w[H:] for w in masks], 1)

    @staticmethod
    def backward(ctx, g):
        K, H = ctx.shape
        gm = g.view(H, 2, K, H).permute(2, 1, 0, 3).reshape(K, 2 * H, H)       # one copy; the K gradients are row blocks of it
        return tuple(gm[k] for k in range(K))


def mask_weights(masks):
    return _MaskWeights.apply(*masks)


def nc_fused_aggregate(x, P, Q, graph, kinds, acts, drop=None, reduce_k=False):
    """Fused K-mask aggregation (K <= 8 per call) -> m (K, graph.N, H), or sum_k m[k] (graph.N, H) with reduce_k.

    x: (graph.n_src, H) feature table whose first graph.N rows are the targets; P = x[:N] @ Wtop (N, K*H),
    Q = x @ Wbot (n_src, K*H) with Wtop/Wbot the column-concatenated top/bottom halves of the K mask weights;
    kinds/acts: MMA_KIND_* / MMA_ACT_* codes per mask."""
    return _NCFused.apply(x, P, Q, graph, tuple(kinds), tuple(acts), drop or DropoutSpec(0.0), bool(reduce_k))


def _spmm_call(rowptr, col, val, items, hubs, n_slots, B, rows_per_block, K, bias, out, n_rows, C, n_wave_items=None):
    if K == 1 and items is not None:
        partial = torch.empty((n_slots, C), device=B.device, dtype=torch.float32) if n_slots else None
        call("mma_csr_spmm_items", ptr(col), ptr(val), ptr(B), B.stride(0), ptr(bias), ptr(out), C, ptr(items), items.shape[0],
             items.shape[0] if n_wave_items is None else n_wave_items,
             ptr(hubs) if n_slots else None, hubs.shape[0], ptr(partial), n_slots, C, stream_ptr())
    else:
        call("mma_csr_spmm", ptr(rowptr), ptr(col), ptr(val), ptr(B), B.stride(0), rows_per_block, K,
             ptr(bias), ptr(out), C, n_rows, C, stream_ptr())


class _CsrSpmm(torch.autograd.Function):
    """out = sum_k A @ B[k] + bias  ( = torch.spmm(cat((adj,)*K, 1), B.view(K*N, C)) + bias, layers.py:861-865 )"""

    @staticmethod
    def forward(ctx, B, bias, sg, K):
        require_gpu(B, bias)
        B = B.contiguous()
        rows_per_block = B.shape[0] // K
        assert B.shape[0] == K * rows_per_block and rows_per_block == sg.n_cols
        C = B.shape[1]
        out = torch.empty((sg.n_rows, C), device=B.device, dtype=torch.float32)
        # per stored element: column id + value + the gathered row; per output row: row pointer / item + the row written
        with _span("csr_spmm_fwd", nbytes=K * sg.col.numel() * (8 + 4 * C) + sg.n_rows * (16 + 4 * C), flops=2 * K * sg.col.numel() * C):
            _spmm_call(sg.rowptr, sg.col, sg.val, sg.items, sg.hubs, sg.n_slots, B, rows_per_block, K, bias, out, sg.n_rows, C,
                       sg.n_wave_items)
        ctx.sg, ctx.K, ctx.has_bias = sg, K, bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        sg, K = ctx.sg, ctx.K
        g = g.contiguous()
        C = g.shape[1]
        gB1 = torch.empty((sg.n_cols, C), device=g.device, dtype=torch.float32)
        with _span("csr_spmm_bwd", nbytes=sg.t_col.numel() * (8 + 4 * C) + sg.n_cols * (16 + 4 * C), flops=2 * sg.t_col.numel() * C):
            _spmm_call(sg.t_rowptr, sg.t_col, sg.t_val, sg.t_items, sg.t_hubs, sg.t_n_slots, g, sg.n_rows, 1, None, gB1, sg.n_cols, C,
                       sg.t_n_wave_items)
        gB = gB1 if K == 1 else gB1.unsqueeze(0).expand(K, -1, -1).reshape(K * sg.n_cols, C)  # every k-block sees the same A^T g
        from .dense import col_sum
        return gB, (col_sum(g) if ctx.has_bias else None), None, None


def csr_spmm(B, bias, sg, K=1):
    return _CsrSpmm.apply(B, bias, sg, K)


# ---- graph-regression path (K3/K4/K6) ----------------------------------------------------------------------
GR_AGGR = {"sum": 0, "mean": 1, "min": 2, "max": 3, "var": 4, "std": 5}
GR_SCALER = {"identity": 0, "amplification": 1, "attenuation": 2, "linear": 3, "inverse_linear": 4}


class DeviceCSR:
    """Edges grouped by `key` (stable): rowptr (N+1), perm (E) original positions, other (E) = other[perm]."""

    def __init__(self, key, other, N, long_list=False):
        require_gpu(key)
        E = key.numel()
        dev = key.device
        key = key.contiguous()
        assert key.dtype == torch.int64 and (other is None or (other.dtype == torch.int64 and other.numel() == E))
        self.N, self.E = int(N), int(E)
        self.rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        self.other = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if other is not None else self.perm
        nbytes = _lib.lib().mma_csr_workspace_bytes(E, N)
        if nbytes < 0:
            raise _lib.MMALibraryError("graph too large for int32 CSR: E=%d N=%d" % (E, N))
        ws = torch.empty(int(nbytes) + 1024, dtype=torch.uint8, device=dev)   # per call: cheap (caching allocator), capture-safe
        # [count, ids...] of the groups above MMA_GR_LONG_SEGMENT entries: what K3/K4 leave to their wave-per-node pass
        self.long_nodes = torch.empty(int(_lib.lib().mma_gr_long_nodes_len(E)), dtype=torch.int32, device=dev) if long_list else None
        with _span("csr_build"):
            call("mma_build_csr", ptr(key), ptr(other.contiguous()) if other is not None else None, E, N, ptr(self.rowptr),
                 ptr(self.perm), ptr(self.other) if other is not None else None, ptr(self.long_nodes), ptr(ws), ws.numel(), stream_ptr())


def _csr_check(self):
    """Host-side check of the long-segment list's error flag (one small device-to-host copy: call it OUTSIDE timed or captured regions).
    K3 / K4 skip a list whose count word or ids cannot be right instead of walking off it - and leave a flag so that the skip is not
    silent (round-4 ADVICE)."""
    if self.long_nodes is None:
        return
    flag = int(self.long_nodes[-1].item())
    if flag:
        raise _lib.MMALibraryError("mma_amd: the long-segment list of this CSR was clobbered (%s): aggregates of segments above 64 edges "
                                   "were skipped" % ("count beyond the list's capacity" if flag == 1 else "an id that is not a node"))


DeviceCSR.check = _csr_check


class GRGraph:
    """Both groupings of one edge_index (by target for the forward, by source for dV), built lazily, cached per
    edge_index tensor by the caller."""

    def __init__(self, edge_index, N):
        self.edge_index, self.N, self.E = edge_index, int(N), int(edge_index.shape[1])
        self.by_target = DeviceCSR(edge_index[1], edge_index[0], N, long_list=True)
        self._by_source = None
        self._inv_perm = None
        self._by_source_pos = None

    @property
    def by_source(self):
        if self._by_source is None:
            self._by_source = DeviceCSR(self.edge_index[0], None, self.N)
        return self._by_source

    @property
    def perm(self):
        """(E,) int32: original edge id at each target-sorted position."""
        return self.by_target.perm[:self.E]

    @property
    def perm_long(self):
        """perm as int64 (an index_select index), converted once per graph - not once per layer call."""
        if self.__dict__.get("_perm_long") is None:
            self._perm_long = self.perm.long()
        return self._perm_long

    @property
    def inv_perm(self):
        """(E,) int64: target-sorted position of each original edge."""
        if self._inv_perm is None:
            inv = torch.empty(self.E, dtype=torch.int64, device=self.perm.device)
            inv[self.perm.long()] = torch.arange(self.E, device=self.perm.device)
            self._inv_perm = inv
        return self._inv_perm

    def type_onehot(self, z_index, n_types):
        """(E, 32 | 64 | ...) fp32 one-hot rows of a uint8 type index (cached for the last index tensor): the operand that turns
        the per-edge message gradient into the gradient of the (n_types, D) table with one fixed-order TN product."""
        key = (z_index.data_ptr(), z_index._version, int(n_types))
        hit = self.__dict__.get("_type_onehot")
        if hit is None or hit[0] != key:
            width = -(-int(n_types) // 32) * 32
            oh = torch.zeros((self.E, width), device=z_index.device, dtype=torch.float32)
            oh.scatter_(1, z_index.long().unsqueeze(1), 1.0)
            hit = (key, oh, z_index)              # the index tensor is kept alive with its one-hot
            self.__dict__["_type_onehot"] = hit
        return hit[1]

    def type_onehot_row_max(self, z_index):
        """(E,) ones: max |row| of the one-hot operand (cached with the graph) - the x_row_max of its three-product TN form."""
        hit = self.__dict__.get("_type_onehot_rm")
        if hit is None or hit.device != z_index.device:
            hit = torch.ones((self.E,), device=z_index.device, dtype=torch.float32)
            self.__dict__["_type_onehot_rm"] = hit
        return hit

    @property
    def by_source_pos(self):
        """(E,) int32: the by-source grouping expressed in target-sorted POSITIONS (rows of a by_pos message-gradient buffer)."""
        if self._by_source_pos is None:
            self._by_source_pos = self.inv_perm[self.by_source.perm[:self.E].long()].to(torch.int32)
        return self._by_source_pos


_GR_GRAPHS = {}          # (data_ptr, version, shape, N) -> GRGraph, most recent last; shared by every layer that sees the same edge_index


def gr_graph(edge_index, N, keep=4):
    """The GRGraph of an edge_index tensor, built once per tensor (and in-place version): the four MMAConv layers of the
    reference's Net (mma.py:91-97) all receive the same edge_index, so one CSR build (K6) serves them all.  The cached graph
    holds its edge_index, so the address in the key cannot be reused by another tensor while the entry is alive."""
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(N), str(edge_index.device))
    g = _GR_GRAPHS.pop(key, None)
    if g is None:
        g = GRGraph(edge_index, N)
    _GR_GRAPHS[key] = g
    while len(_GR_GRAPHS) > keep:
        _GR_GRAPHS.pop(next(iter(_GR_GRAPHS)))
    return g


class _PermuteRows(torch.autograd.Function):
    """rows[idx] for a PERMUTATION idx; backward gathers with the inverse permutation (deterministic, no index_add)."""

    @staticmethod
    def forward(ctx, x, idx, inv):
        ctx.inv = inv
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        return g.index_select(0, ctx.inv), None, None


def rows_by_position(x, graph):
    """x (E, ...) by original edge id -> rows in target-sorted position order (so that a GEMM on it yields Z by position)."""
    if graph.E == 0:
        return x
    return _PermuteRows.apply(x, graph.perm_long, graph.inv_perm)


def _gr_call(fn, csr, U, V, Z, by_pos, inputs, extra, N, E, T, F, aggr, scalers, avg_log, avg_lin, drop, z_index=None):
    D = T * F
    lduv = U.stride(0) if U is not None else 0
    call(fn, ptr(csr.rowptr), ptr(csr.other), ptr(csr.perm), ptr(U), ptr(V), lduv, ptr(Z), Z.stride(0) if Z is not None else 0,
         1 if by_pos else 0, ptr(z_index), ptr(inputs), D if inputs is not None else 0, *extra, N, E, T, F, host_codes(aggr), len(aggr),
         host_codes(scalers), len(scalers), float(avg_log), float(avg_lin), drop.mode, drop.thr, drop.seed, ptr(drop.seed_tensor),
         stream_ptr())


class _GRAggregate(torch.autograd.Function):
    """K aggregators + compounding degree scalers over target segments (mma_conv.py:159-196), messages either given
    (`inputs`, the public aggregate() API) or formed in-kernel as drop(U[i] + V[j] + Z[r]) (fused forward).
    Fused mode: UV (N, 2*T*F) holds U in its left and V in its right column half (one GEMM made both), Z is (E, T*F)
    or None - with by_pos its rows (and the rows of the gradient handed back for it) are in target-sorted position order;
    the gradient of UV comes back as one buffer whose halves the kernel and the by-source segment sum fill."""

    @staticmethod
    def forward(ctx, inputs, UV, Z, graph, T, F, aggr, scalers, avg_log, avg_lin, drop, by_pos, z_index=None):
        fused = inputs is None
        csr = graph.by_target
        N, E, D = graph.N, graph.E, T * F
        K, S = len(aggr), len(scalers)
        ref = UV if fused else inputs
        require_gpu(ref)
        dev = ref.device
        U = V = None
        if fused:
            # row-pitched inputs are fine (the kernels take lduv / ldz): only the columns of a row must be contiguous
            UV = UV if UV.stride(1) == 1 and UV.stride(0) % 4 == 0 else UV.contiguous()
            Z = Z if Z is None or (Z.stride(1) == 1 and Z.stride(0) % 4 == 0) else Z.contiguous()
            if z_index is not None:      # categorical edge features: Z is the (n_types, D) table, z_index (E,) uint8 its row per edge
                assert Z is not None and Z.shape[1] == D and Z.shape[0] <= 256 and z_index.dtype == torch.uint8 and z_index.shape == (E,)
                z_index = z_index.contiguous()
            assert UV.shape == (N, 2 * D) and (Z is None or z_index is not None or Z.shape == (E, D))
            U, V = UV[:, :D], UV[:, D:]
        else:
            inputs = inputs.contiguous()
            assert inputs.shape == (E, T, F)
            by_pos = False
        out = torch.empty((N, T, S * K * F), device=dev, dtype=torch.float32)
        need = any(ctx.needs_input_grad[:3])
        # argmin / argmax: one BYTE per column (offset inside the target's segment) + an int32 side table for the rare
        # segments of >= 256 edges (see include/mma_amd.h) - a quarter of the int32 edge ids round 1 wrote and re-read
        side_rows = int(_lib.lib().mma_gr_arg_side_rows(E))
        amin = torch.empty((N, D), dtype=torch.uint8, device=dev) if need and 2 in aggr else None
        amax = torch.empty((N, D), dtype=torch.uint8, device=dev) if need and 3 in aggr else None
        amin_s = torch.empty((side_rows, D), dtype=torch.int32, device=dev) if amin is not None else None
        amax_s = torch.empty((side_rows, D), dtype=torch.int32, device=dev) if amax is not None else None
        stats = need and (4 in aggr or 5 in aggr)
        mean = torch.empty((N, D), device=dev) if stats else None
        var = torch.empty((N, D), device=dev) if stats else None
        ctx.cfg = (graph, T, F, aggr, scalers, avg_log, avg_lin, drop, fused, Z is not None, by_pos)
        with _span("gr_fused_fwd"):
            _gr_call("mma_gr_fused_fwd", csr, U, V, Z, by_pos, inputs,
                     (ptr(out), ptr(amin), ptr(amax), ptr(amin_s), ptr(amax_s), ptr(mean), ptr(var), D, ptr(csr.long_nodes)),
                     N, E, T, F, aggr, scalers, avg_log, avg_lin, drop, z_index)
        ctx.save_for_backward(inputs, UV, Z, amin, amax, amin_s, amax_s, mean, var, z_index)
        return out

    @staticmethod
    def backward(ctx, gout):
        graph, T, F, aggr, scalers, avg_log, avg_lin, drop, fused, has_z, by_pos = ctx.cfg
        inputs, UV, Z, amin, amax, amin_s, amax_s, mean, var, z_index = ctx.saved_tensors
        csr = graph.by_target
        N, E, D = graph.N, graph.E, T * F
        gout = gout.contiguous()
        from . import dense
        tall = dense.X3_LINEAR and fused
        # the gradients of [U|V] and Z feed tall GEMMs: as views of zero-padded, registered buffers dense.linear_tall takes them as they are -
        # [r5] with their row maxima (K4 and the dV segment sum merge max |row| into one zeroed (E + N,) array): the GEMMs behind take three
        # products instead of six
        pad_m, pad_u = tall and has_z and E >= dense.X3_LINEAR_MIN_ROWS, tall and fused and N >= dense.X3_LINEAR_MIN_ROWS
        rm = torch.zeros((E + N,), device=gout.device, dtype=torch.float32) if (pad_m or pad_u) and dense.X3_ROW_MAX and E > 0 else None
        gmsg = (dense.padded_empty(E, D, gout.device, row_max=rm[:E] if rm is not None else None) if pad_m
                else torch.empty((E, D), device=gout.device, dtype=torch.float32))
        if E == 0:
            if not fused:
                return (gmsg.view(E, T, F),) + (None,) * 12
            gz = (torch.zeros_like(Z) if z_index is not None else gmsg) if has_z else None
            return (None, torch.zeros((N, 2 * D), device=gout.device, dtype=torch.float32), gz) + (None,) * 10
        U, V = (UV[:, :D], UV[:, D:]) if fused else (None, None)
        # dU[i] = sum of its target segment: produced by K4 itself (it walks exactly those segments); dV[j] = sum over the
        # edges leaving j: one segment sum (K5 kernel) over the by-source grouping.  Both land in the halves of one (N, 2D) buffer.
        gUV = None
        if fused:
            gUV = (dense.padded_empty(N, 2 * D, gout.device, row_max=rm[E:] if rm is not None else None) if pad_u
                   else torch.empty((N, 2 * D), device=gout.device, dtype=torch.float32))
        with _span("gr_fused_bwd"):
            _gr_call("mma_gr_fused_bwd", csr, U, V, Z, by_pos, inputs,
                     (ptr(gout), ptr(amin), ptr(amax), ptr(amin_s), ptr(amax_s), ptr(mean), ptr(var), D, ptr(csr.long_nodes), ptr(gmsg),
                      gmsg.stride(0), ptr(gUV), gUV.stride(0) if fused else 0, ptr(rm[:E]) if rm is not None else None,
                      ptr(rm[E:]) if rm is not None else None),
                     N, E, T, F, aggr, scalers, avg_log, avg_lin, drop, z_index)
        if not fused:
            return (gmsg.view(E, T, F),) + (None,) * 12
        cs = graph.by_source
        rows = graph.by_source_pos if by_pos else cs.perm      # gmsg rows: positions (by_pos) or original edge ids
        with _span("gr_segsum", nbytes=E * (4 + 4 * D) + N * (4 + 4 * D), flops=E * D):
            if rm is not None:          # the dV rows' maxima join dU's: one bound for the whole [dU | dV] row
                call("mma_csr_spmm_rm", ptr(cs.rowptr), ptr(rows), None, ptr(gmsg), gmsg.stride(0), E, 1, None, ptr(gUV[:, D:]), gUV.stride(0), N, D,
                     ptr(rm[E:]), stream_ptr())
            else:
                call("mma_csr_spmm", ptr(cs.rowptr), ptr(rows), None, ptr(gmsg), gmsg.stride(0), E, 1, None, ptr(gUV[:, D:]), gUV.stride(0), N, D,
                     stream_ptr())
        gz = gmsg if has_z else None
        if has_z and z_index is not None:
            # dL/dZ_table = onehot(z_index)^T gmsg: a fixed-order reduction (index_add_ would use atomics), one row per edge type
            gp = dense._padded_parent(gmsg, dense._round_up(D, 128))          # the zero-padded buffer itself: bf16x3 TN kernel, no copy
            gz = dense.xt_g(graph.type_onehot(z_index, Z.shape[0]), gp if gp is not None else gmsg, graph.type_onehot_row_max(z_index),
                            rm[:E] if (rm is not None and gp is not None) else None)[:Z.shape[0], :D]
        return (None, gUV, gz) + (None,) * 10


class _TowerPost(torch.autograd.Function):
    """MMAConv's per-tower post-NN on the UNSCALED aggregates (K13 forward, K14 + one TN product per tower backward):
        y[n, t*O + o] = sum_q pre_q(deg_n) sum_kf agg[n,t,kf] Wo[t][o][q*KF + kf]
    = post_nns[t] applied to the `out` of mma_conv.py:181-196 without ever forming that (N,T,S*K*F) tensor or its gradient: the
    degree scalers are per-target row factors (pre_q = their running product), so they move from the aggregates to the products."""

    @staticmethod
    def forward(ctx, agg, Wo, rowptr, scalers, avg_log, avg_lin):
        require_gpu(agg, Wo)
        N, T, KF = agg.shape
        O, S = Wo.shape[1], len(scalers)
        assert Wo.shape == (T, O, S * KF) and O <= 16 and KF % 4 == 0 and S <= 5       # (MMAConv pads the tower width to 4 floats)
        agg = agg if agg.is_contiguous() else agg.contiguous()
        KFp = int(_lib.lib().mma_tower_post_kfp(KF))
        # the weight columns twice, zero-padded: Wa (T, KFp, S*16) [kf][q*16+o] for the forward, Wb (T, S*16, KFp+16) [q*16+o][kf]
        # for the backward (the tower's weights are staged in LDS in exactly these layouts)
        Wb = torch.empty((T, S * 16, KFp + 16), device=agg.device, dtype=torch.float32)
        Wa = torch.empty((T, KFp, S * 16), device=agg.device, dtype=torch.float32)
        call("mma_tower_post_weights", ptr(Wo.contiguous()), T, O, S, KF, ptr(Wa), ptr(Wb), stream_ptr())      # one launch (a fill + two copies before)
        y = torch.empty((N, T * O), device=agg.device, dtype=torch.float32)
        # the scaler products of every node, once: K13 / K14 / K15 read them - and once per GRAPH: the four layers of the reference's Net
        # share edge_index, scalers and the degree statistics (mma.py:91-97), so the table rides on the plan's rowptr tensor
        key = (N, tuple(scalers), float(avg_log), float(avg_lin))
        cache = getattr(rowptr, "_mma_post_pre", None)
        pre = cache[1] if cache is not None and cache[0] == key else None
        # agg rows in, y rows out, the scaler table; S x 16 padded outputs per (node, tower, kf) on the fp32 matrix cores
        # (K13 runs on three bf16 pieces per operand, six piece products, where its split weights fit the LDS and MMA_POST_EXACT is unset)
        x3 = os.environ.get("MMA_POST_EXACT") != "1" and (KFp // 32) * S * 3 * 1024 <= 160 * 1024
        with _span("tower_post_fwd", nbytes=4 * N * (T * KF + T * O + 8), flops=2 * N * T * KFp * S * 16, mfma="bf16x6" if x3 else "f32"):
            if pre is None:
                pre = torch.empty((N, 8), device=agg.device, dtype=torch.float32)
                call("mma_tower_post_pre", ptr(rowptr), ptr(pre), N, S, host_codes(scalers), float(avg_log), float(avg_lin), stream_ptr())
                try:
                    rowptr._mma_post_pre = (key, pre)
                except AttributeError:
                    pass
            call("mma_tower_post_fwd", ptr(agg), T * KF, ptr(pre), ptr(Wa), ptr(y), T * O, N, T, KF, S, O, host_codes(scalers),
                 float(avg_log), float(avg_lin), stream_ptr())
        ctx.save_for_backward(agg, Wb, pre)
        ctx.cfg = (O, scalers, avg_log, avg_lin)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import dense
        agg, Wb, pre = ctx.saved_tensors
        O, scalers, avg_log, avg_lin = ctx.cfg
        N, T, KF = agg.shape
        S = len(scalers)
        gy = gy.contiguous()
        gagg = torch.empty_like(agg)
        need_w = ctx.needs_input_grad[1]
        KFp = Wb.shape[2] - 16
        with _span("tower_post_bwd", nbytes=4 * N * (T * KF + T * O + 8), flops=2 * N * T * KFp * S * 16, mfma="f32"):
            call("mma_tower_post_bwd", ptr(gy), T * O, ptr(pre), ptr(Wb), ptr(gagg), T * KF, None, 0, N, T, KF, S, O,
                 host_codes(scalers), float(avg_log), float(avg_lin), stream_ptr())
        gWo = None
        if need_w:
            # K15: gW[t][q*16+o][kf] = sum_n pre_q gy agg on the fp32 matrix cores (node = reduction index), per-wave partial tiles
            # summed in a fixed order by K8
            kfp16 = -(-KF // 16) * 16
            n_chunks = int(_lib.lib().mma_tower_post_gw_chunks(N, T))
            part = torch.empty((n_chunks, T * S * 16 * kfp16), device=agg.device, dtype=torch.float32)
            with _span("tower_post_gw", nbytes=4 * N * (T * KF + T * O + 8) + 4 * part.numel(), flops=2 * N * T * kfp16 * S * 16, mfma="f32"):
                call("mma_tower_post_gw", ptr(gy), T * O, ptr(agg), T * KF, ptr(pre), ptr(part), n_chunks, N, T, KF, S, O,
                     host_codes(scalers), float(avg_log), float(avg_lin), stream_ptr())
                if n_chunks <= 256 or (n_chunks <= 1024 and part.shape[1] > 2048):      # (where K8 takes its single-block order) the
                    # partial tiles summed straight into the weight layout: one launch, not two
                    gWo = torch.empty((T, O, S * KF), device=agg.device, dtype=torch.float32)
                    call("mma_tower_post_gw_reduce", ptr(part), n_chunks, T, S, O, KF, ptr(gWo), stream_ptr())
                else:
                    gWq = dense.col_sum(part).view(T, S, 16, kfp16)
                    gWo = gWq[:, :, :O, :KF].permute(0, 2, 1, 3).reshape(T, O, S * KF)
        return gagg, gWo, None, None, None, None


def tower_post(agg, Wo, rowptr, scalers, avg_log, avg_lin):
    """agg (N,T,K*F) unscaled aggregates, Wo (T,O,S*K*F) post-NN weight columns of `out` (q-major, then k, then f) -> y (N, T*O)."""
    return _TowerPost.apply(agg, Wo, rowptr, tuple(GR_SCALER[s] for s in scalers), avg_log, avg_lin)


def gr_aggregate(inputs, graph, aggregators, scalers, avg_log, avg_lin):
    """aggregate() on given messages (E,T,F) -> (N,T,S*K*F)."""
    E, T, F = inputs.shape
    return _GRAggregate.apply(inputs, None, None, graph, T, F, tuple(GR_AGGR[a] for a in aggregators),
                              tuple(GR_SCALER[s] for s in scalers), avg_log, avg_lin, DropoutSpec(0.0), False)


def gr_fused_conv(UV, Z, graph, T, F, aggregators, scalers, avg_log, avg_lin, drop, z_by_pos=False, z_index=None):
    """message + aggregate fused: messages drop(U[i] + V[j] + Z[r]) never materialise.  UV = [U | V] (N, 2*T*F).
    z_by_pos: Z's rows are in target-sorted position order (made from rows_by_position(edge_attr, graph)): the kernels then
    stream Z and the message gradients contiguously instead of gathering / scattering them by original edge id."""
    return _GRAggregate.apply(None, UV, Z, graph, T, F, tuple(GR_AGGR[a] for a in aggregators),
                              tuple(GR_SCALER[s] for s in scalers), avg_log, avg_lin, drop, bool(z_by_pos), z_index)
