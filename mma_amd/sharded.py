"""1-D node-sharded MMA layer: one process per GPU, RCCL all-to-all halo exchange over xGMI.

The reference has no distributed code; this is the MI355X-first scale-out of its hot path.
Targets are independent units, so rank r owns a contiguous, EDGE-balanced range of target nodes, their CSR
rows, their feature rows and their outputs.  Sources outside the range are halo rows.  Per layer call:

  forward   all-to-all #1: x rows of the halo (H floats each) - Q of the halo rows is recomputed locally by the
            GEMM, so only x travels ((K+1)x fewer bytes than sending [x|Q]);   fused aggregate on [own | halo];
            all-to-all #2: the (N,C) tail rows the K-stacked SpMM needs from the halo (C floats each);
  backward  the two reverse all-to-alls, summed at the owner (per-peer unpack-add, unique rows => no atomics);
            grad of the (tiny) parameters is all-reduced by `allreduce_grads()`.

xGMI is point-to-point (7 links per GPU): an all-to-all drives all links at once, which is why the exchange is
an all-to-all-v (torch.distributed.all_to_all_single on the "nccl" = RCCL backend) and not a ring collective.
A contiguous target range is also a contiguous range of the global CSR, so the dropout hash keeps the global
edge ids (NCGraph.edge_base) and the sharded result equals the single-GPU one for the same seed.

With the "gloo" backend (tests; CPU-only rendezvous) GPU buffers are staged through the host."""
import numpy as np
import torch
import torch.distributed as dist

from . import functional as Fn
from .dense import mm, mm_into, rows_mm_add_, xt_g
from ._lib import call, ptr, require_gpu, stream_ptr
from .graph import DEFAULT_CHUNK, NCGraph, SpmmGraph
from .layers import _AGG
from .scalers import scaler_row_factor, true_degree_row_factor


DEVICE_PLAN = __import__("os").environ.get("MMA_DEVICE_PLAN", "1") != "0"     # 0: the host numpy plan builders (round 2)


def partition_bounds(rowptr, world, row_cost=0.0):
    """Contiguous target ranges with (nearly) equal cost: bounds[r] .. bounds[r+1].  cost(node) = in-degree + row_cost;
    row_cost = 0 balances the edges alone (the default: under a locality-free node order every range then also holds ~N/world
    rows); a positive row_cost (in edge units) keeps a degree-ordered or BFS-ordered graph from handing one rank most rows."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    N = len(rowptr) - 1
    cost = rowptr.astype(np.float64) + row_cost * np.arange(N + 1, dtype=np.float64)
    cuts = np.searchsorted(cost, np.floor(cost[-1] * np.arange(1, world, dtype=np.float64) / world), side="left")
    bounds = np.concatenate([[0], np.clip(cuts, 0, N), [N]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def halo_report(rowptr, col, world, row_cost=0.0):
    """Per rank of a `world`-way partition: (own rows, halo rows = distinct remote sources, edges, edges with an own source).
    numpy only - what tools/halo_report.py prints and tests/test_sharded.py bounds."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    b = partition_bounds(rowptr, world, row_cost)
    out = []
    for r in range(world):
        lo, hi = int(b[r]), int(b[r + 1])
        c = np.asarray(col[int(rowptr[lo]):int(rowptr[hi])])
        own = (c >= lo) & (c < hi)
        out.append((hi - lo, int(len(np.unique(c[~own]))), int(len(c)), int(own.sum())))
    return out


def _backend(group=None):
    return dist.get_backend(group) if dist.is_initialized() else None


def all_to_all_rows(send, send_counts, recv_counts, group=None):
    """All-to-all-v of row blocks: rows [sum(send_counts[:q]) ..) of `send` go to rank q."""
    width = send.shape[1:]
    n_recv = int(sum(recv_counts))
    if _backend(group) == "gloo" and send.is_cuda:      # test path: stage through the host
        r = torch.empty((n_recv,) + tuple(width), dtype=send.dtype)
        dist.all_to_all_single(r, send.cpu(), [int(c) for c in recv_counts], [int(c) for c in send_counts], group=group)
        return r.to(send.device)
    recv = torch.empty((n_recv,) + tuple(width), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send.contiguous(), [int(c) for c in recv_counts], [int(c) for c in send_counts], group=group)
    return recv


class ExchangeLog:
    """What the halo exchanges of this rank moved and how long they took (bench.py's per-rank diagnosis, round-3 VERDICT item 3).
    bytes: payload rows x row bytes, by direction, counted at every all_to_all_rows_start().  With `timed` set (bench.py, during the
    timed steps) every exchange also gets two events: e0 on the compute stream where the collective is enqueued (its send buffer
    is packed by then), e1 on a SIDE stream that waits for nothing but the collective - so e0 -> e1 is the exchange itself
    (launch to last byte received, waiting for slower peers included), whatever compute the layer queues meanwhile."""

    def __init__(self):
        self.reset()
        self.timed = False
        self._side = {}

    def reset(self):
        self.calls = 0
        self.bytes_sent = 0
        self.bytes_received = 0
        self.events = []

    def side_stream(self, device):
        key = str(device)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def exchange_ms(self):
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.events)


EXCHANGE_LOG = ExchangeLog()


class _A2AHandle:
    """An all-to-all-v in flight.  NCCL/RCCL: runs on the communicator's own stream beside the compute stream;
    wait() makes the compute stream wait for it (no host sync).  gloo (tests): already done when returned."""

    def __init__(self, work, recv, keep, done=None):
        self.work, self.recv, self.keep, self.done = work, recv, keep, done

    def wait(self):
        if self.done is not None:                # timed form: the side stream already waits for the collective; wait for ITS event
            torch.cuda.current_stream().wait_event(self.done)
            self.done = self.work = None
        elif self.work is not None:
            self.work.wait()
            self.work = None
        self.keep = None
        return self.recv


def all_to_all_rows_start(send, send_counts, recv_counts, group=None, out=None):
    """Start an all-to-all-v of row blocks; `out` (n_recv, W) may be a row slice of a larger buffer."""
    n_recv = int(sum(recv_counts))
    log = EXCHANGE_LOG
    row_bytes = send.element_size() * int(np.prod(send.shape[1:])) if send.dim() > 1 else send.element_size()
    log.calls += 1
    log.bytes_sent += int(sum(send_counts)) * row_bytes
    log.bytes_received += n_recv * row_bytes
    timed = log.timed and send.is_cuda
    if timed:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    if _backend(group) == "gloo" and send.is_cuda:      # test path: synchronous, staged through the host
        r = all_to_all_rows(send, send_counts, recv_counts, group)
        if out is not None:
            out.copy_(r)
            r = out
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            log.events.append((e0, e1))
        return _A2AHandle(None, r, None)
    recv = out if out is not None else torch.empty((n_recv,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    send = send.contiguous()
    work = dist.all_to_all_single(recv, send, [int(c) for c in recv_counts], [int(c) for c in send_counts], group=group,
                                  async_op=True)
    if not timed:
        return _A2AHandle(work, recv, send)
    # e1 on a side stream that waits for the collective only (Work.wait() blocks the CURRENT stream, here the side stream); the
    # compute stream later waits for e1 - the same dependency as work.wait() on it, one event further
    side = log.side_stream(send.device)
    e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        work.wait()
        e1.record()
    log.events.append((e0, e1))
    return _A2AHandle(work, recv, send, done=e1)


def all_reduce_sum(t, group=None):
    if _backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def allreduce_grads(params, group=None, average=False, bucket_bytes=256 << 20):
    """Sum (or average) the gradients of `params` over the ranks in as few collectives as possible: the gradients are
    packed into flat fp32 buckets of at most `bucket_bytes` (xGMI rings are per-link bound, so few large messages beat
    many small ones; the hot path's parameter sets are a few MB and travel as ONE bucket).  Used by the node-sharded NC
    layer (ShardedMMA.allreduce_grads) and by data-parallel replicas of the graph-regression path, whose molecule batches
    are independent: no halo, only this exchange."""
    grads = [p.grad for p in params if getattr(p, "grad", None) is not None]
    if not grads:
        return 0
    world = dist.get_world_size(group)
    n_buckets, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (j == i or size + grads[j].numel() * 4 <= bucket_bytes):
            size += grads[j].numel() * 4
            j += 1
        bucket = grads[i:j]
        flat = torch.cat([g.reshape(-1) for g in bucket]) if len(bucket) > 1 else bucket[0].reshape(-1).clone()
        all_reduce_sum(flat, group)
        if average:
            flat.div_(world)
        torch._foreach_copy_(bucket, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in bucket]), bucket)])
        n_buckets += 1
        i = j
    return n_buckets


class HaloPlan:
    """Who sends which rows to whom, for one rank.  Built once per (graph, partition) with one index exchange."""

    def __init__(self, rowptr_own, col_global, bounds, rank, world, comm_device="cpu", group=None, plan_device=None):
        """plan_device (a cuda device): the shard's heavy index work - the distinct remote sources (sort-unique), the local column
        ids (binary search), the reverse-exchange lists (stable radix sort) - runs on the GPU and the local CSR stays there
        (`rowptr_dev`, `col_dev`: ShardedMMA then builds its NCGraph / SpmmGraph plans on the device too).  Default: host numpy."""
        bounds = np.asarray(bounds, dtype=np.int64)
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
        self.n_own = self.hi - self.lo
        rowptr_own = np.asarray(rowptr_own, dtype=np.int64)
        assert len(rowptr_own) == self.n_own + 1 and rowptr_own[0] == 0 and rowptr_own[-1] == len(col_global)
        self.rowptr_dev = self.col_dev = None
        self.plan_device = torch.device(plan_device) if plan_device is not None else None
        if self.plan_device is not None and self.plan_device.type == "cuda":
            pd = self.plan_device
            cg = torch.from_numpy(np.ascontiguousarray(col_global, dtype=np.int64)).to(pd)
            own = (cg >= self.lo) & (cg < self.hi)
            halo_t = torch.unique(cg[~own])                                # sorted => grouped by owner
            col_local = torch.where(own, cg - self.lo, self.n_own + torch.searchsorted(halo_t, cg))
            self.halo_ids = halo_t.cpu().numpy()
            self.col_dev, self.rowptr_dev = col_local, torch.from_numpy(rowptr_own).to(pd)
            self.rowptr, self.col = rowptr_own, None                      # the local columns live on the device only
        else:
            col_global = np.asarray(col_global, dtype=np.int64)
            own = (col_global >= self.lo) & (col_global < self.hi)
            self.halo_ids = np.unique(col_global[~own])                   # ascending => grouped by owner
            # local column ids: own rows first, then halo rows in halo_ids order
            col_local = np.where(own, col_global - self.lo, 0)
            col_local[~own] = self.n_own + np.searchsorted(self.halo_ids, col_global[~own])
            self.rowptr, self.col = rowptr_own, col_local
        self.n_halo = len(self.halo_ids)
        self.n_src = self.n_own + self.n_halo
        owner = np.searchsorted(bounds, self.halo_ids, side="right") - 1
        self.recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
        assert self.recv_counts[rank] == 0
        # index exchange: tell every owner which of its rows this rank needs
        dev = torch.device(comm_device)
        rc = torch.from_numpy(self.recv_counts).to(dev)
        sc = torch.empty_like(rc)
        dist.all_to_all_single(sc, rc, group=group)
        self.send_counts = sc.cpu().numpy().astype(np.int64)
        want = torch.from_numpy(self.halo_ids).to(dev)
        give = torch.empty(int(self.send_counts.sum()), dtype=torch.int64, device=dev)
        dist.all_to_all_single(give, want, [int(c) for c in self.send_counts], [int(c) for c in self.recv_counts], group=group)
        give = give.cpu().numpy()
        assert len(give) == 0 or (give.min() >= self.lo and give.max() < self.hi), "peer asked for rows this rank does not own"
        self.n_total = int(bounds[-1])                                   # global node count (the scalers' quirk Q1 uses N)
        self.send_idx = (give - self.lo).astype(np.int64)               # local row ids, concatenated per peer
        self.send_offsets = np.concatenate([[0], np.cumsum(self.send_counts)]).astype(np.int64)
        self.build_unpack()

    @classmethod
    def from_full_graph(cls, rowptr, col, rank, world, bounds=None):
        """The plan of `rank` of `world` WITHOUT the index exchange, for a caller that holds the whole CSR (tools/shard_sim.py and
        tests/test_c5_rank_gpu.py rehearse one rank of eight on one GPU): what the peers would ask for is read off their rows of the
        full graph.  Same fields as the collective constructor (host numpy form).  Returns (plan, e0 = global position of the shard's
        first edge: the dropout hash is keyed by global edge ids)."""
        rowptr = np.asarray(rowptr, dtype=np.int64)
        bounds = partition_bounds(rowptr, world) if bounds is None else np.asarray(bounds, dtype=np.int64)
        p = object.__new__(cls)
        p.rank, p.world, p.group = rank, world, None
        p.plan_device = p.rowptr_dev = p.col_dev = None
        p.lo, p.hi = int(bounds[rank]), int(bounds[rank + 1])
        p.n_own = p.hi - p.lo
        e0, e1 = int(rowptr[p.lo]), int(rowptr[p.hi])
        cg = np.asarray(col[e0:e1], dtype=np.int64)
        own = (cg >= p.lo) & (cg < p.hi)
        p.halo_ids = np.unique(cg[~own])
        p.n_halo = len(p.halo_ids)
        p.n_src = p.n_own + p.n_halo
        owner = np.searchsorted(bounds, p.halo_ids, side="right") - 1
        p.recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
        cl = np.where(own, cg - p.lo, 0)
        cl[~own] = p.n_own + np.searchsorted(p.halo_ids, cg[~own])
        p.rowptr, p.col = rowptr[p.lo:p.hi + 1] - e0, cl
        sc, give = [], []
        for q in range(world):                                   # the rows of this rank every peer reads
            if q == rank:
                sc.append(0)
                continue
            lo, hi = int(bounds[q]), int(bounds[q + 1])
            cq = np.asarray(col[int(rowptr[lo]):int(rowptr[hi])], dtype=np.int64)
            need = np.unique(cq[(cq >= p.lo) & (cq < p.hi)])
            sc.append(len(need))
            give.append(need)
        p.send_counts = np.array(sc, dtype=np.int64)
        p.send_idx = (np.concatenate(give) - p.lo).astype(np.int64) if give else np.zeros(0, np.int64)
        p.send_offsets = np.concatenate([[0], np.cumsum(p.send_counts)]).astype(np.int64)
        p.n_total = int(bounds[-1])
        p.build_unpack()
        return p, e0

    def build_unpack(self):
        """The reverse exchange in ONE launch: every local row that is sent to anyone, with the positions of its copies in the
        concatenated receive buffer (ascending = by peer rank: a fixed summation order) - mma_unpack_add_rows_csr."""
        pd = getattr(self, "plan_device", None)
        if pd is not None and pd.type == "cuda" and len(self.send_idx):
            # the same lists from ONE stable radix sort of the send positions by local row (K6): rows with at least one copy, the
            # segment of each in the sorted order, the positions themselves
            t = Fn.DeviceCSR(torch.from_numpy(self.send_idx).to(pd), None, max(self.n_own, 1))
            rp = t.rowptr.long()
            cnt = rp[1:] - rp[:-1]
            rows = torch.nonzero(cnt > 0).flatten()
            self.unpack_rows = rows.cpu().numpy().astype(np.int64)
            self.unpack_segptr = torch.cat([rp[rows], rp[-1:]]).cpu().numpy().astype(np.int64)
            self.unpack_pos = t.perm[:len(self.send_idx)].cpu().numpy().astype(np.int64)
            return
        order = np.argsort(self.send_idx, kind="stable")
        self.unpack_rows, first = np.unique(self.send_idx[order], return_index=True)
        self.unpack_segptr = np.concatenate([first, [len(order)]]).astype(np.int64)
        self.unpack_pos = order.astype(np.int64)


def unpack_add(mod, rows, dst):
    """dst[r] += the rows received for r (all peers), one launch, fixed order."""
    W = rows.shape[1]
    with Fn._span("halo_unpack"):
        call("mma_unpack_add_rows_csr", ptr(rows), rows.stride(0), ptr(mod.unpack_rows), ptr(mod.unpack_segptr), ptr(mod.unpack_pos),
             mod.unpack_rows.shape[0], ptr(dst), dst.stride(0), W, stream_ptr())


class _ShardedTail(torch.autograd.Function):
    """out = A_own S + A_halo S_halo + bias: the K-stacked SpMM of layers.py:861-865 on a shard, with the exchange of the
    (n, C) tail rows hidden behind the own-source part of the SpMM (forward) and of its transpose (backward).
    forward : pack -> all-to-all of the S rows others read (async) || A_own S + bias ; wait ; += A_halo S_halo
    backward: A_halo^T g -> reverse all-to-all (async) || A_own^T g, column sum for the bias ; wait ; unpack-add."""

    @staticmethod
    def forward(ctx, S, bias, mod):
        require_gpu(S)
        plan, sgo, sgh = mod.plan, mod.sg_own, mod.sg_halo
        S = S.contiguous()
        n, C = S.shape
        dev = S.device
        n_send = int(plan.send_counts.sum())
        send = torch.empty((n_send, C), device=dev, dtype=torch.float32)
        with Fn._span("halo_pack"):
            call("mma_pack_rows", ptr(S), C, ptr(mod.send_idx), n_send, ptr(send), C, C, stream_ptr())
        h = all_to_all_rows_start(send, plan.send_counts, plan.recv_counts, plan.group)
        out = torch.empty((n, C), device=dev, dtype=torch.float32)
        with Fn._span("csr_spmm_fwd"):
            Fn._spmm_call(sgo.rowptr, sgo.col, sgo.val, sgo.items, sgo.hubs, sgo.n_slots, S, sgo.n_cols, 1, bias, out, n, C, sgo.n_wave_items)
        with Fn._span("halo_wait"):
            S_halo = h.wait()
        if plan.n_halo:
            part = torch.empty((n, C), device=dev, dtype=torch.float32)
            with Fn._span("csr_spmm_fwd"):
                Fn._spmm_call(sgh.rowptr, sgh.col, sgh.val, sgh.items, sgh.hubs, sgh.n_slots, S_halo.contiguous(), sgh.n_cols, 1, None,
                              part, n, C, sgh.n_wave_items)
            out += part
        ctx.mod, ctx.has_bias = mod, bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        plan, sgo, sgh = mod.plan, mod.sg_own, mod.sg_halo
        g = g.contiguous()
        n, C = g.shape
        dev = g.device
        back = None
        if plan.world > 1:        # a collective: every rank joins, with zero rows if it has no halo
            gh = torch.empty((plan.n_halo, C), device=dev, dtype=torch.float32)
            if plan.n_halo:
                with Fn._span("csr_spmm_bwd"):
                    Fn._spmm_call(sgh.t_rowptr, sgh.t_col, sgh.t_val, sgh.t_items, sgh.t_hubs, sgh.t_n_slots, g, n, 1, None, gh, plan.n_halo, C, sgh.t_n_wave_items)
            back = all_to_all_rows_start(gh, plan.recv_counts, plan.send_counts, plan.group)
        gS = torch.empty((n, C), device=dev, dtype=torch.float32)
        with Fn._span("csr_spmm_bwd"):
            Fn._spmm_call(sgo.t_rowptr, sgo.t_col, sgo.t_val, sgo.t_items, sgo.t_hubs, sgo.t_n_slots, g, n, 1, None, gS, n, C, sgo.t_n_wave_items)
        from .dense import col_sum
        gb = col_sum(g) if (ctx.has_bias and n > 0) else (torch.zeros(C, device=dev) if ctx.has_bias else None)
        if back is not None:
            with Fn._span("halo_wait"):
                rows = back.wait()
            unpack_add(mod, rows, gS)
        return gS, gb, None


class _ShardedAggregate(torch.autograd.Function):
    """Fused K-mask aggregate on a shard, with the halo traffic hidden behind independent work.

    forward : pack -> all-to-all of x halo rows (async) || P = x_own Wtop, Q_own = x_own Wbot ;  wait ;  Q_halo = x_halo Wbot ;  K1
    backward: K2b on the HALO sources ; their dL/dx (direct + through Q) -> reverse all-to-all (async) ||
              K2b on the own sources (node-level backward in its epilogue), dL/dx_own, dL/dWtop, dL/dWbot ;  wait ;  unpack-add.
    Memory (round 3): P (n, K*H) and Q (S, K*H) are SEPARATE tables - a halo row has no target role, so its P half (and dL/dP half) was
    never read: one (S, 2*K*H) buffer each way wasted 2 x n_halo x K*H floats per rank (2 x 25 GB at C5, where a rank's halo is ~3x its
    own rows).  The own rows of x are not copied either when the caller's tensor already heads the (S, H) source table
    (ShardedMMA.feature_buffer())."""

    @staticmethod
    def forward(ctx, x_own, wtop, wbot, mod, kinds, acts, drop):
        require_gpu(x_own)
        plan, graph = mod.plan, mod.graph
        n, S, H = plan.n_own, plan.n_src, x_own.shape[1]
        K = len(kinds)
        dev = x_own.device
        x_own = x_own.contiguous()
        x_src = mod.source_table(x_own)                          # (S,H); its first n rows ARE x_own when the caller used feature_buffer()
        n_send = int(plan.send_counts.sum())
        send = torch.empty((n_send, H), device=dev, dtype=torch.float32)
        with Fn._span("halo_pack"):
            call("mma_pack_rows", ptr(x_own), H, ptr(mod.send_idx), n_send, ptr(send), H, H, stream_ptr())
        h = all_to_all_rows_start(send, plan.send_counts, plan.recv_counts, plan.group, out=x_src[n:])
        if x_src.data_ptr() != x_own.data_ptr():
            x_src[:n].copy_(x_own)
        KH = K * H
        P = torch.empty((n, KH), device=dev, dtype=torch.float32)
        Q = torch.empty((S, KH), device=dev, dtype=torch.float32)
        need = any(ctx.needs_input_grad[:3])
        box_own, box_halo = ([], []) if need else (None, None)              # row maxima of x from the forward GEMMs (three-product TN form)
        mm_into(x_own, wtop, P, row_max_box=box_own)
        mm_into(x_own, wbot, Q[:n])
        with Fn._span("halo_wait"):
            h.wait()
        mm_into(x_src[n:], wbot, Q[n:], row_max_box=box_halo)
        msum, T, sel, crow = Fn.nc_fwd_launch(x_src, P, Q, graph, kinds, acts, drop, True, need)
        ctx.mod, ctx.kinds, ctx.acts, ctx.drop = mod, kinds, acts, drop
        ctx.save_for_backward(x_src, P, Q, T, sel, crow, wtop, wbot, box_own[0] if box_own else None, box_halo[0] if box_halo else None)
        return msum

    @staticmethod
    def backward(ctx, g):
        mod, kinds, acts, drop = ctx.mod, ctx.kinds, ctx.acts, ctx.drop
        plan, graph = mod.plan, mod.graph
        x_src, P, Q, T, sel, crow, wtop, wbot, xrm_own, xrm_halo = ctx.saved_tensors
        n, S, H = plan.n_own, plan.n_src, x_src.shape[1]
        K = len(kinds)
        KH = K * H
        dev = g.device
        g = g.contiguous()
        shared = crow is not None
        fuse = shared and Fn.FUSE_NODE_BWD          # K2a in the epilogue of the OWN-source launch (halo sources have no target role)
        gP = torch.empty((n, KH), device=dev, dtype=torch.float32)
        gQ = torch.empty((S, KH), device=dev, dtype=torch.float32)
        from .dense import f16x2_n128_ok, rows_mm_add_scaled_
        # row maxima of gP / gQ for the three-product dL/dx GEMMs (own rows: the larger of both; halo rows: gQ only)
        row_max = torch.zeros((S,), device=dev, dtype=torch.float32) if f16x2_n128_ok(max(n, S - n), KH, H) and K <= 8 else None
        gs = gxs = None
        if not fuse:
            gs, _gP, gxs = Fn.nc_bwd_node_launch(g, True, sel, crow, T, graph, kinds, H, shared, gP=gP, row_max=row_max)
        epi = dict(T=T, gP=gP) if fuse else {}
        gx = torch.empty((S, H), device=dev, dtype=torch.float32)
        partial = (torch.empty((graph.t_n_slots, (K + 1) * H), device=dev, dtype=torch.float32) if graph.t_n_slots else None)
        # The reverse exchange is a COLLECTIVE: every rank of a multi-rank plan joins it, also one that has no halo rows of
        # its own (a directed graph, or an empty shard) - its peers may still owe it gradient rows (send_counts > 0), and a
        # rank that skipped the call would leave them blocked or matched against its next collective.
        back = None
        if S > n:
            halo_part, own_part = graph.t_parts
            Fn.nc_bwd_edges_launch(x_src, P, Q, gs, g, crow, gxs, graph, kinds, acts, drop, gQ, gx, partial, halo_part, row_max=row_max, **epi)
            gxh = rows_mm_add_scaled_(gx[n:], gQ[n:], wbot.t(), row_max[n:] if row_max is not None else None)   # halo rows: direct + via Q
            back = all_to_all_rows_start(gxh, plan.recv_counts, plan.send_counts, plan.group)
            Fn.nc_bwd_edges_launch(x_src, P, Q, gs, g, crow, gxs, graph, kinds, acts, drop, gQ, gx, partial, own_part, row_max=row_max, **epi)
        else:
            if plan.world > 1:
                back = all_to_all_rows_start(gx[n:], plan.recv_counts, plan.send_counts, plan.group)    # sends (0,H), still receives
            Fn.nc_bwd_edges_launch(x_src, P, Q, gs, g, crow, gxs, graph, kinds, acts, drop, gQ, gx, partial, row_max=row_max, **epi)
        rm_own = row_max[:n] if row_max is not None else None
        gx_own = rows_mm_add_scaled_(gx[:n], gP, wtop.t(), rm_own)             # own rows: direct + through P ...
        gx_own = rows_mm_add_scaled_(gx_own, gQ[:n], wbot.t(), rm_own)         # ... + through Q
        gwtop = gwbot = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            rm = row_max                                          # K2b's row maxima also scale the three-product TN form
            gwtop = xt_g(x_src[:n], gP, xrm_own, rm[:n] if rm is not None else None)
            gwbot = xt_g(x_src[:n], gQ[:n], xrm_own, rm[:n] if rm is not None else None)
            if S > n:
                gwbot = gwbot + xt_g(x_src[n:], gQ[n:], xrm_halo, rm[n:] if rm is not None else None)
        if back is not None:
            with Fn._span("halo_wait"):
                rows = back.wait()                                                       # (n_send, H)
            unpack_add(mod, rows, gx_own)
        return gx_own, gwtop, gwbot, None, None, None, None


class ShardedMMA(torch.nn.Module):
    """The MMA layer (layers.py:54-872 semantics, as mma_amd.MMA) on this rank's shard of the graph."""

    def __init__(self, plan, device, H, C, names, masks, weight, bias, dropout, activation="new_sigmoid", edge_base=0,
                 chunk=DEFAULT_CHUNK, adj_val=None, strict_reference=True, scalers=None, compound_scalers=False, avg_d=None,
                 group_below=None, t_group_below=None):
        super().__init__()
        # strict_reference=False: true-degree scalers as in mma_amd.MMA; avg_d must then hold the GLOBAL means {'log','lin'}
        self.strict_reference, self.scaler_names, self.compound_scalers, self.avg_d = strict_reference, scalers, compound_scalers, avg_d
        assert strict_reference or avg_d is not None, "sharded true-degree scalers need the global degree means (avg_d)"
        self._row_factor = None
        self.plan, self.names, self.activation, self.dropout = plan, list(names), activation, dropout
        self.H, self.C = H, C
        self.lo, self.hi = plan.lo, plan.hi
        dev = torch.device(device)
        col_dev = getattr(plan, "col_dev", None)
        if col_dev is not None and adj_val is None:
            # the shard's plans on the device (K6 sorts + scans), from the local CSR the HaloPlan left in HBM
            rp = plan.rowptr_dev.to(dev).long()
            cl = col_dev.to(dev)
            self.graph = NCGraph.from_device_csr(rp, cl, n_src=plan.n_src, chunk=chunk, edge_base=edge_base, H=H, group_below=group_below,
                                                  t_group_below=t_group_below)
            own = cl < plan.n_own
            cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=cl.device), torch.cumsum(own.to(torch.int64), 0)])
            rp_own = cs[rp]                                            # edges with an own source in front of each row
            self.sg_own = SpmmGraph.from_device_csr(rp_own, cl[own], n_cols=plan.n_own)
            self.sg_halo = SpmmGraph.from_device_csr(rp - rp_own, cl[~own] - plan.n_own, n_cols=plan.n_halo)
        else:
            col_np = plan.col if plan.col is not None else col_dev.cpu().numpy()
            self.graph = NCGraph(plan.rowptr, col_np, dev, n_src=plan.n_src, chunk=chunk, edge_base=edge_base, H=H, group_below=group_below,
                                 t_group_below=t_group_below)
            dst = np.repeat(np.arange(plan.n_own, dtype=np.int64), np.diff(plan.rowptr))
            # the tail SpMM in two parts: own sources (runs while the S rows of the halo are on the wire) and halo sources
            own = col_np < plan.n_own
            val = None if adj_val is None else np.asarray(adj_val, dtype=np.float32)
            self.sg_own = SpmmGraph(dst[own], col_np[own], None if val is None else val[own], plan.n_own, plan.n_own, dev)
            self.sg_halo = SpmmGraph(dst[~own], col_np[~own] - plan.n_own, None if val is None else val[~own], plan.n_own, plan.n_halo, dev)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        self.send_idx = i32(plan.send_idx)
        self.unpack_rows, self.unpack_segptr, self.unpack_pos = i32(plan.unpack_rows), i32(plan.unpack_segptr), i32(plan.unpack_pos)
        self.masks = masks          # dict name -> Parameter (2H,H), owned by the caller, identical on every rank
        self.weight, self.bias = weight, bias
        self.owned = [weight, bias] + [masks[n] for n in self.names]
        self.local_edges = int(plan.rowptr[-1])
        self.n_total = plan.n_total  # global node count: the scalers' quirk Q1 evaluates its factor with N, on every rank alike
        self.drop_override = None
        self._x_src = None           # (S,H) source table [own | halo], kept between calls when the caller writes into feature_buffer()

    def feature_buffer(self):
        """A leaf tensor (n_own, H) that HEADS this rank's (S, H) source table: put the shard's features here (copy_ under
        no_grad, or hand it to the producing op as `out=`) and pass it to forward() - the halo rows are then received right behind
        it and K1 reads [own | halo] as one table without the per-call copy of the own rows (0.3 ms per layer call at C5)."""
        if self._x_src is None:
            self._x_src = torch.empty((self.plan.n_src, self.H), device=self.weight.device, dtype=torch.float32)
        return self._x_src[:self.plan.n_own].detach().requires_grad_(True)

    def source_table(self, x_own):
        if self._x_src is not None and x_own.data_ptr() == self._x_src.data_ptr() and x_own.shape[0] == self.plan.n_own:
            return self._x_src
        return torch.empty((self.plan.n_src, x_own.shape[1]), device=x_own.device, dtype=torch.float32)

    @classmethod
    def build(cls, rowptr, col, rank, world, device, H, C, names, dropout, seed=42, chunk=DEFAULT_CHUNK, group=None,
              strict_reference=True, scalers=None, compound_scalers=False, group_below=None, t_group_below=None):
        """Convenience for bench/tests: every rank holds the full CSR and slices its shard; parameters are
        initialised identically on all ranks (layers.py:143-198 distributions)."""
        rowptr = np.asarray(rowptr, dtype=np.int64)
        bounds = partition_bounds(rowptr, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        e0, e1 = int(rowptr[lo]), int(rowptr[hi])
        comm_dev = "cpu" if _backend(group) == "gloo" else device
        plan = HaloPlan(rowptr[lo:hi + 1] - e0, np.asarray(col[e0:e1]), bounds, rank, world, comm_dev, group,
                        plan_device=device if (DEVICE_PLAN and torch.device(device).type == "cuda") else None)
        g = torch.Generator().manual_seed(seed)
        b = 1.0 / np.sqrt(H)
        P = lambda *s: torch.nn.Parameter(((torch.rand(*s, generator=g) * 2 - 1) * b).to(device))
        masks = {n: P(2 * H, H) for n in names}
        weight, bias = P(H, C), P(C)
        avg_d = None
        if not strict_reference:      # PNA's delta over the WHOLE graph (every rank holds the full rowptr here)
            d = np.maximum(np.diff(rowptr), 1).astype(np.float32)
            avg_d = {"log": float(torch.log(torch.from_numpy(d) + 1).mean()), "lin": float(torch.from_numpy(d).mean())}
        return cls(plan, device, H, C, names, masks, weight, bias, dropout, edge_base=e0, chunk=chunk,
                   strict_reference=strict_reference, scalers=scalers, compound_scalers=compound_scalers, avg_d=avg_d,
                   group_below=group_below, t_group_below=t_group_below)

    def _drop(self):
        return self.drop_override if self.drop_override is not None else Fn.DropoutSpec(self.dropout)

    def forward(self, x_own):
        require_gpu(x_own)
        H, K, n = self.H, len(self.names), self.plan.n_own
        kinds = [Fn.KIND[_AGG[a][0]] for a in self.names]
        acts = [Fn.ACT_RAW if (_AGG[a][1] and self.activation == "new_sigmoid") else Fn.ACT_SIGMOID for a in self.names]
        ws = [self.masks[a] for a in self.names]
        msum = _ShardedAggregate.apply(x_own, torch.cat([w[:H] for w in ws], 1), torch.cat([w[H:] for w in ws], 1), self,
                                       tuple(kinds), tuple(acts), self._drop())                 # (n_own, H)
        if self.strict_reference:
            c3 = scaler_row_factor(self.n_total, x_own.device)                         # Q1: identical rows
        else:
            if self._row_factor is None:
                deg = self.graph.rowptr[1:] - self.graph.rowptr[:-1]                    # a rank owns ALL in-edges of its targets
                names_ = self.scaler_names if self.scaler_names is not None else ["identity", "amplification", "attenuation"]
                self._row_factor = true_degree_row_factor(deg, names_, self.compound_scalers, self.avg_d)
            c3 = self._row_factor
        # sum_k A (m_k W) == A ((sum_k m_k) W): only the (n,C) rows travel and enter the SpMM
        S = mm(msum, self.weight) * c3
        return _ShardedTail.apply(S, self.bias, self)

    def allreduce_grads(self):
        """Sum the parameter gradients over the ranks: ONE collective on a flat bucket (six tiny all-reduces cost six
        collective latencies per step, which at 8 ranks is a tenth of the step)."""
        allreduce_grads(self.owned, self.plan.group)
