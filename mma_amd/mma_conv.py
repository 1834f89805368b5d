"""Drop-in `MMAConv` (reference graph_regression/mma_conv.py:20-201, a PNAConv fork on PyG MessagePassing).

Same constructor, `forward(x, edge_index, edge_attr=None)`, `message`, `aggregate(inputs, index, dim_size=None)`,
public `dropout` attribute and error behaviour, without torch_geometric / torch_scatter.  forward() runs the fused
HIP path: two dense GEMMs ([U | V] = x [W_i | W_j]^T + [b | 0] for all towers at once, Z = e (W_e W_enc)^T + W_e b_enc with
the edge encoder folded in) and ONE kernel that forms the per-edge message U[i] + V[j] + Z[e], applies the always-on
dropout and reduces all K aggregators with their degree scalers (K3); the post-NN runs as one batched GEMM forward and
the K9 kernel backward; `aggregate()` on given messages uses the same K3.  Reference quirks kept:
  G1  only the LAST aggregator's pre_nns is applied; all K aggregators reduce the same message;
  G2  pre_nns is a plain dict of ModuleLists => mask Linears are unregistered (not in parameters()/state_dict);
  G3  reset_parameters() iterates dict keys => no-op for pre_nns;   G4  dropout p=0.5 hard-coded, always on;
  G5  the whole aggregator string goes to scatter => only sum|mean|min|max are valid through forward();
  G6  var/std reachable only through aggregate();   G7  scalers compound;   G8  avg_deg from the histogram values;
  G11 x replicated across towers (shared GEMM);   G13 deg.clamp_(1), empty targets give 0.
"""
from typing import Dict, List, Optional

import math

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn import ModuleList, ReLU, Sequential

from . import dense
from . import functional as Fn
from ._lib import require_gpu
from .mask_aggr import MaskAggregateLinear
from .pyg_compat import Linear, reset

_SCATTER_REDUCE = ("sum", "mean", "min", "max")
FACTOR_SCALERS = __import__("os").environ.get("MMA_FACTOR_SCALERS", "1") != "0"     # 0: materialise `out` (N,T,S*K*F) as in round 2


class CategoricalEdges:
    """Edge features that are rows of a small table: `edge_attr = table[types]`, as the reference's Net makes them from bond types
    (mma.py:88 `Embedding(4, 50)`, :103 `edge_attr = self.edge_emb(edge_attr)`).  Passing this object as MMAConv's `edge_attr`
    is the same computation as passing `table[types]` - an opt-in extension of the drop-in surface: the fused kernels then read
    an (n_types, T*F) table and one byte per edge instead of an (E, T*F) stream, the edge GEMM shrinks from E rows to n_types,
    and the table's gradient is a fixed-order one-hot reduction instead of an index_add."""

    def __init__(self, types, table):
        assert types.dim() == 1 and table.dim() == 2 and table.shape[0] <= 256
        self.types, self.table = types, table
        self._by_pos = None

    def dense(self):
        return self.table.index_select(0, self.types.long())

    def types_by_position(self, graph):
        """(E,) uint8 in target-sorted position order (the order K3/K4 walk the edges in)."""
        key = (id(graph), self.types.data_ptr(), self.types._version)
        if self._by_pos is None or self._by_pos[0] != key:
            if self.types.numel() and not torch.cuda.is_current_stream_capturing():
                # an out-of-range type would read past the table inside the kernel: device-side check, no sync (skipped inside a
                # hipGraph capture, whose eager warm-up has just run it on the same buffers)
                torch._assert_async(((self.types >= 0) & (self.types < self.table.shape[0])).all())
            t = self.types.to(torch.uint8)
            self._by_pos = (key, t if graph.E == 0 else t.index_select(0, graph.perm_long).contiguous())
        return self._by_pos[1]


class _WeightPlan:
    """The block table of K18 (mma_pack_blocks) for one MMAConv: which (row, column) block of which Parameter lands where in the padded
    matrices the fused path takes, and the same blocks addressed inside ONE flat gradient buffer for the way back."""

    def __init__(self, params, blocks, outs, device, n_acc=0):
        # params: the source Parameters; blocks: (param index, column offset, rows, cols, out index, b_off, ldb, b_rows, b_cols);
        # outs: shapes of the packed outputs (index = out index)
        # n_acc: the first n_acc params are the UNREGISTERED mask Linears (G2).  Nobody zeroes their .grad (optimizer.zero_grad() does
        # not know them), so it accumulates for the whole run - here inside the unpack launch, onto one persistent buffer whose views
        # are their .grad, instead of one add launch per tensor per step
        self.ptrs = tuple(q.data_ptr() for q in params)
        self.shapes = [tuple(q.shape) for q in params]
        self.n_acc = n_acc
        self.g_off = [0]
        for q in params:
            self.g_off.append(self.g_off[-1] + q.numel())
        self.acc = torch.zeros((self.g_off[n_acc],), device=device, dtype=torch.float32) if n_acc else None
        self.acc_views = [self.acc[self.g_off[i]:self.g_off[i + 1]].view(self.shapes[i]) for i in range(n_acc)]
        self.params = list(params)
        fwd, bwd = [], []
        for (pi, c0, rows, cols, oi, b_off, ldb, b_rows, b_cols) in blocks:
            q = params[pi]
            assert q.is_contiguous() and q.dtype == torch.float32
            lda = q.shape[-1]
            fwd.append([q.data_ptr() + 4 * c0, lda, rows, cols, oi, b_off, ldb, b_rows, b_cols, 0])
            if pi < n_acc:
                bwd.append([self.acc.data_ptr() + 4 * (self.g_off[pi] + c0), lda, rows, cols, oi, b_off, ldb, b_rows, b_cols, 3])
            else:
                bwd.append([self.g_off[pi] - self.g_off[n_acc] + c0, lda, rows, cols, oi, b_off, ldb, b_rows, b_cols, 0])
        self.n_blocks = len(blocks)
        self.used = {blk[4] for blk in blocks}                                # outputs some block lands in (the rest are placeholders)
        self.fwd = torch.tensor(fwd, dtype=torch.int64).to(device)
        self.bwd = torch.tensor(bwd, dtype=torch.int64).to(device)
        self.outs = outs
        self.o_off = [0]
        for shp in outs:
            n = 1
            for d in shp:
                n *= d
            self.o_off.append(self.o_off[-1] + (n + 3) // 4 * 4)          # 16-byte aligned outputs


class _PackWeights(torch.autograd.Function):
    """The padded matrices of the fused path from the layer's Parameters, one launch; their gradients back into the Parameters'
    layout, one launch (K18) - instead of stack / slice / pad / cat and the zero-fill + copy + add each of those costs in backward."""

    @staticmethod
    def forward(ctx, plan, *params):
        from ._lib import call, ptr, stream_ptr
        flat = torch.empty((plan.o_off[-1],), device=params[0].device, dtype=torch.float32)
        outs = [flat[plan.o_off[i]:plan.o_off[i] + math.prod(shp)].view(shp) for i, shp in enumerate(plan.outs)]
        b = [ptr(o) for o in outs] + [None] * (8 - len(outs))
        call("mma_pack_blocks", ptr(plan.fwd), plan.n_blocks, None, *b, 0, stream_ptr())
        ctx.plan = plan
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        from ._lib import call, ptr, stream_ptr
        plan = ctx.plan
        gouts = [(torch.zeros(shp, device=plan.fwd.device) if i in plan.used else None) if g is None else g.contiguous()
                 for i, (g, shp) in enumerate(zip(gouts, plan.outs))]
        n_acc, base = plan.n_acc, plan.g_off[plan.n_acc]
        for q, view in zip(plan.params[:n_acc], plan.acc_views):      # the accumulating .grad of the unregistered Linears
            if q.grad is not view:
                if q.grad is None:
                    view.zero_()                                       # somebody reset it: start again from zero
                else:
                    view.copy_(q.grad)                                 # somebody else's running sum: carry it over
                q.grad = view
        flat = torch.empty((plan.g_off[-1] - base,), device=plan.fwd.device, dtype=torch.float32)
        b = gouts + [None] * (8 - len(gouts))
        call("mma_pack_blocks", ptr(plan.bwd), plan.n_blocks, ptr(flat), *b, 1, stream_ptr())
        return (None,) * (1 + n_acc) + tuple(flat[plan.g_off[i] - base:plan.g_off[i + 1] - base].view(plan.shapes[i])
                                             for i in range(n_acc, len(plan.shapes)))


class _EdgeFold(torch.autograd.Function):
    """K19: (wz, bz) = (We @ Wenc, We @ benc) - the edge encoder folded into the pre-NN's edge block - in one launch, and
    (gWe, gWenc, gbenc) back in one (as torch ops: a GEMM + a GEMV forward, five launches backward, ~5 us each)."""

    @staticmethod
    def forward(ctx, We, Wenc, benc):
        from ._lib import call, ptr, stream_ptr
        We = We if We.stride(1) == 1 else We.contiguous()
        Wenc, benc = Wenc.contiguous(), benc.contiguous()
        TF, F_ = We.shape
        ED = Wenc.shape[1]
        wz = torch.empty((TF, ED), device=We.device, dtype=torch.float32)
        bz = torch.empty((TF,), device=We.device, dtype=torch.float32)
        call("mma_edge_fold_fwd", ptr(We), We.stride(0), ptr(Wenc), ptr(benc), ptr(wz), ptr(bz), TF, F_, ED, stream_ptr())
        ctx.save_for_backward(We, Wenc, benc)
        return wz, bz

    @staticmethod
    def backward(ctx, gwz, gbz):
        from ._lib import call, ptr, stream_ptr
        We, Wenc, benc = ctx.saved_tensors
        TF, F_ = We.shape
        ED = Wenc.shape[1]
        gwz = torch.zeros((TF, ED), device=We.device) if gwz is None else gwz.contiguous()
        gbz = torch.zeros((TF,), device=We.device) if gbz is None else gbz.contiguous()
        gWe = torch.empty((TF, F_), device=We.device, dtype=torch.float32)
        gWenc = torch.empty((F_, ED), device=We.device, dtype=torch.float32)
        gbenc = torch.empty((F_,), device=We.device, dtype=torch.float32)
        call("mma_edge_fold_bwd", ptr(We), We.stride(0), ptr(Wenc), ptr(benc), ptr(gwz), ptr(gbz), ptr(gWe), ptr(gWenc), ptr(gbenc), TF, F_, ED,
             stream_ptr())
        return gWe, gWenc, gbenc


EDGE_FOLD = __import__("os").environ.get("MMA_EDGE_FOLD", "1") != "0"
PACK_WEIGHTS = __import__("os").environ.get("MMA_PACK_WEIGHTS", "1") != "0"
ACCUMULATE_UNREGISTERED = __import__("os").environ.get("MMA_ACC_UNREGISTERED", "1") != "0"


class MMAConv(torch.nn.Module):
    def __init__(self, in_channels: int, out_channels: int, aggregators: List[str], scalers: List[str], deg: Tensor,
                 edge_dim: Optional[int] = None, towers: int = 1, pre_layers: int = 1, post_layers: int = 1,
                 mask: bool = True, divide_input: bool = False, **kwargs):
        kwargs.setdefault('aggr', None)      # accepted for signature compatibility with MessagePassing(node_dim=0, **kwargs)
        super().__init__()
        assert out_channels % towers == 0
        assert not divide_input or in_channels % towers == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggregators, self.scalers = aggregators, scalers
        self.edge_dim, self.towers, self.divide_input = edge_dim, towers, divide_input
        self.mask, self.dropout = mask, 0.5                                     # G4
        self.pre_layers, self.post_layers = pre_layers, post_layers
        self.F_in = in_channels // towers if divide_input else in_channels
        self.F_out = out_channels // towers

        hist = deg.to(torch.float)                                              # G8: statistics of the histogram VALUES
        self.avg_deg: Dict[str, float] = {'lin': float(hist.mean()), 'log': float(torch.log(hist + 1).mean()),
                                          'exp': float(torch.exp(hist).mean())}
        if edge_dim is not None:
            self.edge_encoder = Linear(edge_dim, self.F_in)

        # module creation order = the reference's (tower-major: K pre stacks, then the tower's post stack)
        self.pre_nns = {aggr: ModuleList() for aggr in aggregators}             # plain dict (G2)
        self.post_nns = ModuleList()
        for _ in range(towers):
            for aggr in aggregators:
                self.pre_nns[aggr].append(self._pre_stack(aggr))
            self.post_nns.append(self._post_stack())
        self.lin = Linear(out_channels, out_channels)
        self.reset_parameters()

        self.drop_override = None     # tests: a functional.DropoutSpec with a fixed seed
        self.graph_capturable = False  # True: the dropout seed is re-drawn on the device each call (hipGraph replays)
        self._seeds = None
        self._seed_buf = None
        self._wplan = None

    def _pre_stack(self, aggr):
        """Sequential(MaskAggregateLinear([x_i|x_j|e] -> F_in), (ReLU, MaskAggregateLinear) x (pre_layers-1))."""
        def mal(width):
            return MaskAggregateLinear(width, self.F_in, self.aggregators, aggr, mask=self.mask)
        blocks = [mal((3 if self.edge_dim else 2) * self.F_in)]
        for _ in range(1, self.pre_layers):
            blocks.extend((ReLU(), mal(self.F_in)))
        return Sequential(*blocks)

    def _post_stack(self):
        """Sequential(Linear([x | K*S aggregates] -> F_out), (ReLU, Linear) x (post_layers-1))."""
        width = (1 + len(self.aggregators) * len(self.scalers)) * self.F_in
        blocks = [Linear(width, self.F_out)]
        for _ in range(1, self.post_layers):
            blocks.extend((ReLU(), Linear(self.F_out, self.F_out)))
        return Sequential(*blocks)

    def reset_parameters(self):
        if self.edge_dim is not None:
            self.edge_encoder.reset_parameters()
        for key in self.pre_nns:        # the KEYS are strings; walking their characters resets nothing (G3)
            for ch in key:
                reset(ch)
        for stack in self.post_nns:
            reset(stack)
        self.lin.reset_parameters()

    def unregistered_parameters(self):
        """The Parameters of the mask Linears, which live in plain dicts (G2): absent from parameters() / state_dict(), never stepped by
        the optimizer, their .grad never zeroed by optimizer.zero_grad()."""
        out = []
        for stacks in self.pre_nns.values():
            for seq in stacks:
                for m in seq:
                    for lin in getattr(m, "aggregation_layers", {}).values():
                        if lin is not None:
                            out.extend(lin.parameters())
        return out

    # ---- graph plan ---------------------------------------------------------------------------------------
    def _graph(self, edge_index, N):
        return Fn.gr_graph(edge_index, N)        # shared by all layers that see this edge_index (mma.py:91-97: four of them)

    def _check_aggregators(self):
        for aggregator in self.aggregators:     # mma_conv.py:153-154 accepts a prefix match, torch_scatter then rejects
            if aggregator not in _SCATTER_REDUCE:   # anything but the four exact names (G5): same ValueError either way
                raise ValueError(f'Unknown aggregator "{aggregator}".')

    def _fusable(self):
        return self.pre_layers == 1 and self.mask != "no_linear"

    def _drop(self, device=None):
        if self.drop_override is not None:
            return self.drop_override
        if self.graph_capturable and device is not None:
            if self._seeds is None or self._seeds.device != device:
                self._seeds = Fn.DeviceSeeds(1, device)
                self._seed_buf = self._seeds.seeds
            return Fn.DropoutSpec(self.dropout, seed_tensor=self._seeds.advance())
        return Fn.DropoutSpec(self.dropout)

    # ---- weight plumbing (K18) ----------------------------------------------------------------------------
    def _packed_weights(self, lins, has_edge, device, plan_only=False):
        """(Wij (2*T*Fw, F), b2 (2*T*Fw,) | None, We (T*Fw, F) | None, Wx (T*F_out, F), Wo (T, F_out, K*S*Fw), bp (T*F_out,)) - the
        matrices forward() builds from the per-tower Linears, zero padding included - in one launch (and one for all their gradients)."""
        T, Fi, Fw, Fo = self.towers, self.F_in, self.fused_width(), self.F_out
        KS = len(self.aggregators) * len(self.scalers)
        TF = T * Fw
        posts = [seq[0] for seq in self.post_nns]
        has_b = lins[0].bias is not None
        params = [l.weight for l in lins] + ([l.bias for l in lins] if has_b else []) + [q.weight for q in posts] + [q.bias for q in posts]
        key = (str(device), has_edge, tuple(q.data_ptr() for q in params))
        if self._wplan is None or self._wplan[0] != key:
            iw, ib = 0, T
            ipw = T + (T if has_b else 0)
            ipb = ipw + T
            outs = [(2 * TF, Fi), (2 * TF,) if has_b else (4,), (TF, Fi) if has_edge else (4,), (T * Fo, Fi), (T, Fo, KS * Fw), (T * Fo,)]
            blocks = []
            for t in range(T):
                blocks.append((iw + t, 0, Fi, Fi, 0, t * Fw * Fi, Fi, Fw, Fi))                       # Wi rows of tower t
                blocks.append((iw + t, Fi, Fi, Fi, 0, (TF + t * Fw) * Fi, Fi, Fw, Fi))               # Wj
                if has_edge:
                    blocks.append((iw + t, 2 * Fi, Fi, Fi, 2, t * Fw * Fi, Fi, Fw, Fi))              # We
                if has_b:
                    blocks.append((ib + t, 0, 1, Fi, 1, t * Fw, Fw, 1, Fw))                          # the bias lands in U only
                blocks.append((ipw + t, 0, Fo, Fi, 3, t * Fo * Fi, Fi, Fo, Fi))                      # Wx
                for ks in range(KS):
                    blocks.append((ipw + t, (1 + ks) * Fi, Fo, Fi, 4, t * Fo * KS * Fw + ks * Fw, KS * Fw, Fo, Fw))
                blocks.append((ipb + t, 0, 1, Fo, 5, t * Fo, Fo, 1, Fo))
            if has_b:
                blocks.append((0, 0, 0, 0, 1, TF, TF, 1, TF))                                        # V's half of the bias: zeros
            self._wplan = (key, _WeightPlan(params, blocks, outs, device, n_acc=ipw if ACCUMULATE_UNREGISTERED else 0))
        if plan_only:
            return self._wplan[1]
        Wij, b2, We, Wx, Wo, bp = _PackWeights.apply(self._wplan[1], *params)
        return Wij, (b2 if has_b else None), (We if has_edge else None), Wx, Wo, bp

    # ---- forward ------------------------------------------------------------------------------------------
    def forward(self, x: Tensor, edge_index, edge_attr: Optional[Tensor] = None) -> Tensor:
        require_gpu(x)
        T, Fi = self.towers, self.F_in
        # G11: without divide_input every tower sees the same x; a stride-0 view stands in for the reference's repeat()
        x2 = None if self.divide_input else x.view(-1, Fi)       # the shared rows themselves: `x[:, 0]` of the expanded view would
        x = x.view(-1, T, Fi) if self.divide_input else x2.view(-1, 1, Fi).expand(-1, T, -1)   # cost a zero-filled (N,T,F) in backward
        N = x.shape[0]
        graph = self._graph(edge_index, N)
        Fw = Fi                     # width of one aggregate block inside `out`
        if self._fusable():
            self._check_aggregators()
            Fw = self.fused_width()
            last = self.aggregators[-1]                                         # G1
            lins = [seq[0].active_linear() for seq in self.pre_nns[last]]       # T Linears (F_in, 3F|2F)
            TF = T * Fw

            packed = None
            if PACK_WEIGHTS and self.post_layers == 1 and not self.divide_input and self.post_nns[0][0].bias is not None:
                packed = self._packed_weights(lins, edge_attr is not None, x.device)        # K18: one launch each way
            Wall = None if packed else torch.stack([l.weight for l in lins])    # (T, F, 3F|2F): ONE stack, then three slices

            def rows(lo, hi):       # the T per-tower (F, hi-lo) weight blocks as rows of one (T*Fw, hi-lo) matrix
                return self._pad_dim(Wall[:, :, lo:hi], 1, Fw).reshape(TF, hi - lo)
            has_b = lins[0].bias is not None
            if packed:
                UV = dense.linear_tall(x2, packed[0], packed[1])
            elif self.divide_input:
                Wi, Wj = rows(0, Fi), rows(Fi, 2 * Fi)
                b = self._pad_dim(torch.stack([l.bias for l in lins]), 1, Fw).reshape(TF) if has_b else None   # lands in U only
                U = torch.einsum('ntf,tgf->ntg', x, Wi.view(T, Fw, Fi)).reshape(N, TF)
                V = torch.einsum('ntf,tgf->ntg', x, Wj.view(T, Fw, Fi)).reshape(N, TF)
                UV = torch.cat([dense.bias_add(U, b) if has_b else U, V], 1)
            else:                                                               # towers share x -> ONE GEMM for U | V
                Wi, Wj = rows(0, Fi), rows(Fi, 2 * Fi)
                b = self._pad_dim(torch.stack([l.bias for l in lins]), 1, Fw).reshape(TF) if has_b else None
                UV = dense.linear_tall(x2, torch.cat([Wi, Wj]), torch.cat([b, torch.zeros_like(b)]) if has_b else None)
            Z = z_index = None
            if edge_attr is not None:
                # enc(e) W_e^T = e (W_e W_enc)^T + W_e b_enc: the (E,F) encoding never materialises (mma_conv.py:141-146)
                # The (E, edge_dim) rows are put in target-sorted position order BEFORE the GEMM (50 floats per edge), so that Z
                # - and in backward the (E, T*Fw) message gradients - stream contiguously through K3/K4.
                We, enc = (packed[2] if packed else rows(2 * Fi, 3 * Fi)), self.edge_encoder
                if (EDGE_FOLD and enc.bias is not None and We.is_cuda and We.dtype == torch.float32 and max(We.shape[1], enc.weight.shape[1]) <= 512
                        and enc.weight.dtype == torch.float32):
                    wz, bz = _EdgeFold.apply(We, enc.weight, enc.bias)                   # K19: one launch each way
                else:
                    wz, bz = We @ enc.weight, (We @ enc.bias if enc.bias is not None else None)
                if isinstance(edge_attr, CategoricalEdges):
                    # edge_attr = table[types] (an Embedding: mma.py:88,103): Z has only n_types distinct rows, so the kernels get
                    # the (n_types, T*Fw) table and one byte per edge instead of a (E, T*Fw) stream
                    Z = dense.linear(edge_attr.table, wz, bz)
                    z_index = edge_attr.types_by_position(graph)
                else:
                    # (E, T*Fw) by position; [r5] on the zero-padded path the permutation rides on the pad launch (no permuted copy of edge_attr)
                    Z = dense.linear_tall_rows(edge_attr, graph.perm, graph.inv_perm, wz, bz) if graph.E else None
                    if Z is None:
                        Z = dense.linear_tall(Fn.rows_by_position(edge_attr, graph), wz, bz)
            # Round 3: when the post-NN is a single small Linear, the degree scalers - per-target row factors - move from the
            # aggregates into it (Fn.tower_post): K3 then runs with the identity scaler only and leaves the K UNSCALED aggregates
            # (N,T,K*Fw), a third of `out` at S = 3; the (N,T,S*K*F) tensor of mma_conv.py:196 and its gradient never exist
            factored = FACTOR_SCALERS and self.post_layers == 1 and self.F_out <= 16 and len(self.scalers) <= 5 and \
                all(s_ in Fn.GR_SCALER for s_ in self.scalers) and \
                dense.tower_post_fits(len(self.aggregators) * Fw, len(self.scalers))    # K*Fw <= 512 and the LDS fit: the kernels' own limits
            out = Fn.gr_fused_conv(UV, Z, graph, T, Fw, self.aggregators, ["identity"] if factored else self.scalers,
                                   self.avg_deg['log'], self.avg_deg['lin'], self._drop(x.device), z_by_pos=True, z_index=z_index)
        else:
            factored, packed = False, None
            src, dst = edge_index[0], edge_index[1]
            if isinstance(edge_attr, CategoricalEdges):
                edge_attr = edge_attr.dense()
            hs = self.message(x.index_select(0, dst), x.index_select(0, src), edge_attr)
            out = self.aggregate(hs, dst, N, _graph=graph)

        KS = len(self.aggregators) * len(self.scalers)
        if self.post_layers == 1:
            # post_nns[t](cat[x_t, out_t]) = x_t Wx_t^T + out_t Wo_t^T + b_t  (mma_conv.py:132-134) as ONE strided-batched
            # GEMM over the towers: neither the (N,T,(K*S+1)*F) concatenation nor the per-tower slices are materialised.
            if packed:
                Wx, Wo, bp = packed[3].view(T, self.F_out, Fi), packed[4], packed[5]
            else:
                Wp = torch.stack([seq[0].weight for seq in self.post_nns])               # (T, F_out, (K*S+1)*F_in)
                bp = torch.cat([seq[0].bias for seq in self.post_nns])                   # (T*F_out,)
                Wx = Wp[:, :, :Fi]
                Wo = self._pad_dim(Wp[:, :, Fi:].reshape(T, self.F_out, KS, Fi), 3, Fw).reshape(T, self.F_out, KS * Fw)
            if factored:
                y = Fn.tower_post(out, Wo, graph.by_target.rowptr, self.scalers, self.avg_deg['log'], self.avg_deg['lin']).view(N, T, self.F_out)
            else:
                y = dense.tower_linear(out, Wo)                                          # (N, T, F_out), no copy of `out`
            if self.divide_input:
                y = y + torch.bmm(x.transpose(0, 1), Wx.transpose(1, 2)).transpose(0, 1)
                out = dense.bias_add(y.reshape(N, T * self.F_out), bp)
            else:                                                                        # bias rides on the shared-x GEMM
                # (the addition as K16's epilogue - dense.linear(..., addend=y) - was built and measured in round 5: the kernel's dword-wise
                # epilogue pays 0.157 ms for what the plain kernel + a separate add launch do in 0.092 + 0.027: not used)
                out = y.reshape(N, T * self.F_out) + dense.linear(x2, Wx.reshape(T * self.F_out, Fi), bp)
        else:
            if Fw != Fi:
                out = out.view(N, T, KS, Fw)[..., :Fi].reshape(N, T, KS * Fi)
            out = torch.cat([x, out], dim=-1)
            outs = [nn(out[:, i]) for i, nn in enumerate(self.post_nns)]
            out = torch.cat(outs, dim=1)
        return self.lin(out)

    def fused_width(self):
        """Per-tower feature width inside the fused path: F_in rounded up to 4 floats so that every (node|edge, tower) row
        segment is 16-byte aligned (ZINC's F=75 -> 76).  The pad columns carry zero weights, hence zero messages, zero
        aggregates and zero gradients; `out`'s pad columns meet zero rows of the post-NN weight.  The counter-hash dropout
        stream is indexed by the padded column t*fused_width()+f."""
        return -(-self.F_in // 4) * 4

    @staticmethod
    def _pad_dim(t, dim, size):
        extra = size - t.shape[dim]
        if extra == 0:
            return t
        pad = [0, 0] * (t.dim() - dim)
        pad[-1] = extra
        return F.pad(t, pad)

    def message(self, x_i: Tensor, x_j: Tensor, edge_attr: Optional[Tensor]) -> Tensor:
        """mma_conv.py:138-157 with dense torch ops (only used when the fused path does not apply)."""
        parts = [x_i, x_j]
        if edge_attr is not None:
            enc = self.edge_encoder(edge_attr)
            parts.append(enc.unsqueeze(1).expand(-1, self.towers, -1))
        h = torch.cat(parts, dim=-1)                                            # (E, T, 2F|3F)
        hs = None
        for aggregator in self.aggregators:                                     # every pass overwrites hs: the last wins (G1)
            if not aggregator.startswith(_SCATTER_REDUCE):
                raise ValueError(f'Unknown aggregator "{aggregator}".')
            hs = torch.stack([stack(h[:, t]) for t, stack in enumerate(self.pre_nns[aggregator])], dim=1)
        return F.dropout(hs, self.dropout)

    def aggregate(self, inputs: Tensor, index: Tensor, dim_size: Optional[int] = None, _graph=None) -> Tensor:
        """mma_conv.py:159-196 on given messages (E,T,F): the HIP kernel in given-messages mode."""
        require_gpu(inputs)
        for aggregator in self.aggregators:
            if not (aggregator in _SCATTER_REDUCE or aggregator in ('var', 'std')):
                raise ValueError(f'Unknown aggregator "{aggregator}".')
        for scaler in self.scalers:
            if scaler not in Fn.GR_SCALER:
                raise ValueError(f'Unknown scaler "{scaler}".')
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        graph = _graph
        if graph is None:
            ei = torch.stack([index, index])        # sources are not needed for given messages
            graph = Fn.GRGraph(ei, dim_size)
        return Fn.gr_aggregate(inputs, graph, self.aggregators, self.scalers, self.avg_deg['log'], self.avg_deg['lin'])

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, towers={self.towers}, '
                f'edge_dim={self.edge_dim})')
