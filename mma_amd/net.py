"""Model glue for the graph-regression path (reference graph_regression/mma.py:63-127 `Net`): atom / bond embeddings,
4 x (MMAConv 75->75, towers=5, edge_dim=50  +  BatchNorm), global add pooling, MLP head.  No torch_geometric: BatchNorm is
torch's BatchNorm1d (PyG's BatchNorm wraps exactly that) and global_add_pool is an index_add over the batch vector."""
import torch
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Embedding, Linear, ModuleList, ReLU, Sequential

from .mma_conv import CategoricalEdges, MMAConv


def global_add_pool(x, batch, size=None, assume_sorted=False):
    """global_add_pool (mma.py:124) for ANY batch vector, in a fixed summation order.  assume_sorted: the caller guarantees a
    non-decreasing batch vector (GraphedNetStep checks it when a batch is loaded) - the pooled row of a graph is then the sum of one
    contiguous node range and no grouping pass is needed.  Otherwise the nodes are grouped by graph id with the stable K6 sort
    (PyG's global_add_pool accepts an unsorted vector): same kernel, a few more launches, no host sync either way."""
    size = int(batch.max()) + 1 if size is None else size
    if x.is_cuda and x.dtype == torch.float32:
        return _SegmentPool.apply(x, batch, size, bool(assume_sorted))
    return torch.zeros((size, x.shape[1]), device=x.device, dtype=x.dtype).index_add_(0, batch, x)


class _SegmentPool(torch.autograd.Function):
    """Pooling as the K5 segment-sum kernel over the nodes grouped by graph: a fixed order instead of index_add_'s float atomics, whose
    summation order changes from run to run (and between a hipGraph replay and the eager step it was captured from).  Sorted batch
    vectors (PyG mini-batches number their nodes graph by graph) need no grouping: the row pointers are a searchsorted and the columns
    the identity; unsorted ones are grouped by the stable device sort of mma_build_csr (functional.DeviceCSR)."""

    @staticmethod
    def forward(ctx, x, batch, size, assume_sorted):
        from . import functional as Fn
        from ._lib import call, ptr, stream_ptr
        x = x.contiguous()
        N, C = x.shape
        if assume_sorted:
            rowptr = torch.searchsorted(batch, torch.arange(size + 1, device=x.device, dtype=batch.dtype)).to(torch.int32)
            col = torch.arange(N, device=x.device, dtype=torch.int32)
        else:
            if N and not torch.cuda.is_current_stream_capturing():
                torch._assert_async(((batch >= 0) & (batch < size)).all())      # index_add_ would raise for these (device-side, no sync)
            csr = Fn.DeviceCSR(batch.to(torch.int64), None, size)
            rowptr, col = csr.rowptr, csr.perm
        out = torch.empty((size, C), device=x.device, dtype=torch.float32)
        with Fn._span("pool_segsum"):
            call("mma_csr_spmm", ptr(rowptr), ptr(col), None, ptr(x), C, N, 1, None, ptr(out), C, size, C, stream_ptr())
        ctx.save_for_backward(batch)
        return out

    @staticmethod
    def backward(ctx, g):
        batch, = ctx.saved_tensors
        return g.index_select(0, batch), None, None, None


class _EmbedRows(torch.autograd.Function):
    """Embedding lookup (mma.py:87,116 node_emb) whose weight gradient is a fixed-order one-hot TN product (as for the bond-type
    table, functional.GRGraph.type_onehot) instead of torch's scatter: deterministic, so a replayed step equals the eager one."""

    @staticmethod
    def forward(ctx, idx, weight):
        ctx.save_for_backward(idx)
        ctx.n_types = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        from . import dense
        idx, = ctx.saved_tensors
        width = -(-ctx.n_types // 32) * 32
        oh = torch.zeros((idx.numel(), width), device=g.device, dtype=torch.float32)
        oh.scatter_(1, idx.unsqueeze(1), 1.0)
        return None, dense.xt_g(oh, g.contiguous())[:ctx.n_types]


def masked_batch_norm(x, bn, n_valid):
    """BatchNorm1d over the first n_valid rows of x (n_valid: 0-dim device tensor): the padded rows of a static-shape batch neither
    enter the statistics nor the running averages (PyG's BatchNorm = torch's BatchNorm1d on the real nodes, mma.py:97,121).  The
    padded rows are normalised with the same statistics (their values are never read by a real node or graph)."""
    cnt = n_valid.to(torch.float32)
    w = (torch.arange(x.shape[0], device=x.device) < n_valid).to(torch.float32).unsqueeze(1)
    if bn.training or not bn.track_running_stats:
        mean = (x * w).sum(0) / cnt
        d = (x - mean) * w
        var = (d * d).sum(0) / cnt
        if bn.track_running_stats:
            with torch.no_grad():
                m = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - m).add_(mean.detach() * m)
                bn.running_var.mul_(1 - m).add_(var.detach() * (cnt / torch.clamp(cnt - 1, min=1)) * m)
                bn.num_batches_tracked.add_(1)
    else:
        mean, var = bn.running_mean, bn.running_var
    y = (x - mean) * torch.rsqrt(var + bn.eps)
    if bn.affine:
        y = y * bn.weight + bn.bias
    return y


class _MaskedBNReLU(torch.autograd.Function):
    """K17: training-mode BatchNorm1d over the first n_valid rows + ReLU, one launch each way (running statistics updated in place)."""

    @staticmethod
    def forward(ctx, x, n_valid, weight, bias, bn):
        from ._lib import call, ptr, stream_ptr
        x = x.contiguous()
        N, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((C,), device=x.device, dtype=torch.float32)
        rstd = torch.empty((C,), device=x.device, dtype=torch.float32)
        track = bn.track_running_stats
        m = bn.momentum if bn.momentum is not None else 0.1
        call("mma_masked_bn_relu_fwd", ptr(x), C, ptr(n_valid), ptr(weight), ptr(bias), ptr(y), C, ptr(mean), ptr(rstd),
             ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None, ptr(bn.num_batches_tracked) if track else None,
             float(m), float(bn.eps), N, C, 1, stream_ptr())
        ctx.save_for_backward(x, y, mean, rstd, weight, n_valid)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        from ._lib import call, ptr, stream_ptr
        x, y, mean, rstd, weight, n_valid = ctx.saved_tensors
        N, C = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gw = torch.empty((C,), device=x.device, dtype=torch.float32) if weight is not None else None
        gb = torch.empty((C,), device=x.device, dtype=torch.float32) if ctx.has_bias else None
        call("mma_masked_bn_relu_bwd", ptr(gy), C, ptr(y), C, ptr(x), C, ptr(mean), ptr(rstd), ptr(weight), ptr(n_valid), ptr(gx), C,
             ptr(gw), ptr(gb), N, C, 1, stream_ptr())
        return gx, None, gw, gb, None


def masked_bn_relu(x, bn, n_valid):
    """F.relu(batch_norm(x)) with the statistics over the first n_valid rows: the fused K17 kernels in training mode on the GPU, the
    torch formulation (masked_batch_norm) otherwise."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and (bn.training or not bn.track_running_stats) and FUSED_BN:
        return _MaskedBNReLU.apply(x, n_valid, bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn)
    return F.relu(masked_batch_norm(x, bn, n_valid))


FUSED_BN = __import__("os").environ.get("MMA_FUSED_BN", "1") != "0"


class Net(torch.nn.Module):
    categorical_edges = True      # hand the bond types and the embedding table to the layers (same math as the embedded rows)

    def __init__(self, aggregator_list, scaler_list, deg, mask=True, layers=4):
        super().__init__()
        self.node_emb = Embedding(21, 75)
        self.edge_emb = Embedding(4, 50)
        self.convs = ModuleList()
        self.batch_norms = ModuleList()
        for _ in range(layers):                                   # mma.py:91-97 (4 layers, towers=5, hard-coded)
            self.convs.append(MMAConv(in_channels=75, out_channels=75, aggregators=aggregator_list, scalers=scaler_list,
                                      deg=deg, edge_dim=50, towers=5, pre_layers=1, post_layers=1, mask=mask,
                                      divide_input=False))
            self.batch_norms.append(BatchNorm1d(75))
        self.mlp = Sequential(Linear(75, 50), ReLU(), Linear(50, 25), ReLU(), Linear(25, 1))

    def forward(self, x, edge_index, edge_attr, batch, n_valid=None, n_graphs=None):
        """mma.py:103-127.  n_valid / n_graphs (extension, used by train_step.GraphedNetStep): the batch is PADDED to a static shape -
        rows [n_valid, N) are dummy nodes of a dummy graph with id n_graphs; BatchNorm then takes its statistics over the real rows
        only and the pooled output has n_graphs + 1 rows, the last one the dummy graph's."""
        idx = x.squeeze(-1) if x.dim() > 1 else x
        x = _EmbedRows.apply(idx, self.node_emb.weight) if idx.is_cuda else self.node_emb(idx)
        if self.categorical_edges and x.is_cuda and edge_attr.dim() == 1:
            edge_attr = CategoricalEdges(edge_attr, self.edge_emb.weight)     # = self.edge_emb(edge_attr), never materialised
        else:
            edge_attr = self.edge_emb(edge_attr)
        for conv, batch_norm in zip(self.convs, self.batch_norms):
            h = conv(x, edge_index, edge_attr)
            x = F.relu(batch_norm(h)) if n_valid is None else masked_bn_relu(h, batch_norm, n_valid)
        # the padded step (n_graphs given) takes the contiguous-range pooling, which needs a SORTED batch vector: GraphedNetStep.load checks
        # its callers' batches; any other caller of the padded forward is checked here (device-side assert, no host sync) whenever the
        # stream is not being captured (round-4 ADVICE: an unsorted vector would silently pool wrong sums)
        if n_graphs is not None and batch.is_cuda and batch.numel() > 1 and not torch.cuda.is_current_stream_capturing():
            torch._assert_async((batch[1:] >= batch[:-1]).all())
        x = global_add_pool(x, batch, None if n_graphs is None else n_graphs + 1, assume_sorted=n_graphs is not None)
        return self.mlp(x)
