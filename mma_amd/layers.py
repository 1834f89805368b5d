"""Drop-in `MMA` / `GraphConvolution` modules: the reference's node_classification/layers.py surface
(same constructor arguments, same public `learnable_*` methods, same `forward(input, adj)`), with the
per-node Python loops replaced by the fused HIP kernels of libmma_amd.so.

Reference behaviour reproduced by default:
  Q1/Q2  all three scalers always on, and - because the reference hands them the sparse adj - every
         factor is 1.0 to an ulp (scalers.py:22-62 via layers.py:856); we compute the same fp32 factors;
  Q5     activation == "new_sigmoid" leaves RAW logits as the mask in mean3/max/min/softmax/softmin;
  Q6     mask dropout is always on (F.dropout default training=True), also in eval();
  Q7-Q9  max/min are element-wise against x_i, mean divides (x_i+s) by d_i, softmax/softmin == s;
  Q14    reset_parameters: weight/bias U(+-1/sqrt(weight.size(0))), masks U(+-1/sqrt(mask.size(1))).
Parameters stay owned by the caller (models.py:17-60 creates them and the optimizer holds them).
"""
import math

import torch
from torch.nn.modules.module import Module

from . import functional as Fn
from .dense import mm
from ._lib import require_gpu
from .graph import DEFAULT_CHUNK, NCGraph, SpmmGraph
from .scalers import SCALERS, scaler_row_factor, true_degree_row_factor

FOLD_ROW_FACTOR = __import__("os").environ.get("MMA_FOLD_ROW_FACTOR", "1") != "0"

# aggregator name -> (combine kind, raw logits under activation == "new_sigmoid")   layers.py:201-728
_AGG = {
    "sum": ("sum", False), "sum2": ("sum", False), "sum3": ("sum", False), "sum4": ("sum", False),
    "mean": ("mean", False), "mean2": ("mean", False), "mean3": ("mean", True), "mean4": ("mean", False),
    "max": ("max", True), "max2": ("max", False), "max3": ("max", False), "max4": ("max", False),
    "min": ("min", True), "min2": ("min", False), "min3": ("min", False), "min4": ("min", False),
    "softmax": ("softmax", True), "softmin": ("softmin", True),
}
_UNUSABLE = ("std", "normalized_mean", "moment_3")   # layers.py:731-851: O(N^2) with a wrong shape / NameError
_MASK_NAMES = ["moment_3", "sum", "sum2", "sum3", "sum4", "mean", "mean2", "mean3", "mean4", "max", "max2", "max3",
               "max4", "min", "min2", "min3", "min4", "softmax", "softmin", "std", "normalized_mean"]


class GraphConvolution(Module):
    """spmm(adj, x @ W) + b  (layers.py:12-51, pygcn).  The SpMM is libmma_amd's CSR kernel."""

    SPARSE_BELOW = 0.10      # feature matrices with fewer non-zeros than this fraction go through the CSR kernel

    def __init__(self, in_features, out_features, weight, bias, device):
        super().__init__()
        self.in_features, self.out_features, self.device = in_features, out_features, device
        self.weight, self.bias = weight, bias
        self._sg = None
        self._xg = None          # (input tensor, version, SpmmGraph of it or None): sparse-feature first layer
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        require_gpu(input if input.layout == torch.strided else self.weight)
        if self._sg is None or self._sg[0] is not adj:
            self._sg = (adj, adj if isinstance(adj, SpmmGraph) else SpmmGraph.from_torch_sparse(adj, self.weight.device))
        xg = self._feature_graph(input)
        # x @ W: a Planetoid feature matrix is ~1 % dense (Cora: 49 k non-zeros of 3.9 M, models.py:54 feeds it to gc1 as a
        # dense FloatTensor) - as a CSR operand of K5 the product reads the non-zeros only, and its weight gradient x^T g is
        # the same kernel over the transposed CSR; dense / differentiable inputs take the GEMM
        support = Fn.csr_spmm(self.weight, None, xg, 1) if xg is not None else mm(input, self.weight)
        return Fn.csr_spmm(support, self.bias, self._sg[1], 1)

    def _feature_graph(self, input):
        """SpmmGraph of a static, sparse enough feature matrix (decided once per tensor: one host sync), else None."""
        if input.layout != torch.strided:
            if self._xg is None or self._xg[0] is not input:
                self._xg = (input, 0, SpmmGraph.from_torch_sparse(input.to_sparse_coo() if input.layout != torch.sparse_coo else input,
                                                                  self.weight.device))
            return self._xg[2]
        if input.requires_grad or input.dim() != 2:
            return None
        if self._xg is None or self._xg[0] is not input or self._xg[1] != input._version:
            nnz = int(torch.count_nonzero(input))
            sg = None
            if nnz < self.SPARSE_BELOW * input.numel():
                idx = input.nonzero(as_tuple=True)
                sg = SpmmGraph(idx[0].cpu().numpy(), idx[1].cpu().numpy(), input[idx].cpu().numpy(), input.shape[0], input.shape[1],
                               input.device)
            self._xg = (input, input._version, sg)
        return self._xg[2]

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class MMA(Module):
    """Multi-Mask Aggregator layer (layers.py:54-872)."""

    def __init__(self, add_all, activation, k, in_features, out_features, weight, bias,
                 weight_moment_3, weight_sum, weight_sum2, weight_sum3, weight_sum4, weight_mean,
                 weight_mean2, weight_mean3, weight_mean4, weight_max, weight_max2, weight_max3,
                 weight_max4, weight_min, weight_min2, weight_min3, weight_min4, weight_softmax,
                 weight_softmin, weight_std, weight_normalized_mean, dropout, aggregator_list, device,
                 chunk=DEFAULT_CHUNK, strict_reference=True, scalers=None, compound_scalers=False, avg_d=None):
        """Positional arguments: the reference's (layers.py:57-61).  Keyword extensions (defaults = reference behaviour):
        strict_reference=False evaluates the degree scalers with the TRUE degrees len(add_all[i]) instead of the degenerate
        factor 1.0 the reference computes (quirk Q1): `scalers` (default identity, amplification, attenuation - the
        reference's three, scalers.py:64) may also name linear / inverse_linear, and compound_scalers=True chains them like
        mma_conv.py:181-196 (BASELINE configs[4]: "K=8 aggregators + all scalers")."""
        super().__init__()
        self.activation = activation
        self.k = k
        self.in_features = in_features
        self.out_features = out_features
        self.add_all = add_all
        self.dropout = dropout
        self.device = device
        self.weight = weight
        self.bias = bias
        masks = [weight_moment_3, weight_sum, weight_sum2, weight_sum3, weight_sum4, weight_mean, weight_mean2,
                 weight_mean3, weight_mean4, weight_max, weight_max2, weight_max3, weight_max4, weight_min,
                 weight_min2, weight_min3, weight_min4, weight_softmax, weight_softmin, weight_std,
                 weight_normalized_mean]
        for name, w in zip(_MASK_NAMES, masks):
            setattr(self, "mask_" + name, w)          # same attribute names as the reference (layers.py:114-134)

        self.all_aggregators = {name: getattr(self, "learnable_" + name) for name in _MASK_NAMES}
        self.AGGREGATORS = dict()
        for aggr in aggregator_list:
            self.AGGREGATORS[aggr] = self.all_aggregators[aggr]   # KeyError on unknown names (layers.py:106)
        self.aggregator_names = list(self.AGGREGATORS)
        self.aggregators = [self.AGGREGATORS[a] for a in self.AGGREGATORS]
        self.scalers = [SCALERS[s] for s in SCALERS]
        self.num_aggregators = len(self.aggregators)

        self.reset_parameters()
        self.avg_d = avg_d          # strict_reference=False: optional {'log','lin'} means (default: over this graph's degrees)
        self.self_loop = None

        self.strict_reference = bool(strict_reference)
        self.scaler_names = list(scalers) if scalers is not None else list(SCALERS)
        self.compound_scalers = bool(compound_scalers)
        if self.strict_reference and (scalers is not None or compound_scalers):
            raise ValueError("scalers / compound_scalers are extensions: pass strict_reference=False to use them")
        self._row_factor = None          # (device, (N,1) tensor) cache of the true-degree factor

        self._chunk = chunk
        # NCGraph, built once (the reference captures add_all at construction time).  Extension: a ready-made
        # NCGraph may be passed as `add_all` (large graphs never materialise a Python list of arrays).
        self._graph = add_all if isinstance(add_all, NCGraph) else None
        self._sg = None         # (adj object, SpmmGraph) cache for the tail spmm
        self._sg_scaled = None  # (SpmmGraph, factor tensor, its copy with the scaler row factor folded into the edge values)
        self.drop_override = None   # tests: a DropoutSpec (explicit keep mask / fixed seed) used instead of p
        # hipGraph capture (torch.cuda.graph) of the layer: the dropout seed then lives in a device buffer that is re-drawn
        # by one captured launch (Fn.DeviceSeeds) on every replay, instead of being baked into the kernel arguments at capture time
        self.graph_capturable = False
        self._seeds = None
        self._seed_buf = None

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(0))
        self.weight.data.uniform_(-stdv, stdv)
        for name in _MASK_NAMES:
            w = getattr(self, "mask_" + name)
            s = 1. / math.sqrt(w.size(1))
            setattr(self, "mask_stdv_" + name, s)
            w.data.uniform_(-s, s)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    # ---- plan -------------------------------------------------------------------------------------------
    def graph(self, device):
        if self._graph is None or self._graph.device != device:
            self._graph = NCGraph.from_add_all(self.add_all, device, chunk=self._chunk, H=self.in_features)
        return self._graph

    def _scaler_factor(self, N, device):
        """Row factor of the scaler stage (layers.py:856-860 with the weight stacked once per scaler): the reference's
        degenerate constant (Q1), or - strict_reference=False - sum_s c_s(d_i) from the true degrees."""
        if self.strict_reference:
            return scaler_row_factor(N, device)
        if self._row_factor is None or self._row_factor[0] != device:
            g = self.graph(device)
            deg = (g.rowptr[1:] - g.rowptr[:-1])
            self._row_factor = (device, true_degree_row_factor(deg, self.scaler_names, self.compound_scalers, self.avg_d))
        return self._row_factor[1]

    def _codes(self, names):
        kinds, acts = [], []
        for n in names:
            if n in _UNUSABLE:
                raise NotImplementedError(
                    "aggregator %r is unusable in the reference (layers.py:731-851: learnable_std calls the full-graph "
                    "mean inside the node loop, normalized_mean/moment_3 raise NameError); not implemented" % n)
            kind, rawq = _AGG[n]
            kinds.append(Fn.KIND[kind])
            acts.append(Fn.ACT_RAW if (rawq and self.activation == "new_sigmoid") else Fn.ACT_SIGMOID)
        return kinds, acts

    def _drop(self, names, device=None):
        if self.drop_override is not None:
            return self.drop_override
        if self.graph_capturable and self.dropout > 0 and device is not None:
            n_groups = -(-len(names) // 8)      # one device seed per 8-mask launch group: groups must not share dropout bits
            if self._seeds is None or self._seeds.device != device or self._seeds.n != n_groups:
                self._seeds = Fn.DeviceSeeds(n_groups, device)
                self._seed_buf = self._seeds.seeds
            return Fn.DropoutSpec(self.dropout, seed_tensor=self._seeds.advance())
        return Fn.DropoutSpec(self.dropout)

    def _aggregate(self, names, input, drop=None, reduce_k=False):
        """All aggregators in `names` (<= 8) in one fused launch -> (K, N, H), or their sum (N, H) with reduce_k."""
        require_gpu(input)
        H = input.shape[1]
        kinds, acts = self._codes(names)
        masks = [getattr(self, "mask_" + n) for n in names]
        # [x_i || x_j] @ W_k  ==  x_i @ W_k[:H] + x_j @ W_k[H:]: dense GEMMs (matrix cores) shared by all K masks
        graph = self.graph(input.device)
        if reduce_k:
            return Fn.nc_local_layer(input, Fn.mask_weights(masks), None, graph, kinds, acts, drop or self._drop(names, input.device))
        wtop, wbot = torch.cat([w[:H] for w in masks], 1), torch.cat([w[H:] for w in masks], 1)      # (H, K*H) each
        return Fn.nc_fused_aggregate(input, mm(input, wtop), mm(input, wbot), graph, kinds, acts,
                                     drop or self._drop(names, input.device))

    def _aggregate_all(self, names, input, reduce_k=False):
        outs = []
        base = self._drop(names, input.device)
        for g0 in range(0, len(names), 8):
            grp = names[g0:g0 + 8]
            drop = base
            if base.keep is not None:   # explicit (K,E,H) mask: hand each group its slice
                drop = Fn.DropoutSpec(base.p, keep=base.keep[g0:g0 + 8].contiguous())
            elif g0 and base.seed_tensor is not None and base.seed_tensor.numel() > g0 // 8:
                drop = Fn.DropoutSpec(base.p, seed_tensor=base.seed_tensor[g0 // 8:])       # this group's own device seed
            elif g0 and base.seed_tensor is None:
                drop = Fn.DropoutSpec(base.p, seed=base.seed + g0)
            outs.append(self._aggregate(grp, input, drop, reduce_k))
        if reduce_k:
            return outs[0] if len(outs) == 1 else torch.stack(outs).sum(0)
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    # ---- forward ------------------------------------------------------------------------------------------
    def forward(self, input, adj):
        N = input.shape[0]
        # layers.py:855-862 computes  spmm(cat((adj,)*K,1), cat_k(m_k) scaled @ [W;W;W]).  The scalers are row factors
        # the reference evaluates to 1.0 (Q1) and everything after the aggregators is linear, so
        #     sum_k A (c * m_k W)  ==  A (c * (sum_k m_k) W):
        # the fused kernel emits sum_k m_k (N,H) directly - K x fewer bytes through the GEMM and the SpMM.
        msum = self._aggregate_all(self.aggregator_names, input, reduce_k=True)
        if self._sg is None or self._sg[0] is not adj:   # extension: a ready-made SpmmGraph is accepted as `adj`
            self._sg = (adj, adj if isinstance(adj, SpmmGraph) else SpmmGraph.from_torch_sparse(adj, input.device))
        if not FOLD_ROW_FACTOR:
            support = mm(msum, self.weight) * self._scaler_factor(N, input.device)
            return Fn.csr_spmm(support, self.bias, self._sg[1], 1)              # layers.py:861-867
        # the row factor is a constant of the graph: it rides on the SpMM's edge values (A_ij * c_j), forward and transposed, instead of
        # an element-wise launch each way
        factor = self._scaler_factor(N, input.device)
        if self._sg_scaled is None or self._sg_scaled[0] is not self._sg[1] or self._sg_scaled[1] is not factor:
            self._sg_scaled = (self._sg[1], factor, self._sg[1].scaled_by_source(factor))
        return Fn.csr_spmm(mm(msum, self.weight), self.bias, self._sg_scaled[2], 1)                  # layers.py:861-867

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


def _make_learnable(name):
    """learnable_<name>(input, adj): the public per-aggregator methods of layers.py:201-851, same signatures - the max* /
    min* forms carry the reference's (unused) `min_value=-inf` / `max_value=inf` keyword (layers.py:430,540)."""
    def run(self, input):
        return self._aggregate([name], input)[0]
    if name.startswith("max"):
        def learnable(self, input, adj, min_value=-math.inf):
            return run(self, input)
    elif name.startswith("min"):
        def learnable(self, input, adj, max_value=math.inf):
            return run(self, input)
    else:
        def learnable(self, input, adj):
            return run(self, input)
    learnable.__name__ = "learnable_" + name
    learnable.__doc__ = "Fused HIP form of layers.py learnable_%s(input, adj) -> (N, H)." % name
    return learnable


for _n in _MASK_NAMES:
    setattr(MMA, "learnable_" + _n, _make_learnable(_n))
