#!/usr/bin/env python3
"""Golden vectors for the graph-regression (GR) hot path from the REFERENCE'S OWN MODULE CODE, run on CPU in the build container.

What this pins and what it does not.  `graph_regression/mma_conv.py` and `mask_aggr.py` are imported from /root/reference and EXECUTED:
`MMAConv.__init__` (tower split, `avg_deg` from the histogram tensor: G8; the dict of mask Linears: G2), `forward` (:121-136), `message`
(:138-157: the overwritten `hs`, G1; the always-on dropout, G4), `aggregate` (:159-196: aggregator loop, degree clamp, COMPOUNDING
scalers, G7), `MaskAggregateLinear.forward` (mask_aggr.py:53-68).  Their third-party imports are ABSENT from this image (torch_scatter,
torch_geometric; unpinned by the reference - README "PyTorch 1.9" era: torch-scatter 2.0.7-2.0.9, PyG 2.0.x) and are satisfied here by
harness-side STAND-INS restating the published semantics of exactly the six names the reference imports:
  torch_scatter.scatter(src, index, dim, out, dim_size, reduce)   sum / mean (sum / clamp(count, 1)) / min / max, zero-initialised output,
        an empty target gives 0, min / max hand their gradient to the FIRST extremal edge (the CPU kernel's strict-compare update loop)
  torch_geometric.nn.conv.MessagePassing                          propagate(): flow source -> target, x_j = x[edge_index[0]],
        x_i = x[edge_index[1]], aggregate(inputs, index = edge_index[1], dim_size = N), update = identity; node_dim = 0
  torch_geometric.nn.dense.linear.Linear                          F.linear with a (out, in) weight and a bias
  torch_geometric.utils.degree, torch_geometric.nn.inits.reset, torch_geometric.typing.{Adj, OptTensor}
So the fixtures pin oracle/gr_oracle.py (and through it the HIP path) to the reference's own control flow and quirks; the arithmetic INSIDE
scatter / propagate remains a restatement of published semantics (anchored by the hand-computed known answers of tests/test_gr_oracle.py).
DESIGN.md 6 says "GR: pinned to the reference's module code over stand-ins for its absent third-party primitives" - not more.

The dropout of `message()` (`F.dropout(hs, self.dropout)`, training=True always) is replayed from a saved keep mask; the mask is the one the
HIP kernels' counter hash generates for (seed, p) (oracle/dropout_rng.py, laid out in the fused path's 4-float-padded tower width), so the
`-m gpu` test can run the HIP path in HASH mode against these outputs.  Nothing from the reference is copied: the fixtures hold data only.

    python tests/golden/gen_gr_golden.py            # writes tests/golden/gr_*.npz (a few seconds)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/graph_regression"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True          # never emit .pyc of the reference
sys.path.insert(0, ROOT)


# ---- stand-ins for the absent third-party modules (harness code; see the docstring) ---------------------------------
class _ScatterExtreme(torch.autograd.Function):
    """min / max over index groups along dim 0: values of the published CPU kernel (sequential strict-compare updates in edge order, so
    the FIRST extremal edge is the arg; a group without edges gives 0), gradient to the arg edge only."""

    @staticmethod
    def forward(ctx, src, index, dim_size, is_max):
        E = src.shape[0]
        flat = src.reshape(E, -1)
        big = torch.finfo(src.dtype).max
        out = torch.full((dim_size, flat.shape[1]), -big if is_max else big, dtype=src.dtype)
        arg = torch.full((dim_size, flat.shape[1]), E, dtype=torch.int64)
        for e in range(E):                                   # the kernel's own order: fixtures are small
            i = int(index[e])
            upd = flat[e] > out[i] if is_max else flat[e] < out[i]
            out[i] = torch.where(upd, flat[e], out[i])
            arg[i] = torch.where(upd, torch.full_like(arg[i], e), arg[i])
        out[arg == E] = 0
        ctx.save_for_backward(arg)
        ctx.shape = src.shape
        return out.reshape((dim_size,) + tuple(src.shape[1:]))

    @staticmethod
    def backward(ctx, g):
        arg, = ctx.saved_tensors
        E = ctx.shape[0]
        gs = torch.zeros((E + 1, arg.shape[1]), dtype=g.dtype)
        gs.scatter_(0, arg, g.reshape(arg.shape))            # every (group, column) has one arg: no collisions
        return gs[:E].reshape(ctx.shape), None, None, None


def _scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and out is None and index.dim() == 1
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    if reduce in ("sum", "add", "mean"):
        res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)
        if reduce == "mean":
            cnt = torch.zeros(dim_size, dtype=src.dtype).index_add(0, index, torch.ones(len(index), dtype=src.dtype)).clamp(min=1)
            res = res / cnt.view((-1,) + (1,) * (src.dim() - 1))
        return res
    if reduce in ("min", "max"):
        return _ScatterExtreme.apply(src, index, dim_size, reduce == "max")
    raise ValueError(reduce)


class _PygLinear(torch.nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None, bias_initializer=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels))
        self.bias = torch.nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if self.bias is not None:
            bound = 1.0 / max(self.in_channels, 1) ** 0.5
            torch.nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias)


class _MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        assert flow == "source_to_target" and node_dim == 0
        self.aggr, self.node_dim = aggr, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        x, edge_attr = kwargs["x"], kwargs.get("edge_attr")
        src, dst = edge_index[0], edge_index[1]
        msg = self.message(x_i=x.index_select(0, dst), x_j=x.index_select(0, src), edge_attr=edge_attr)
        return self.update(self.aggregate(msg, dst, dim_size=x.shape[0]))

    def update(self, inputs):
        return inputs


def _degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    return torch.zeros(n, dtype=dtype or torch.float32).index_add(0, index, torch.ones(len(index), dtype=dtype or torch.float32))


def _reset(value):
    if hasattr(value, "reset_parameters"):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, "children") else []:
            _reset(child)


def install_stand_ins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    from typing import Optional, Union
    mod("torch_scatter", scatter=_scatter)
    mod("torch_geometric")
    mod("torch_geometric.typing", Adj=Union[torch.Tensor], OptTensor=Optional[torch.Tensor])
    mod("torch_geometric.nn")
    mod("torch_geometric.nn.conv", MessagePassing=_MessagePassing)
    mod("torch_geometric.nn.dense")
    mod("torch_geometric.nn.dense.linear", Linear=_PygLinear)
    mod("torch_geometric.utils", degree=_degree)
    mod("torch_geometric.nn.inits", reset=_reset)


CASES = {
    # name: ctor arguments of the reference's MMAConv + batch size; zinc = mma.py:92-95's layer
    "gr_zinc_t5_f75": dict(aggregators=["min", "max"], scalers=["identity", "amplification", "linear"], towers=5, F=75, edge_dim=50, n_graphs=3, out=75),
    "gr_sum_mean_t2": dict(aggregators=["sum", "mean"], scalers=["identity", "attenuation", "inverse_linear"], towers=2, F=6, edge_dim=None, n_graphs=4),
    "gr_last_wins_t3": dict(aggregators=["max", "sum", "min"], scalers=["linear", "identity"], towers=3, F=4, edge_dim=5, n_graphs=4, divide_input=True),
    "gr_stacks_t2": dict(aggregators=["mean", "max"], scalers=["identity", "amplification"], towers=2, F=8, edge_dim=3, n_graphs=4, pre_layers=2, post_layers=2),
}
P_DROP, SEED = 0.5, 0x6A09E667F3BCC908


def main():
    install_stand_ins()
    sys.path.insert(0, REF)
    import mma_conv as R                                     # the reference (imports its own mask_aggr)
    import torch.nn.functional as F
    from oracle.dropout_rng import keep_mask
    from tools.synth import molecule_batch
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, c in CASES.items():
        if only not in name:
            continue
        rng = np.random.default_rng(sum(map(ord, name)))
        T, Fi = c["towers"], c["F"]
        div = c.get("divide_input", False)
        cin = Fi * T if div else Fi
        cout = c.get("out", cin if div else Fi * T)           # zinc: MMAConv(75, 75, towers=5) - 15 outputs per tower (mma.py:92-95)
        ei, N = molecule_batch(rng, c["n_graphs"])
        E = ei.shape[1]
        hist = np.bincount(np.bincount(ei[1], minlength=N), minlength=5)
        torch.manual_seed(sum(map(ord, name)))
        conv = R.MMAConv(cin, cout, c["aggregators"], c["scalers"], torch.tensor(hist), edge_dim=c["edge_dim"], towers=T,
                         pre_layers=c.get("pre_layers", 1), post_layers=c.get("post_layers", 1), divide_input=div)
        x = torch.from_numpy(rng.standard_normal((N, cin)).astype(np.float32)).requires_grad_(True)
        ea = torch.from_numpy(rng.standard_normal((E, c["edge_dim"])).astype(np.float32)).requires_grad_(True) if c["edge_dim"] else None
        cot = torch.from_numpy(rng.standard_normal((N, cout)).astype(np.float32))
        Fw = -(-Fi // 4) * 4                                 # the fused path's padded tower width: its dropout stream is indexed by t*Fw + f
        keep = keep_mask(SEED, int(P_DROP * 256), 1, E, T * Fw)[0].reshape(E, T, Fw)[:, :, :Fi].astype(np.float32)
        last = c["aggregators"][-1]
        pre = [[m.aggregation_layers[last] for m in seq if hasattr(m, "aggregation_layers")] for seq in conv.pre_nns[last]]
        post = [[m for m in seq if hasattr(m, "weight")] for seq in conv.post_nns]
        params = {}
        for t in range(T):
            for li, l in enumerate(pre[t]):
                params["pre_w/%d/%d" % (t, li)], params["pre_b/%d/%d" % (t, li)] = l.weight, l.bias
            for li, l in enumerate(post[t]):
                params["post_w/%d/%d" % (t, li)], params["post_b/%d/%d" % (t, li)] = l.weight, l.bias
        params["lin_w"], params["lin_b"] = conv.lin.weight, conv.lin.bias
        if c["edge_dim"]:
            params["enc_w"], params["enc_b"] = conv.edge_encoder.weight, conv.edge_encoder.bias
        out_blob = {"edge_index": ei.astype(np.int64), "hist": hist.astype(np.int64), "x": x.detach().numpy(), "cot": cot.numpy(), "keep": keep.astype(np.uint8),
                    "p": np.float32(P_DROP), "seed": np.uint64(SEED), "avg_deg_lin": np.float64(conv.avg_deg["lin"]), "avg_deg_log": np.float64(conv.avg_deg["log"]),
                    "meta": np.array(repr({k: v for k, v in c.items()}))}
        if ea is not None:
            out_blob["edge_attr"] = ea.detach().numpy()
        for k, v in params.items():
            out_blob["param/" + k] = v.detach().numpy().copy()
        real_dropout = F.dropout
        for tag, p_run in (("p0", 0.0), ("p50", P_DROP)):
            conv.dropout = p_run                              # a public attribute (mma_conv.py:67); 0.5 is what the reference hard-codes
            km = torch.from_numpy(keep)

            def replay(inp, p=0.5, training=True, inplace=False):
                if p == 0.0:
                    return inp
                assert inp.shape == km.shape and p == P_DROP
                return inp * km / (1.0 - p)
            F.dropout = replay
            try:
                for q in [x] + ([ea] if ea is not None else []) + list(params.values()):
                    q.grad = None
                out = conv(x, torch.from_numpy(ei), ea)
                (out * cot).sum().backward()
            finally:
                F.dropout = real_dropout
            out_blob[tag + "/out"] = out.detach().numpy()
            out_blob[tag + "/gx"] = x.grad.numpy().copy()
            if ea is not None:
                out_blob[tag + "/gea"] = ea.grad.numpy().copy()
            for k, v in params.items():
                out_blob[tag + "/g/" + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().copy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out_blob)
        print("wrote %s: N=%d E=%d, %d arrays, %.0f KB, |out| max %.3g" % (path, N, E, len(out_blob), os.path.getsize(path) / 1024,
                                                                      float(np.abs(out_blob["p50/out"]).max())))
    if only in "graggr_all":
        # MMAConv.aggregate() called directly (mma_conv.py:159-196): the only way to its var / std branch (forward() raises on
        # those names first) - all six aggregators x all five scalers on given messages, targets 5 and 11 empty, one tie-heavy column
        rng = np.random.default_rng(77)
        aggs = ["sum", "mean", "min", "max", "var", "std"]
        scal = ["identity", "amplification", "attenuation", "linear", "inverse_linear"]
        T, Fi, N, E = 2, 6, 14, 60
        conv = R.MMAConv(Fi, Fi * T, ["sum"], scal, torch.tensor([0, 3, 9, 5, 2]), towers=T)
        conv.aggregators = aggs                                  # aggregate() walks the attribute
        index = torch.from_numpy(rng.choice([i for i in range(N) if i not in (5, 11)], E))
        inp = rng.standard_normal((E, T, Fi)).astype(np.float32)
        inp[:, 0, 0] = rng.integers(-2, 3, E)                    # ties: the first extremal edge must win, alone
        inputs = torch.from_numpy(inp).requires_grad_(True)
        out = conv.aggregate(inputs, index, N)
        cot = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
        (out * cot).sum().backward()
        path = os.path.join(HERE, "graggr_all.npz")
        np.savez_compressed(path, inputs=inp, index=index.numpy(), N=np.int64(N), cot=cot.numpy(), out=out.detach().numpy(), ginputs=inputs.grad.numpy(),
                            avg_deg_lin=np.float64(conv.avg_deg["lin"]), avg_deg_log=np.float64(conv.avg_deg["log"]),
                            meta=np.array(repr(dict(aggregators=aggs, scalers=scal, towers=T, F=Fi, hist=[0, 3, 9, 5, 2]))))
        print("wrote %s: out %s, |out| max %.3g" % (path, tuple(out.shape), float(out.abs().max())))


if __name__ == "__main__":
    main()
