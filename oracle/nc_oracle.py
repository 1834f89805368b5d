"""CPU oracle for the node-classification (NC) MMA hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it.  `mma_amd/` never does.

It restates, with plain torch-CPU fp32 ops (so autograd supplies the gradients), what
/root/reference/node_classification computes:

  layers.py:201-302   learnable_sum*    m = x_i + s            s = sum_j drop(act(z)) * x_j
  layers.py:305-427   learnable_mean*   m = (x_i + s) / d_i    z = [x_i || x_j] @ W_k
  layers.py:430-538   learnable_max*    m = max(x_i, s)        (element-wise, NOT a neighbour max)
  layers.py:540-651   learnable_min*    m = min(x_i, s)
  layers.py:653-728   learnable_softmax/softmin   softmax over a singleton dim == s (NaN on exp overflow)
  layers.py:381-385,445-449,555-559,668-672,708-712   activation=="new_sigmoid" leaves RAW logits as
                      the mask in mean3/max/min/softmax/softmin (the expression result is discarded)
  layers.py:219 ...   F.dropout(mask0, p) with training=True always
  layers.py:853-867   MMA.forward: cat(K) -> 3 scalers (cat dim 1) -> mm [W;W;W] -> spmm K-stacked adj -> +bias
  scalers.py:22-64    scalers are handed the SPARSE adj, so every "degree" is N and the factor is 1.0 (to an ulp)

Pinned against the reference itself: tests/test_oracle_golden.py checks every function here
against tests/golden/*.npz, which tests/golden/gen_golden.py produced by importing the
reference on CPU (parity pinned).

Two forms are provided:
  * `aggregate_loop`   - per-node loop, same op sequence as the reference (small cases; also
                         the "faithful" CPU timing in bench.py)
  * `aggregate`        - vectorised over edges (index_select + index_add_), same arithmetic
"""
import math

import numpy as np
import torch

# name -> (combine kind, uses raw logits when activation == "new_sigmoid")
AGGREGATORS = {
    "sum": ("sum", False), "sum2": ("sum", False), "sum3": ("sum", False), "sum4": ("sum", False),
    "mean": ("mean", False), "mean2": ("mean", False), "mean3": ("mean", True), "mean4": ("mean", False),
    "max": ("max", True), "max2": ("max", False), "max3": ("max", False), "max4": ("max", False),
    "min": ("min", True), "min2": ("min", False), "min3": ("min", False), "min4": ("min", False),
    "softmax": ("softmax", True), "softmin": ("softmin", True),
}
BROKEN = ("std", "normalized_mean", "moment_3")  # layers.py:731-851: O(N^2)/NameError, unusable


def uses_raw_logits(name, activation):
    return AGGREGATORS[name][1] and activation == "new_sigmoid"


def csr_from_add_all(add_all):
    rowptr = np.zeros(len(add_all) + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([len(a) for a in add_all])
    col = np.concatenate([np.asarray(a, dtype=np.int64) for a in add_all]) if len(add_all) else np.zeros(0, np.int64)
    return rowptr, col


def _combine(kind, xi, s, deg):
    if kind == "sum":
        return xi + s
    if kind == "mean":
        return (xi + s) / deg
    if kind == "max":
        return torch.max(xi, s)
    if kind == "min":
        return torch.min(xi, s)
    if kind in ("softmax", "softmin"):
        e = torch.exp(s if kind == "softmax" else -s)  # layers.py:678,717
        return (e / e) * s                              # sum over a singleton dim: e/e_sum == e/e
    raise KeyError(kind)


def aggregate(name, x, W, rowptr, col, activation="new_sigmoid", p=0.0, keep=None):
    """Vectorised learnable_<name>(x): x (N,H), W (2H,H), CSR of add_all, keep (E,H) in {0,1} or None."""
    kind, _ = AGGREGATORS[name]
    N, H = x.shape
    rp = torch.as_tensor(np.asarray(rowptr), dtype=torch.int64)
    cj = torch.as_tensor(np.asarray(col), dtype=torch.int64)
    deg = (rp[1:] - rp[:-1])
    dst = torch.repeat_interleave(torch.arange(N), deg)
    xi_e = x.index_select(0, dst)
    xj_e = x.index_select(0, cj)
    z = torch.cat([xi_e, xj_e], 1) @ W                      # layers.py:215-216
    a = z if uses_raw_logits(name, activation) else torch.sigmoid(z)
    if p > 0.0:
        assert keep is not None, "explicit keep mask required for p>0 (reference RNG is not reproducible)"
        a = a * torch.as_tensor(keep, dtype=x.dtype) / (1.0 - p)   # F.dropout, training=True
    s = torch.zeros_like(x).index_add_(0, dst, a * xj_e)    # layers.py:221 sum over neighbours
    # degree-0 nodes crash the reference (Q12); the build defines s = 0 and divides by max(d,1)
    return _combine(kind, x, s, deg.clamp(min=1).to(x.dtype).unsqueeze(1))


def aggregate_loop(name, x, W, add_all, activation="new_sigmoid", p=0.0, keep=None):
    """Per-node loop in the reference's own op order (layers.py:205-226)."""
    kind, _ = AGGREGATORS[name]
    outs, e0 = [], 0
    raw = uses_raw_logits(name, activation)
    for i in range(len(add_all)):
        nb = torch.as_tensor(np.asarray(add_all[i]), dtype=torch.int64)
        aa = x[i:i + 1]
        bb = x.index_select(0, nb)
        z = torch.cat([aa.expand(len(nb), -1), bb], 1) @ W
        a = z if raw else torch.sigmoid(z)
        if p > 0.0:
            a = a * torch.as_tensor(keep[e0:e0 + len(nb)], dtype=x.dtype) / (1.0 - p)
        e0 += len(nb)
        s = torch.sum(a * bb, 0, keepdim=True)
        outs.append(_combine(kind, aa, s, float(len(nb))))
    return torch.cat(outs, 0)


def scaler_factors(N):
    """scalers.py:22-62 as invoked from layers.py:856 (with the sparse adj => all 'degrees' == N)."""
    all_degrees = torch.full((N,), N, dtype=torch.int64)
    lg = torch.log(all_degrees + 1)
    avg = torch.mean(lg)
    return (lg / avg).unsqueeze(-1), (avg / lg).unsqueeze(-1)   # amplification, attenuation: (N,1)


def true_degree_scaled(m, rowptr, K, scalers, compound, avg_d=None):
    """The scaler stage with TRUE degrees (the build's strict_reference=False extension; BASELINE configs[4] "all
    scalers"): m (K*N,H) -> (K*N, S*H).  compound=False: each scaler applied to m on its own, as scalers.py:22-62 would
    with add_all handed in (PNA); compound=True: the running product of mma_conv.py:181-196 (`out = out * ...`, every
    stage appends the running `out`).  deg = len(add_all[i]).clamp(1) (mma_conv.py:179), tiled K times (scalers.py:33-40)."""
    rp = torch.as_tensor(np.asarray(rowptr), dtype=torch.int64)
    deg = (rp[1:] - rp[:-1]).clamp(min=1).to(torch.float32)
    avg_log = torch.log(deg + 1).mean() if avg_d is None else avg_d["log"]
    avg_lin = deg.mean() if avg_d is None else avg_d["lin"]
    deg = torch.cat([deg] * K, 0).unsqueeze(-1)
    outs, out = [], m
    for scaler in scalers:
        if scaler == "identity":
            f = None
        elif scaler == "amplification":
            f = torch.log(deg + 1) / avg_log
        elif scaler == "attenuation":
            f = avg_log / torch.log(deg + 1)
        elif scaler == "linear":
            f = deg / avg_lin
        elif scaler == "inverse_linear":
            f = avg_lin / deg
        else:
            raise ValueError('Unknown scaler "%s".' % scaler)
        if compound:
            out = out if f is None else out * f
            outs.append(out)
        else:
            outs.append(m if f is None else m * f)
    return torch.cat(outs, 1)


def mma_forward(names, x, Ws, weight, bias, rowptr, col, adj_row, adj_col, adj_val,
                activation="new_sigmoid", p=0.0, keeps=None, return_m=False, true_degree_scalers=None, compound=False,
                avg_d=None):
    """MMA.forward (layers.py:853-867), literal op sequence.  true_degree_scalers: list of scaler names -> the
    strict_reference=False extension (see true_degree_scaled) instead of the reference's degenerate three."""
    N = x.shape[0]
    K = len(names)
    ms = [aggregate(n, x, Ws[n], rowptr, col, activation, p, None if keeps is None else keeps[n]) for n in names]
    m = torch.cat(ms, 0)                                              # (K*N, H)
    if true_degree_scalers is not None:
        m3 = true_degree_scaled(m, rowptr, K, true_degree_scalers, compound, avg_d)
        w3 = torch.cat([weight] * len(true_degree_scalers), 0)
    else:
        amp, att = scaler_factors(N)
        amp, att = torch.cat([amp] * K, 0), torch.cat([att] * K, 0)
        m3 = torch.cat([m, amp * m, att * m], 1)                      # identity, amplification, attenuation
        w3 = torch.cat([weight, weight, weight], 0)
    support = m3 @ w3                                                 # (K*N, C)
    r = torch.as_tensor(np.asarray(adj_row), dtype=torch.int64)
    c = torch.as_tensor(np.asarray(adj_col), dtype=torch.int64)
    v = torch.as_tensor(np.asarray(adj_val), dtype=x.dtype)
    out = torch.zeros(N, weight.shape[1], dtype=x.dtype)
    for k in range(K):                                                # spmm(cat((adj,)*K, 1), support)
        out = out.index_add(0, r, v.unsqueeze(1) * support[k * N:(k + 1) * N].index_select(0, c))
    if bias is not None:
        out = out + bias
    return (out, ms) if return_m else out


def init_like_reference(H, C, names, seed):
    """U(+-1/sqrt(H)) masks (layers.py:148-192), U(+-1/sqrt(in)) weight/bias (layers.py:145,170,197)."""
    g = torch.Generator().manual_seed(seed)
    b = 1.0 / math.sqrt(H)
    Ws = {n: (torch.rand(2 * H, H, generator=g) * 2 - 1) * b for n in names}
    weight = (torch.rand(H, C, generator=g) * 2 - 1) * b
    bias = (torch.rand(C, generator=g) * 2 - 1) * b
    return Ws, weight, bias


def gcn_forward(x, weight, bias, adj_row, adj_col, adj_val):
    """GraphConvolution.forward (layers.py:38-45): spmm(adj, x @ W) + b."""
    support = x @ weight
    r = torch.as_tensor(np.asarray(adj_row), dtype=torch.int64)
    c = torch.as_tensor(np.asarray(adj_col), dtype=torch.int64)
    v = torch.as_tensor(np.asarray(adj_val), dtype=x.dtype)
    out = torch.zeros(x.shape[0], weight.shape[1], dtype=x.dtype).index_add(0, r, v.unsqueeze(1) * support.index_select(0, c))
    return out + bias if bias is not None else out


def model_forward(features, prm, names, rowptr, col, adj_row, adj_col, adj_val, activation, p, hidden_keep, keeps):
    """models.MMAConv.forward (models.py:64-68) in training mode: gc1 -> relu -> dropout(explicit keep mask, models.py:66)
    -> gc2 = MMA.forward (explicit per-aggregator masks) -> log_softmax.  prm: weight0, bias0, weight1, bias1, weight_<agg>."""
    h = torch.relu(gcn_forward(features, prm["weight0"], prm["bias0"], adj_row, adj_col, adj_val))
    if hidden_keep is not None:
        h = h * torch.as_tensor(hidden_keep, dtype=h.dtype) / (1.0 - p)
    out = mma_forward(names, h, {a: prm["weight_" + a] for a in names}, prm["weight1"], prm["bias1"], rowptr, col,
                      adj_row, adj_col, adj_val, activation, p, keeps)
    return torch.log_softmax(out, dim=1)
