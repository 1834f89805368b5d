#!/usr/bin/env python3
"""Generate golden vectors for the node-classification (NC) hot path by running the
REFERENCE implementation itself on CPU.

Runs ONLY in the build container (needs /root/reference, which never travels to the
GPU box).  It imports the reference's `node_classification/layers.py` with the three
harness-side shims of SURVEY.md Appendix B (dead scipy import alias; networkx
csr_array -> csr_matrix; `.to('cuda:*')` -> cpu), constructs `layers.MMA` directly
with CPU Parameters, and dumps inputs (as seeds + checksums), per-aggregator outputs,
the full `MMA.forward` output and autograd gradients into small `.npz` fixtures next
to this script.  Nothing from the reference is copied: the fixtures hold data only.

    python tests/golden/gen_golden.py            # all cases (~2-3 min)
    python tests/golden/gen_golden.py toy        # only cases whose name contains "toy"

Reference entry points exercised (file:line in /root/reference/node_classification):
  layers.py:201-651   learnable_{sum,mean,max,min}{,2,3,4}
  layers.py:653-728   learnable_softmax / learnable_softmin
  layers.py:853-867   MMA.forward  (+ scalers.py:22-64 through it)
"""
import hashlib
import os
import pickle
import sys
import types

import numpy as np

REF = "/root/reference/node_classification"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True  # never emit .pyc of the reference

import networkx as nx  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import scipy.sparse.linalg as sla  # noqa: E402
import torch  # noqa: E402

# ---- Appendix-B shims (harness code) -------------------------------------------------
_pkg = types.ModuleType("scipy.sparse.linalg.eigen")
_pkg.__path__ = []
_arp = types.ModuleType("scipy.sparse.linalg.eigen.arpack")
_arp.eigsh = sla.eigsh
sys.modules[_pkg.__name__] = _pkg
sys.modules[_arp.__name__] = _arp
_adj = nx.adjacency_matrix
nx.adjacency_matrix = lambda g, *a, **k: sp.csr_matrix(_adj(g, *a, **k))
_to = torch.Tensor.to
torch.Tensor.to = lambda s, *a, **k: _to(
    s, *tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a), **k)
sys.path.insert(0, REF)
import layers  # noqa: E402  (the reference)

sys.path.insert(0, HERE)
from inputs import ALL_MASK_NAMES, WORKING, make_inputs, keep_mask, sha  # noqa: E402


# ---- graphs ---------------------------------------------------------------------------
def graph_from_pickle(name):
    """Same construction as utils.py:71,97-100 (raw 0/1 adjacency, ascending neighbour order)."""
    with open(os.path.join(REF, "data", "ind.%s.graph" % name), "rb") as f:
        graph = pickle.load(f, encoding="latin1")
    adj = nx.adjacency_matrix(nx.from_dict_of_lists(graph))
    return adj


def toy_graph():
    # 6 nodes: node 0 is a hub (deg 5), node 5 has degree 1, 1-2-3 form a triangle
    edges = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (1, 2), (2, 3), (1, 3), (3, 4)]
    a = np.zeros((6, 6), dtype=np.float32)
    for u, v in edges:
        a[u, v] = a[v, u] = 1
    return sp.csr_matrix(a)


def ring_hub_graph(n=40):
    # ring + one hub touching every 3rd node + a self loop (Pubmed has 3 self loops)
    a = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        a[i, (i + 1) % n] = a[(i + 1) % n, i] = 1
    for i in range(3, n, 3):
        a[0, i] = a[i, 0] = 1
    a[7, 7] = 1
    return sp.csr_matrix(a)


def add_all_of(adj):
    return [adj[i].nonzero()[1] for i in range(adj.shape[0])]  # utils.py:97-100


def torch_sparse(adj):
    m = adj.tocoo().astype(np.float32)
    idx = torch.from_numpy(np.vstack((m.row, m.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(m.data), torch.Size(m.shape))


# ---- explicit dropout: replace layers.F by a namespace whose dropout multiplies a saved mask
class _FShim:
    def __init__(self):
        self.keep = None   # (E,H) float {0,1}, rows in CSR order
        self.rowptr = None
        self.node = 0
        self.p = 0.0

    def start(self, keep, rowptr, p):
        self.keep, self.rowptr, self.node, self.p = keep, rowptr, 0, p

    def dropout(self, x, p):
        # every learnable_* calls F.dropout(mask0, p) once per node, in node order
        lo, hi = self.rowptr[self.node], self.rowptr[self.node + 1]
        self.node += 1
        if self.keep is None:
            assert p == 0.0
            return x
        assert hi - lo == x.shape[0] and abs(p - self.p) < 1e-12
        return x * self.keep[lo:hi] / (1.0 - p)


FSHIM = _FShim()
layers.F = FSHIM


def build_mma(add_all, activation, H, C, masks, weight, bias, p, aggs):
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()))
    mp = {n: P(masks[n]) for n in ALL_MASK_NAMES}
    w, b = P(weight), P(bias)
    mma = layers.MMA(add_all, activation, 2, H, C, w, b, *[mp[n] for n in ALL_MASK_NAMES], p, aggs, "cpu")
    # the ctor re-initialises everything (layers.py:143-198): put our values back
    with torch.no_grad():
        for n in ALL_MASK_NAMES:
            mp[n].copy_(torch.from_numpy(masks[n]))
        w.copy_(torch.from_numpy(weight))
        b.copy_(torch.from_numpy(bias))
    return mma, mp, w, b


def sample_rows(N, deg, n=192, seed=7):
    if N <= n:
        return np.arange(N, dtype=np.int64)
    r = np.random.default_rng(seed)
    hubs = np.argsort(-deg)[:32]
    low = np.argsort(deg)[:16]
    rest = r.choice(N, size=n - 48, replace=False)
    return np.unique(np.concatenate([hubs, low, rest])).astype(np.int64)


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def run_case(name, adj, H, C, seed, agg_sets, activations, p_list, single_aggs=(), full_store=False):
    N = adj.shape[0]
    add_all = add_all_of(adj)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([len(a) for a in add_all])
    col = np.concatenate(add_all).astype(np.int32)
    deg = np.diff(rowptr)
    x, masks, weight, bias, cot = make_inputs(seed, N, H, C)
    tadj = torch_sparse(adj)
    rows = np.arange(N, dtype=np.int64) if full_store else sample_rows(N, deg)
    out = dict(name=name, N=N, H=H, C=C, seed=seed, rowptr=rowptr.astype(np.int32), col=col, rows=rows,
               sha_x=sha(x), sha_w=sha(weight), sha_mask_sum=sha(masks["sum"]), sha_cot=sha(cot),
               adj_row=adj.tocoo().row.astype(np.int32), adj_col=adj.tocoo().col.astype(np.int32),
               adj_val=adj.tocoo().data.astype(np.float32))
    E = len(col)
    keep_all = {}
    for p in p_list:
        if p > 0:
            # one keep mask per aggregator name (K independent masks), Bernoulli(1-p)
            for ai, an in enumerate(WORKING):
                keep_all[(p, an)] = keep_mask(seed, an, E, H, p)
    meta = []
    for act in activations:
        for p in p_list:
            ptag = "p%02d" % int(round(p * 100))
            # individually callable aggregators (layers.py public learnable_*)
            if single_aggs:
                mma, mp, w, b = build_mma(add_all, act, H, C, masks, weight, bias, p, list(single_aggs))
                xt = torch.from_numpy(x.copy())
                for an in single_aggs:
                    FSHIM.start(torch.from_numpy(keep_all[(p, an)]) if p > 0 else None, rowptr, p)
                    with torch.no_grad():
                        mk = mma.AGGREGATORS[an](xt, tadj)
                    key = "single/%s/%s/%s" % (act, ptag, an)
                    out[key] = mk.numpy()[rows].astype(np.float32)
                    out[key + "/stats"] = stats(mk)
                    meta.append(key)
            for aggs in agg_sets:
                mma, mp, w, b = build_mma(add_all, act, H, C, masks, weight, bias, p, list(aggs))
                xt = torch.from_numpy(x.copy()).requires_grad_(True)
                # forward: aggregators are called in list order, each walking all nodes
                ms = []
                for an in aggs:
                    FSHIM.start(torch.from_numpy(keep_all[(p, an)]) if p > 0 else None, rowptr, p)
                    ms.append(mma.AGGREGATORS[an](xt, tadj))
                # full forward (re-runs the aggregators; replay the same masks in order)
                class _Chain:
                    def __init__(s):
                        s.i = -1
                    def dropout(s, t, pp):
                        if FSHIM.node == 0 or FSHIM.node >= N:
                            s.i += 1
                            an = aggs[s.i]
                            FSHIM.start(torch.from_numpy(keep_all[(p, an)]) if p > 0 else None, rowptr, p)
                        return FSHIM.dropout(t, pp)
                FSHIM.node = 0
                layers.F = _Chain()
                outp = mma(xt, tadj)
                layers.F = FSHIM
                loss = (outp * torch.from_numpy(cot)).sum()
                used = [mp[an] for an in aggs]
                grads = torch.autograd.grad(loss, [xt, w, b] + used)
                key = "set/%s/%s/%s" % (act, ptag, ",".join(aggs))
                out[key + "/out"] = outp.detach().numpy()[rows].astype(np.float32)
                out[key + "/out/stats"] = stats(outp)
                for an, mk in zip(aggs, ms):
                    out[key + "/m/" + an] = mk.detach().numpy()[rows].astype(np.float32)
                    out[key + "/m/" + an + "/stats"] = stats(mk)
                out[key + "/gx"] = grads[0].numpy()[rows].astype(np.float32)
                out[key + "/gx/stats"] = stats(grads[0])
                out[key + "/gweight"] = grads[1].numpy().astype(np.float32)
                out[key + "/gbias"] = grads[2].numpy().astype(np.float32)
                for an, g in zip(aggs, grads[3:]):
                    out[key + "/gmask/" + an] = g.numpy().astype(np.float32)
                meta.append(key)
                print("   ", name, key, "out|max|=%.3g" % outp.abs().max().item(), flush=True)
    out["keys"] = np.array(meta)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), flush=True)


CASES = {
    # every working aggregator individually + mixed sets, both activations, p=0 and explicit-mask p=0.5
    "toy6_h8": lambda: run_case("toy6_h8", toy_graph(), 8, 3, 42,
                                agg_sets=[("mean", "mean2"), ("min", "min2", "min3", "min4"),
                                          ("sum", "max", "min", "mean"), ("mean3", "softmax", "softmin")],
                                activations=["sigmoid", "new_sigmoid"], p_list=[0.0, 0.5],
                                single_aggs=WORKING, full_store=True),
    "ringhub40_h20": lambda: run_case("ringhub40_h20", ring_hub_graph(40), 20, 5, 43,
                                      agg_sets=[("sum2", "mean4", "max3", "min"), ("max", "max2")],
                                      activations=["new_sigmoid"], p_list=[0.0, 0.5],
                                      single_aggs=("sum3", "mean3", "max4", "min2"), full_store=True),
    # BASELINE config 1 shape: Cora, H=64, mean,mean2 (README.md:70)
    "cora_h64": lambda: run_case("cora_h64", graph_from_pickle("cora"), 64, 7, 44,
                                 agg_sets=[("mean", "mean2")], activations=["new_sigmoid"], p_list=[0.0, 0.75]),   # --dropout=0.75
    "cora_h16": lambda: run_case("cora_h16", graph_from_pickle("cora"), 16, 7, 45,
                                 agg_sets=[("sum", "max", "min", "mean")], activations=["new_sigmoid"],
                                 p_list=[0.0, 0.5]),
    # BASELINE config 3 shape: Pubmed structure, H=16, min,min2,min3,min4 (README.md:58/64)
    "pubmed_h16": lambda: run_case("pubmed_h16", graph_from_pickle("pubmed"), 16, 3, 46,
                                   agg_sets=[("min", "min2", "min3", "min4")], activations=["new_sigmoid"],
                                   p_list=[0.0, 0.5]),                                                       # --dropout=0.5
    # the headline width pinned to the reference (round-2 VERDICT): Citeseer, H=128, min,min2,min3, --dropout=0.5 (README.md:64)
    "citeseer_h128": lambda: run_case("citeseer_h128", graph_from_pickle("citeseer"), 128, 6, 47,
                                      agg_sets=[("min", "min2", "min3")], activations=["new_sigmoid"], p_list=[0.0, 0.5]),
}

if __name__ == "__main__":
    sel = sys.argv[1:] or [""]
    torch.set_num_threads(8)
    for cname, fn in CASES.items():
        if any(s in cname for s in sel):
            print("== case", cname, flush=True)
            fn()
