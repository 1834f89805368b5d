#!/usr/bin/env python3
"""Golden vectors for the CALLERS of the hot path, produced by running the reference itself on CPU:

  * `layers.GraphConvolution.forward` (layers.py:38-45) on the real Cora features (utils.py:33-119) with seeded weights;
  * three epochs of `train.py:72-80` (model.train(); zero_grad; forward; nll_loss on idx_train; backward; Adam.step) on
    `models.MMAConv` (models.py:10-68) with BASELINE configs[0]'s arguments (README.md:70: --dataset cora
    --aggregators mean,mean2 --hidden 64 --dropout 0.75; lr 0.01, weight_decay 5e-4 = train.py:24-25), every dropout
    (models.py:66 and the always-on mask dropout of layers.py:324) replaced by multiplication with a seeded keep mask so
    that another implementation can replay it: per-epoch loss, the final log-probabilities and parameters.

Build container only (imports /root/reference with the shims of gen_golden.py plus `torch.cuda.FloatTensor ->
torch.FloatTensor` for models.py:17-43).  Writes tests/golden/train3_cora.npz: data only (seeds, masks' seeds, outputs).
    python tests/golden/gen_train_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (runs the Appendix-B shims and imports the reference's layers.py)
from inputs import ALL_MASK_NAMES, keep_mask, rng_uniform, sha  # noqa: E402

torch.cuda.FloatTensor = torch.FloatTensor          # models.py:17-43 allocates with torch.cuda.FloatTensor
import models  # noqa: E402  (the reference)
import utils  # noqa: E402  (the reference)

SEED, HID, P, AGGS, EPOCHS, LR, WD = 77, 64, 0.75, ("mean", "mean2"), 3, 0.01, 5e-4


def hidden_keep(seed, epoch, N, H, p):
    """Keep mask of models.py:66 (F.dropout on the hidden features), one per epoch."""
    return (np.random.default_rng(seed + 7000 + epoch).random((N, H), dtype=np.float32) >= p).astype(np.float32)


def model_params(seed, nfeat, nhid, nclass):
    """Seeded values for the 25 Parameters of models.py:17-43 (name -> array)."""
    prm = {"weight0": rng_uniform(seed + 1, (nfeat, nhid), 1.0 / np.sqrt(nhid)), "bias0": rng_uniform(seed + 2, (nhid,), 1.0 / np.sqrt(nhid)),
           "weight1": rng_uniform(seed + 3, (nhid, nclass), 1.0 / np.sqrt(nhid)), "bias1": rng_uniform(seed + 4, (nclass,), 1.0 / np.sqrt(nhid))}
    for i, n in enumerate(ALL_MASK_NAMES):
        prm["weight_" + n] = rng_uniform(seed + 100 + i, (2 * nhid, nhid), 1.0 / np.sqrt(nhid))
    return prm


class _ModelsF:
    """Stand-in for `F` inside models.py: relu / log_softmax pass through, dropout multiplies the epoch's saved mask."""

    def __init__(self):
        self.keep = None

    relu = staticmethod(torch.nn.functional.relu)
    log_softmax = staticmethod(torch.nn.functional.log_softmax)

    def dropout(self, x, p, training=True):
        assert training and abs(p - P) < 1e-12
        return x * self.keep / (1.0 - p)


class _LayersF:
    """Stand-in for `F` inside layers.py: every learnable_* calls F.dropout(mask0, p) once per node, aggregators in list
    order; each call multiplies the rows of that aggregator's saved (E,H) keep mask that belong to the node."""

    def __init__(self, rowptr, N):
        self.rowptr, self.N = rowptr, N

    def start(self, keeps):
        self.keeps, self.agg, self.node = keeps, 0, 0

    def dropout(self, x, p):
        if self.node == self.N:
            self.agg, self.node = self.agg + 1, 0
        lo, hi = self.rowptr[self.node], self.rowptr[self.node + 1]
        self.node += 1
        return x * self.keeps[self.agg][lo:hi] / (1.0 - p)


def main():
    os.chdir(G.REF)                                   # load_data opens the relative path data/ind.cora.*
    add_all, adj, features, labels, idx_train, idx_val, idx_test = utils.load_data("cora")
    N, nfeat = features.shape
    nclass = int(labels.max()) + 1
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([len(a) for a in add_all])
    col = np.concatenate(add_all).astype(np.int32)
    E = len(col)
    prm = model_params(SEED, nfeat, HID, nclass)

    out = dict(seed=SEED, N=N, nfeat=nfeat, nclass=nclass, hidden=HID, p=P, aggs=np.array(AGGS), epochs=EPOCHS, lr=LR, wd=WD,
               rowptr=rowptr.astype(np.int32), col=col, labels=labels.numpy().astype(np.int16), idx_train=idx_train.numpy().astype(np.int32))
    f = features.numpy()
    fr, fc = np.nonzero(f)
    out.update(feat_row=fr.astype(np.int16), feat_col=fc.astype(np.int16), feat_val=f[fr, fc].astype(np.float32))
    a = adj.coalesce()
    out.update(adj_row=a.indices()[0].numpy().astype(np.int32), adj_col=a.indices()[1].numpy().astype(np.int32),
               adj_val=a.values().numpy().astype(np.float32))
    out["sha_weight0"], out["sha_mask_mean"] = sha(prm["weight0"]), sha(prm["weight_mean"])

    # ---- f-1: GraphConvolution.forward of the reference on the real features
    w0, b0 = torch.nn.Parameter(torch.zeros(nfeat, HID)), torch.nn.Parameter(torch.zeros(HID))
    gc = G.layers.GraphConvolution(nfeat, HID, w0, b0, "cpu")
    with torch.no_grad():
        w0.copy_(torch.from_numpy(prm["weight0"])); b0.copy_(torch.from_numpy(prm["bias0"]))
    g_out = gc(features, adj)
    cot = torch.from_numpy(np.random.default_rng(SEED + 9).standard_normal((N, HID), dtype=np.float32))
    gw0, gb0 = torch.autograd.grad((g_out * cot).sum(), [w0, b0])
    rows = np.sort(np.random.default_rng(SEED + 11).choice(N, 384, replace=False))           # stored rows of the (N, .) results
    wrows = np.sort(np.random.default_rng(SEED + 12).choice(nfeat, 256, replace=False))      # stored rows of the (nfeat, hidden) ones
    out.update(rows=rows.astype(np.int32), wrows=wrows.astype(np.int32), gcn_out=g_out.detach().numpy()[rows].astype(np.float32),
               gcn_out_stats=G.stats(g_out), gcn_gweight=gw0.numpy()[wrows], gcn_gweight_stats=G.stats(gw0), gcn_gbias=gb0.numpy())
    print("GraphConvolution out |max| %.3g" % g_out.abs().max().item(), flush=True)

    # ---- f-3: three training steps of models.MMAConv
    model = models.MMAConv(add_all, "new_sigmoid", 2, nfeat, HID, nclass, P, list(AGGS), "cpu")
    with torch.no_grad():
        for n, v in prm.items():
            getattr(model, n).copy_(torch.from_numpy(v))
    params = [getattr(model, n) for n in prm]          # models.py:45-50's ParameterList, in its order
    opt = torch.optim.Adam(params, lr=LR, weight_decay=WD)          # train.py:69
    mf, lf = _ModelsF(), _LayersF(rowptr, N)
    models.F, G.layers.F = mf, lf
    losses = []
    for ep in range(EPOCHS):
        mf.keep = torch.from_numpy(hidden_keep(SEED, ep, N, HID, P))
        lf.start([torch.from_numpy(keep_mask(SEED + 10000 * (ep + 1), an, E, HID, P)) for an in AGGS])
        model.train()
        opt.zero_grad()
        output = model(features, adj)
        loss = torch.nn.functional.nll_loss(output[idx_train], labels[idx_train])       # train.py:77
        loss.backward()
        opt.step()
        losses.append(loss.item())
        print("epoch %d loss %.6f" % (ep + 1, loss.item()), flush=True)
    G.layers.F = G.FSHIM
    out["losses"] = np.array(losses, dtype=np.float64)
    out["final_logp_train"] = output.detach().numpy()[idx_train.numpy()].astype(np.float32)       # of the last (3rd) forward
    for n in ("weight0", "bias0", "weight1", "bias1", "weight_mean", "weight_mean2"):
        v = getattr(model, n).detach()
        out["final_" + n] = (v.numpy()[wrows] if n == "weight0" else v.numpy()).astype(np.float32)
        out["final_" + n + "_stats"] = G.stats(v)
    path = os.path.join(HERE, "train3_cora.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
