"""The CALLERS of the hot path against the reference (fixture tests/golden/train3_cora.npz made by
tests/golden/gen_train_golden.py from the reference's layers.GraphConvolution, models.MMAConv and the step of train.py:72-80
on the real Cora data, every dropout replayed from seeded keep masks):
  * CPU: the oracle's restatement of the model (oracle/nc_oracle.model_forward) + torch Adam walks the reference's loss
    trajectory - pins the oracle for the whole training step;
  * GPU: the drop-in mma_amd.models.MMAConv (HIP kernels, sparse-feature first layer) does the same."""
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import check_close
from golden.inputs import ALL_MASK_NAMES, keep_mask, rng_uniform, sha

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


class Fix:
    def __init__(self):
        z = self.z = np.load(os.path.join(HERE, "golden", "train3_cora.npz"), allow_pickle=False)
        self.seed, self.N, self.nfeat, self.nclass, self.H = int(z["seed"]), int(z["N"]), int(z["nfeat"]), int(z["nclass"]), int(z["hidden"])
        self.p, self.aggs, self.epochs, self.lr, self.wd = float(z["p"]), [str(a) for a in z["aggs"]], int(z["epochs"]), float(z["lr"]), float(z["wd"])
        self.rowptr, self.col = z["rowptr"].astype(np.int64), z["col"].astype(np.int64)
        self.E = len(self.col)
        f = np.zeros((self.N, self.nfeat), dtype=np.float32)
        f[z["feat_row"].astype(np.int64), z["feat_col"].astype(np.int64)] = z["feat_val"]
        self.features = f
        self.labels, self.idx_train = z["labels"].astype(np.int64), z["idx_train"].astype(np.int64)
        s, nh = self.seed, self.H
        prm = {"weight0": rng_uniform(s + 1, (self.nfeat, nh), 1.0 / np.sqrt(nh)), "bias0": rng_uniform(s + 2, (nh,), 1.0 / np.sqrt(nh)),
               "weight1": rng_uniform(s + 3, (nh, self.nclass), 1.0 / np.sqrt(nh)), "bias1": rng_uniform(s + 4, (self.nclass,), 1.0 / np.sqrt(nh))}
        for i, n in enumerate(ALL_MASK_NAMES):
            prm["weight_" + n] = rng_uniform(s + 100 + i, (2 * nh, nh), 1.0 / np.sqrt(nh))
        assert sha(prm["weight0"]) == str(z["sha_weight0"]) and sha(prm["weight_mean"]) == str(z["sha_mask_mean"])
        self.prm = prm
        self.add_all = [self.col[self.rowptr[i]:self.rowptr[i + 1]] for i in range(self.N)]

    def hidden_keep(self, ep):
        return (np.random.default_rng(self.seed + 7000 + ep).random((self.N, self.H), dtype=np.float32) >= self.p).astype(np.float32)

    def mask_keep(self, ep, agg):
        return keep_mask(self.seed + 10000 * (ep + 1), agg, self.E, self.H, self.p)


@pytest.fixture(scope="module")
def fx():
    return Fix()


_TRUTH = {}


def train_truth(fx):
    """The same three steps by the CPU oracle in FLOAT64: the exact trajectory of the reference's formulas on these inputs.
    |reference fp32 result - this| is the reference's own rounding noise, which sets the slack of the signed-sum comparisons
    (golden_util.check_close): after three Adam steps on un-normalised Cora features the logits reach +-50 and single
    log-probabilities of the reference are ~1e-3 away from exact."""
    if "t" not in _TRUTH:
        from oracle import nc_oracle as O
        z = fx.z
        x = torch.from_numpy(fx.features).double()
        used = ["weight0", "bias0", "weight1", "bias1"] + ["weight_" + a for a in fx.aggs]
        prm = {n: torch.from_numpy(v.copy()).double().requires_grad_(n in used) for n, v in fx.prm.items()}
        opt = torch.optim.Adam([prm[n] for n in fx.prm], lr=fx.lr, weight_decay=fx.wd)
        labels, idx = torch.from_numpy(fx.labels), torch.from_numpy(fx.idx_train)
        losses = []
        for ep in range(fx.epochs):
            opt.zero_grad()
            out = O.model_forward(x, prm, fx.aggs, fx.rowptr, fx.col, z["adj_row"], z["adj_col"], z["adj_val"], "new_sigmoid", fx.p,
                                  fx.hidden_keep(ep), {a: fx.mask_keep(ep, a) for a in fx.aggs})
            loss = torch.nn.functional.nll_loss(out[idx], labels[idx])
            loss.backward()
            opt.step()
            losses.append(loss.item())
        t = {"losses": losses, "final_logp_train": out[idx].detach().numpy()}
        for n in ("weight0", "bias0", "weight1", "bias1", "weight_mean", "weight_mean2"):
            t["final_" + n] = prm[n].detach().numpy()
        _TRUTH["t"] = t
    return _TRUTH["t"]


def _check_final(fx, named, out_idx):
    z, tr = fx.z, train_truth(fx)
    check_close(out_idx, z["final_logp_train"], None, None, what="final log-probabilities (train rows)", signed_sum=True,
                truth=tr["final_logp_train"])
    for n, t in named.items():
        rows = z["wrows"].astype(np.int64) if n == "weight0" else None
        want = z["final_" + n]
        check_close(t, want, rows, z["final_" + n + "_stats"], what="final " + n, signed_sum=True,
                    truth=tr["final_" + n].reshape(-1, want.shape[-1]) if want.ndim > 1 else tr["final_" + n])


def test_oracle_walks_the_reference_training_trajectory(fx):
    from oracle import nc_oracle as O
    z = fx.z
    x = torch.from_numpy(fx.features)
    used = ["weight0", "bias0", "weight1", "bias1"] + ["weight_" + a for a in fx.aggs]
    prm = {n: torch.from_numpy(v.copy()).requires_grad_(n in used) for n, v in fx.prm.items()}
    # f-1: GraphConvolution.forward + its weight / bias gradients
    g = O.gcn_forward(x, prm["weight0"], prm["bias0"], z["adj_row"], z["adj_col"], z["adj_val"])
    cot = torch.from_numpy(np.random.default_rng(fx.seed + 9).standard_normal((fx.N, fx.H), dtype=np.float32))
    gw, gb = torch.autograd.grad((g * cot).sum(), [prm["weight0"], prm["bias0"]])
    check_close(g, z["gcn_out"], z["rows"].astype(np.int64), z["gcn_out_stats"], what="GraphConvolution out", signed_sum=True)
    check_close(gw, z["gcn_gweight"], z["wrows"].astype(np.int64), z["gcn_gweight_stats"], what="GraphConvolution gweight", signed_sum=True)
    check_close(gb, z["gcn_gbias"], None, None, what="GraphConvolution gbias", signed_sum=True)
    # f-3: three steps of train.py:72-80
    opt = torch.optim.Adam([prm[n] for n in fx.prm], lr=fx.lr, weight_decay=fx.wd)       # all 25, like models.py:45-50
    labels, idx = torch.from_numpy(fx.labels), torch.from_numpy(fx.idx_train)
    losses = []
    for ep in range(fx.epochs):
        opt.zero_grad()
        out = O.model_forward(x, prm, fx.aggs, fx.rowptr, fx.col, z["adj_row"], z["adj_col"], z["adj_val"], "new_sigmoid", fx.p,
                              fx.hidden_keep(ep), {a: fx.mask_keep(ep, a) for a in fx.aggs})
        loss = torch.nn.functional.nll_loss(out[idx], labels[idx])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, z["losses"], rtol=1e-5, atol=1e-5), (losses, z["losses"])
    _check_final(fx, {n: prm[n].detach() for n in ("weight0", "bias0", "weight1", "bias1", "weight_mean", "weight_mean2")}, out[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("sparse_first_layer,fused_step", [(True, False), (False, False), (True, True)])
def test_hip_model_walks_the_reference_training_trajectory(fx, sparse_first_layer, fused_step):
    """fused_step: the loss through K10 (fused log_softmax + nll_loss) and the update through K11 (multi-tensor Adam) instead
    of torch's element-wise ops and per-tensor optimizer - SURVEY 8 f-4, checked against the REFERENCE's trajectory."""
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.train_step import FusedAdam
    from mma_amd.layers import GraphConvolution
    from mma_amd.models import MMAConv
    z = fx.z
    x = torch.from_numpy(fx.features).to(DEV)
    idxa = torch.from_numpy(np.stack([z["adj_row"], z["adj_col"]]).astype(np.int64))
    adj = torch.sparse_coo_tensor(idxa, torch.from_numpy(z["adj_val"]), (fx.N, fx.N)).to(DEV)
    model = MMAConv(fx.add_all, "new_sigmoid", 2, fx.nfeat, fx.H, fx.nclass, fx.p, fx.aggs, DEV)
    if not sparse_first_layer:
        model.gc1.SPARSE_BELOW = 0.0                                   # dense GEMM first layer
    with torch.no_grad():
        for n, v in fx.prm.items():
            getattr(model, n).copy_(torch.from_numpy(v))
    # f-1: the first layer alone, against the reference's GraphConvolution.forward and its gradients
    g = model.gc1(x, adj)
    assert (model.gc1._xg[2] is not None) == sparse_first_layer       # Cora's features are 1.3 % dense -> CSR kernel
    cot = torch.from_numpy(np.random.default_rng(fx.seed + 9).standard_normal((fx.N, fx.H), dtype=np.float32)).to(DEV)
    gw, gb = torch.autograd.grad((g * cot).sum(), [model.weight0, model.bias0])
    check_close(g, z["gcn_out"], z["rows"].astype(np.int64), z["gcn_out_stats"], what="GraphConvolution out", signed_sum=True)
    check_close(gw, z["gcn_gweight"], z["wrows"].astype(np.int64), z["gcn_gweight_stats"], what="GraphConvolution gweight", signed_sum=True)
    check_close(gb, z["gcn_gbias"], None, None, what="GraphConvolution gbias", signed_sum=True)
    # f-3: three steps
    params = [getattr(model, n) for n in fx.prm]
    opt = (FusedAdam if fused_step else torch.optim.Adam)(params, lr=fx.lr, weight_decay=fx.wd)
    labels, idx = torch.from_numpy(fx.labels).to(DEV), torch.from_numpy(fx.idx_train).to(DEV)
    losses = []
    model.train()
    for ep in range(fx.epochs):
        model.hidden_keep = torch.from_numpy(fx.hidden_keep(ep)).to(DEV)
        keep = torch.from_numpy(np.stack([fx.mask_keep(ep, a) for a in fx.aggs]).astype(np.uint8)).to(DEV)
        model.gc2.drop_override = Fn.DropoutSpec(fx.p, keep=keep)
        opt.zero_grad()
        if fused_step:
            loss, out = model.nll_loss(x, adj, idx, labels)
        else:
            out = model(x, adj)
            loss = torch.nn.functional.nll_loss(out[idx], labels[idx])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, z["losses"], rtol=1e-5, atol=1e-5), (losses, z["losses"])
    _check_final(fx, {n: getattr(model, n).detach() for n in ("weight0", "bias0", "weight1", "bias1", "weight_mean", "weight_mean2")}, out[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("sparse_first_layer,fused_step", [(True, False), (False, False), (True, True)])
def test_trajectory_noise_is_single_step_rounding_amplified_by_adam(fx, capsys, sparse_first_layer, fused_step):
    """Round-3 VERDICT item 4: the final log-probabilities of the 3-epoch trajectory needed 4.32 (of the 8 then allowed) times the
    reference's own fp32 noise.  WHICH operation moves them?  Three trajectories on the fixture's inputs - the oracle in float64 (exact
    arithmetic of the reference's formulas), the oracle in float32 (the reference's numerics: pinned to it within the strict bar by
    test_oracle_walks_the_reference_training_trajectory), the HIP model - compared epoch by epoch, free-running and TEACHER-FORCED
    (the HIP model restarted every epoch from the float32 oracle's parameters and Adam state, so that it takes exactly one step from
    the same point).  Finding (printed below, quoted in DESIGN.md 6): from the same point the HIP step is as close to the float32
    oracle as that is to exact arithmetic (multiple <= ~1.5 in every epoch, every kernel involved: sparse first layer, fused
    aggregate, SpMM tail, log-softmax, Adam); free-running, the differences of epoch 1 - rounding-level differences in gradient
    elements whose magnitude is near Adam's eps, where the update lr g / (sqrt(v) + eps) turns a relative change of g into an absolute
    change of the weight - are carried and amplified by the next steps like the reference's own noise is.  It is accumulation order
    through an ill-conditioned optimiser step, not a kernel."""
    import mma_amd
    from mma_amd import functional as Fn
    from mma_amd.models import MMAConv
    from oracle import nc_oracle as O
    z = fx.z
    labels_c, idx_c = torch.from_numpy(fx.labels), torch.from_numpy(fx.idx_train)
    used = ["weight0", "bias0", "weight1", "bias1"] + ["weight_" + a for a in fx.aggs]

    def oracle_run(dtype):
        x = torch.from_numpy(fx.features).to(dtype)
        prm = {n: torch.from_numpy(v.copy()).to(dtype).requires_grad_(n in used) for n, v in fx.prm.items()}
        opt = torch.optim.Adam([prm[n] for n in fx.prm], lr=fx.lr, weight_decay=fx.wd)
        snaps, logps = [], []
        for ep in range(fx.epochs):
            snaps.append(({n: prm[n].detach().clone() for n in used},
                          {n: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state[prm[n]].items()} for n in used if prm[n] in opt.state}))
            opt.zero_grad()
            out = O.model_forward(x, prm, fx.aggs, fx.rowptr, fx.col, z["adj_row"], z["adj_col"], z["adj_val"], "new_sigmoid", fx.p,
                                  fx.hidden_keep(ep), {a: fx.mask_keep(ep, a) for a in fx.aggs})
            torch.nn.functional.nll_loss(out[idx_c], labels_c[idx_c]).backward()
            opt.step()
            logps.append(out[idx_c].detach().double().numpy())
        return snaps, logps

    snaps32, logp32 = oracle_run(torch.float32)
    _, logp64 = oracle_run(torch.float64)

    x = torch.from_numpy(fx.features).to(DEV)
    idxa = torch.from_numpy(np.stack([z["adj_row"], z["adj_col"]]).astype(np.int64))
    adj = torch.sparse_coo_tensor(idxa, torch.from_numpy(z["adj_val"]), (fx.N, fx.N)).to(DEV)
    labels, idx = labels_c.to(DEV), idx_c.to(DEV)

    from mma_amd.train_step import FusedAdam
    Opt = FusedAdam if fused_step else torch.optim.Adam

    def hip_model():
        m = MMAConv(fx.add_all, "new_sigmoid", 2, fx.nfeat, fx.H, fx.nclass, fx.p, fx.aggs, DEV)
        if not sparse_first_layer:
            m.gc1.SPARSE_BELOW = 0.0                                   # dense (library) GEMM for the first layer
        with torch.no_grad():
            for n, v in fx.prm.items():
                getattr(m, n).copy_(torch.from_numpy(v))
        m.train()
        return m

    def hip_epoch(m, opt, ep):
        m.hidden_keep = torch.from_numpy(fx.hidden_keep(ep)).to(DEV)
        keep = torch.from_numpy(np.stack([fx.mask_keep(ep, a) for a in fx.aggs]).astype(np.uint8)).to(DEV)
        m.gc2.drop_override = Fn.DropoutSpec(fx.p, keep=keep)
        opt.zero_grad()
        if fused_step:
            loss, out = m.nll_loss(x, adj, idx, labels)
        else:
            out = m(x, adj)
            loss = torch.nn.functional.nll_loss(out[idx], labels[idx])
        loss.backward()
        opt.step()
        return out[idx].detach().double().cpu().numpy()

    def need(got, ep):
        ref, truth = logp32[ep], logp64[ep]
        noise = np.abs(ref - truth).max(1, keepdims=True)
        err = np.abs(got - ref)
        over = np.maximum(err - (1e-5 + 1e-5 * np.abs(ref)), 0.0) / np.maximum(noise, 1e-30)
        return float(over[np.broadcast_to(noise, err.shape) > 0].max()), float(np.abs(ref - truth).max()), float(err.max())

    m = hip_model()
    opt = Opt([getattr(m, n) for n in fx.prm], lr=fx.lr, weight_decay=fx.wd)
    free = [need(hip_epoch(m, opt, ep), ep) for ep in range(fx.epochs)]
    forced = []
    for ep in range(fx.epochs):
        m = hip_model()
        opt = Opt([getattr(m, n) for n in fx.prm], lr=fx.lr, weight_decay=fx.wd)
        prm32, st32 = snaps32[ep]
        with torch.no_grad():
            for n in used:
                getattr(m, n).copy_(prm32[n])
        for n, st in st32.items():                                   # Adam's moments and step of the float32 oracle at this point
            opt.state[getattr(m, n)] = {k: (v.to(DEV).float() if torch.is_tensor(v) else v) for k, v in st.items()}
        forced.append(need(hip_epoch(m, opt, ep), ep))
    with capsys.disabled():
        print("\n[trajectory noise] variant: %s first layer, %s loss + optimiser" % ("sparse (K5)" if sparse_first_layer else "dense (library GEMM)",
                                                                                      "fused K10 + K11" if fused_step else "torch"))
        print("[trajectory noise] epoch | max |fp32 oracle - fp64| of the train log-probabilities | free-running HIP: max |HIP - fp32|, multiple of "
              "the row's reference noise needed | teacher-forced HIP: the same")
        for ep in range(fx.epochs):
            print("[trajectory noise]   %d   | %.3g | %.3g, %.2f | %.3g, %.2f" % (ep + 1, free[ep][1], free[ep][2], free[ep][0], forced[ep][2], forced[ep][0]))
    assert max(f[0] for f in forced) <= 2.0, forced            # ONE step from the same point: within twice the reference's own noise
    assert max(f[0] for f in free) <= 6.0, free
