"""the reference's node-classification training step (train.py:72-86: forward, nll_loss on idx_train,
backward, Adam) captured as one hipGraph (mma_amd/train_step.py).  With dropout off the captured step must walk the same
parameter trajectory as the eager loop; with the reference's always-on dropout it must train and be faster than eager
(the step is launch-bound on Cora-sized graphs)."""
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def _cora_problem(dropout, seed=0):
    import scipy.sparse as sp
    from mma_amd.models import MMAConv
    from mma_amd.utils import sparse_mx_to_torch_sparse_tensor
    z = np.load(os.path.join(HERE, "golden", "cora_h64.npz"))
    rowptr, col = z["rowptr"].astype(np.int64), z["col"].astype(np.int64)
    N, C, nfeat = len(rowptr) - 1, 7, 96
    add_all = [col[rowptr[i]:rowptr[i + 1]] for i in range(N)]
    adj_sp = sp.csr_matrix((np.ones(len(col), np.float32), col, rowptr), shape=(N, N))
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, C, N)
    feats = (np.eye(C)[labels] @ rng.standard_normal((C, nfeat)) + 2.0 * rng.standard_normal((N, nfeat))).astype(np.float32)
    torch.manual_seed(42)
    model = MMAConv(add_all, "new_sigmoid", 2, nfeat, 64, C, dropout, ["mean", "mean2"], DEV).to(DEV)
    for p in model.parameters():                               # the reference leaves init to the layers; unused masks too
        if p.dim() == 2 and not torch.isfinite(p).all():
            torch.nn.init.uniform_(p, -0.1, 0.1)
    adj = sparse_mx_to_torch_sparse_tensor(adj_sp).to(DEV)
    x, y = torch.from_numpy(feats).to(DEV), torch.from_numpy(labels).to(DEV)
    idx_train = torch.arange(0, 1500, device=DEV)
    return model, adj, x, y, idx_train


def _used(model):
    return [p for p in model.parameters() if p.requires_grad]


def test_graphed_step_matches_eager_without_dropout():
    from mma_amd.train_step import GraphedTrainStep
    finals = []
    for graphed in (False, True):
        model, adj, x, y, idx = _cora_problem(0.0)
        model.train()
        opt = torch.optim.Adam(_used(model), lr=0.01, weight_decay=5e-4, capturable=True)
        loss_fn = lambda: F.nll_loss(model(x, adj)[idx], y[idx])
        if graphed:
            step = GraphedTrainStep(model, opt, loss_fn, warmup=3)      # 3 eager warm-up steps, UNDONE again: construction
            losses = [step().item() for _ in range(6)]                  # leaves the model where it was, a call = one epoch of train.py:72-80
        else:
            losses = []
            for _ in range(6):
                opt.zero_grad(set_to_none=False)
                loss = loss_fn(); loss.backward(); opt.step()
                losses.append(loss.item())
        finals.append((losses[-1], [p.detach().clone() for p in (model.weight0, model.weight1, model.weight_mean, model.bias1)]))
    (le, pe), (lg, pg) = finals
    assert abs(le - lg) <= 1e-4 * abs(le), (le, lg)                    # 6 optimizer steps either way
    for a, b in zip(pe, pg):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_graphed_step_trains_and_beats_eager_launch_overhead():
    from mma_amd.train_step import GraphedTrainStep
    model, adj, x, y, idx = _cora_problem(0.5)
    model.train()
    opt = torch.optim.Adam(_used(model), lr=0.01, weight_decay=5e-4, capturable=True)
    loss_fn = lambda: F.nll_loss(model(x, adj)[idx], y[idx])

    def eager():
        opt.zero_grad(set_to_none=False)
        loss = loss_fn(); loss.backward(); opt.step()
        return loss
    first = [eager().item() for _ in range(5)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        eager()
    torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / 20
    step = GraphedTrainStep(model, opt, loss_fn)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        loss = step()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 100
    last = [step().item() for _ in range(5)]
    print("train step on Cora structure: eager %.3f ms, hipGraph replay %.3f ms" % (t_eager * 1e3, t_graph * 1e3))
    assert np.isfinite(last).all() and np.mean(last) < 0.8 * np.mean(first), (first, last)
    assert len({round(v, 6) for v in last}) > 1                        # fresh dropout bits per replay
    assert t_graph < 0.7 * t_eager, (t_eager, t_graph)


@pytest.mark.parametrize("N,C,n_idx", [(300, 7, 140), (50, 100, 50), (5, 3, 1)])
def test_fused_nll_loss_matches_torch(N, C, n_idx):
    """K10 against plain torch fp32: log_softmax(dim=1) -> nll_loss(output[idx], labels[idx]) (models.py:68, train.py:77)."""
    from mma_amd.train_step import fused_nll_loss
    g = torch.Generator().manual_seed(N + C)
    logits = (torch.randn(N, C, generator=g) * 5).to(DEV).requires_grad_(True)
    labels = torch.randint(0, C, (N,), generator=g).to(DEV)
    idx = torch.randperm(N, generator=g)[:n_idx].to(DEV)
    loss, logp = fused_nll_loss(logits, idx, labels)
    gx, = torch.autograd.grad(loss * 3.0, [logits])
    ref_in = logits.detach().clone().requires_grad_(True)
    ref_logp = F.log_softmax(ref_in, dim=1)
    ref_loss = F.nll_loss(ref_logp[idx], labels[idx])
    ref_g, = torch.autograd.grad(ref_loss * 3.0, [ref_in])
    assert torch.allclose(logp, ref_logp, rtol=1e-6, atol=1e-6) and torch.allclose(loss, ref_loss, rtol=1e-6, atol=1e-6)
    assert torch.allclose(gx, ref_g, rtol=1e-5, atol=2e-6)       # softmax - 1 on the label column cancels: a few 1e-7 absolute either way
    assert torch.equal(gx[~torch.isin(torch.arange(N, device=DEV), idx)], torch.zeros(N - n_idx, C, device=DEV))


def test_fused_adam_matches_torch_adam():
    """K11 against torch.optim.Adam (train.py:69: lr 0.01, weight_decay 5e-4) over several steps on tensors of awkward sizes."""
    from mma_amd.train_step import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(1433, 64), (64,), (64, 7), (7,), (128, 64), (5000,), (1,), (4097,)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(6)]
    res = []
    for fused in (False, True):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
        opt = (FusedAdam if fused else torch.optim.Adam)(ps, lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        for p in ps:
            p.grad = torch.zeros_like(p)
        for step_g in grads:
            for p, gg in zip(ps, step_g):
                p.grad.copy_(gg)
            opt.step()
        res.append([p.detach().clone() for p in ps])
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (a - b).abs().max()


def test_fused_adam_takes_fresh_gradient_tensors_every_step():
    """zero_grad(set_to_none=True) + NEW gradient tensors per step (what autograd hands over): the gradient pointers travel with
    the K11 launch, so nothing has to stay allocated - same trajectory as torch.optim.Adam; more tensors than fit the launch
    arguments fall back to the table (gradients must then stay in place, and a moved one raises)."""
    from mma_amd.train_step import FusedAdam
    g = torch.Generator().manual_seed(2)
    shapes = [(129, 3), (4097,), (64, 16), (1,), (75, 225)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(5)]
    res = []
    for fused in (False, True):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
        opt = (FusedAdam if fused else torch.optim.Adam)(ps, lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        if fused:
            assert opt.fresh_gradients_ok()
        keep = []
        for step_g in grads:
            opt.zero_grad(set_to_none=True)
            assert all(p.grad is None for p in ps)
            for p, gg in zip(ps, step_g):
                p.grad = gg.to(DEV).clone()
                keep.append(p.grad)                       # old gradient tensors stay alive: the new ones cannot reuse their addresses
            opt.step()
        res.append([p.detach().clone() for p in ps])
        if fused:
            assert float(opt.state[ps[0]]["step"]) == len(grads)
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (a - b).abs().max()
    # beyond the by-value limit: the table form, which needs the gradients to stay where they were
    many = [torch.nn.Parameter(torch.randn(5, device=DEV)) for _ in range(130)]
    opt = FusedAdam(many, lr=0.01)
    assert not opt.fresh_gradients_ok()
    for p in many:
        p.grad = torch.ones_like(p)
    opt.step()
    opt.zero_grad(set_to_none=True)                       # kept allocated on purpose
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0 for p in many)
    many[3].grad = torch.ones_like(many[3])
    with pytest.raises(RuntimeError, match="gradient buffer moved"):
        opt.step()


def test_fused_adam_checkpoint_round_trip_like_torch_adam():
    """FusedAdam checkpoints and resumes like the torch.optim.Adam it replaces.  Three steps, state_dict(), then
    (a) a FRESH FusedAdam loads it before its first step, (b) a LIVE FusedAdam (tables built, two unrelated steps taken) loads it
    - the loaded moments and step count must land in the buffers the device table points at -, (c) a FusedAdam loads a
    torch.optim.Adam checkpoint; all continue for three more steps exactly like an uninterrupted torch.optim.Adam."""
    import copy
    from mma_amd.train_step import FusedAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(33, 7), (4097,), (64, 16), (1,)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(6)]
    hyper = dict(lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)

    def make(cls, values):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in values]
        for q in ps:
            q.grad = torch.zeros_like(q)
        return ps, cls(ps, **hyper)

    def run(ps, opt, steps):
        for step_g in steps:
            for q, gg in zip(ps, step_g):
                q.grad.copy_(gg)
            opt.step()

    ref_ps, ref_opt = make(torch.optim.Adam, init)
    run(ref_ps, ref_opt, grads[:3])
    ref_mid = [q.detach().clone() for q in ref_ps]
    torch_ckpt = copy.deepcopy(ref_opt.state_dict())
    run(ref_ps, ref_opt, grads[3:])

    ps, opt = make(FusedAdam, init)
    run(ps, opt, grads[:3])
    ckpt = copy.deepcopy(opt.state_dict())
    mid = [q.detach().clone() for q in ps]
    assert len(ckpt["state"]) == len(shapes) and all(float(st["step"]) == 3.0 for st in ckpt["state"].values())

    def check(ps2, what):
        for a, b in zip(ps2, ref_ps):
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (what, (a - b).abs().max())

    # (a) fresh optimizer, load before the first step
    ps_a, opt_a = make(FusedAdam, mid)
    opt_a.load_state_dict(copy.deepcopy(ckpt))
    run(ps_a, opt_a, grads[3:])
    check(ps_a, "fresh load")
    # (b) live optimizer: its table exists and its state is elsewhere; load must overwrite the live buffers in place
    ps_b, opt_b = make(FusedAdam, mid)
    run(ps_b, opt_b, grads[4:])                   # two unrelated steps: moments and step count are now wrong on purpose
    with torch.no_grad():
        for q, t in zip(ps_b, mid):
            q.copy_(t)
    before = [opt_b.state[q]["exp_avg"].data_ptr() for q in ps_b]
    opt_b.load_state_dict(copy.deepcopy(ckpt))
    assert before == [opt_b.state[q]["exp_avg"].data_ptr() for q in ps_b]          # same buffers, new contents
    run(ps_b, opt_b, grads[3:])
    check(ps_b, "live load")
    # (c) a torch.optim.Adam checkpoint resumes under FusedAdam
    ps_c, opt_c = make(FusedAdam, ref_mid)
    opt_c.load_state_dict(torch_ckpt)
    run(ps_c, opt_c, grads[3:])
    check(ps_c, "torch checkpoint")


def test_fused_step_kernels_inside_the_graphed_step():
    """K10 + K11 are capture-safe: the hipGraph of (forward, fused nll, backward, FusedAdam) replays the eager fused trajectory
    (dropout off), and the fused step needs fewer kernels per replay than torch's loss + per-tensor Adam."""
    from mma_amd.train_step import FusedAdam, GraphedTrainStep
    runs = {}
    for graphed in (False, True):
        model, adj, x, y, idx = _cora_problem(0.0)
        model.train()
        opt = FusedAdam(_used(model), lr=0.01, weight_decay=5e-4)
        loss_fn = lambda: model.nll_loss(x, adj, idx, y)[0]
        if graphed:
            step = GraphedTrainStep(model, opt, loss_fn, warmup=3)
            assert float(opt.state[_used(model)[0]]["step"]) == 0.0       # the warm-up steps were undone: Adam has taken no step yet
            losses = [step().item() for _ in range(6)]
            assert float(opt.state[_used(model)[0]]["step"]) == 6.0
        else:
            losses = []
            for _ in range(6):
                opt.zero_grad(set_to_none=False)
                loss = loss_fn(); loss.backward(); opt.step()
                losses.append(loss.item())
        runs[graphed] = (losses[-1], model.weight1.detach().clone())
    (le, we), (lg, wg) = runs[False], runs[True]
    assert abs(le - lg) <= 1e-4 * abs(le), (le, lg)
    assert torch.allclose(we, wg, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n", [1, 64, 257, 10000])
def test_fused_l1_loss_matches_torch(n):
    """K12 against plain torch fp32: (out.squeeze() - y).abs().mean() (graph_regression/mma.py:156), incl. exact zeros (sign(0) = 0)."""
    from mma_amd.train_step import fused_l1_loss
    g = torch.Generator().manual_seed(n)
    pred = torch.randn(n, 1, generator=g).to(DEV).requires_grad_(True)
    y = torch.randn(n, generator=g).to(DEV)
    with torch.no_grad():
        y[::7] = pred[::7, 0]                                   # exact ties: gradient 0 there
    loss = fused_l1_loss(pred.squeeze(-1), y)
    gp, = torch.autograd.grad(loss * 2.5, [pred])
    ref_in = pred.detach().clone().requires_grad_(True)
    ref = (ref_in.squeeze(-1) - y).abs().mean()
    rg, = torch.autograd.grad(ref * 2.5, [ref_in])
    assert torch.allclose(loss, ref, rtol=1e-6, atol=1e-7)
    assert torch.allclose(gp, rg, rtol=1e-6, atol=0) and gp.shape == pred.shape
    assert (gp[::7] == 0).all()
