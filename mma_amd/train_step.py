"""The training step of the reference's node-classification loop as ONE hipGraph.

`train.py:72-86` runs, per epoch: zero_grad, model(features, adj), `F.nll_loss(output[idx_train], labels[idx_train])`,
backward, Adam step.  On Cora / Pubmed every kernel of that step moves a few MB (2-8 us at the HBM roofline), so the step
is launch-bound on any GPU: ~150 launches of 5-10 us host time each.  The MI355X answer is not a fused-kernel zoo but a
captured graph: the full-batch graph, the features and the labels are static, the drop-in layers are capture-safe
(stream-ordered C-ABI calls, no host sync, dropout seed re-drawn on the device inside the graph when
`graph_capturable=True`), and `torch.optim.Adam(capturable=True)` keeps its step count on the device - so forward, loss,
backward and the optimizer update replay as one `hipGraphLaunch`.

    step = GraphedTrainStep(model, optimizer, lambda: F.nll_loss(model(features, adj)[idx_train], labels[idx_train]))
    for epoch in range(200):
        loss = step()            # a 0-dim device tensor (call .item() only when you want to print it)
"""
import struct

import torch

from . import _lib
from ._lib import call, ptr, require_gpu, stream_ptr


_CHECKED_TARGETS = {}        # (idx ptr, version, labels ptr, version, N, C) -> True: one host check per (idx, labels) pair


def _check_targets(idx, labels, N, C):
    """What F.nll_loss / advanced indexing would have raised for (train.py:77 `F.nll_loss(output[idx_train], labels[idx_train])`):
    a row id outside [0,N), a class outside [0,C) - torch's ignore_index=-100 included, the reference never uses it - and, a K10
    precondition, DUPLICATE rows (the backward assigns one gradient row per training row).  Checked once per tensor pair and
    in-place version (one host sync, at the first eager call: the warm-up of a captured step), never inside a capture."""
    key = (idx.data_ptr(), idx._version, labels.data_ptr(), labels._version, int(N), int(C), idx.numel())
    if key in _CHECKED_TARGETS:
        return
    if torch.cuda.is_current_stream_capturing():
        return          # first call inside a capture: nothing may sync here; the kernels skip out-of-range rows / classes
    if idx.numel():
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= N:
            raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (lo if lo < 0 else hi, N))
        if int(torch.unique(idx).numel()) != idx.numel():
            raise ValueError("fused_nll_loss: duplicate rows in idx (the fused backward writes one gradient row per index; "
                             "use F.nll_loss for a multiset of rows)")
        lab = labels[idx]
        lo, hi = int(lab.min()), int(lab.max())
        if lo < 0 or hi >= C:
            raise IndexError("Target %d is out of bounds." % (lo if lo < 0 else hi))
    if len(_CHECKED_TARGETS) > 64:
        _CHECKED_TARGETS.clear()
    _CHECKED_TARGETS[key] = True


class _FusedNLL(torch.autograd.Function):
    """K10: log_softmax over the classes + mean nll over the rows `idx` (models.py:68, train.py:77)."""

    @staticmethod
    def forward(ctx, logits, idx, labels):
        require_gpu(logits, idx, labels)
        logits = logits.contiguous()
        N, C = logits.shape
        assert idx.dtype == torch.int64 and labels.dtype == torch.int64 and labels.numel() == N
        _check_targets(idx, labels, N, C)
        logp = torch.empty_like(logits)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        call("mma_logsoftmax_nll_fwd", ptr(logits), C, ptr(idx), ptr(labels), idx.numel(), ptr(logp), C, ptr(loss), N, C, stream_ptr())
        ctx.save_for_backward(logp, idx, labels)
        ctx.mark_non_differentiable(logp)
        return loss, logp

    @staticmethod
    def backward(ctx, gloss, _glogp):
        logp, idx, labels = ctx.saved_tensors
        N, C = logp.shape
        gx = torch.empty_like(logp)
        gl = gloss.contiguous().to(torch.float32)
        call("mma_logsoftmax_nll_bwd", ptr(logp), C, ptr(idx), ptr(labels), idx.numel(), ptr(gl), ptr(gx), C, N, C, stream_ptr())
        return gx, None, None


class _FusedL1(torch.autograd.Function):
    """K12: mean |pred - target| (graph_regression/mma.py:156) in one launch each way."""

    @staticmethod
    def forward(ctx, pred, target):
        require_gpu(pred, target)
        assert pred.numel() == target.numel() and pred.numel() >= 1 and pred.dtype == torch.float32
        p, t = pred.contiguous().view(-1), target.to(torch.float32).contiguous().view(-1)
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        call("mma_l1_loss_fwd", ptr(p), ptr(t), p.numel(), ptr(loss), stream_ptr())
        ctx.save_for_backward(p, t)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, gloss):
        p, t = ctx.saved_tensors
        g = torch.empty_like(p)
        gl = gloss.contiguous().to(torch.float32)
        call("mma_l1_loss_bwd", ptr(p), ptr(t), p.numel(), ptr(gl), ptr(g), stream_ptr())
        return g.view(ctx.shape), None


def fused_l1_loss(pred, target):
    """(pred - target).abs().mean() - the loss of the graph-regression script (mma.py:156) - as one kernel forward, one backward."""
    return _FusedL1.apply(pred, target)


def fused_nll_loss(logits, idx, labels):
    """-> (mean nll over the rows idx, log_softmax(logits)); labels is indexed by node (the reference's labels[idx_train])."""
    return _FusedNLL.apply(logits, idx.contiguous(), labels.contiguous())


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) semantics (train.py:69; no amsgrad) with ONE K11 launch for all
    parameter tensors and the step counter in device memory (hipGraph-capturable).  fp32 CUDA parameters only; parameters
    whose .grad is None at the first step are left out for good (the reference's unused masks never get a gradient).

    State lives where torch.optim.Adam keeps it - `state[p] = {"step", "exp_avg", "exp_avg_sq"}` - so `state_dict()` /
    `load_state_dict()` checkpoint and resume like the optimizer it replaces, also from a torch.optim.Adam checkpoint; the step
    counter is ONE 0-dim fp32 device tensor per group, shared by that group's parameters.  Loading after the first step copies
    INTO the buffers the device table (and any captured hipGraph) already points at; they are never replaced."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))
        self._tables = None
        self._max_by_value = _lib.query("mma_adam_max_grads_by_value")

    @staticmethod
    def _rec(p, st):
        return (p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())

    def fresh_gradients_ok(self):
        """True when every group's gradient pointers travel with the launch (<= 120 tensors per group): zero_grad(set_to_none=True)
        is then honoured after the first step too, and a step costs neither a zero-fill nor an accumulating add per parameter."""
        return all(len(g["params"]) <= self._max_by_value for g in self.param_groups)

    def _build(self):
        self._tables = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            require_gpu(*ps)
            dev = ps[0].device if ps else "cpu"
            # one counter per group; a state loaded before the first step brings its own (all parameters of a group step together)
            loaded = [self.state[p]["step"] for p in ps if "step" in self.state[p]]
            step = torch.zeros((), device=dev, dtype=torch.float32)
            if loaded:
                vals = {float(t) for t in loaded}
                if len(vals) != 1:
                    raise RuntimeError("FusedAdam: the parameters of one group carry different step counts %s" % sorted(vals))
                step.fill_(vals.pop())
            recs, chunk_ids, chunk0 = b"", [], 0
            for t, p in enumerate(ps):
                assert p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                st = self.state[p]
                for key in ("exp_avg", "exp_avg_sq"):
                    if key not in st:
                        st[key] = torch.zeros_like(p)
                    elif st[key].device != p.device or st[key].dtype != torch.float32 or not st[key].is_contiguous():
                        st[key] = st[key].to(device=p.device, dtype=torch.float32).contiguous()
                st["step"] = step
                n_chunks = _lib.query("mma_adam_chunks", p.numel())
                recs += struct.pack("<QQQQqq", p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                    p.numel(), chunk0)
                chunk_ids += [t] * n_chunks
                chunk0 += n_chunks
            raw = recs + struct.pack("<%di" % len(chunk_ids), *chunk_ids) + struct.pack("<i", 0)        # + the launch's ticket counter
            assert len(raw) == _lib.query("mma_adam_table_bytes", len(ps), chunk0)
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev) if ps else None
            by_value = len(ps) <= self._max_by_value
            self._tables.append((gi, ps, table, chunk0, step, [self._rec(p, self.state[p]) for p in ps], by_value,
                                 [p.grad.data_ptr() for p in ps]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._tables is None:
            self._build()
        for gi, ps, table, n_chunks, step, ptrs, by_value, gptrs in self._tables:
            if not ps:
                continue
            group = self.param_groups[gi]        # looked up per step: load_state_dict() replaces the group dicts
            if any(self._rec(p, self.state[p]) != q for p, q in zip(ps, ptrs)):
                raise RuntimeError("FusedAdam: a parameter or moment buffer moved; load checkpoints with load_state_dict()")
            b1, b2 = group["betas"]
            if by_value:            # this step's gradient tensors, wherever autograd put them
                raw = bytearray()
                for p in ps:
                    g = p.grad
                    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                        raise RuntimeError("FusedAdam: a parameter that had a gradient at the first step has none now (or a non-contiguous "
                                           "/ non-fp32 one)")
                    raw += struct.pack("<Q", g.data_ptr())
                call("mma_adam_step_grads", ptr(table), len(ps), n_chunks, ptr(step), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                     bytes(raw), stream_ptr())
            else:
                if any(p.grad is None or p.grad.data_ptr() != q for p, q in zip(ps, gptrs)):
                    raise RuntimeError("FusedAdam: a gradient buffer moved; with more than %d tensors in a group keep gradients allocated "
                                       "(zero_grad(set_to_none=False))" % self._max_by_value)
                call("mma_adam_step", ptr(table), len(ps), n_chunks, ptr(step), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                     stream_ptr())
        return loss

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Before the first step: plain load, `_build` adopts the loaded moments and step.  After it: the loaded values are copied
        into the live buffers (the device table and a captured graph hold their addresses), which stay in `self.state`."""
        live = None
        if self._tables is not None:
            live = {p: dict(self.state[p]) for _, ps, *_ in self._tables for p in ps}
        super().load_state_dict(state_dict)
        if live is None:
            return
        for _gi, ps, _table, _n, step, *_rest in self._tables:
            steps = set()
            for p in ps:
                new, old = self.state.get(p, {}), live[p]
                for key in ("exp_avg", "exp_avg_sq"):
                    if key in new:
                        old[key].copy_(new[key])
                    else:
                        old[key].zero_()
                steps.add(float(new["step"]) if "step" in new else 0.0)
                self.state[p] = old
            if len(steps) > 1:
                raise RuntimeError("FusedAdam: the parameters of one group carry different step counts %s" % sorted(steps))
            if steps:
                step.fill_(steps.pop())

    def zero_grad(self, set_to_none=False):
        keep = self._tables is not None and not all(t[6] for t in self._tables)      # a table that holds gradient addresses
        super().zero_grad(set_to_none=False if keep else set_to_none)


def _set_capturable(model, flag):
    for m in model.modules():
        if hasattr(m, "graph_capturable"):
            m.graph_capturable = flag


class _WarmupState:
    """Everything a warm-up step changes, saved before the warm-up and put back after it: the reference's loops take ONE optimizer step
    per batch (train.py:72-80, mma.py:150-160), and capturing a hipGraph needs a few eager runs first.  Restored IN PLACE (the device
    tables of FusedAdam and the captured graph hold the addresses): parameters, module buffers (BatchNorm's running statistics and
    batch counter), optimizer state (moments and step counters; state created by the warm-up is zeroed = its initial value), the
    dropout seed streams, and the never-zeroed `.grad` of MMAConv's unregistered mask Linears (quirk G2: it accumulates for the whole
    run, so three warm-up backward passes would stay in it)."""

    def __init__(self, model, optimizer):
        self.optimizer = optimizer
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.param_vals = [p.detach().clone() for p in self.params]
        self.buffers = [(b, b.detach().clone()) for b in model.buffers()]
        self.opt_state = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()}
                          for p in self.params if p in optimizer.state and optimizer.state[p]}
        self.seeds, seen = [], set()
        for m in model.modules():
            sd = getattr(m, "_seeds", None)
            sd = getattr(sd, "shared", sd)                 # a SeedSlot shares one DeviceSeeds
            if sd is not None and id(sd) not in seen:
                seen.add(id(sd))
                self.seeds.append((sd.buf, sd.buf.clone()))
        self.model = model
        self.loose = [(q, None if q.grad is None else q.grad.detach().clone()) for q in self._unregistered(model)]

    @staticmethod
    def _unregistered(model):
        out = []
        for m in model.modules():
            f = getattr(m, "unregistered_parameters", None)
            if f is not None:
                out.extend(f())
        return out

    @torch.no_grad()
    def restore(self):
        for p, v in zip(self.params, self.param_vals):
            p.copy_(v)
        for b, v in self.buffers:
            b.copy_(v)
        for p in self.params:
            st = self.optimizer.state.get(p)
            if not st:
                continue
            was = self.opt_state.get(p, {})
            for k, v in st.items():
                if torch.is_tensor(v):
                    if k in was:
                        v.copy_(was[k])
                    else:
                        v.zero_()                          # created by the warm-up: back to its initial value
        for buf, v in self.seeds:
            buf.copy_(v)
        # seed streams created DURING the warm-up (first call of a capturable layer) were drawn from torch's generator then: they
        # simply continue - the reference has no notion of "the seed before the first step" either
        had = {id(q): g for q, g in self.loose}
        for q in self._unregistered(self.model):
            if q.grad is not None:
                g0 = had.get(id(q))
                if g0 is None:
                    q.grad.zero_()                         # None before: an all-zero running sum is the same starting point
                else:
                    q.grad.copy_(g0)


class GraphedTrainStep:
    """Captures `optimizer.zero_grad(); loss = loss_fn(); loss.backward(); optimizer.step()` once and replays it.

    loss_fn: closure over STATIC device tensors (full-batch features / adjacency / labels / index sets).  The optimizer must
    be graph-capturable (`mma_amd.FusedAdam`, `torch.optim.Adam(..., capturable=True)`).  Gradients are fresh tensors every step
    (`set_to_none=True`: no zero-fill and no accumulating add per parameter - the capture's private pool gives them the same addresses
    on every replay) unless the optimizer needs them to stay in place (FusedAdam with more tensors than fit its launch arguments)."""

    def __init__(self, model, optimizer, loss_fn, warmup=3, restore_after_warmup=True):
        """warmup: eager runs of the step before the capture (allocator, lazily built plans, autotuned library kernels).  They are real
        steps, so with restore_after_warmup (default) parameters, buffers, optimizer state and dropout seed streams are put back
        afterwards (`_WarmupState`): construction leaves the model where it was and every later call is exactly one optimizer step,
        like one epoch of train.py:72-80.  restore_after_warmup=False keeps the warm-up steps (round-3 behaviour)."""
        params = [p for g in optimizer.param_groups for p in g["params"]]
        require_gpu(*params)
        for g in optimizer.param_groups:
            if "capturable" in g and not g["capturable"]:
                raise ValueError("GraphedTrainStep needs a capturable optimizer, e.g. torch.optim.Adam(..., capturable=True)")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        _set_capturable(model, True)
        saved = _WarmupState(model, optimizer) if restore_after_warmup and warmup > 0 else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream, as capture requires
            for _ in range(warmup):
                self._eager()
            if saved is not None:
                saved.restore()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=getattr(self.optimizer, "fresh_gradients_ok", lambda: True)())
        loss = self.loss_fn()
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self):
        self.graph.replay()
        return self.loss


class GraphedNetStep:
    """One TRAINING step of the graph-regression Net (graph_regression/mma.py:150-160: forward, L1 loss, backward, Adam) as ONE
    hipGraph over PADDED, static-shape batch buffers - so the per-batch CSR build (K6) is inside the graph too.

    A ZINC mini-batch of 64 molecules (mma.py:52-54) is ~1 500 nodes / 3 200 edges: every kernel of the step moves a few hundred KB
    and the eager step is ~700 launches of 5-10 us host time each (7 ms).  Molecule batches change shape every step, which a captured
    graph cannot; so the batch is copied into buffers of a fixed bucket size (n_pad nodes, e_pad edges) and padded with DUMMY nodes
    that belong to a dummy graph (id = n_graphs) and dummy self-loop edges on those nodes:
      * a dummy node is isolated from every real node, so no real aggregate, message or gradient sees it;
      * BatchNorm takes its statistics over the first n_valid rows only (net.masked_batch_norm);
      * pooling yields n_graphs + 1 rows, the loss reads the first n_graphs.
    The result for the real graphs is what the unpadded step computes (up to BatchNorm's reduction order); `step_eager()` runs the
    SAME padded step without the graph - the replay equals it bit for bit (tests/test_graph_capture_gpu.py).

        step = GraphedNetStep(net, optimizer, n_graphs=64, n_pad=1792, e_pad=3840)
        for data in loader:
            loss = step(data.x, data.edge_index, data.edge_attr, data.batch, data.y)     # 0-dim device tensor
    A batch that does not fit the bucket (or has another graph count) raises; use one step object per bucket."""

    def __init__(self, net, optimizer, n_graphs, n_pad, e_pad, device="cuda:0", warmup=3, restore_after_warmup=True):
        """warmup: eager runs of the padded step on the FIRST batch before the capture.  With restore_after_warmup (default) their
        effects are undone (`_WarmupState`: parameters, BatchNorm statistics, Adam moments / step, dropout seeds, the accumulating
        gradients of the unregistered mask Linears), so the first call is ONE optimizer step on its batch - mma.py:150-160 takes one per
        batch - and the trajectory equals `step_eager()` called once per batch, bit for bit at p = 0."""
        from . import functional as Fn
        self._restore = bool(restore_after_warmup)
        self.net, self.optimizer, self.n_graphs, self.n_pad, self.e_pad = net, optimizer, int(n_graphs), int(n_pad), int(e_pad)
        dev = torch.device(device)
        for g in optimizer.param_groups:
            if "capturable" in g and not g["capturable"]:
                raise ValueError("GraphedNetStep needs a capturable optimizer (mma_amd.FusedAdam, or torch.optim.Adam(capturable=True))")
        self.x = torch.zeros((n_pad, 1), dtype=torch.int64, device=dev)
        self.ei = torch.zeros((2, e_pad), dtype=torch.int64, device=dev)
        self.ea = torch.zeros((e_pad,), dtype=torch.int64, device=dev)
        self.batch = torch.full((n_pad,), self.n_graphs, dtype=torch.int64, device=dev)
        self.y = torch.zeros((self.n_graphs,), dtype=torch.float32, device=dev)
        self.n_valid = torch.zeros((), dtype=torch.int64, device=dev)
        self._fn = Fn
        self.graph = None
        self._warmup = warmup
        _set_capturable(net, True)
        # one dropout-seed launch per step for all MMAConv layers (they run in module order; slot 0 advances the shared set)
        from .mma_conv import MMAConv
        convs = [m for m in net.modules() if isinstance(m, MMAConv)]
        if len(convs) > 1 and dev.type == "cuda":
            shared = Fn.DeviceSeeds(len(convs), dev)
            for i, m in enumerate(convs):
                m._seeds = Fn.SeedSlot(shared, i)
                m._seed_buf = m._seeds.seeds

    def load(self, x, edge_index, edge_attr, batch, y):
        """Copy a batch into the static buffers and pad it (device-side copies on the current stream, no sync)."""
        N, E = int(x.shape[0]), int(edge_index.shape[1])
        if N + 1 > self.n_pad or E > self.e_pad or int(y.numel()) != self.n_graphs:
            raise ValueError("batch of %d nodes / %d edges / %d graphs does not fit the bucket (%d, %d, %d graphs); one dummy node is needed"
                             % (N, E, int(y.numel()), self.n_pad, self.e_pad, self.n_graphs))
        if N > 1:      # the pooling of the padded step takes contiguous node ranges: the batch vector must be sorted (PyG loaders'
            torch._assert_async((batch[1:] >= batch[:-1]).all())              # are).  load() is never captured: checked on EVERY batch
        self.x[:N].copy_(x.reshape(N, 1)); self.x[N:].zero_()
        self.batch[:N].copy_(batch); self.batch[N:].fill_(self.n_graphs)
        self.ei[:, :E].copy_(edge_index); self.ea[:E].copy_(edge_attr); self.ea[E:].zero_()
        if E < self.e_pad:       # dummy self-loops, dealt round-robin over the dummy nodes (short segments: the block kernels' shape)
            d = N + torch.arange(self.e_pad - E, device=self.ei.device) % (self.n_pad - N)
            self.ei[0, E:] = d; self.ei[1, E:] = d
        self.y.copy_(y.reshape(-1).to(torch.float32))
        self.n_valid.fill_(N)

    def forward_backward(self):
        """Loss and gradients of the loaded (padded) batch, no optimizer step."""
        self._fn._GR_GRAPHS.clear()            # the CSR of THIS batch is built inside the step (and inside the captured graph)
        # fresh gradient tensors every step where the optimizer takes them (FusedAdam: pointers travel with the launch; torch's
        # capturable Adam reads p.grad at capture time): no zero-fill and no accumulating add per parameter - 110 launches of the ~430
        self.optimizer.zero_grad(set_to_none=getattr(self.optimizer, "fresh_gradients_ok", lambda: True)())
        out = self.net(self.x, self.ei, self.ea, self.batch, n_valid=self.n_valid, n_graphs=self.n_graphs)
        loss = fused_l1_loss(out[:self.n_graphs].squeeze(-1), self.y)
        loss.backward()
        return loss.detach()

    def _step(self):
        loss = self.forward_backward()
        self.optimizer.step()
        return loss

    def step_eager(self):
        """The padded step without the graph (what the captured graph replays)."""
        return self._step()

    def _capture(self):
        saved = _WarmupState(self.net, self.optimizer) if self._restore and self._warmup > 0 else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self._step()
            if saved is not None:
                saved.restore()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step()

    def __call__(self, x=None, edge_index=None, edge_attr=None, batch=None, y=None):
        if x is not None:
            self.load(x, edge_index, edge_attr, batch, y)
        if self.graph is None:
            self._capture()          # warm-up steps on the loaded batch, undone again (restore_after_warmup); the capture itself runs nothing
        self.graph.replay()
        return self.loss
