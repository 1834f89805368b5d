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


class _FusedNLL(torch.autograd.Function):
    """K10: log_softmax over the classes + mean nll over the rows `idx` (models.py:68, train.py:77)."""

    @staticmethod
    def forward(ctx, logits, idx, labels):
        require_gpu(logits, idx, labels)
        logits = logits.contiguous()
        N, C = logits.shape
        assert idx.dtype == torch.int64 and labels.dtype == torch.int64 and labels.numel() == N
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
    whose .grad is None at the first step are left out for good (the reference's unused masks never get a gradient)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))
        self._tables = None

    def _build(self):
        self._tables = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            require_gpu(*ps)
            recs, chunk_ids, chunk0 = b"", [], 0
            for t, p in enumerate(ps):
                assert p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                n_chunks = _lib.query("mma_adam_chunks", p.numel())
                recs += struct.pack("<QQQQqq", p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                    p.numel(), chunk0)
                chunk_ids += [t] * n_chunks
                chunk0 += n_chunks
            dev = ps[0].device if ps else "cpu"
            raw = recs + struct.pack("<%di" % len(chunk_ids), *chunk_ids)
            assert len(raw) == _lib.query("mma_adam_table_bytes", len(ps), chunk0)
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev) if ps else None
            step = torch.zeros((), device=dev, dtype=torch.float32)
            self._tables.append((group, ps, table, chunk0, step, [(p.data_ptr(), p.grad.data_ptr()) for p in ps]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._tables is None:
            self._build()
        for group, ps, table, n_chunks, step, ptrs in self._tables:
            if not ps:
                continue
            if any((p.data_ptr(), p.grad.data_ptr()) != q for p, q in zip(ps, ptrs)):
                raise RuntimeError("FusedAdam: a parameter or gradient buffer moved; keep gradients allocated (zero_grad(set_to_none=False))")
            b1, b2 = group["betas"]
            call("mma_adam_step", ptr(table), len(ps), n_chunks, ptr(step), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                 stream_ptr())
        return loss

    def zero_grad(self, set_to_none=False):
        super().zero_grad(set_to_none=False if self._tables is not None else set_to_none)


def _set_capturable(model, flag):
    for m in model.modules():
        if hasattr(m, "graph_capturable"):
            m.graph_capturable = flag


class GraphedTrainStep:
    """Captures `optimizer.zero_grad(); loss = loss_fn(); loss.backward(); optimizer.step()` once and replays it.

    loss_fn: closure over STATIC device tensors (full-batch features / adjacency / labels / index sets).  The optimizer must
    be graph-capturable (`torch.optim.Adam(..., capturable=True)`); gradients are kept allocated (`set_to_none=False`)
    so their addresses stay valid across replays."""

    def __init__(self, model, optimizer, loss_fn, warmup=3):
        params = [p for g in optimizer.param_groups for p in g["params"]]
        require_gpu(*params)
        for g in optimizer.param_groups:
            if "capturable" in g and not g["capturable"]:
                raise ValueError("GraphedTrainStep needs a capturable optimizer, e.g. torch.optim.Adam(..., capturable=True)")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        _set_capturable(model, True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream, as capture requires
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=False)
        loss = self.loss_fn()
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self):
        self.graph.replay()
        return self.loss
