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
import torch

from ._lib import require_gpu


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
