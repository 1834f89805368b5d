"""Model glue for the node-classification path (reference node_classification/models.py:10-68): GCN layer -> ReLU ->
dropout -> MMA layer -> log_softmax, with all 25 Parameters owned by the model exactly like the reference - minus its
hard-coded `torch.cuda.FloatTensor` / 'cuda:2' device pins.  (a caller of the hot path, provided so that
train.py-style scripts run end to end on the HIP kernels.)"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import _MASK_NAMES, GraphConvolution, MMA


class MMAConv(nn.Module):
    def __init__(self, add_all, activation, k, nfeat, nhid, nclass, dropout, aggregator_list, device):
        super().__init__()
        self.device = device
        new = lambda *shape: nn.Parameter(torch.empty(*shape, device=device))
        self.weight0, self.bias0 = new(nfeat, nhid), new(nhid)
        self.weight1, self.bias1 = new(nhid, nclass), new(nclass)
        for name in _MASK_NAMES:                      # weight_moment_3, weight_sum, ... (models.py:23-43)
            setattr(self, "weight_" + name, new(2 * nhid, nhid))
        self.add_all = add_all
        self.gc1 = GraphConvolution(nfeat, nhid, self.weight0, self.bias0, device)
        self.gc2 = MMA(self.add_all, activation, k, nhid, nclass, self.weight1, self.bias1,
                       *[getattr(self, "weight_" + name) for name in _MASK_NAMES], dropout, aggregator_list, device)
        self.dropout = dropout
        self.hidden_keep = None      # tests: an explicit (N, nhid) keep mask replayed instead of F.dropout's RNG (models.py:66)

    def logits(self, x, adj):
        x = F.relu(self.gc1(x, adj))
        if self.hidden_keep is not None and self.training:
            x = x * self.hidden_keep / (1.0 - self.dropout)
        else:
            x = F.dropout(x, self.dropout, training=self.training)
        return self.gc2(x, adj)

    def forward(self, x, adj):
        return F.log_softmax(self.logits(x, adj), dim=1)

    def nll_loss(self, x, adj, idx, labels):
        """(loss, output): `F.nll_loss(model(x, adj)[idx], labels[idx])` and the model output of train.py:76-77, with
        log_softmax + nll_loss (and their backward) as the fused K10 kernels."""
        from .train_step import fused_nll_loss
        return fused_nll_loss(self.logits(x, adj), idx, labels)
