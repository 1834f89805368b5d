"""The few torch_geometric pieces the reference's graph-regression path leans on, restated so the drop-in
modules need no torch_geometric install: `Linear` (torch_geometric.nn.dense.linear.Linear: y = x W^T + b, weight
(out,in), kaiming-uniform(a=sqrt 5) / uniform(1/sqrt(in)) default init), `reset`, `degree`."""
import math

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn.parameter import Parameter


class Linear(torch.nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None, bias_initializer=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer, self.bias_initializer = weight_initializer, bias_initializer
        self.weight = Parameter(torch.empty(out_channels, in_channels))
        self.bias = Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == "glorot":
            a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
            self.weight.data.uniform_(-a, a)
        elif self.weight_initializer in (None, "kaiming_uniform"):
            bound = 1.0 / math.sqrt(self.in_channels) if self.in_channels > 0 else 0.0   # kaiming_uniform(fan_in, a=sqrt(5))
            self.weight.data.uniform_(-bound, bound)
        else:
            raise RuntimeError("Linear layer weight initializer '%s' is not supported" % self.weight_initializer)
        if self.bias is not None:
            if self.bias_initializer == "zeros":
                self.bias.data.zero_()
            elif self.bias_initializer is None:
                bound = 1.0 / math.sqrt(self.in_channels) if self.in_channels > 0 else 0.0
                self.bias.data.uniform_(-bound, bound)
            else:
                raise RuntimeError("Linear layer bias initializer '%s' is not supported" % self.bias_initializer)

    def forward(self, x: Tensor) -> Tensor:
        if x.is_cuda:                       # same GEMM, backward through the split-reduction dW and the K8 column sum
            from .dense import linear
            return linear(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)

    def __repr__(self):
        return "%s(%d, %d, bias=%s)" % (self.__class__.__name__, self.in_channels, self.out_channels, self.bias is not None)


def reset(value):
    """torch_geometric.nn.inits.reset: recurse into children, call reset_parameters where present."""
    if hasattr(value, "reset_parameters"):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, "children") else []:
            reset(child)


def degree(index, num_nodes=None, dtype=None):
    """torch_geometric.utils.degree (= scatter_add of ones); plumbing used by callers to build the histogram."""
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((n,), dtype=dtype or torch.get_default_dtype(), device=index.device)
    return out.scatter_add_(0, index, torch.ones((index.numel(),), dtype=out.dtype, device=index.device))
