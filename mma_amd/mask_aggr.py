"""Drop-in `MaskAggregateLinear` (behaviour of the reference's graph_regression/mask_aggr.py:7-68).

Quirks kept on purpose: the class derives from `Linear` but its own weight is never applied (G12);
the per-aggregation Linears sit in a plain dict, i.e. they are NOT registered (missing from parameters()/state_dict(),
never trained, ignored by .to()) and are therefore placed on the GPU at construction (G2); an aggregation name that is not
in the list raises ValueError at call time (mask_aggr.py:64); mask == "no_linear" makes the module the identity."""
from typing import List, Optional

import torch

from .pyg_compat import Linear

_NO_LINEAR = "no_linear"


class MaskAggregateLinear(Linear):
    def __init__(self, in_channels: int, out_channels: int, aggregation_list: List[str], aggregation: str,
                 mask: bool = True, bias: bool = True, weight_initializer: Optional[str] = None,
                 bias_initializer: Optional[str] = None):
        super().__init__(in_channels, out_channels, bias, weight_initializer, bias_initializer)
        self.device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.mask, self.aggregation = mask, aggregation

        def one_linear():
            if mask == _NO_LINEAR:
                return None
            return Linear(in_channels, out_channels, bias, weight_initializer, bias_initializer).to(self.device)

        # a dict on purpose: keeps the K Linears out of the module's registered parameters, as in the reference
        self.aggregation_layers = {str(name): one_linear() for name in aggregation_list}

    def active_linear(self):
        """The Linear that forward() applies (None for mask == "no_linear")."""
        try:
            return self.aggregation_layers[self.aggregation]
        except KeyError:
            raise ValueError("Invalid aggregation type: {}".format(self.aggregation)) from None

    def forward(self, input):
        layer = self.active_linear()
        return input if self.mask == _NO_LINEAR else layer(input).to(self.device)
