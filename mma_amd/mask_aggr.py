"""Drop-in `MaskAggregateLinear` (reference graph_regression/mask_aggr.py:7-68).

Same constructor and behaviour, quirks included: it subclasses `Linear` but never uses the inherited weight (G12);
the K per-aggregation Linears live in a plain dict, so they are NOT registered parameters (absent from
state_dict/parameters(), never trained, not moved by .to()) and are moved to the GPU eagerly (G2);
an unknown aggregation raises ValueError (mask_aggr.py:64); mask == "no_linear" returns the input."""
from typing import List, Optional

import torch

from .pyg_compat import Linear


class MaskAggregateLinear(Linear):
    def __init__(self, in_channels: int, out_channels: int, aggregation_list: List[str], aggregation: str,
                 mask: bool = True, bias: bool = True, weight_initializer: Optional[str] = None,
                 bias_initializer: Optional[str] = None):
        super().__init__(in_channels, out_channels, bias, weight_initializer, bias_initializer)
        self.device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.mask = mask
        self.aggregation = aggregation
        self.aggregation_layers = {}
        for i, aggr in enumerate(aggregation_list):
            aggregation_name = "{}".format(aggr)
            if self.mask == "no_linear":
                self.aggregation_layers[aggregation_name] = None
            else:
                linear = Linear(in_channels, out_channels, bias, weight_initializer, bias_initializer).to(self.device)
                self.aggregation_layers[aggregation_name] = linear

    def active_linear(self):
        """The Linear that forward() applies (None for mask == "no_linear")."""
        if self.aggregation not in self.aggregation_layers:
            raise ValueError("Invalid aggregation type: {}".format(self.aggregation))
        return self.aggregation_layers[self.aggregation]

    def forward(self, input):
        lin = self.active_linear()
        if self.mask == "no_linear":
            return input
        return lin(input).to(self.device)
