"""mma_amd - MI355X (gfx950) kernels for the Multi-Mask-Aggregator message-passing hot path,
behind the reference's own module surfaces (asarigun/mma: node_classification/layers.py `MMA`,
graph_regression/mma_conv.py `MMAConv`, graph_regression/mask_aggr.py `MaskAggregateLinear`).

The compute path is libmma_amd.so (hand-written HIP, C ABI in include/mma_amd.h).  There is no CPU
fallback: importing works anywhere, but every op raises if the library or a GPU is missing."""
from . import _lib  # noqa: F401
from .graph import NCGraph  # noqa: F401
from .functional import nc_fused_aggregate, csr_spmm  # noqa: F401
from .layers import MMA, GraphConvolution  # noqa: F401
from .mask_aggr import MaskAggregateLinear  # noqa: F401
from .mma_conv import CategoricalEdges, MMAConv  # noqa: F401
from .train_step import FusedAdam, GraphedNetStep, GraphedTrainStep, fused_l1_loss, fused_nll_loss  # noqa: F401

__all__ = ["MMA", "GraphConvolution", "MMAConv", "CategoricalEdges", "MaskAggregateLinear", "NCGraph", "nc_fused_aggregate", "csr_spmm",
           "GraphedTrainStep", "GraphedNetStep", "FusedAdam", "fused_nll_loss", "fused_l1_loss"]
