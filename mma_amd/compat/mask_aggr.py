"""Flat-import shim for `from mask_aggr import MaskAggregateLinear` (reference mma_conv.py:13)."""
from mma_amd.mask_aggr import MaskAggregateLinear  # noqa: F401
