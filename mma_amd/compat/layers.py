"""Flat-import shim: the reference's scripts do `from layers import GraphConvolution, MMA` (models.py:4)."""
from mma_amd.layers import MMA, GraphConvolution  # noqa: F401
