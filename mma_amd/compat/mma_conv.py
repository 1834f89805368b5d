"""Flat-import shim for `from mma_conv import MMAConv` (reference graph_regression/mma.py)."""
from mma_amd.mma_conv import MMAConv  # noqa: F401
