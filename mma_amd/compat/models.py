"""Flat-import shim for `from models import MMAConv` (reference node_classification/train.py:13)."""
from mma_amd.models import MMAConv  # noqa: F401
