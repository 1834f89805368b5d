"""Flat-import shim for `from utils import load_data, accuracy` (reference node_classification/train.py:12)."""
from mma_amd.utils import *  # noqa: F401,F403
from mma_amd.utils import load_data, accuracy  # noqa: F401
