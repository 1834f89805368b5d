"""Flat-import shim for `from scalers import SCALERS` (reference layers.py:10)."""
from mma_amd.scalers import *  # noqa: F401,F403
from mma_amd.scalers import SCALERS  # noqa: F401
