"""Portable seeded inputs shared by tests/golden/gen_golden.py (the generator, build container only)
and the tests (anywhere).  Data only: no reference code involved."""
import hashlib

import numpy as np

ALL_MASK_NAMES = ["moment_3", "sum", "sum2", "sum3", "sum4", "mean", "mean2", "mean3", "mean4",
                  "max", "max2", "max3", "max4", "min", "min2", "min3", "min4",
                  "softmax", "softmin", "std", "normalized_mean"]  # ctor order, reference layers.py:57-61
WORKING = ["sum", "sum2", "sum3", "sum4", "mean", "mean2", "mean3", "mean4",
           "max", "max2", "max3", "max4", "min", "min2", "min3", "min4", "softmax", "softmin"]


def rng_normal(seed, shape):
    return np.random.default_rng(seed).standard_normal(size=shape, dtype=np.float32)


def rng_uniform(seed, shape, bound):
    return ((np.random.default_rng(seed).random(size=shape, dtype=np.float32) * 2 - 1) * bound).astype(np.float32)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_inputs(seed, N, H, C):
    x = np.maximum(rng_normal(seed, (N, H)), 0).astype(np.float32)  # stands in for relu(gc1), models.py:65
    masks = {n: rng_uniform(seed + 1000 + i, (2 * H, H), 1.0 / np.sqrt(H)) for i, n in enumerate(ALL_MASK_NAMES)}
    weight = rng_uniform(seed + 2000, (H, C), 1.0 / np.sqrt(H))
    bias = rng_uniform(seed + 2001, (C,), 1.0 / np.sqrt(H))
    cot = rng_normal(seed + 3000, (N, C))
    return x, masks, weight, bias, cot


def keep_mask(seed, agg_name, E, H, p):
    """Explicit Bernoulli(1-p) keep mask for one aggregator, rows in CSR edge order."""
    ai = WORKING.index(agg_name)
    return (np.random.default_rng(seed + 5000 + ai).random((E, H), dtype=np.float32) >= p).astype(np.float32)
