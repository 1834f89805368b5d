"""numpy restatement of the kernels' counter-based dropout keep bits (mma_amd/csrc/common.h: drop_edge_key,
drop_col_key, drop_mix, drop_base_word, drop_mask_word).  TEST INFRASTRUCTURE ONLY - lets the parity tests hand the oracle the exact keep mask
the HIP kernels generate for a (seed, thr).  The reference itself uses torch's global RNG through F.dropout
(layers.py:219), which no other implementation can reproduce bit-for-bit; parity of the dropout path is
therefore tested by feeding both sides the same mask."""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix(a):
    a = a.astype(np.uint64)
    a ^= a >> np.uint64(16); a = (a * np.uint64(0x7FEB352D)) & _M32
    a ^= a >> np.uint64(15); a = (a * np.uint64(0x846CA68B)) & _M32
    a ^= a >> np.uint64(16)
    return a


_MULT = [1, 0x85EBCA6B, 0xC2B2AE35, 0x27D4EB2F, 0x165667B1, 0xCC9E2D51, 0x1B873593, 0xE6546B65]     # common.h drop_mask_mult


def mask_words(seed, K, E, HQ, edge_ids=None):
    """(K,E,HQ) uint64 holding the 32-bit word of every (mask, edge, feature quad): ONE full hash h per (edge, quad), mask 0 uses
    h, mask k >= 1 the folded 64-bit product hi32(h*M_k) ^ lo32(h*M_k) (common.h: drop_base_word, drop_mask_word).  K <= 8:
    the masks of one launch group (more masks are issued as groups of 8, each with its own seed)."""
    assert 1 <= K <= 8
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    e = np.arange(E, dtype=np.uint64) if edge_ids is None else np.asarray(edge_ids).astype(np.uint64)
    ek = ((e * np.uint64(0x9E3779B1)) + seed_lo) & _M32                                   # (E,)
    q = np.arange(HQ, dtype=np.uint64)
    ck = ((q * np.uint64(0x85EBCA77)) + seed_hi) & _M32                                   # (HQ,)
    h = _mix(ek[:, None] ^ ck[None, :])                                                   # (E,HQ)
    out = [h]
    for k in range(1, K):
        t = h * np.uint64(_MULT[k])                                                       # < 2^64: exact in uint64
        out.append((t & _M32) ^ (t >> np.uint64(32)))
    return np.stack(out)


def keep_mask(seed, thr, K, E, H, edge_ids=None):
    """(K,E,H) uint8 in {0,1}: element kept iff its hash byte >= thr.  scale for survivors: 256/(256-thr).
    edge_ids: the (global) edge positions to generate for (default arange(E))."""
    HQ = (H + 3) // 4
    r = mask_words(seed, K, E, HQ, edge_ids)                                              # (K,E,HQ)
    h = np.arange(H)
    byte = (r[:, :, h >> 2] >> (np.uint64(8) * (h & 3).astype(np.uint64))) & np.uint64(0xFF)
    return np.ascontiguousarray((byte >= np.uint64(thr)).astype(np.uint8))


def splitmix64(state):
    """One step of the seed stream behind a graph-capturable layer (mma_seed_advance, csrc/train_step.hip): (new state, seed)."""
    mask = (1 << 64) - 1
    state = (state + 0x9E3779B97F4A7C15) & mask
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
    return state, z ^ (z >> 31)
