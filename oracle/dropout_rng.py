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


_MULT2 = [0x9E3779B9, 0xB5297A4D, 0x68E31DA5, 0x1B56C4E9, 0xD6E8FEB9, 0xA3D95FA9, 0x7F4A7C15, 0x94D049BB]    # common.h drop_mask_mult2


def low_words(seed, K, E, HQ, edge_ids=None):
    """(K,E,HQ) uint64: the SECOND word of every (mask, edge, quad) - fold(h * M2_k), every mask (common.h: drop_low_word).  Its bytes
    are the low halves of the 16-bit values a threshold that is no multiple of 256 is compared with."""
    h = mask_words(seed, 1, E, HQ, edge_ids)[0]
    out = []
    for k in range(K):
        t = h * np.uint64(_MULT2[k])
        out.append((t & _M32) ^ (t >> np.uint64(32)))
    return np.stack(out)


def keep_mask16(seed, thr16, K, E, H, edge_ids=None):
    """(K,E,H) uint8 in {0,1}: element kept iff its 16-bit value (byte of the mask word << 8 | the same byte of the second word) >= thr16:
    P(drop) = thr16 / 65536, scale for survivors 65536 / (65536 - thr16) (common.h: drop_unpack / drop_unpack16; for thr16 = 256 t the
    low byte never decides and the kernels compare the high byte with t).  edge_ids: the (global) edge positions (default arange(E))."""
    assert 0 <= int(thr16) < 65536
    HQ = (H + 3) // 4
    r = mask_words(seed, K, E, HQ, edge_ids)                                              # (K,E,HQ)
    lo = low_words(seed, K, E, HQ, edge_ids)
    h = np.arange(H)
    sh = np.uint64(8) * (h & 3).astype(np.uint64)
    v = (((r[:, :, h >> 2] >> sh) & np.uint64(0xFF)) << np.uint64(8)) | ((lo[:, :, h >> 2] >> sh) & np.uint64(0xFF))
    return np.ascontiguousarray((v >= np.uint64(thr16)).astype(np.uint8))


def keep_mask(seed, thr, K, E, H, edge_ids=None):
    """keep_mask16 for a threshold in units of 1/256 (the README's p = 0.5, 0.75): element kept iff its hash byte >= thr."""
    return keep_mask16(seed, int(thr) * 256, K, E, H, edge_ids)


def threshold16(p):
    """The 16-bit threshold mma_amd.functional.DropoutSpec hands the kernels for probability p (|thr16 / 65536 - p| <= 2^-17)."""
    return min(65535, max(1, int(round(float(p) * 65536)))) if p > 0 else 0


def splitmix64(state):
    """One step of the seed stream behind a graph-capturable layer (mma_seed_advance, csrc/train_step.hip): (new state, seed)."""
    mask = (1 << 64) - 1
    state = (state + 0x9E3779B97F4A7C15) & mask
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
    return state, z ^ (z >> 31)
