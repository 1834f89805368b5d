"""numpy restatement of the kernels' counter-based dropout keep bits (mma_amd/csrc/common.h: drop_edge_key,
drop_col_key, drop_mix).  TEST INFRASTRUCTURE ONLY - lets the parity tests hand the oracle the exact keep mask
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


def keep_mask(seed, thr, K, E, H, edge_ids=None):
    """(K,E,H) uint8 in {0,1}: element kept iff its hash byte >= thr.  scale for survivors: 256/(256-thr).
    edge_ids: the (global) edge positions to generate for (default arange(E))."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    HQ = (H + 3) // 4
    e = np.arange(E, dtype=np.uint64) if edge_ids is None else np.asarray(edge_ids).astype(np.uint64)
    ek = ((e * np.uint64(0x9E3779B1)) + seed_lo) & _M32                                   # (E,)
    kq = np.arange(K * HQ, dtype=np.uint64)
    ck = ((kq * np.uint64(0x85EBCA77)) + seed_hi) & _M32                                  # (K*HQ,)
    r = _mix(ek[None, :, None] ^ ck.reshape(K, 1, HQ))                                    # (K,E,HQ)
    h = np.arange(H)
    byte = (r[:, :, h >> 2] >> (np.uint64(8) * (h & 3).astype(np.uint64))) & np.uint64(0xFF)
    return np.ascontiguousarray((byte >= np.uint64(thr)).astype(np.uint8))
