// Device/host helpers shared by the libmma_amd.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mma_amd.h"

namespace mma {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup, one item stream per wave
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 workgroups: grid-stride beyond this

// ---- error reporting (thread-local, read back through mma_last_error) -----------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

#define MMA_REQUIRE(cond, ...) \
  do { if (!(cond)) return ::mma::fail(1, __VA_ARGS__); } while (0)

// ---- small fixed-width vectors that live in registers ----------------------------------------------
template <int VEC> struct Vec { float v[VEC]; };

template <int VEC> __device__ __forceinline__ Vec<VEC> vzero() {
  Vec<VEC> r;
#pragma unroll
  for (int i = 0; i < VEC; ++i) r.v[i] = 0.f;
  return r;
}
template <int VEC> __device__ __forceinline__ Vec<VEC> ldv(const float* p);
template <> __device__ __forceinline__ Vec<4> ldv<4>(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  Vec<4> r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
}
template <> __device__ __forceinline__ Vec<1> ldv<1>(const float* p) { Vec<1> r; r.v[0] = *p; return r; }

template <int VEC> __device__ __forceinline__ void stv(float* p, const Vec<VEC>& a);
template <> __device__ __forceinline__ void stv<4>(float* p, const Vec<4>& a) {
  *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
template <> __device__ __forceinline__ void stv<1>(float* p, const Vec<1>& a) { *p = a.v[0]; }

// streaming forms: data written once and not read again by this kernel / read exactly once (the `nt` policy keeps them from
// displacing the gathered rows in L2)
typedef float mma_f32x4 __attribute__((ext_vector_type(4)));
template <int VEC> __device__ __forceinline__ void stv_nt(float* p, const Vec<VEC>& a);
template <> __device__ __forceinline__ void stv_nt<4>(float* p, const Vec<4>& a) {
  mma_f32x4 v = {a.v[0], a.v[1], a.v[2], a.v[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<mma_f32x4*>(p));
}
template <> __device__ __forceinline__ void stv_nt<1>(float* p, const Vec<1>& a) { __builtin_nontemporal_store(a.v[0], p); }
template <int VEC> __device__ __forceinline__ Vec<VEC> ldv_nt(const float* p);
template <> __device__ __forceinline__ Vec<4> ldv_nt<4>(const float* p) {
  const mma_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const mma_f32x4*>(p));
  Vec<4> r; r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3]; return r;
}
template <> __device__ __forceinline__ Vec<1> ldv_nt<1>(const float* p) { Vec<1> r; r.v[0] = __builtin_nontemporal_load(p); return r; }

// 1 byte per element (selection codes, explicit keep masks)
template <int VEC> __device__ __forceinline__ uint32_t ldb(const uint8_t* p);
template <> __device__ __forceinline__ uint32_t ldb<4>(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
template <> __device__ __forceinline__ uint32_t ldb<1>(const uint8_t* p) { return *p; }
template <int VEC> __device__ __forceinline__ void stb(uint8_t* p, uint32_t codes);
template <> __device__ __forceinline__ void stb<4>(uint8_t* p, uint32_t c) { *reinterpret_cast<uint32_t*>(p) = c; }
template <int VEC> __device__ __forceinline__ void stb_nt(uint8_t* p, uint32_t codes);
template <> __device__ __forceinline__ void stb_nt<4>(uint8_t* p, uint32_t c) { __builtin_nontemporal_store(c, reinterpret_cast<uint32_t*>(p)); }
template <> __device__ __forceinline__ void stb_nt<1>(uint8_t* p, uint32_t c) { __builtin_nontemporal_store((uint8_t)c, p); }
template <int VEC> __device__ __forceinline__ uint32_t ldb_nt(const uint8_t* p);
template <> __device__ __forceinline__ uint32_t ldb_nt<4>(const uint8_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p)); }
template <> __device__ __forceinline__ uint32_t ldb_nt<1>(const uint8_t* p) { return __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ void stb<1>(uint8_t* p, uint32_t c) { *p = (uint8_t)c; }

// ---- mask activation ---------------------------------------------------------------------------------
// sigmoid on the hardware transcendental pipe: v_exp_f32 + v_rcp_f32 (1 ulp each)
__device__ __forceinline__ float sigmoid_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

// ---- dropout keep bits: counter-based, stateless, identical in forward and backward -------------------
// ONE full 32-bit hash per (edge position e, feature quad q) - h = mix32(edge_key(e) ^ col_key(q)) - serves all K masks of
// a launch group (round 3; round 2 ran one full hash per mask): mask 0 uses h itself, mask k >= 1 the folded 64-bit product
// r_k = hi32(h * M_k) ^ lo32(h * M_k) with a fixed odd multiplier per mask (one v_mad_u64_u32 + one xor instead of two
// v_mul_lo_u32 and seven ALU operations).  A word yields 4 bytes, one per feature of the quad.  The drop probability is thr16/65536
// (round 5: any p of F.dropout(mask0, p), layers.py:219, to 2^-17): an element is KEPT iff its 16-bit value v >= thr16, v = byte of the
// mask word (high half) | the same byte of a SECOND word c_k = fold(h * M2_k) (low half); survivors scale by 65536/(65536-thr16).  When
// thr16 is a multiple of 256 (the README's 0.5 and 0.75) the low half never decides - v >= 256 t  <=>  byte >= t - and the kernels run the
// one-word form (internal mode HASH, thr = thr16 >> 8): the same bits at round 4's cost; otherwise mode HASH16 (+1 multiply-fold and two
// v_perm_b32 per mask and quad).  The numpy restatement the parity tests use is
// oracle/dropout_rng.py; tests/test_dropout_rng.py checks keep rate and the independence of masks / bytes / neighbouring
// edges and quads (a cheaper derivation, r_k = fold16(h * M_k), failed exactly that test: the XOR of two bytes of one mask
// was 800 sigma correlated with the same XOR of another mask).
__host__ __device__ __forceinline__ uint32_t drop_edge_key(uint32_t e, uint32_t seed_lo) { return e * 0x9E3779B1u + seed_lo; }
__host__ __device__ __forceinline__ uint32_t drop_col_key(uint32_t q, uint32_t seed_hi) { return q * 0x85EBCA77u + seed_hi; }
__host__ __device__ __forceinline__ uint32_t drop_mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7FEB352Du; a ^= a >> 15; a *= 0x846CA68Bu; a ^= a >> 16; return a;
}
// multiplier of mask k (k = 0: unused, the base word itself)
__host__ __device__ __forceinline__ uint32_t drop_mask_mult(int k) {
  constexpr uint32_t m[8] = {1u, 0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu, 0x165667B1u, 0xCC9E2D51u, 0x1B873593u, 0xE6546B65u};
  return m[k & 7];
}
__host__ __device__ __forceinline__ uint32_t drop_mask_word(uint32_t h, int k_abs, uint32_t mult) {
  if (k_abs == 0) return h;
  const uint64_t t = (uint64_t)h * (uint64_t)mult;
  return (uint32_t)t ^ (uint32_t)(t >> 32);
}
// HASH16: multiplier of mask k's SECOND word (low halves of the 16-bit values); every mask folds, mask 0 too
__host__ __device__ __forceinline__ uint32_t drop_mask_mult2(int k) {
  constexpr uint32_t m[8] = {0x9E3779B9u, 0xB5297A4Du, 0x68E31DA5u, 0x1B56C4E9u, 0xD6E8FEB9u, 0xA3D95FA9u, 0x7F4A7C15u, 0x94D049BBu};
  return m[k & 7];
}
__host__ __device__ __forceinline__ uint32_t drop_low_word(uint32_t h, uint32_t mult2) {
  const uint64_t t = (uint64_t)h * (uint64_t)mult2;
  return (uint32_t)t ^ (uint32_t)(t >> 32);
}
// internal kernel mode beside the MMA_DROP_* of the header: HASH with a threshold that is no multiple of 256
#define MMA_DROP_HASH16 3

struct DropParams {
  int mode;             // MMA_DROP_* or MMA_DROP_HASH16 (set by drop_make from the caller's HASH + thr16)
  uint32_t thr;         // HASH: 0..255 (= thr16 >> 8, thr16 a multiple of 256); HASH16: 1..65535
  float scale;          // 65536/(65536-thr16) (HASH, HASH16) or 1/(1-p) = the same formula (EXPLICIT)
  uint32_t seed_lo, seed_hi;
  const uint64_t* seed_dev;   // optional: the seed lives in device memory (graph replays draw a fresh one without re-capture)
  const uint8_t* keep;  // EXPLICIT: (K_total, E, H)
  int64_t E;
  uint32_t edge_base;   // HASH: key = edge position + edge_base (a shard's edges keep their global ids)
};

// kernel entry: a seed in device memory overrides the one passed by value (one uniform 8-byte load per wave)
__device__ __forceinline__ DropParams drop_resolve(DropParams d) {
  if ((d.mode == MMA_DROP_HASH || d.mode == MMA_DROP_HASH16) && d.seed_dev != nullptr) {
    const uint64_t s = *d.seed_dev;
    d.seed_lo = (uint32_t)s; d.seed_hi = (uint32_t)(s >> 32);
  }
  return d;
}

// the base word of (edge e, quad q): shared by every mask of the launch group
__device__ __forceinline__ uint32_t drop_base_word(const DropParams& d, uint32_t e, int q) {
  return drop_mix(drop_edge_key(e + d.edge_base, d.seed_lo) ^ drop_col_key((uint32_t)q, d.seed_hi));
}
template <int VEC>
__device__ __forceinline__ void drop_unpack(const DropParams& d, uint32_t r, int c, float (&f)[VEC]) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const uint32_t byte = (r >> (8 * ((c + i) & 3))) & 0xFFu;
    f[i] = byte >= d.thr ? d.scale : 0.f;
  }
}
// HASH16: r = the mask word (high bytes), lo = its second word (low bytes); element i of the quad takes byte (c+i)&3 of both
template <int VEC>
__device__ __forceinline__ void drop_unpack16(const DropParams& d, uint32_t r, uint32_t lo, int c, float (&f)[VEC]) {
  if constexpr (VEC == 4) {          // c % 4 == 0: two v_perm_b32 interleave the bytes into four 16-bit values
    const uint32_t v01 = __builtin_amdgcn_perm(r, lo, 0x05010400u);      // [r1 lo1 | r0 lo0]
    const uint32_t v23 = __builtin_amdgcn_perm(r, lo, 0x07030602u);      // [r3 lo3 | r2 lo2]
    f[0] = (v01 & 0xFFFFu) >= d.thr ? d.scale : 0.f;
    f[1] = (v01 >> 16) >= d.thr ? d.scale : 0.f;
    f[2] = (v23 & 0xFFFFu) >= d.thr ? d.scale : 0.f;
    f[3] = (v23 >> 16) >= d.thr ? d.scale : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int sh = 8 * ((c + i) & 3);
      const uint32_t v = (((r >> sh) & 0xFFu) << 8) | ((lo >> sh) & 0xFFu);
      f[i] = v >= d.thr ? d.scale : 0.f;
    }
  }
}
template <int VEC>
__device__ __forceinline__ void drop_explicit(const DropParams& d, uint32_t e, int k_abs, int c, int H, float (&f)[VEC]) {
  const uint32_t bits = ldb<VEC>(d.keep + ((size_t)k_abs * (size_t)d.E + e) * (size_t)H + c);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = ((bits >> (8 * i)) & 0xFFu) ? d.scale : 0.f;
}

// keep multiplier (0 or scale) for the VEC features starting at column c of mask k_abs on edge e (run-time mode; the fused NC
// kernels use the pieces above with the mode as a template parameter and ONE base word for all their masks)
template <int VEC>
__device__ __forceinline__ void drop_factors(const DropParams& d, uint32_t e, int k_abs, int c, int H, int HQ,
                                             float (&f)[VEC]) {
  if (d.mode == MMA_DROP_HASH) {
    drop_unpack<VEC>(d, drop_mask_word(drop_base_word(d, e, c >> 2), k_abs, drop_mask_mult(k_abs)), c, f);
  } else if (d.mode == MMA_DROP_HASH16) {
    const uint32_t h = drop_base_word(d, e, c >> 2);
    drop_unpack16<VEC>(d, drop_mask_word(h, k_abs, drop_mask_mult(k_abs)), drop_low_word(h, drop_mask_mult2(k_abs)), c, f);
  } else {  // EXPLICIT
    drop_explicit<VEC>(d, e, k_abs, c, H, f);
  }
}

// the caller's (mode, thr16) -> the kernels' (mode, thr, scale): HASH with thr16 % 256 == 0 keeps the one-word form
__host__ inline void drop_set_threshold(DropParams* d, int mode, uint32_t thr16) {
  d->scale = 65536.0f / (65536.0f - (float)thr16);
  if (mode == MMA_DROP_HASH && (thr16 & 0xFFu)) { d->mode = MMA_DROP_HASH16; d->thr = thr16; }
  else { d->mode = mode; d->thr = mode == MMA_DROP_HASH ? (thr16 >> 8) : thr16; }
}

// max over the lanes of a FULLY ACTIVE wavefront of a non-negative value, in every lane: four DPP steps inside the 16-lane row (two quad
// permutes, two mirrors), one swizzle across the two rows of a half, one bpermute across the halves.  ONLY with all 64 lanes active: a
// mirror step hands a lane nothing when its partner is inactive, and the maximum then never reaches the lanes below (found in round 5:
// with lanes 60..63 off, a maximum held by lanes 56..59 did not reach lane 0 - callers test `__ballot(1) == ~0ull` and fall back to
// per-lane merges)
__device__ __forceinline__ float wave_max_nonneg(float m) {
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0xB1, 0xF, 0xF, true)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x4E, 0xF, 0xF, true)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x141, 0xF, 0xF, true)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x140, 0xF, 0xF, true)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(m), 0x401F)));
  return fmaxf(m, __shfl_xor(m, 32, 64));
}

// the upper 16 bits of a non-negative fp32 pattern, rounded UP: as an unsigned 16-bit number it orders like the value, and (v << 16) read as
// fp32 is >= the value and within 2^-7 of it (inf and NaN patterns stay what they are)
__device__ __forceinline__ uint32_t up16_nonneg(const float m) {
  const uint32_t b = __float_as_uint(m);
  return b >= 0x7F800000u ? b >> 16 : (b + 0xFFFFu) >> 16;
}
// lane-wise maximum of two packed unsigned 16-bit fields over a FULLY ACTIVE wavefront (see wave_max_nonneg), result in every lane
__device__ __forceinline__ uint32_t wave_pkmax_u16(uint32_t v) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  auto pk = [](uint32_t a, uint32_t b) {
    const us2 r = __builtin_elementwise_max(*reinterpret_cast<const us2*>(&a), *reinterpret_cast<const us2*>(&b));
    return *reinterpret_cast<const uint32_t*>(&r);
  };
  v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));
  v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));
  v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true));
  v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true));
  v = pk(v, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));
  return pk(v, (uint32_t)__shfl_xor((int)v, 32, 64));
}

__host__ inline int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace mma
