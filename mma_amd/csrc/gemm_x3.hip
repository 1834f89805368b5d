// fp32-accurate tall-skinny GEMM on the bf16 matrix cores ("bf16x3"):  C (M,N) = A (M,K) @ B (K,N),  M >> N, K.
//
// The dense pre/post transforms of the hot path (P = x Wtop, Q = x Wbot and dL/dx = g W^T) have a
// million rows and K, N in {128, 512}.  On gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate (157 TFLOP/s) and
// there is no xf32/TF32 path, while bf16 MFMA is 16x faster per instruction.  Each fp32 value is therefore split
// exactly into three bf16 pieces, a = a1 + a2 + a3 (8+8+8 mantissa bits), and the product is summed from the six
// piece products that matter,   a b ~= a1b3 + a2b2 + a3b1 + a1b2 + a2b1 + a1b1   (dropped terms < 2^-24 |ab|),
// every piece product being exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  The result is as accurate as
// an fp32 GEMM (measured max error 2.4e-7 vs 3.0e-7 for the fp32 library GEMM, both relative to sum|a||b| against
// fp64) at 6/16 of its matrix-core time.  Four kernels: gemm_x3_colgroup_kernel (K == 128, whole 128-column groups: the
// forward [P|Q] = x W, 1.43-1.48 ms at C4), gemm_x3_n128_kernel (N == 128, long K: dL/dx, 1.53 ms), gemm_x3_tn_kernel
// (x^T g, 1.39-1.45 ms) and the plain gemm_x3_kernel<NCT> below for every other shape and the ragged tails; rocBLAS fp32
// takes 2.1-2.6 ms for each.  On real data all of them are held by the chip's power management (1.83-1.96 GHz effective
// clock, 1.0-1.1 PFLOP/s of bf16 MFMA; the same binaries on zero operands: 1.12-1.35 ms) - DESIGN.md 3.
// What was measured about gemm_x3_kernel in round 1 (so that the next attempt does not repeat it): s_memtime stamps put
// 26 % of a tile in the issue of the 16 C stores and 23 % in the issue of the 6 slab loads queued behind them (K == 128
// form); turning the tile through LDS into 4 dwordx4 stores changes nothing (the wait is the 4.3 GB themselves - 0.63 ms
// at the 6.8 TB/s a plain fill reaches - not the instruction count; without any store the kernel still takes 1.43 ms); prefetching the next K chunk of A in the
// persistent form changes nothing; -fno-slp-vectorize changes nothing; two independent workgroups per CU beat one 8-wave
// workgroup with explicit multiply / split roles.
//
// Layout: one workgroup = 4 waves = 128 rows of A; a wave keeps its 32 rows x 128 k of A in registers, already split
// (3 x 8 k-steps x 8 bf16 = 96 VGPRs) and walks the 32-column tiles of B; the three bf16 pieces of the B tile
// ([piece][col][k], k contiguous, rows padded by 16 B against LDS bank conflicts) are staged through a double-buffered
// LDS slab shared by the 4 waves, prefetched one tile ahead.  K > 128 is processed in chunks of 128 with the (at most
// four) accumulator tiles kept in registers, so either K == 128 or N <= 128 is required.
#include <algorithm>
#include <cstdlib>
#include "common.h"

// Measurement builds only (tools/build_ablation.sh compiles this file with -DMMA_ABL=<bits> into scratch/ and the micro-benchmark loads
// that library through MMA_LIB_OVERRIDE): bit 1 = no MFMAs, bit 2 = no C stores, bit 3 = no A loads, bit 4 = no split, bit 5 = no B slab
// staging, in the forward column-group kernel and the pipelined dL/dx kernel.  A compile-time constant, 0 in the product library: the
// ablated code does not exist there (run-time switches changed the register allocation of the kernels they were meant to measure).
#ifndef MMA_ABL
#define MMA_ABL 0
#endif

namespace mma {

constexpr int kAbl = MMA_ABL;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kKC = 128;                 // k per chunk when the A chunk is the only thing kept in registers (K == 128 form)
constexpr int kKCPersist = 64;           // k per chunk when up to 4 accumulator tiles persist (K > 128 form): 48 instead of
                                         // 96 A registers, so that two workgroups fit a CU
struct GemmParams {
  const float* A; int64_t lda;
  const __bf16* Bt;          // (3, N, K) bf16: piece p of B^T, k contiguous
  float* C; int64_t ldc;
  int64_t M; int N, K;
  int accumulate;            // C += A B instead of C = A B (the tile's old values are fetched before its MFMAs start)
};

struct Bf3 { __bf16 a, b, c; };
__device__ __forceinline__ Bf3 split3(const float v) {
  Bf3 r;
  r.a = (__bf16)v;
  const float r1 = v - (float)r.a;
  r.b = (__bf16)r1;
  r.c = (__bf16)(r1 - (float)r.b);
  return r;
}

template <int KC> struct SlabGeom {
  static constexpr int kRowPitch = KC * 2 + 16;   // bytes per LDS row: KC bf16 + 16 B pad (conflict-free ds_read_b128)
  static constexpr int kPiece = 32 * kRowPitch;   // one piece of one 32-column tile
  static constexpr int kSlab = 3 * kPiece;
  static constexpr int kCpr = KC / 8;             // 16-byte chunks per row
  static constexpr int kPerThread = 3 * 32 * kCpr / kBlock;   // 6 (KC = 128) or 3 (KC = 64)
};

// One thread moves kPerThread of the 16-byte chunks of a 3 x 32 x KC bf16 slab.  Scalars, not an array: an indexed
// register array that is written and read under run-time conditions stays in scratch memory.
struct SlabRegs { uint4 r0, r1, r2, r3, r4, r5; };
template <int KC>
__device__ __forceinline__ const uint4* slab_src(const GemmParams& p, int q, int kc, int ct) {
  constexpr int cpr = SlabGeom<KC>::kCpr;
  const int piece = q / (32 * cpr), rem = q % (32 * cpr), col = rem / cpr, kq = rem % cpr;
  return reinterpret_cast<const uint4*>(p.Bt + ((size_t)piece * p.N + (size_t)(ct * 32 + col)) * p.K + kc * KC + kq * 8);
}
template <int KC>
__device__ __forceinline__ uint4* slab_dst(unsigned char* dst, int q) {
  constexpr int cpr = SlabGeom<KC>::kCpr;
  const int piece = q / (32 * cpr), rem = q % (32 * cpr), col = rem / cpr, kq = rem % cpr;
  return reinterpret_cast<uint4*>(dst + piece * SlabGeom<KC>::kPiece + col * SlabGeom<KC>::kRowPitch + kq * 16);
}
template <int KC>
__device__ __forceinline__ SlabRegs slab_load(const GemmParams& p, int tid, int kc, int ct) {
  SlabRegs s;
  s.r0 = *slab_src<KC>(p, tid, kc, ct);              s.r1 = *slab_src<KC>(p, tid + kBlock, kc, ct);
  s.r2 = *slab_src<KC>(p, tid + 2 * kBlock, kc, ct);
  if (SlabGeom<KC>::kPerThread > 3) {
    s.r3 = *slab_src<KC>(p, tid + 3 * kBlock, kc, ct);
    s.r4 = *slab_src<KC>(p, tid + 4 * kBlock, kc, ct); s.r5 = *slab_src<KC>(p, tid + 5 * kBlock, kc, ct);
  }
  return s;
}
template <int KC>
__device__ __forceinline__ void slab_store(unsigned char* dst, int tid, const SlabRegs& s) {
  *slab_dst<KC>(dst, tid) = s.r0;              *slab_dst<KC>(dst, tid + kBlock) = s.r1;
  *slab_dst<KC>(dst, tid + 2 * kBlock) = s.r2;
  if (SlabGeom<KC>::kPerThread > 3) {
    *slab_dst<KC>(dst, tid + 3 * kBlock) = s.r3;
    *slab_dst<KC>(dst, tid + 4 * kBlock) = s.r4; *slab_dst<KC>(dst, tid + 5 * kBlock) = s.r5;
  }
}

// NCT > 0: K > 128, the NCT (= N/32 <= 4) accumulator tiles persist across the K chunks (compile-time indices only).
// NCT == 0: K == 128, any number of column tiles, one accumulator tile at a time.
// FULL: every row of the block exists (no guards: the C stores are unconditional, so the compiler can count them in
// vmcnt and never waits for a store); the last partial block is a separate guarded launch.
// ACC: C += A B (a template parameter: as a run-time flag the 16 old values cost the plain K == 128 form 12 spilled VGPRs).
template <int NCT, bool FULL, bool ACC>
__global__ __launch_bounds__(kBlock, 2) void gemm_x3_kernel(const GemmParams p, int64_t blk0, int64_t n_blocks) {
  constexpr bool PERSIST = NCT > 0;
  constexpr int KC = PERSIST ? kKCPersist : kKC;
  constexpr int KSTEPS = KC / 16;
  constexpr int kRowPitch = SlabGeom<KC>::kRowPitch, kPiece = SlabGeom<KC>::kPiece, kSlab = SlabGeom<KC>::kSlab;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kSlab];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r31 = lane & 31, h = lane >> 5;
  const int n_kc = p.K / KC, n_ct = PERSIST ? NCT : p.N / 32;
  const int n_it = n_kc * n_ct;

  for (int64_t blk = blk0 + blockIdx.x; blk < blk0 + n_blocks; blk += gridDim.x) {
    const int64_t row0 = blk * 128 + wave * 32;
    const int64_t arow = FULL ? row0 + r31 : min(row0 + r31, p.M - 1);   // rows past M re-read the last row, never stored
    const float* ap = p.A + arow * p.lda + 8 * h;

    f32x16 acc[PERSIST ? NCT : 1];
#pragma unroll
    for (int t = 0; t < (PERSIST ? NCT : 1); ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    SlabRegs nxt = slab_load<KC>(p, tid, 0, 0);              // slab prefetch registers: 3 or 6 x 16 B per thread
    slab_store<KC>(lds, tid, nxt);
    if (n_it > 1) nxt = slab_load<KC>(p, tid, 1 / n_ct, 1 % n_ct);
    __syncthreads();

    bf16x8 af[KSTEPS][3];
    int it = 0;
// one 32-column tile of one K chunk: prefetch the next slab, 8 k-steps x 6 piece products, (last chunk) the C store,
// publish the prefetched slab.  A macro, not a lambda: register arrays captured by a closure end up in scratch.
// Order inside a tile: MFMAs on slab `it` | publish slab it+1 (loaded one tile ago) | issue the loads of slab it+2 |
// C stores | barrier.  vmcnt is in-order and counts stores: a load issued AFTER a tile's stores could only be waited
// for together with them, so the prefetch is issued BEFORE the stores and its wait (a tile later) leaves them in flight.
#define MMA_X3_TILE(C_, CT_, LAST_)                                                                          \
    {                                                                                                        \
      /* C += A B: the tile's 16 old values per lane are fetched in ONE batch before its last MFMA block (round 1 loaded,  \
         waited, added and stored them one by one after it: 64 serialised HBM round trips per row block of dL/dx) */      \
      float oldv[16];                                                                                        \
      if ((LAST_) && ACC) {                                                                         \
        const float* op = p.C + (size_t)((CT_) * 32 + r31);                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                     \
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;                                         \
          oldv[r] = (FULL || row < p.M) ? op[row * p.ldc] : 0.f;                                             \
        }                                                                                                    \
      }                                                                                                      \
      const unsigned char* sb = lds + (it & 1) * kSlab + r31 * kRowPitch + h * 16;                           \
      _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ++ks) {                                                \
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(sb + 0 * kPiece + ks * 32);                       \
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(sb + 1 * kPiece + ks * 32);                       \
        const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(sb + 2 * kPiece + ks * 32);                       \
        C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], b3, C_, 0, 0, 0);                            \
        C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], b2, C_, 0, 0, 0);                            \
        C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][2], b1, C_, 0, 0, 0);                            \
        C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], b2, C_, 0, 0, 0);                            \
        C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], b1, C_, 0, 0, 0);                            \
        C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], b1, C_, 0, 0, 0);                            \
      }                                                                                                      \
      if (it + 1 < n_it) slab_store<KC>(lds + ((it + 1) & 1) * kSlab, tid, nxt);                             \
      if (it + 2 < n_it) {                                                                                   \
        const int nk = (it + 2) / n_ct;                                                                      \
        nxt = slab_load<KC>(p, tid, nk, it + 2 - nk * n_ct);                                                 \
      }                                                                                                      \
      if (LAST_) { /* acc reg r holds row (r&3) + 8*(r>>2) + 4*h, column r31 of the tile */   \
        float* cp = p.C + (size_t)((CT_) * 32 + r31);                                                        \
        if (ACC) {                                                                                           \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                   \
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;                                       \
            if (FULL || row < p.M) cp[row * p.ldc] = oldv[r] + C_[r];                                        \
          }                                                                                                  \
        } else {                                                                                             \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                   \
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;                                       \
            if (FULL || row < p.M) cp[row * p.ldc] = C_[r];                                                  \
          }                                                                                                  \
        }                                                                                                    \
      }                                                                                                      \
      /* raw barrier: __syncthreads() would also drain vmcnt, i.e. wait for the C stores to reach memory */         \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
      __builtin_amdgcn_s_barrier();                                                                          \
      asm volatile("" ::: "memory");                                                                         \
      ++it;                                                                                                  \
    }

    for (int kc = 0; kc < n_kc; ++kc) {
      // this wave's 32 rows x 128 k of A: load, split into three bf16 pieces, keep in registers for all column tiles
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const float4 lo = *reinterpret_cast<const float4*>(ap + kc * KC + ks * 16);
        const float4 hi = *reinterpret_cast<const float4*>(ap + kc * KC + ks * 16 + 4);
        { const Bf3 t = split3(lo.x); af[ks][0][0] = t.a; af[ks][1][0] = t.b; af[ks][2][0] = t.c; }
        { const Bf3 t = split3(lo.y); af[ks][0][1] = t.a; af[ks][1][1] = t.b; af[ks][2][1] = t.c; }
        { const Bf3 t = split3(lo.z); af[ks][0][2] = t.a; af[ks][1][2] = t.b; af[ks][2][2] = t.c; }
        { const Bf3 t = split3(lo.w); af[ks][0][3] = t.a; af[ks][1][3] = t.b; af[ks][2][3] = t.c; }
        { const Bf3 t = split3(hi.x); af[ks][0][4] = t.a; af[ks][1][4] = t.b; af[ks][2][4] = t.c; }
        { const Bf3 t = split3(hi.y); af[ks][0][5] = t.a; af[ks][1][5] = t.b; af[ks][2][5] = t.c; }
        { const Bf3 t = split3(hi.z); af[ks][0][6] = t.a; af[ks][1][6] = t.b; af[ks][2][6] = t.c; }
        { const Bf3 t = split3(hi.w); af[ks][0][7] = t.a; af[ks][1][7] = t.b; af[ks][2][7] = t.c; }
      }
      if (PERSIST) {
#pragma unroll
        for (int ct = 0; ct < (PERSIST ? NCT : 1); ++ct) MMA_X3_TILE(acc[ct], ct, kc == n_kc - 1)
      } else {
        for (int ct = 0; ct < n_ct; ++ct) {
          MMA_X3_TILE(acc[0], ct, true)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        }
      }
    }
  }
}

#undef MMA_X3_TILE

// ---- K == 128, N % 128 == 0: "column group" form (round 2) ------------------------------------------------------------
// A workgroup owns 128 columns of C for the whole launch: their three-piece B slab (4 tiles x 3 x 32 x 128 bf16 = 102 KB
// with the row padding) is staged into LDS ONCE, so the steady state has no slab loads, no LDS writes and no barrier - a
// wave only streams its A rows in, runs 4 x 48 MFMAs against the resident slab (fragments of the next k-step requested
// before the current step's MFMAs) and stores its 4 tiles with buffer stores (SGPR row offsets: no per-store address VALU).
// 8 waves = 256 rows per workgroup step; one workgroup per CU (LDS), 2 waves per SIMD.  The N/128 workgroups that take
// the SAME rows sit on one XCD (block ids congruent mod 8 share an XCD and its L2) and share A through that L2.
// Measured (C4 forward, M = 2^20, N = 1024; same box, back to back): gemm_x3_kernel<0> 1.65-1.67 ms, this kernel 1.51-1.53 ms.
// With ZERO-filled operands the same two binaries take 1.31 and 1.09 ms: on real data the kernel is held by the chip's
// power management (PMC: 1.84 GHz effective clock instead of 2.4), not by its schedule - a 12-wave variant (3 per SIMD)
// and a variant that prefetches the next A block into registers both land on the same 1.51-1.53 ms.
constexpr int kCgWaves = 8, kCgThreads = 64 * kCgWaves, kCgRows = 32 * kCgWaves;
// nt on the C stores: C streams out once; with the default policy the 4 MB of C per step flush the A rows the other column
// groups still need out of the XCD's L2 (PMC: 2.5 GB of A re-reads per launch, 1.6 GB with nt; A itself is 0.54 GB)
constexpr int kCgStoreAux = 2;
constexpr int kCgPitch = 128 * 2 + 16, kCgPiece = 32 * kCgPitch, kCgTile = 3 * kCgPiece, kCgLds = 4 * kCgTile;
constexpr int kCgSlotsPerXcd = 32;                  // 256 CUs / 8 XCDs

struct BFrag { bf16x8 b1, b2, b3; };
__device__ __forceinline__ BFrag cg_frag(const unsigned char* sb, int ks) {
  BFrag f;
  f.b1 = *reinterpret_cast<const bf16x8*>(sb + 0 * kCgPiece + ks * 32);
  f.b2 = *reinterpret_cast<const bf16x8*>(sb + 1 * kCgPiece + ks * 32);
  f.b3 = *reinterpret_cast<const bf16x8*>(sb + 2 * kCgPiece + ks * 32);
  return f;
}

__global__ __launch_bounds__(kCgThreads, 1) void gemm_x3_colgroup_kernel(const GemmParams p, int64_t n_units, int n_groups) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kCgLds];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int streams_per_xcd = kCgSlotsPerXcd / n_groups;
  if (slot >= streams_per_xcd * n_groups) return;       // N/128 does not divide 32: the last slots of every XCD stay idle (6 groups: 2 of 32)
  const int g = slot % n_groups;
  const int64_t stream = xcd * streams_per_xcd + slot / n_groups, n_streams = 8 * streams_per_xcd;

  for (int q = tid; q < 4 * 3 * 32 * 16; q += kCgThreads) {          // 6144 16-byte chunks, 12 per thread
    const int kq = q & 15, col = (q >> 4) & 31, tp = q >> 9, piece = tp % 3, tile = tp / 3;
    *reinterpret_cast<uint4*>(lds + tile * kCgTile + piece * kCgPiece + col * kCgPitch + kq * 16) =
        *reinterpret_cast<const uint4*>(p.Bt + ((size_t)piece * p.N + (size_t)(g * 128 + tile * 32 + col)) * 128 + kq * 8);
  }
  __syncthreads();

  // addresses = wave-uniform base (SGPRs) + one 32-bit lane offset: 16 precomputed 64-bit row pointers would cost 32 VGPRs
  const uint32_t a_off = (uint32_t)r31 * (uint32_t)p.lda + 8u * h;
  const uint32_t c_off = 4u * h * (uint32_t)p.ldc + (uint32_t)r31;
  const unsigned char* sb0 = lds + r31 * kCgPitch + h * 16;

  BFrag cur = cg_frag(sb0, 0);
  for (int64_t u = stream; u < n_units; u += n_streams) {
    float4 raw[16];                                          // this lane's 64 A values (prefetching the next block's changed nothing)
    {
      const float* ap = p.A + (u * kCgRows + wave * 32) * p.lda;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        raw[2 * ks] = *reinterpret_cast<const float4*>(ap + a_off + ks * 16);
        raw[2 * ks + 1] = *reinterpret_cast<const float4*>(ap + a_off + ks * 16 + 4);
      }
    }
    bf16x8 af[8][3];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float4 lo = raw[2 * ks], hi = raw[2 * ks + 1];
      { const Bf3 t = split3(lo.x); af[ks][0][0] = t.a; af[ks][1][0] = t.b; af[ks][2][0] = t.c; }
      { const Bf3 t = split3(lo.y); af[ks][0][1] = t.a; af[ks][1][1] = t.b; af[ks][2][1] = t.c; }
      { const Bf3 t = split3(lo.z); af[ks][0][2] = t.a; af[ks][1][2] = t.b; af[ks][2][2] = t.c; }
      { const Bf3 t = split3(lo.w); af[ks][0][3] = t.a; af[ks][1][3] = t.b; af[ks][2][3] = t.c; }
      { const Bf3 t = split3(hi.x); af[ks][0][4] = t.a; af[ks][1][4] = t.b; af[ks][2][4] = t.c; }
      { const Bf3 t = split3(hi.y); af[ks][0][5] = t.a; af[ks][1][5] = t.b; af[ks][2][5] = t.c; }
      { const Bf3 t = split3(hi.z); af[ks][0][6] = t.a; af[ks][1][6] = t.b; af[ks][2][6] = t.c; }
      { const Bf3 t = split3(hi.w); af[ks][0][7] = t.a; af[ks][1][7] = t.b; af[ks][2][7] = t.c; }
    }
    // C goes out through buffer stores: wave-uniform descriptor + SGPR row offset + one VGPR lane offset (global stores made
    // the compiler keep 16 64-bit row pointers in VGPRs and add to each per tile)
    const __amdgpu_buffer_rsrc_t crow =
        __builtin_amdgcn_make_buffer_rsrc(p.C + (u * kCgRows + wave * 32) * p.ldc + g * 128, 0, 0x7fffffff, 0x00020000);
#pragma unroll 1
    for (int ct = 0; ct < 4; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const unsigned char* sb = sb0 + ct * kCgTile;
      const unsigned char* sbn = sb0 + ((ct + 1) & 3) * kCgTile;       // B is static: the first fragments of the next tile too
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        // the fragments of the NEXT k-step are requested before this step's MFMAs (left alone the compiler reads, waits,
        // multiplies: ~100 cycles of LDS latency exposed per 192 cycles of MFMA)
        const BFrag nxt = ks < 7 ? cg_frag(sb, ks + 1) : cg_frag(sbn, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], cur.b3, acc, 0, 0, 0);     // same product order as gemm_x3_kernel:
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], cur.b2, acc, 0, 0, 0);     // bitwise the same C
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][2], cur.b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], cur.b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], cur.b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], cur.b1, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), crow, c_off * 4u,
                                              (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + ct * 32) * 4u, kCgStoreAux);
    }
  }
}

// ---- K == 128, three products: fp16 x 2 pieces with row / column scales (round 2) -------------------------------------------
// a = s_row (a_hi + 2^-11 a_lo), b = s_col (b_hi + 2^-11 b_lo) with fp16 pieces (11 + 11 mantissa bits) and POWER-OF-TWO scales that
// put every row maximum of A and column maximum of B into [2^14, 2^15): fp16's exponent range then never limits a value
// that matters (29 binades below the maximum keep full precision), the scaling is exact, and
//     a b ~= hi hi + 2^-11 (hi lo + lo hi)        (dropped: lo lo, 2^-22 relative)
// needs THREE MFMAs per k-step instead of six (two accumulators).  Measured max error 1.1e-7 sum|a||b| (fp32 library GEMM:
// 4.1e-7).  The row scale is free here: in the column-group form a wave holds whole rows of A (K = 128) in registers.  Same
// structure as gemm_x3_colgroup_kernel (resident B slab: 70 KB, pipelined fragments, nt buffer stores); handles a ragged M itself.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
struct HFrag { f16x8 b1, b2; };
// [r5] KS k-steps of 16: K = 16 KS in {64, 96, 128}.  The B slab of a 128-column group is 2 pieces x 128 columns x (32 KS + 16) bytes - 68 KB at
// K = 128, 36 KB at K = 64, 52 KB at K = 96 - so the narrow forms keep THREE groups resident (G2 = 3: 108 / 156 KB).
template <int KS> struct HgGeom {
  static constexpr int pitch = KS * 32 + 16, piece = 32 * pitch, tile = 2 * piece, group = 4 * tile;
};
template <int KS>
__device__ __forceinline__ HFrag hg_frag(const unsigned char* sb, int ks) {
  HFrag f;
  f.b1 = *reinterpret_cast<const f16x8*>(sb + 0 * HgGeom<KS>::piece + ks * 32);
  f.b2 = *reinterpret_cast<const f16x8*>(sb + 1 * HgGeom<KS>::piece + ks * 32);
  return f;
}
// G2 = 2: a workgroup owns TWO adjacent 128-column groups (136 KB of B in LDS): a wave's rows are loaded, scaled and split once for
// 256 columns of C instead of once per 128 - half the A traffic through L2 and half the split work per output (N/128 even).
// [r5] KS < 8: the zero-padded tall Linears of graph regression (50 + 1 -> 64 columns, 75 + 1 -> 96) no longer multiply, load, split and
// write pad columns up to 128: at C2L the edge-feature product (4e5 x 51 -> 380) read its 205 MB padded operand three times (G2 = 1, 384 = 3
// groups) for 614 MB of output; K = 64 with G2 = 3 reads 102 MB once.
template <int G2, int KS = 8>
__global__ __launch_bounds__(kCgThreads, 1) void gemm_f16x2_colgroup_kernel(const GemmParams p, const float* col_unscale, int64_t n_units,
                                                                           int n_groups, float* a_row_max) {
  constexpr int dbg = kAbl;
  using Geo = HgGeom<KS>;
  constexpr int K = 16 * KS;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G2 * Geo::group];
  constexpr int NT = 4 * G2;                                           // 32-column tiles per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int streams_per_xcd = kCgSlotsPerXcd / n_groups;
  if (slot >= streams_per_xcd * n_groups) return;
  const int g = slot % n_groups;
  const int64_t stream = xcd * streams_per_xcd + slot / n_groups, n_streams = 8 * streams_per_xcd;
  const _Float16* Bh = reinterpret_cast<const _Float16*>(p.Bt);            // (2, N, K) fp16: hi, lo * 2^11
  constexpr int CH = 2 * KS;                                               // 16-byte chunks per column and piece
  for (int q = tid; q < NT * 2 * 32 * CH; q += kCgThreads) {
    const int kq = q % CH, col = (q / CH) & 31, tp = q / (CH * 32), piece = tp % 2, tile = tp / 2;
    *reinterpret_cast<uint4*>(lds + tile * Geo::tile + piece * Geo::piece + col * Geo::pitch + kq * 16) =
        *reinterpret_cast<const uint4*>(Bh + ((size_t)piece * p.N + (size_t)(g * 32 * NT + tile * 32 + col)) * K + kq * 8);
  }
  __syncthreads();
  const uint32_t c_off = 4u * h * (uint32_t)p.ldc + (uint32_t)r31;
  const unsigned char* sb0 = lds + r31 * Geo::pitch + h * 16;
  HFrag cur = hg_frag<KS>(sb0, 0);
  // this lane's column in each of the 4 tiles: log2 of its un-scale (an exact power of two), applied together with the row's by ONE
  // ldexp (two multiplications overflow / underflow in between for rows or columns near the ends of the fp32 range).  Loaded
  // ONCE: a load inside the tile loop would sit behind the previous tile's 16 stores in the in-order vmcnt queue.
  int cues[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) cues[t] = (int)((__float_as_uint(col_unscale[g * 32 * NT + 32 * t + r31]) >> 23) & 0xFF) - 127;
  // the finished tile waiting to be stored (its stores ride between the next tile's MFMAs): values, descriptor, column offset.  Before the
  // first tile the descriptor covers ZERO bytes: the hardware's range check drops those stores, so the loop needs no "is there one" branch
  float prev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
  __amdgpu_buffer_rsrc_t crow_p = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);
  uint32_t pcol = 0;
  // The rows of a unit are requested at the START of their own unit.  vmcnt is one in-order counter for loads and stores, so waiting for
  // them also waits for every store the wave issued before them (measurement builds, C4: 1.04 ms with the loads, 0.81 without, 0.75 for
  // the stores alone).
  // [r4, measured and NOT kept] requesting a unit's rows one unit ahead (inline-asm loads, hand-placed vmcnt(63): hipcc otherwise merges
  // the loop's entry and back edge into a vmcnt(14) .. vmcnt(0) ladder, a full drain of the wave's stores per unit) changed nothing:
  // 1.04 -> 1.06-1.12 ms at C4.  Measurement builds (MMA_ABL): the stores alone 0.75 ms, everything but the A loads 0.81, with them
  // 1.04 - the loads cost by BEING there, not by being waited for: a fragment-shaped load (32 rows x 32 bytes per instruction) is ~32 line
  // requests to the texture addresser, time the store stream does not get.  The cure is a coalesced load + a transpose through LDS, for
  // which there is no LDS left beside the resident B slab (136 KB).
  for (int64_t u = stream; u < n_units; u += n_streams) {
    const int64_t row0 = u * kCgRows + wave * 32;
    if (row0 >= p.M) continue;                                      // the last unit may be ragged: rows past M are re-read (the last
    const int64_t rows_here = min((int64_t)32, p.M - row0);          // row) and their stores dropped by the buffer range check
    float4 raw[2 * KS];
    if (dbg & 8) {
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) raw[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    } else {
      const float* ap = p.A + min(row0 + r31, p.M - 1) * p.lda + 8 * h;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        raw[2 * ks] = *reinterpret_cast<const float4*>(ap + ks * 16);
        raw[2 * ks + 1] = *reinterpret_cast<const float4*>(ap + ks * 16 + 4);
      }
    }
    // row scale: a power of two that puts the row maximum into [2^14, 2^15)
    float rmax = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) rmax = fmaxf(rmax, fmaxf(fmaxf(fabsf(raw[i].x), fabsf(raw[i].y)), fmaxf(fabsf(raw[i].z), fabsf(raw[i].w))));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    if (a_row_max && g == 0 && h == 0 && r31 < rows_here) a_row_max[row0 + r31] = rmax;     // for the weight-gradient product (TN form)
    const int ex = (int)((__float_as_uint(rmax) >> 23) & 0xFF);
    const int sce = min(max(14 - (ex - 127), -126), 127);            // log2 of the row scale
    const float sc = __uint_as_float((uint32_t)(sce + 127) << 23);
    int rse[16];                                                   // accumulator register r holds row (r&3) + 8 (r>>2) + 4h: that row's
#pragma unroll                                                     // scale exponent, fetched ONCE per block (not per tile and store)
    for (int r = 0; r < 16; ++r) rse[r] = __shfl(sce, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
    f16x8 af[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float v[8] = {raw[2*ks].x, raw[2*ks].y, raw[2*ks].z, raw[2*ks].w, raw[2*ks+1].x, raw[2*ks+1].y, raw[2*ks+1].z, raw[2*ks+1].w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = v[i] * sc;
        const _Float16 hi = (_Float16)x;
        af[ks][0][i] = hi;
        af[ks][1][i] = (_Float16)((x - (float)hi) * 2048.f);
      }
    }
    const __amdgpu_buffer_rsrc_t crow =
        __builtin_amdgcn_make_buffer_rsrc(p.C + row0 * p.ldc, 0, (int)min((int64_t)0x7fffffff, rows_here * p.ldc * 4), 0x00020000);
#pragma unroll 1
    for (int ct = 0; ct < NT; ++ct) {
      int cue = cues[0];
#pragma unroll
      for (int t = 1; t < NT; ++t) cue = ct == t ? cues[t] : cue;       // selects: a run-time index would put the array in scratch
      f32x16 acc, acl;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acl[r] = 0.f; }
      const unsigned char* sb = sb0 + ct * Geo::tile;
      const unsigned char* sbn = sb0 + (ct + 1 == NT ? 0 : ct + 1) * Geo::tile;
      constexpr int SPK = (16 + KS - 1) / KS;                          // stores of the previous tile behind every k-step
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (!(dbg & 2)) {
          const HFrag nxt = ks < KS - 1 ? hg_frag<KS>(sb, ks + 1) : hg_frag<KS>(sbn, 0);
          __builtin_amdgcn_sched_barrier(0);
          acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b2, acl, 0, 0, 0);
          acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][1], cur.b1, acl, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b1, acc, 0, 0, 0);
          cur = nxt;
        } else {
          asm volatile("" :: "v"(af[ks][0]), "v"(af[ks][1]));      // keep the split alive
        }
        // [r4] two stores of the PREVIOUS tile behind every k-step of this one (in program order: a wave issues in order).  With the 16
        // stores of a tile issued in one burst after its MFMAs, all eight waves of the CU multiplied together and then all sat in front
        // of the full store queue together: stores alone 0.74 ms, everything but the stores 0.64 ms, both 1.05 ms (measurement
        // builds, C4) - nothing overlapped.  Spread out, a wave that waits for a store slot has its MFMAs in between.
        if (KS == 8 && (dbg & 64)) {             // measurement: the tile as FOUR 16-byte stores (8 whole 128-byte rows each), wrong data
          if (ks & 1) {
            const int q4 = ks >> 1;
            typedef unsigned int su4 __attribute__((ext_vector_type(4)));
            const su4 v = {__float_as_uint(prev[4 * q4]), __float_as_uint(prev[4 * q4 + 1]), __float_as_uint(prev[4 * q4 + 2]), __float_as_uint(prev[4 * q4 + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, crow_p, ((uint32_t)(lane >> 3) * (uint32_t)p.ldc + (uint32_t)(lane & 7) * 4u) * 4u,
                                                   (uint32_t)(8 * q4 * (int)p.ldc) * 4u + pcol, kCgStoreAux);
          }
        } else if (!(dbg & 4)) {
#pragma unroll
          for (int j = 0; j < SPK; ++j) {
            const int r = SPK * ks + j;
            if (r < 16)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off * 4u,
                                                    (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol, kCgStoreAux);
          }
        } else {
          asm volatile("" :: "v"(prev[(2 * ks) & 15]), "v"(prev[(2 * ks + 1) & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // row r of the tile belongs to lane-row (r&3)+8*(r>>2)+4h: its scale lives in the lane with that r31
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = ldexpf(acc[r] + acl[r] * (1.f / 2048.f), cue - rse[r]);
      crow_p = crow;
      pcol = (uint32_t)(g * 32 * NT + ct * 32) * 4u;
    }
  }
  // the last tile of the last unit
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off * 4u, (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol,
                                          kCgStoreAux);
}

// [r5] The two forward forms below (W-stationary, loader waves) are MEASUREMENT kernels: built, bit-equal to the column-group kernels,
// and slower (C4 1.37 / 1.19-1.27 ms against 1.04).  They are compiled only with -DMMA_EXPERIMENTAL_FWD (make EXTRA=-DMMA_EXPERIMENTAL_FWD;
// round-4 ADVICE: ~400 lines of slower kernels and an environment switch do not belong in the product library's hot path); without it
// mma_gemm_f16x2_ws reports that, and MMA_FWD_WS is not read.
#ifdef MMA_EXPERIMENTAL_FWD
// ---- forward, W-STATIONARY (round 4): C (M, N) = x (M, K) W (K, N), K = 128 or 256, N % 256 == 0 ------------------------------------------
// The column-group kernels above keep W in LDS and stream x through registers in the MFMA's fragment shape: 32 rows x 32 bytes per load
// instruction - ~32 line requests to the texture addresser for 1 KB, and every 256-column (K = 256: 128-column) group of the same
// rows loads them again.  Measurement builds at C4: the stores alone 0.75 ms, everything but the x loads 0.81, with them 1.04 - the
// loads cost by being there, whether waited for or not (an asm-prefetched form changed nothing).  Here the roles are swapped:
//   * a WAVE keeps its own 32 columns of W in registers for the whole launch (both fp16 pieces of all K/16 k-steps: 64 VGPRs at
//     K = 128, 128 at K = 256) - there is no B slab in LDS, no fragment read of B, and LDS is free for x;
//   * the eight waves of a workgroup (256 columns) load each 64-row (K = 256: 32-row) slice of x TOGETHER, row-major: one load
//     instruction covers two whole rows (K = 256: one), 8 lines instead of 32; a row's maximum is a butterfly over the lanes that hold
//     it, the power-of-two row scale and the split are done ONCE per row (not once per wave that multiplies it), and the fp16 pieces go to
//     LDS in the MFMA's A-fragment order, [block][piece][k-step][lane][8 halves], which every wave then reads with lane-linear
//     ds_read_b128 (conflict-free);
//   * two LDS slots: iteration i multiplies slot i % 2 while the slice for iteration i+1 - requested at the top of iteration i, so that
//     it is in flight behind the multiplications and is both issued and waited for inside ONE iteration (no load crosses the loop's
//     back edge: hipcc merges entry and back edge into a full vmcnt drain otherwise) - is split and written to the other slot at the
//     bottom; one barrier per iteration;
//   * the stores of a finished 32 x 32 tile ride between the MFMAs of the next one (as in the column-group kernels, [r4]).
// Same products in the same order as gemm_f16x2_colgroup_kernel (hi lo, lo hi into one accumulator, hi hi into the other, lo pieces
// pre-scaled by 2^11): the results are bit-identical to it.  Bt2 = (2, N, K) as for mma_gemm_f16x2.
template <int KS, int RBI>
__global__ __launch_bounds__(kCgThreads, 1) void gemm_f16x2_ws_kernel(const GemmParams p, const float* col_unscale, int64_t n_rgroups,
                                                                     int n_cgroups, float* a_row_max) {
  constexpr int K = 16 * KS;
  constexpr int RI = 32 * RBI;                             // rows per iteration (a "row group")
  constexpr int kKsP = 1024 + 32;                          // bytes per (piece, k-step) block of A fragments; the pad spreads the k-steps over the banks
  constexpr int kPieceP = KS * kKsP, kBlkP = 2 * kPieceP;  // piece / 32-row block pitch
  constexpr int kSlot = RBI * kBlkP + RI * 4;              // + the rows' scale exponents
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kSlot];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int streams_per_xcd = kCgSlotsPerXcd / n_cgroups;
  if (slot_id >= streams_per_xcd * n_cgroups) return;      // (before any barrier)
  const int cg = slot_id % n_cgroups;
  const int64_t stream = xcd * streams_per_xcd + slot_id / n_cgroups, n_streams = 8 * streams_per_xcd;
  const int col0 = cg * 256 + wave * 32;                   // this wave's 32 columns

  // W: this wave's columns, both pieces, every k-step - B[k][n]: lane (n = r31, h) holds k = 16 ks + 8 h + 0..7 of column col0 + r31
  f16x8 wh[KS], wl[KS];
  {
    const _Float16* Bh = reinterpret_cast<const _Float16*>(p.Bt);
    const _Float16* wp = Bh + (size_t)(col0 + r31) * K + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wh[ks] = *reinterpret_cast<const f16x8*>(wp + ks * 16);
      wl[ks] = *reinterpret_cast<const f16x8*>(wp + (size_t)p.N * K + ks * 16);
    }
  }
  const int cue = (int)((__float_as_uint(col_unscale[col0 + r31]) >> 23) & 0xFF) - 127;     // log2 of this lane's column un-scale
  const uint32_t c_off = (4u * h * (uint32_t)p.ldc + (uint32_t)r31) * 4u;
  const uint32_t pcol = (uint32_t)col0 * 4u;

  // producer role: this wave prepares rows [wave * 4 RBI, +4 RBI) of every row group; 4 float4 per lane, row-major
  constexpr int C4 = K / 4;                                // float4 per row: 32 or 64
  int p_row[4], p_k0[4];                                   // row inside the group and first k of the lane's float4 number i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = lane + 64 * i;
    p_row[i] = wave * (4 * RBI) + f / C4;
    p_k0[i] = 4 * (f % C4);
  }
  float4 raw[4];
#define MMA_WS_LOAD(G_)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
    const int64_t row = min((G_) * RI + p_row[i], p.M - 1);          /* rows past M re-read the last row; never stored */ \
    raw[i] = *reinterpret_cast<const float4*>(p.A + row * p.lda + p_k0[i]);                        \
  }
// raw (rows of group G_) -> row maxima, scale exponents and fragment-ordered fp16 pieces in LDS slot S_
#define MMA_WS_PRODUCE1(G_, S_, I_)                                                               \
  {                                                                                                \
    unsigned char* sl_ = lds + (S_) * kSlot;                                                       \
    /* pins the part HERE: arithmetic on the loaded values is free to move up to the loads otherwise (IR-level code motion does not */ \
    /* see sched_barrier) - and with it the wait for them, in front of the iteration's stores (seen in the K = 256 ISA: vmcnt(3)) */  \
    asm volatile("" : "+v"(raw[I_].x), "+v"(raw[I_].y), "+v"(raw[I_].z), "+v"(raw[I_].w));          \
    float rmax = fmaxf(0.f, fmaxf(fmaxf(fabsf(raw[I_].x), fabsf(raw[I_].y)), fmaxf(fabsf(raw[I_].z), fabsf(raw[I_].w))));   \
    _Pragma("unroll") for (int o = 1; o < C4; o <<= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o, 64));            \
    const int ex = (int)((__float_as_uint(rmax) >> 23) & 0xFF);                                    \
    const int sce = min(max(14 - (ex - 127), -126), 127);            /* the row maximum lands in [2^14, 2^15) */ \
    const float sc = __uint_as_float((uint32_t)(sce + 127) << 23);                                 \
    if ((lane & (C4 - 1)) == 0) {                                    /* one lane per row */        \
      reinterpret_cast<int*>(sl_ + RBI * kBlkP)[p_row[I_]] = sce;                                  \
      const int64_t grow = (G_) * RI + p_row[I_];                                                  \
      if (a_row_max && cg == 0 && grow < p.M) a_row_max[grow] = rmax;        /* for the weight-gradient product (TN form) */ \
    }                                                                                              \
    const float v[4] = {raw[I_].x * sc, raw[I_].y * sc, raw[I_].z * sc, raw[I_].w * sc};           \
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));                                       \
    h4 hi, lo;                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
      hi[j] = (_Float16)v[j];                                                                      \
      lo[j] = (_Float16)((v[j] - (float)hi[j]) * 2048.f);                                          \
    }                                                                                              \
    const int blk_ = p_row[I_] >> 5, m_ = p_row[I_] & 31, k0 = p_k0[I_];                           \
    unsigned char* d = sl_ + blk_ * kBlkP + (k0 >> 4) * kKsP + (m_ + 32 * ((k0 >> 3) & 1)) * 16 + (k0 & 7) * 2;  \
    *reinterpret_cast<h4*>(d) = hi;                                                                \
    *reinterpret_cast<h4*>(d + kPieceP) = lo;                                                      \
  }
#define MMA_WS_PRODUCE(G_, S_) { MMA_WS_PRODUCE1(G_, S_, 0) MMA_WS_PRODUCE1(G_, S_, 1) MMA_WS_PRODUCE1(G_, S_, 2) MMA_WS_PRODUCE1(G_, S_, 3) }

  // the finished tile waiting to be stored: values, descriptor (ZERO bytes before the first tile: the range check drops those stores)
  float prev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
  __amdgpu_buffer_rsrc_t crow_p = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);

  if (stream < n_rgroups) {
    MMA_WS_LOAD(stream)
    MMA_WS_PRODUCE(stream, 0)
  }
  __syncthreads();
  int it = 0;
  for (int64_t gI = stream; gI < n_rgroups; gI += n_streams, ++it) {
    const int64_t gN = gI + n_streams < n_rgroups ? gI + n_streams : gI;        // the last group requests its own rows again (4 loads, unused)
    MMA_WS_LOAD(gN)
    const unsigned char* sl = lds + (it & 1) * kSlot;
#pragma unroll
    for (int blk = 0; blk < RBI; ++blk) {
      const int64_t row0 = gI * RI + blk * 32;
      const int64_t rows_here = min((int64_t)32, p.M - row0);
      const __amdgpu_buffer_rsrc_t crow = __builtin_amdgcn_make_buffer_rsrc(
          p.C + min(row0, p.M - 1) * p.ldc, 0, rows_here > 0 ? (int)min((int64_t)0x7fffffff, rows_here * p.ldc * 4) : 0, 0x00020000);
      f32x16 acc, acl;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acl[r] = 0.f; }
      const unsigned char* fa = sl + blk * kBlkP + lane * 16;
      f16x8 ah = *reinterpret_cast<const f16x8*>(fa), al = *reinterpret_cast<const f16x8*>(fa + kPieceP);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f16x8 nh = ah, nl = al;
        if (ks + 1 < KS) {
          nh = *reinterpret_cast<const f16x8*>(fa + (ks + 1) * kKsP);
          nl = *reinterpret_cast<const f16x8*>(fa + kPieceP + (ks + 1) * kKsP);
        }
        __builtin_amdgcn_sched_barrier(0);
        acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[ks], acl, 0, 0, 0);
        acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[ks], acl, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ks], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16 / KS; ++j) {                          // 2 stores of the previous tile per k-step (K = 128), 1 (K = 256)
          const int r = (16 / KS) * ks + j;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off,
                                                (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol, kCgStoreAux);
        }
        // the NEXT group's rows, one load's worth (a row or two) every KS/4 k-steps of the group's last block: row maximum, scale, split
        // and the LDS writes sit in the MFMAs' shadow like the stores do - at the bottom of the iteration all eight waves did them
        // at once, with the matrix pipe and the store queue idle (1.36 ms at C4 against the column-group kernel's 1.10)
        if (blk == RBI - 1 && (ks % (KS / 4)) == KS / 4 - 1) {
          constexpr int dummy_ = 0; (void)dummy_;
          switch (ks / (KS / 4)) {
            case 0: MMA_WS_PRODUCE1(gN, (it + 1) & 1, 0) break;
            case 1: MMA_WS_PRODUCE1(gN, (it + 1) & 1, 1) break;
            case 2: MMA_WS_PRODUCE1(gN, (it + 1) & 1, 2) break;
            default: MMA_WS_PRODUCE1(gN, (it + 1) & 1, 3) break;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        ah = nh; al = nl;
      }
      // the scale exponents of the 16 rows this lane's accumulator registers hold, rows (r&3) + 8 (r>>2) + 4h - fetched now, not before
      // the k-steps: at K = 256 (128 registers of W) sixteen more live registers through the loop meant spills
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int4 e4 = *reinterpret_cast<const int4*>(sl + RBI * kBlkP + (blk * 32 + 8 * q + 4 * h) * 4);
        const int rse[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) prev[4 * q + j] = ldexpf(acc[4 * q + j] + acl[4 * q + j] * (1.f / 2048.f), cue - rse[j]);
      }
      crow_p = crow;
    }
    __syncthreads();
  }
#undef MMA_WS_LOAD
#undef MMA_WS_PRODUCE
#undef MMA_WS_PRODUCE1
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off, (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol,
                                          kCgStoreAux);
}

// ---- forward, LOADER WAVES (round 4): the W-stationary kernel above with its two jobs given to different waves ---------------------------
// What the W-stationary kernel taught: vmcnt is ONE in-order counter for loads and stores, so a wave that
// stores tiles and also waits for rows of x waits for its own stores first.  Here waves 0-7 MULTIPLY (W in registers, A fragments from
// LDS, tile stores between the MFMAs) and never issue a global load after their prologue - nothing they wait for stands behind a store;
// waves 8-11 LOAD: each brings 16 of the row group's 64 rows in, row-major (whole lines), forms the row maxima, splits and writes the
// fp16 pieces to the other LDS slot - their vmcnt queue holds loads only (plus one 4-byte row-maximum store per row).  One raw barrier per iteration (lgkmcnt only: a
// __syncthreads() would also drain the multipliers' stores).  K = 128, N % 256 == 0; same products in the same order: the bits of
// mma_gemm_f16x2.
constexpr int kLwThreads = 768;          // 8 multiplier waves + 4 loader waves
typedef float lw_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kLwThreads, 1) void gemm_f16x2_lw_kernel(const GemmParams p, const float* col_unscale, int64_t n_rgroups,
                                                                     int n_cgroups, float* a_row_max) {
  constexpr int KS = 8, K = 128, RBI = 2, RI = 64;
  constexpr int kKsP = 1024 + 32;
  constexpr int kPieceP = KS * kKsP, kBlkP = 2 * kPieceP;
  constexpr int kSlot = RBI * kBlkP + RI * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kSlot];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int streams_per_xcd = kCgSlotsPerXcd / n_cgroups;
  if (slot_id >= streams_per_xcd * n_cgroups) return;      // (before any barrier)
  const int cg = slot_id % n_cgroups;
  const int64_t stream = xcd * streams_per_xcd + slot_id / n_cgroups, n_streams = 8 * streams_per_xcd;
  const int64_t n_it = stream < n_rgroups ? (n_rgroups - stream + n_streams - 1) / n_streams : 0;     // row groups of this workgroup
#define MMA_LW_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

  if (wave >= 8) {
    // ---------------- loader: rows [32 lw, 32 lw + 32) of every row group; load i covers rows 2i, 2i+1 (lane >> 5), 16 bytes per lane
    const int lw = wave - 8;                                  // 0..3: rows [16 lw, 16 lw + 16) of the group, load i = rows 2i, 2i+1
    __builtin_amdgcn_s_setprio(3);                            // the loaders' short instruction stream goes ahead of the multipliers' MFMAs:
                                                              // loaders + multipliers without stores 0.79 -> 0.71 ms, everything 1.27 -> 1.19
    const int row_in = 16 * lw + h, k0 = 4 * r31;             // + 2 i
    const uint32_t lda_b = (uint32_t)p.lda * 4u;
    const uint32_t a_voff = (uint32_t)h * lda_b + (uint32_t)k0 * 4u, a_voff_c = (uint32_t)k0 * 4u;
    lw_f4 rawA[8], rawB[8];
// The loader's loads cross the loop's back edge (requested one iteration before they are split): hipcc then drains vmcnt at the loop head
// (seen in the ISA: vmcnt(0) once per pair of iterations - the prefetch waited for on the spot), so they are inline asm with hand-placed
// counted waits, as in gemm_f16x2_nlp_kernel: a wave-uniform SGPR base per row pair + one 32-bit lane offset, `s_nop 4` in front (the
// base may come straight from a v_readlane: 5 wait states before a vector-memory instruction reads it), early-clobber outputs; the
// wait names the registers it releases.  Rows past M: the base row is clamped to M-1 and the lane offset of the pair's second row to
// the first (they are never stored).
#define MMA_LW_LOAD(R_, IT_)                                                                       \
    {                                                                                              \
      const int64_t g_ = stream + min((int64_t)(IT_), n_it - 1) * n_streams;      /* past the end: the last group again (unused) */ \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                              \
        const int64_t r_ = g_ * RI + 16 * lw + 2 * i;                                              \
        const float* b_ = p.A + min(r_, p.M - 1) * p.lda;                                          \
        const uint32_t vo_ = (r_ + 1 < p.M) ? a_voff : a_voff_c;                                   \
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(R_[i]) : "v"(vo_), "s"(b_) : "memory");          \
      }                                                                                            \
    }
// at most 8 younger operations in flight: releases the raw set R_ (and everything older)
#define MMA_LW_WAIT8(R_)                                                                           \
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(R_[0]), "+v"(R_[1]), "+v"(R_[2]), "+v"(R_[3]), "+v"(R_[4]), "+v"(R_[5]), "+v"(R_[6]), "+v"(R_[7]) :: "memory");
// three passes over the 8 row pairs, so that their dependency chains interleave: with the one-lane store of the row maximum inside the
// per-pair loop every pair was its own basic block and the wave walked eight ~40-instruction chains one after the other
#define MMA_LW_PRODUCE(R_, IT_, S_)                                                                \
    {                                                                                              \
      unsigned char* sl_ = lds + (S_) * kSlot;                                                     \
      const int64_t g_ = stream + (int64_t)(IT_) * n_streams;                                      \
      const bool real_ = (IT_) < n_it;                                                             \
      float rmax_[8]; int sce_[8];                                                                 \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                              \
        float rmax = fmaxf(0.f, fmaxf(fmaxf(fabsf(R_[i][0]), fabsf(R_[i][1])), fmaxf(fabsf(R_[i][2]), fabsf(R_[i][3]))));   \
        /* maximum over the 32 lanes that hold the row: four DPP steps inside the 16-lane rows (quad xor 1, xor 2, half-row mirror, row */ \
        /* mirror) and one swizzle across the two rows - as five ds_bpermute round trips per row the LOADERS set the kernel's pace */       \
        rmax = fmaxf(rmax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(rmax), 0xB1, 0xF, 0xF, true)));   \
        rmax = fmaxf(rmax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(rmax), 0x4E, 0xF, 0xF, true)));   \
        rmax = fmaxf(rmax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(rmax), 0x141, 0xF, 0xF, true)));  \
        rmax = fmaxf(rmax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(rmax), 0x140, 0xF, 0xF, true)));  \
        rmax = fmaxf(rmax, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(rmax), 0x401F)));              \
        const int ex = (int)((__float_as_uint(rmax) >> 23) & 0xFF);                                \
        rmax_[i] = rmax;                                                                           \
        sce_[i] = min(max(14 - (ex - 127), -126), 127);                                            \
      }                                                                                            \
      if (r31 == 0) {                                                /* one lane per row */        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                            \
          const int pr = row_in + 2 * i;                             /* row inside the group */    \
          reinterpret_cast<int*>(sl_ + RBI * kBlkP)[pr] = sce_[i];                                 \
          const int64_t grow = g_ * RI + pr;                                                       \
          if (a_row_max && cg == 0 && real_ && grow < p.M) a_row_max[grow] = rmax_[i];             \
        }                                                                                          \
      }                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                              \
        const float sc = __uint_as_float((uint32_t)(sce_[i] + 127) << 23);                         \
        const float v[4] = {R_[i][0] * sc, R_[i][1] * sc, R_[i][2] * sc, R_[i][3] * sc};           \
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));                                   \
        h4 hi, lo;                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
          hi[j] = (_Float16)v[j];                                                                  \
          lo[j] = (_Float16)((v[j] - (float)hi[j]) * 2048.f);                                      \
        }                                                                                          \
        const int pr = row_in + 2 * i, m_ = pr & 31;                                               \
        unsigned char* d = sl_ + (pr >> 5) * kBlkP + (k0 >> 4) * kKsP + (m_ + 32 * ((k0 >> 3) & 1)) * 16 + (k0 & 7) * 2;   \
        *reinterpret_cast<h4*>(d) = hi;                                                            \
        *reinterpret_cast<h4*>(d + kPieceP) = lo;                                                  \
      }                                                                                            \
    }
    // two raw sets: the rows of group it+2 are requested before the rows of group it+1 are split (a full iteration ahead).  Four loader
    // waves: with two, and the rows requested just before the barrier, the LOADERS set the pace (6.2 us per 64 rows against the ~3 us
    // the multipliers' stores need: 1.59 ms at C4)
    if (n_it > 0 && !(kAbl & 8)) {
      MMA_LW_LOAD(rawA, 0)
      MMA_LW_LOAD(rawB, 1)
      MMA_LW_WAIT8(rawA)
      MMA_LW_PRODUCE(rawA, 0, 0)
    }
    MMA_LW_BARRIER()
    // whole pairs of iterations, then the odd one: no path on which a raw set is requested twice without a wait in between (the wait
    // audit, tools/check_asm_waits.py, walks every path of the control-flow graph)
    int64_t it = 0;
    for (; it + 1 < n_it; it += 2) {
      // multipliers are on group `it` (slot 0): group it+1 goes to slot 1, the rows of group it+2 are requested
      if (!(kAbl & 8)) {
        MMA_LW_LOAD(rawA, it + 2)
        MMA_LW_WAIT8(rawB)
        MMA_LW_PRODUCE(rawB, it + 1, 1)
      }
      MMA_LW_BARRIER()
      if (!(kAbl & 8)) {
        MMA_LW_LOAD(rawB, it + 3)
        MMA_LW_WAIT8(rawA)
        MMA_LW_PRODUCE(rawA, it + 2, 0)
      }
      MMA_LW_BARRIER()
    }
    if (it < n_it) MMA_LW_BARRIER()                         // the multipliers' last (odd) iteration: nothing left to bring in
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawA[0]), "+v"(rawA[1]), "+v"(rawA[2]), "+v"(rawA[3]), "+v"(rawA[4]), "+v"(rawA[5]), "+v"(rawA[6]), "+v"(rawA[7]),
                 "+v"(rawB[0]), "+v"(rawB[1]), "+v"(rawB[2]), "+v"(rawB[3]), "+v"(rawB[4]), "+v"(rawB[5]), "+v"(rawB[6]), "+v"(rawB[7]) :: "memory");   // nothing in flight into dead registers at the end
#undef MMA_LW_LOAD
#undef MMA_LW_WAIT8
#undef MMA_LW_PRODUCE
    return;
  }

  // ---------------- multipliers
  const int col0 = cg * 256 + wave * 32;
  f16x8 wh[KS], wl[KS];
  {
    const _Float16* Bh = reinterpret_cast<const _Float16*>(p.Bt);
    const _Float16* wp = Bh + (size_t)(col0 + r31) * K + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wh[ks] = *reinterpret_cast<const f16x8*>(wp + ks * 16);
      wl[ks] = *reinterpret_cast<const f16x8*>(wp + (size_t)p.N * K + ks * 16);
    }
  }
  const int cue = (int)((__float_as_uint(col_unscale[col0 + r31]) >> 23) & 0xFF) - 127;
  const uint32_t c_off = (4u * h * (uint32_t)p.ldc + (uint32_t)r31) * 4u;
  const uint32_t pcol = (uint32_t)col0 * 4u;
  float prev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
  __amdgpu_buffer_rsrc_t crow_p = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // W and the column un-scale have landed: no load is waited for below
#define MMA_LW_CONSUME(IT_, S_)                                                                    \
  {                                                                                                \
    const unsigned char* sl = lds + (S_) * kSlot;                                                  \
    const int64_t gI = stream + (int64_t)(IT_) * n_streams;                                        \
    _Pragma("unroll") for (int blk = 0; blk < RBI; ++blk) {                                        \
      const int64_t row0 = gI * RI + blk * 32;                                                     \
      const int64_t rows_here = min((int64_t)32, p.M - row0);                                      \
      const __amdgpu_buffer_rsrc_t crow = __builtin_amdgcn_make_buffer_rsrc(                       \
          p.C + min(row0, p.M - 1) * p.ldc, 0, rows_here > 0 ? (int)min((int64_t)0x7fffffff, rows_here * p.ldc * 4) : 0, 0x00020000); \
      f32x16 acc, acl;                                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acl[r] = 0.f; }               \
      const unsigned char* fa = sl + blk * kBlkP + lane * 16;                                      \
      f16x8 ah = *reinterpret_cast<const f16x8*>(fa), al = *reinterpret_cast<const f16x8*>(fa + kPieceP); \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                          \
        f16x8 nh = ah, nl = al;                                                                    \
        if (ks + 1 < KS) {                                                                         \
          nh = *reinterpret_cast<const f16x8*>(fa + (ks + 1) * kKsP);                              \
          nl = *reinterpret_cast<const f16x8*>(fa + kPieceP + (ks + 1) * kKsP);                    \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (!(kAbl & 2)) {                                                                         \
        acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[ks], acl, 0, 0, 0);                    \
        acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[ks], acl, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ks], acc, 0, 0, 0);                    \
        } else { asm volatile("" :: "v"(ah), "v"(al), "v"(wh[ks]), "v"(wl[ks])); }                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
          const int r = 2 * ks + j;                                                                \
          if (!(kAbl & 4)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off, \
                                                (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol, kCgStoreAux); \
          else asm volatile("" :: "v"(prev[r]));                                                   \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        ah = nh; al = nl;                                                                          \
      }                                                                                            \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
        const int4 e4 = *reinterpret_cast<const int4*>(sl + RBI * kBlkP + (blk * 32 + 8 * q + 4 * h) * 4); \
        const int rse[4] = {e4.x, e4.y, e4.z, e4.w};                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                              \
          prev[4 * q + j] = ldexpf(acc[4 * q + j] + acl[4 * q + j] * (1.f / 2048.f), cue - rse[j]); \
      }                                                                                            \
      crow_p = crow;                                                                               \
    }                                                                                              \
  }
  MMA_LW_BARRIER()                                                                                 // slot 0 holds the first group
  int64_t it = 0;
  for (; it + 1 < n_it; it += 2) {
    MMA_LW_CONSUME(it, 0)
    MMA_LW_BARRIER()
    MMA_LW_CONSUME(it + 1, 1)
    MMA_LW_BARRIER()
  }
  if (it < n_it) {
    MMA_LW_CONSUME(it, 0)
    MMA_LW_BARRIER()
  }
#undef MMA_LW_CONSUME
#undef MMA_LW_BARRIER
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off, (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol,
                                          kCgStoreAux);
}

#endif  // MMA_EXPERIMENTAL_FWD

// ---- K == 256, three products (round 2, late): the forward [P|Q] = x [Wtop|Wbot] of hidden width 256 (C5) -----------------------------
// The column-group form with a 256-deep reduction: a workgroup owns 128 columns, their B slab (2 pieces x 128 columns x 256 k fp16 = 135 KB
// with the row padding) stays in LDS for the whole launch, a wave's 32 rows (256 floats each) are loaded once, scaled by the power of
// two the caller's row maximum implies (mma_row_absmax / a producer's bound: 32 float4 per lane cannot all wait for an in-kernel
// maximum beside their own fp16 pieces) and split into 16 k-steps x 2 pieces (128 registers), then 4 tiles x 48 MFMAs.  A is read
// once per column group through the XCD's L2 (the N/128 groups of one row stream share an XCD); the chunked N = 128 kernel, one
// launch per 128 columns, re-read A from HBM 32 times at C5 (N = 4096).
constexpr int kH2Pitch = 256 * 2 + 16, kH2Piece = 32 * kH2Pitch, kH2Tile = 2 * kH2Piece, kH2Lds = 4 * kH2Tile;      // 135 168 B
__global__ __launch_bounds__(kCgThreads, 1) void gemm_f16x2_colgroup_k256_kernel(const GemmParams p, const float* row_max, const float* col_unscale,
                                                                                int64_t n_units, int n_groups) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kH2Lds];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int streams_per_xcd = kCgSlotsPerXcd / n_groups;
  if (slot >= streams_per_xcd * n_groups) return;
  const int g = slot % n_groups;
  const int64_t stream = xcd * streams_per_xcd + slot / n_groups, n_streams = 8 * streams_per_xcd;
  const _Float16* Bh = reinterpret_cast<const _Float16*>(p.Bt);            // (2, N, 256) fp16: hi, lo * 2^11
  for (int q = tid; q < 4 * 2 * 32 * 32; q += kCgThreads) {                // 16-byte chunks: 32 per (piece, column) row
    const int kq = q & 31, col = (q >> 5) & 31, tp = q >> 10, piece = tp % 2, tile = tp / 2;
    *reinterpret_cast<uint4*>(lds + tile * kH2Tile + piece * kH2Piece + col * kH2Pitch + kq * 16) =
        *reinterpret_cast<const uint4*>(Bh + ((size_t)piece * p.N + (size_t)(g * 128 + tile * 32 + col)) * 256 + kq * 8);
  }
  __syncthreads();
  const uint32_t c_off = 4u * h * (uint32_t)p.ldc + (uint32_t)r31;
  const unsigned char* sb0 = lds + r31 * kH2Pitch + h * 16;
  int cues[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) cues[t] = (int)((__float_as_uint(col_unscale[g * 128 + 32 * t + r31]) >> 23) & 0xFF) - 127;
  float prev[16];                                  // the finished tile whose stores ride between the next tile's MFMAs (zero-byte descriptor at first)
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
  __amdgpu_buffer_rsrc_t crow_p = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);
  uint32_t pcol = 0;
  for (int64_t u = stream; u < n_units; u += n_streams) {
    const int64_t row0 = u * kCgRows + wave * 32;
    if (row0 >= p.M) break;                                          // units ascend: every later one starts past M as well
    const int64_t rows_here = min((int64_t)32, p.M - row0);          // ragged last unit: rows past M re-read the last row, stores dropped
    const int64_t myrow = row0 + min((int64_t)r31, rows_here - 1);
    const float* ap = p.A + myrow * p.lda + 8 * h;
    const float rmax = row_max[myrow];
    const int ex = (int)((__float_as_uint(rmax) >> 23) & 0xFF);
    const int sce = min(max(14 - (ex - 127), -126), 127);            // log2 of the row scale
    const float sc = __uint_as_float((uint32_t)(sce + 127) << 23);
    int rse[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rse[r] = __shfl(sce, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
    f16x8 af[16][2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {                            // 16 float4 at a time: raw pieces turn into fp16 pieces in place
      float4 raw[16];
      if (kAbl & 8) {                                                 // measurement: no A loads
#pragma unroll
        for (int i = 0; i < 16; ++i) raw[i] = make_float4(rmax, 1.f, 1.f, 1.f);
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          raw[2 * ks] = *reinterpret_cast<const float4*>(ap + (half * 8 + ks) * 16);
          raw[2 * ks + 1] = *reinterpret_cast<const float4*>(ap + (half * 8 + ks) * 16 + 4);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float v[8] = {raw[2*ks].x, raw[2*ks].y, raw[2*ks].z, raw[2*ks].w, raw[2*ks+1].x, raw[2*ks+1].y, raw[2*ks+1].z, raw[2*ks+1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = v[i] * sc;
          const _Float16 hi = (_Float16)x;
          af[half * 8 + ks][0][i] = hi;
          af[half * 8 + ks][1][i] = (kAbl & 16) ? hi : (_Float16)((x - (float)hi) * 2048.f);      // 16: measurement, no lo piece arithmetic
        }
      }
    }
    const __amdgpu_buffer_rsrc_t crow =
        __builtin_amdgcn_make_buffer_rsrc(p.C + row0 * p.ldc, 0, (int)min((int64_t)0x7fffffff, rows_here * p.ldc * 4), 0x00020000);
#pragma unroll 1
    for (int ct = 0; ct < 4; ++ct) {
      int cue = cues[0];
#pragma unroll
      for (int t = 1; t < 4; ++t) cue = ct == t ? cues[t] : cue;
      f32x16 acc, acl;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acl[r] = 0.f; }
      const unsigned char* sb = sb0 + ct * kH2Tile;
      HFrag cur;
      cur.b1 = *reinterpret_cast<const f16x8*>(sb);
      cur.b2 = *reinterpret_cast<const f16x8*>(sb + kH2Piece);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        HFrag nxt = cur;
        if (ks < 15) {
          nxt.b1 = *reinterpret_cast<const f16x8*>(sb + (ks + 1) * 32);
          nxt.b2 = *reinterpret_cast<const f16x8*>(sb + kH2Piece + (ks + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(kAbl & 2)) {
          acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b2, acl, 0, 0, 0);
          acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][1], cur.b1, acl, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b1, acc, 0, 0, 0);
        } else {
          asm volatile("" :: "v"(af[ks][0]), "v"(af[ks][1]), "v"(cur.b1), "v"(cur.b2));
        }
        // [r4] one store of the PREVIOUS tile behind every k-step (see gemm_f16x2_colgroup_kernel): the 17 GB this product writes at
        // hidden width 256 leave in a steady stream beside the MFMAs instead of in bursts between them
        if (!(kAbl & 4))
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[ks]), crow_p, c_off * 4u,
                                                (uint32_t)(((ks & 3) + 8 * (ks >> 2)) * (int)p.ldc) * 4u + pcol, kCgStoreAux);
        else
          asm volatile("" :: "v"(prev[ks]));
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = ldexpf(acc[r] + acl[r] * (1.f / 2048.f), cue - rse[r]);
      crow_p = crow;
      pcol = (uint32_t)(g * 128 + ct * 32) * 4u;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off * 4u, (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol,
                                          kCgStoreAux);
}

// ---- [r5] K == 256 with a PACKED A operand ------------------------------------------------------------------------------------------
// Measurement builds of the kernel above at the C5 shard shape (2^20 x 256 -> 4096: 7.7 ms): without its MFMAs 6.4 ms, without its stores
// 6.3, without the A loads 5.4, without the lo-piece arithmetic 7.2 - the matrix pipe is not what it waits for.  Every one of the 32 column
// groups loads the same rows in the MFMA's fragment shape (one instruction = 32 rows x 32 bytes: 32 line requests for 1 KB, every line
// requested by four instructions) and splits them again: 2 048 line requests and ~800 VALU instructions per wave and 32-row unit, 32 times.
// mma_pack_f16x2_k256 does that ONCE: per 32-row unit the 16 k-steps x 2 fp16 pieces in fragment order, [unit][k-step][piece][lane][8
// halves] (32 KB per unit, the same bytes as the fp32 rows), plus the row's scale exponent and maximum.  The product kernel then reads a
// fragment with ONE fully coalesced 1 KB load per (k-step, piece) and does no arithmetic on A at all: 7.75 -> 6.9 ms (pack pass 0.3 included).
// What is left (rocprofv3 SQ counters, profiles/r05_c5_gemm_sq.md): the matrix pipe 40 % busy, no LDS bank conflicts, LDS issue stalls 3 % of
// the wave cycles; measurement builds: without MFMAs 4.6 ms, without stores 5.0.  A wave's fragment loads sit behind the 64 stores of its
// previous unit in the one in-order vmcnt queue, so their wait is a wait for store acknowledgements - both waves of a SIMD at once.
__global__ __launch_bounds__(256) void pack_f16x2_k256_kernel(const float* __restrict__ A, int64_t lda, int64_t M, uint4* __restrict__ Ap,
                                                              int* __restrict__ sce_out, float* __restrict__ row_max_out) {
  const int lane = threadIdx.x & 63, r31 = lane & 31, h = lane >> 5;
  const int64_t n_units = (M + 31) / 32;
  for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < n_units; u += (int64_t)gridDim.x * 4) {
    const int64_t row = u * 32 + r31;
    const bool valid = row < M;
    const float* ap = A + min(row, M - 1) * lda + 8 * h;
    float4 raw[32];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      raw[2 * ks] = *reinterpret_cast<const float4*>(ap + ks * 16);
      raw[2 * ks + 1] = *reinterpret_cast<const float4*>(ap + ks * 16 + 4);
    }
    float rmax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) rmax = fmaxf(rmax, fmaxf(fmaxf(fabsf(raw[i].x), fabsf(raw[i].y)), fmaxf(fabsf(raw[i].z), fabsf(raw[i].w))));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    const int ex = (int)((__float_as_uint(rmax) >> 23) & 0xFF);
    const int sce = min(max(14 - (ex - 127), -126), 127);            // log2 of the row scale (the same rule as the kernels above)
    const float sc = valid ? __uint_as_float((uint32_t)(sce + 127) << 23) : 0.f;      // rows past M: zero pieces
    if (h == 0 && valid) { sce_out[row] = sce; if (row_max_out) row_max_out[row] = rmax; }
    uint4* dst = Ap + (size_t)u * (16 * 2 * 64) + lane;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float v[8] = {raw[2*ks].x, raw[2*ks].y, raw[2*ks].z, raw[2*ks].w, raw[2*ks+1].x, raw[2*ks+1].y, raw[2*ks+1].z, raw[2*ks+1].w};
      f16x8 hi8, lo8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = v[i] * sc;
        const _Float16 hi = (_Float16)x;
        hi8[i] = hi;
        lo8[i] = (_Float16)((x - (float)hi) * 2048.f);
      }
      dst[(ks * 2 + 0) * 64] = *reinterpret_cast<const uint4*>(&hi8);
      dst[(ks * 2 + 1) * 64] = *reinterpret_cast<const uint4*>(&lo8);
    }
  }
}

__global__ __launch_bounds__(kCgThreads, 1) void gemm_f16x2_colgroup_k256p_kernel(const GemmParams p, const uint4* __restrict__ Ap,
                                                                                 const int* __restrict__ sce_arr, const float* col_unscale,
                                                                                 int64_t n_units, int n_groups) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kH2Lds];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int streams_per_xcd = kCgSlotsPerXcd / n_groups;
  if (slot >= streams_per_xcd * n_groups) return;
  const int g = slot % n_groups;
  const int64_t stream = xcd * streams_per_xcd + slot / n_groups, n_streams = 8 * streams_per_xcd;
  const _Float16* Bh = reinterpret_cast<const _Float16*>(p.Bt);            // (2, N, 256) fp16: hi, lo * 2^11
  for (int q = tid; q < 4 * 2 * 32 * 32; q += kCgThreads) {
    const int kq = q & 31, col = (q >> 5) & 31, tp = q >> 10, piece = tp % 2, tile = tp / 2;
    *reinterpret_cast<uint4*>(lds + tile * kH2Tile + piece * kH2Piece + col * kH2Pitch + kq * 16) =
        *reinterpret_cast<const uint4*>(Bh + ((size_t)piece * p.N + (size_t)(g * 128 + tile * 32 + col)) * 256 + kq * 8);
  }
  __syncthreads();
  const uint32_t c_off = 4u * h * (uint32_t)p.ldc + (uint32_t)r31;
  const unsigned char* sb0 = lds + r31 * kH2Pitch + h * 16;
  int cues[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) cues[t] = (int)((__float_as_uint(col_unscale[g * 128 + 32 * t + r31]) >> 23) & 0xFF) - 127;
  float prev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
  __amdgpu_buffer_rsrc_t crow_p = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);
  uint32_t pcol = 0;
  // [r5, measured and NOT kept] the next unit's fragments requested during the last tile of this one, each k-step's two pieces right behind
  // the last MFMAs that read them (clean code: 192 MFMAs, four vmcnt waits per unit, no copies): 7.10 ms against 6.89 with the loads at the
  // top of their own unit - the wait for the rows is not what this kernel loses its time to.
  for (int64_t u = stream; u < n_units; u += n_streams) {
    const int64_t row0 = u * kCgRows + wave * 32;
    if (row0 >= p.M) break;
    const int64_t rows_here = min((int64_t)32, p.M - row0);
    const uint4* ap = Ap + (size_t)(row0 >> 5) * (16 * 2 * 64) + lane;
    f16x8 af[16][2];
    if (kAbl & 8) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) { af[ks][0][i] = (_Float16)1.f; af[ks][1][i] = (_Float16)0.5f; }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const uint4 a0 = ap[(ks * 2 + 0) * 64], a1 = ap[(ks * 2 + 1) * 64];
        af[ks][0] = *reinterpret_cast<const f16x8*>(&a0);
        af[ks][1] = *reinterpret_cast<const f16x8*>(&a1);
      }
    }
    const int sce = sce_arr[row0 + min((int64_t)r31, rows_here - 1)];
    int rse[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rse[r] = __shfl(sce, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
    const __amdgpu_buffer_rsrc_t crow =
        __builtin_amdgcn_make_buffer_rsrc(p.C + row0 * p.ldc, 0, (int)min((int64_t)0x7fffffff, rows_here * p.ldc * 4), 0x00020000);
#pragma unroll 1
    for (int ct = 0; ct < 4; ++ct) {
      int cue = cues[0];
#pragma unroll
      for (int t = 1; t < 4; ++t) cue = ct == t ? cues[t] : cue;
      f32x16 acc, acl;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acl[r] = 0.f; }
      const unsigned char* sb = sb0 + ct * kH2Tile;
      HFrag cur;
      cur.b1 = *reinterpret_cast<const f16x8*>(sb);
      cur.b2 = *reinterpret_cast<const f16x8*>(sb + kH2Piece);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        HFrag nxt = cur;
        if (ks < 15) {
          nxt.b1 = *reinterpret_cast<const f16x8*>(sb + (ks + 1) * 32);
          nxt.b2 = *reinterpret_cast<const f16x8*>(sb + kH2Piece + (ks + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(kAbl & 2)) {
          // (the hi product between the two lo products or behind them, with or without the scheduling fences: 6.92-6.99 ms, no difference)
          acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b2, acl, 0, 0, 0);
          acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][1], cur.b1, acl, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b1, acc, 0, 0, 0);
        } else {
          asm volatile("" :: "v"(af[ks][0]), "v"(af[ks][1]), "v"(cur.b1), "v"(cur.b2));
        }
        if (!(kAbl & 4))
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[ks]), crow_p, c_off * 4u,
                                                (uint32_t)(((ks & 3) + 8 * (ks >> 2)) * (int)p.ldc) * 4u + pcol, kCgStoreAux);
        else
          asm volatile("" :: "v"(prev[ks]));
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = ldexpf(acc[r] + acl[r] * (1.f / 2048.f), cue - rse[r]);
      crow_p = crow;
      pcol = (uint32_t)(g * 128 + ct * 32) * 4u;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[r]), crow_p, c_off * 4u, (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc) * 4u + pcol,
                                          kCgStoreAux);
}

// ---- N == 128, long K: the dL/dx products g [Wtop|Wbot]^T (round 2) ---------------------------------------------------
// The accumulator tiles of all 128 columns persist in registers over the K chunks (as in gemm_x3_kernel<4>), but a workgroup
// is 8 waves = 256 rows and the slab of one K chunk holds ALL four column tiles (4 x 3 x 32 x 64 bf16, 54 KB, double
// buffered): one barrier per 96 MFMAs instead of one per 24, and B crosses L2 -> LDS once per 256 rows instead of once per
// 128.  Fragments of the next k-step are requested before the current step's MFMAs, C moves through buffer loads / stores
// with SGPR row offsets.  A of the next chunk is prefetched raw (32 VGPRs) during the MFMAs and split after them.
constexpr int kNlKC = 64, kNlRows = 256, kNlThreads = 512;
constexpr int kNlPitch = kNlKC * 2 + 16;                 // 144 B per (piece, column) row: conflict-free ds_read_b128
constexpr int kNlPiece = 128 * kNlPitch, kNlSlab = 3 * kNlPiece;       // 55 296 B per K chunk

__device__ __forceinline__ SlabRegs nl_slab_load(const __bf16* src, int K) {
  SlabRegs r;
  r.r0 = *reinterpret_cast<const uint4*>(src);                      r.r1 = *reinterpret_cast<const uint4*>(src + (size_t)64 * K);
  r.r2 = *reinterpret_cast<const uint4*>(src + (size_t)128 * K);    r.r3 = *reinterpret_cast<const uint4*>(src + (size_t)192 * K);
  r.r4 = *reinterpret_cast<const uint4*>(src + (size_t)256 * K);    r.r5 = *reinterpret_cast<const uint4*>(src + (size_t)320 * K);
  return r;
}
__device__ __forceinline__ void nl_slab_store(unsigned char* dst, const SlabRegs& r) {
  *reinterpret_cast<uint4*>(dst) = r.r0;                            *reinterpret_cast<uint4*>(dst + 64 * kNlPitch) = r.r1;
  *reinterpret_cast<uint4*>(dst + 128 * kNlPitch) = r.r2;           *reinterpret_cast<uint4*>(dst + 192 * kNlPitch) = r.r3;
  *reinterpret_cast<uint4*>(dst + 256 * kNlPitch) = r.r4;           *reinterpret_cast<uint4*>(dst + 320 * kNlPitch) = r.r5;
}

template <bool ACC>
__global__ __launch_bounds__(kNlThreads, 1) void gemm_x3_n128_kernel(const GemmParams p, int64_t n_units) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kNlSlab];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int n_kc = p.K / kNlKC;
  const uint32_t a_off = (uint32_t)r31 * (uint32_t)p.lda + 8u * h;
  const uint32_t c_off = (4u * h * (uint32_t)p.ldc + (uint32_t)r31) * 4u;
  const unsigned char* fb = lds + r31 * kNlPitch + h * 16;             // + slab + piece * kNlPiece + ct * 32 * kNlPitch + ks * 32
  // slab staging: 384 rows (piece, column) x 8 chunks of 16 B = 3072 chunks, 6 per thread; row = q >> 3 is also the row of Bt3
  const int s_row = tid >> 3, s_kq = tid & 7;
  const __bf16* s_src = p.Bt + (size_t)s_row * p.K + s_kq * 8;          // + i * 64 rows * K + kc * 64
  unsigned char* s_dst = lds + s_row * kNlPitch + s_kq * 16;            // + i * 64 rows * pitch + slab

  for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
    const float* arow = p.A + (u * kNlRows + wave * 32) * p.lda;
    const __amdgpu_buffer_rsrc_t crow = __builtin_amdgcn_make_buffer_rsrc(p.C + (u * kNlRows + wave * 32) * p.ldc, 0, 0x7fffffff, 0x00020000);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    SlabRegs sreg = nl_slab_load(s_src, p.K);               // scalars, not an array (see SlabRegs)
    float4 raw[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      raw[2 * ks] = *reinterpret_cast<const float4*>(arow + a_off + ks * 16);
      raw[2 * ks + 1] = *reinterpret_cast<const float4*>(arow + a_off + ks * 16 + 4);
    }
    __syncthreads();                                     // the previous unit's last chunk has been read by every wave
    nl_slab_store(s_dst, sreg);
    __syncthreads();

    for (int kc = 0; kc < n_kc; ++kc) {
      bf16x8 af[4][3];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float4 lo = raw[2 * ks], hi = raw[2 * ks + 1];
        { const Bf3 t = split3(lo.x); af[ks][0][0] = t.a; af[ks][1][0] = t.b; af[ks][2][0] = t.c; }
        { const Bf3 t = split3(lo.y); af[ks][0][1] = t.a; af[ks][1][1] = t.b; af[ks][2][1] = t.c; }
        { const Bf3 t = split3(lo.z); af[ks][0][2] = t.a; af[ks][1][2] = t.b; af[ks][2][2] = t.c; }
        { const Bf3 t = split3(lo.w); af[ks][0][3] = t.a; af[ks][1][3] = t.b; af[ks][2][3] = t.c; }
        { const Bf3 t = split3(hi.x); af[ks][0][4] = t.a; af[ks][1][4] = t.b; af[ks][2][4] = t.c; }
        { const Bf3 t = split3(hi.y); af[ks][0][5] = t.a; af[ks][1][5] = t.b; af[ks][2][5] = t.c; }
        { const Bf3 t = split3(hi.z); af[ks][0][6] = t.a; af[ks][1][6] = t.b; af[ks][2][6] = t.c; }
        { const Bf3 t = split3(hi.w); af[ks][0][7] = t.a; af[ks][1][7] = t.b; af[ks][2][7] = t.c; }
      }
      const bool more = kc + 1 < n_kc;
      if (more) {                                         // next chunk: slab and A, in flight during this chunk's MFMAs
        sreg = nl_slab_load(s_src + (kc + 1) * kNlKC, p.K);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          raw[2 * ks] = *reinterpret_cast<const float4*>(arow + a_off + (kc + 1) * kNlKC + ks * 16);
          raw[2 * ks + 1] = *reinterpret_cast<const float4*>(arow + a_off + (kc + 1) * kNlKC + ks * 16 + 4);
        }
      }
      const unsigned char* sb = fb + (kc & 1) * kNlSlab;
      BFrag cur;
      cur.b1 = *reinterpret_cast<const bf16x8*>(sb);
      cur.b2 = *reinterpret_cast<const bf16x8*>(sb + kNlPiece);
      cur.b3 = *reinterpret_cast<const bf16x8*>(sb + 2 * kNlPiece);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          BFrag nxt = cur;
          if (ks < 3 || ct < 3) {
            const unsigned char* q = sb + (ks < 3 ? ct : ct + 1) * 32 * kNlPitch + (ks < 3 ? ks + 1 : 0) * 32;
            nxt.b1 = *reinterpret_cast<const bf16x8*>(q);
            nxt.b2 = *reinterpret_cast<const bf16x8*>(q + kNlPiece);
            nxt.b3 = *reinterpret_cast<const bf16x8*>(q + 2 * kNlPiece);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], cur.b3, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], cur.b2, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][2], cur.b1, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], cur.b2, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], cur.b1, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], cur.b1, acc[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          cur = nxt;
        }
      }
      if (more) {
        nl_slab_store(s_dst + ((kc + 1) & 1) * kNlSlab, sreg);
        __syncthreads();                                  // slab kc+1 visible; slab kc (read above) may be overwritten at kc+2
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (ACC)   // C += : ONE fp32 addition per element, done by the L2 atomic unit (every element has exactly one writer, so the
                   // result is the deterministic old + product).  Old values in registers cost 64 VGPRs (spills); starting the
                   // accumulators from them would round the old value once per MFMA instead of once.
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[t][r], crow, c_off,
                                                          (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + t * 32) * 4u, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][r]), crow, c_off,
                                                (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + t * 32) * 4u, 0);
  }
}

// ---- N == 128, long K, three products (round 2): gemm_x3_n128_kernel with fp16 x 2 pieces -------------------------------------
// Same structure (8 waves = 256 rows, whole-chunk slab, one barrier per K chunk, pipelined fragments, atomic C +=), half the
// MFMAs.  The row scale comes from row_max (>= the row's maximum |a|, produced by the kernels that wrote A); both un-scalings
// are one ldexp in the epilogue.  Handles a ragged M itself (rows past M re-read the last row, their stores / atomics are
// dropped by the buffer range check).
constexpr int kHnSlab = 2 * kNlPiece;                    // 36 864 B per K chunk (two fp16 pieces of all 128 columns)

// ACC: 0 = C = A B, 1 = C += A B by one float atomic per element, 2 = C += A B by a plain read - add - store of the tile (every element has
// exactly one writer either way, so both accumulate forms give the same bits: old + product, one rounding).  Round 4: the atomic form
// issues 64 atomics per wave at the end of every 256-row unit, all eight waves of the one workgroup a CU holds at once, and float
// atomics execute at the memory side at ~1.3 TB/s chip-wide, one 256-byte wave-instruction per ~50 ns per CU
// (MI355X_MICROARCH.md "Global float atomics"): 512 of them are ~25 us of a 75 us unit with nothing else running on the CU.  Form 2
// fetches the 64 old values in one batch when the last chunk's MFMAs are done (the A / slab prefetch registers are dead by then).
template <int ACC>
__global__ __launch_bounds__(kNlThreads, 1) void gemm_f16x2_n128_kernel(const GemmParams p, const float* row_max, const float* col_unscale,
                                                                        int64_t n_units) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kHnSlab];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int n_kc = p.K / kNlKC;
  const uint32_t c_off = (4u * h * (uint32_t)p.ldc + (uint32_t)r31) * 4u;
  const unsigned char* fb = lds + r31 * kNlPitch + h * 16;
  // slab staging: 256 rows (piece, column) x 8 chunks of 16 B = 2048 chunks, 4 per thread
  const int s_row = tid >> 3, s_kq = tid & 7;
  const _Float16* s_src = reinterpret_cast<const _Float16*>(p.Bt) + (size_t)s_row * p.K + s_kq * 8;
  unsigned char* s_dst = lds + s_row * kNlPitch + s_kq * 16;

  for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
    const int64_t row0 = u * kNlRows + wave * 32;
    const int64_t rows_here = min((int64_t)32, p.M - row0);          // <= 0: this wave has no row, but it still stages and syncs
    const int64_t arow_i = row0 + min((int64_t)r31, max(rows_here, (int64_t)1) - 1);
    const float* ap = p.A + min(arow_i, p.M - 1) * p.lda + 8 * h;
    // power-of-two row scale from the caller's bound on the row maximum
    const float rmax = row_max[min(arow_i, p.M - 1)];
    const int ex = (int)((__float_as_uint(rmax) >> 23) & 0xFF);
    const int sce = min(max(14 - (ex - 127), -126), 127);
    const float sc = __uint_as_float((uint32_t)(sce + 127) << 23);
    f32x16 acc[4], acl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acl[t][r] = 0.f; }
    uint4 s0 = *reinterpret_cast<const uint4*>(s_src), s1 = *reinterpret_cast<const uint4*>(s_src + (size_t)64 * p.K);
    uint4 s2 = *reinterpret_cast<const uint4*>(s_src + (size_t)128 * p.K), s3 = *reinterpret_cast<const uint4*>(s_src + (size_t)192 * p.K);
    float4 raw[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      raw[2 * ks] = *reinterpret_cast<const float4*>(ap + ks * 16);
      raw[2 * ks + 1] = *reinterpret_cast<const float4*>(ap + ks * 16 + 4);
    }
    __syncthreads();                                     // the previous unit's last chunk has been read by every wave
    *reinterpret_cast<uint4*>(s_dst) = s0;                          *reinterpret_cast<uint4*>(s_dst + 64 * kNlPitch) = s1;
    *reinterpret_cast<uint4*>(s_dst + 128 * kNlPitch) = s2;         *reinterpret_cast<uint4*>(s_dst + 192 * kNlPitch) = s3;
    __syncthreads();

    for (int kc = 0; kc < n_kc; ++kc) {
      f16x8 af[4][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float v[8] = {raw[2*ks].x, raw[2*ks].y, raw[2*ks].z, raw[2*ks].w, raw[2*ks+1].x, raw[2*ks+1].y, raw[2*ks+1].z, raw[2*ks+1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = v[i] * sc;
          const _Float16 hi = (_Float16)x;
          af[ks][0][i] = hi;
          af[ks][1][i] = (_Float16)((x - (float)hi) * 2048.f);
        }
      }
      const bool more = kc + 1 < n_kc;
      if (more) {                                         // next chunk: slab and A, in flight during this chunk's MFMAs
        const _Float16* q = s_src + (kc + 1) * kNlKC;
        s0 = *reinterpret_cast<const uint4*>(q);                           s1 = *reinterpret_cast<const uint4*>(q + (size_t)64 * p.K);
        s2 = *reinterpret_cast<const uint4*>(q + (size_t)128 * p.K);       s3 = *reinterpret_cast<const uint4*>(q + (size_t)192 * p.K);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          raw[2 * ks] = *reinterpret_cast<const float4*>(ap + (kc + 1) * kNlKC + ks * 16);
          raw[2 * ks + 1] = *reinterpret_cast<const float4*>(ap + (kc + 1) * kNlKC + ks * 16 + 4);
        }
      }
      const unsigned char* sb = fb + (kc & 1) * kHnSlab;
      HFrag cur;
      cur.b1 = *reinterpret_cast<const f16x8*>(sb);
      cur.b2 = *reinterpret_cast<const f16x8*>(sb + kNlPiece);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          HFrag nxt = cur;
          if (ks < 3 || ct < 3) {
            const unsigned char* q = sb + (ks < 3 ? ct : ct + 1) * 32 * kNlPitch + (ks < 3 ? ks + 1 : 0) * 32;
            nxt.b1 = *reinterpret_cast<const f16x8*>(q);
            nxt.b2 = *reinterpret_cast<const f16x8*>(q + kNlPiece);
          }
          __builtin_amdgcn_sched_barrier(0);
          acl[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b2, acl[ct], 0, 0, 0);
          acl[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][1], cur.b1, acl[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], cur.b1, acc[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          cur = nxt;
        }
      }
      if (more) {
        unsigned char* d = s_dst + ((kc + 1) & 1) * kHnSlab;
        *reinterpret_cast<uint4*>(d) = s0;                          *reinterpret_cast<uint4*>(d + 64 * kNlPitch) = s1;
        *reinterpret_cast<uint4*>(d + 128 * kNlPitch) = s2;         *reinterpret_cast<uint4*>(d + 192 * kNlPitch) = s3;
        __syncthreads();
      }
    }
    if (rows_here <= 0) continue;                         // (after the barriers)
    const __amdgpu_buffer_rsrc_t crow =
        __builtin_amdgcn_make_buffer_rsrc(p.C + row0 * p.ldc, 0, (int)min((int64_t)0x7fffffff, rows_here * p.ldc * 4), 0x00020000);
    if (ACC == 2) {
      float oldv[4][16];                                  // all 64 loads in flight before the first use
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          oldv[t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crow, c_off, (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + t * 32) * 4u, 0));
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cue = (int)((__float_as_uint(col_unscale[t * 32 + r31]) >> 23) & 0xFF) - 127;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rse = __shfl(sce, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
          const float val = ldexpf(acc[t][r] + acl[t][r] * (1.f / 2048.f), cue - rse);
          const uint32_t so = (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + t * 32) * 4u;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oldv[t][r] + val), crow, c_off, so, 0);
        }
      }
      continue;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int cue = (int)((__float_as_uint(col_unscale[t * 32 + r31]) >> 23) & 0xFF) - 127;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rse = __shfl(sce, (r & 3) + 8 * (r >> 2) + 4 * h, 64);          // once per 256 x 128 block: not worth 16 registers
        const float val = ldexpf(acc[t][r] + acl[t][r] * (1.f / 2048.f), cue - rse);
        const uint32_t so = (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + t * 32) * 4u;
        if (ACC == 1) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(val, crow, c_off, so, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), crow, c_off, so, 0);
      }
    }
  }
}

// ---- N = 32 NT (128 or 256), long K, three products, ONE accumulator set, two raw A chunks in flight (round 4) ---------------------------
// What held gemm_f16x2_n128_kernel at 0.56 of the HBM peak (round-4 measurements: the atomic epilogue was 3 % of it, not more): a wave
// had ONE 64-deep chunk of A in flight, issued after the split of the previous one and drained by the __syncthreads() at the end of every
// chunk - per chunk ~4 000 cycles of MFMA + split and ~5 800 of waiting, i.e. one exposed memory round trip per chunk, with 8 KB per
// wave in flight for about half of the time (Little: 23 GB/s per CU x 2.5 us needs ~60 KB per CU continuously).  This form:
//   * pieces hi = fp16(a s), lo = fp16(a s - hi), PLAIN (not pre-scaled by 2^11), for both operands: hi hi + hi lo + lo hi accumulate into
//     ONE fp32 tile (the TN kernel's numerics: 22 bits for every element within 2^16 of its row / column maximum, an absolute 2^-25 of the
//     scaled maximum below that) - 64 accumulator registers less than the two-set form;
//   * those registers hold a SECOND raw chunk: chunk c+1 is split during the first half of step c - in program order BETWEEN the MFMA
//     groups (a wave issues in order: VALU work hides in an MFMA's shadow only if it sits there) - and its registers are re-requested
//     with chunk c+3 at mid-step, so a chunk has a step and a half to arrive and the memory pipe never runs dry at a chunk boundary;
//   * raw barriers (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() also waits for vmcnt(0), i.e. for the prefetches just issued;
//     the slab registers are requested BEFORE the A chunk, so their wait (vmcnt(8)) leaves the A loads in flight;
//   * NT = 8: all 256 output columns of hidden width 256 (C5) in one pass over A - the two-set form needed two launches, each reading
//     the whole 17 GB of [gP|gQ].
// Bt2 = (2, 32 NT, K) fp16 with PLAIN lo pieces (mma_split_f16x2 with plain_lo = 1).  K % 128 == 0, K >= 256.
typedef float nlp_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int nlp_u4 __attribute__((ext_vector_type(4)));

template <int NT, int ACC>
__global__ __launch_bounds__(kNlThreads, 1) void gemm_f16x2_nlp_kernel(const GemmParams p, const float* row_max, const float* col_unscale,
                                                                       int64_t n_units) {
  constexpr int dbg = kAbl;
  constexpr bool PIPE = NT == 4;                           // two raw chunks in flight + the split under the MFMAs (NT = 8: registers for one)
  constexpr int kPieceB = 32 * NT * kNlPitch;              // one fp16 piece of all 32 NT columns, one K chunk
  constexpr int kSlabB = 2 * kPieceB;                      // 36 864 B (NT = 4) / 73 728 B (NT = 8) per K chunk
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kSlabB];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int n_kc = p.K / kNlKC;                            // even, >= 4 (checked by the launcher)
  const uint32_t c_off = (4u * h * (uint32_t)p.ldc + (uint32_t)r31) * 4u;
  const unsigned char* fb = lds + r31 * kNlPitch + h * 16;
  // slab staging: 64 NT rows (piece, column) x 8 chunks of 16 B = 512 NT chunks, NT per thread; row = (tid >> 3) + 64 i is also the row of Bt2.
  // Every global address is a wave-uniform base (SGPRs) + ONE 32-bit lane offset: 64-bit pointers per load cost the kernel ~20 VGPRs.
  const int s_row = tid >> 3, s_kq = tid & 7;
  const uint32_t s_voff = ((uint32_t)s_row * (uint32_t)p.K + (uint32_t)s_kq * 8u) * 2u;                  // bytes into Bt2
  const _Float16* Bh = reinterpret_cast<const _Float16*>(p.Bt);
  unsigned char* s_dst = lds + s_row * kNlPitch + s_kq * 16;

  for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
    const int64_t row0 = u * kNlRows + wave * 32;
    const int64_t rows_here = min((int64_t)32, p.M - row0);          // <= 0: this wave has no row, but it still stages and syncs
    const int64_t base_row = min(row0, p.M - 1);                     // wave-uniform; rows past M re-read the last row, never stored
    const int drow = (int)(min(row0 + min((int64_t)r31, max(rows_here, (int64_t)1) - 1), p.M - 1) - base_row);       // 0 .. 31
    const uint32_t a_voff = ((uint32_t)drow * (uint32_t)p.lda + 8u * h) * 4u;                             // bytes from the wave's base row
    const uint32_t a_voff_rm = ((uint32_t)(lane >> 4) * (uint32_t)p.lda + 4u * (lane & 15)) * 4u;         // (measurement build 128)
    const float* abase = p.A + base_row * p.lda;                     // + chunk * 64 floats
    const float rmax = row_max[base_row + drow];
    const int ex = (int)((__float_as_uint(rmax) >> 23) & 0xFF);
    const int sce = min(max(14 - (ex - 127), -126), 127);
    const float sc = __uint_as_float((uint32_t)(sce + 127) << 23);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    nlp_u4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7;           // named scalars: an indexed array of staging registers ends up in scratch
    nlp_f4 rawA[8], rawB[8];
    f16x8 afP[4][2], afQ[4][2];
    if (dbg || NT <= 4) { sr4 = sr5 = sr6 = sr7 = nlp_u4{0, 0, 0, 0}; }
    if (dbg || !PIPE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rawB[i] = nlp_f4{1.f, 1.f, 1.f, 1.f};
    }
    if (dbg) {                                                // measurement builds skip loads / splits: defined values everywhere
#pragma unroll
      for (int i = 0; i < 8; ++i) rawA[i] = nlp_f4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 8; ++i) { afP[ks][q][i] = (_Float16)1.f; afQ[ks][q][i] = (_Float16)1.f; }
      sr0 = sr1 = sr2 = sr3 = nlp_u4{0, 0, 0, 0};
    }

// All global loads are INLINE ASM and every wait is written by hand (cdna_hip_programming.md 5.7).  hipcc's own bookkeeping could not
// express this pipeline: it merged the loop's entry and back edge into a full vmcnt(4) .. vmcnt(0) drain at the loop head, and at 256
// registers it spilled a register it had just loaded - a vmcnt(0) in the middle of a step (both seen in the ISA; 1.55 ms).  vmcnt is one
// in-order counter: a wait for an operation is a wait for everything older.  Order inside a step: slab (NT loads) | at mid-step: A chunk
// (8 loads) | at the end vmcnt(8): the slab has landed, and with it every OLDER load - the raw set requested in the middle of the
// previous step, which the next step splits.  Each wait statement names the registers it releases ("+v"), so the compiler can neither
// read them earlier nor reuse them while a load is in flight; tools/check_asm_waits.py audits the ISA for copies in between.
// One asm statement per load group.  `s_nop 4` opens it: the base is an SGPR pair the compiler may have just restored from a spill lane
// with v_readlane - a VALU write of an SGPR needs 5 wait states before a vector-memory instruction reads it, and hipcc pads nothing
// inside an asm string (5.7 item 2; without the pad the first build of this kernel read a stale base: a memory access fault).  Outputs are
// early-clobber: the statement writes its first destination before it has read the offset register for the last time.
#define MMA_NLP_LOADA(R_, C_)                                                                      \
    if (dbg & 128) {   /* measurement: the chunk as 8 ROW-MAJOR loads (4 whole 256-byte row segments each), wrong lane layout */      \
      const float* ab_ = abase + (int64_t)(C_) * kNlKC;                                            \
      const float *b0_ = ab_, *b1_ = ab_ + 4 * p.lda, *b2_ = ab_ + 8 * p.lda, *b3_ = ab_ + 12 * p.lda,                          \
                  *b4_ = ab_ + 16 * p.lda, *b5_ = ab_ + 20 * p.lda, *b6_ = ab_ + 24 * p.lda, *b7_ = ab_ + 28 * p.lda;          \
      asm volatile("s_nop 4\n\t"                                                                   \
                   "global_load_dwordx4 %0, %8, %9\n\t"  "global_load_dwordx4 %1, %8, %10\n\t"     \
                   "global_load_dwordx4 %2, %8, %11\n\t" "global_load_dwordx4 %3, %8, %12\n\t"     \
                   "global_load_dwordx4 %4, %8, %13\n\t" "global_load_dwordx4 %5, %8, %14\n\t"     \
                   "global_load_dwordx4 %6, %8, %15\n\t" "global_load_dwordx4 %7, %8, %16"          \
                   : "=&v"(R_[0]), "=&v"(R_[1]), "=&v"(R_[2]), "=&v"(R_[3]), "=&v"(R_[4]), "=&v"(R_[5]), "=&v"(R_[6]), "=&v"(R_[7])  \
                   : "v"(a_voff_rm), "s"(b0_), "s"(b1_), "s"(b2_), "s"(b3_), "s"(b4_), "s"(b5_), "s"(b6_), "s"(b7_) : "memory");    \
    } else if (!(dbg & 8)) {                                                                       \
      const float* ab_ = abase + (int64_t)(C_) * kNlKC;                                            \
      asm volatile("s_nop 4\n\t"                                                                   \
                   "global_load_dwordx4 %0, %8, %9 offset:0\n\t"   "global_load_dwordx4 %1, %8, %9 offset:16\n\t"   \
                   "global_load_dwordx4 %2, %8, %9 offset:64\n\t"  "global_load_dwordx4 %3, %8, %9 offset:80\n\t"   \
                   "global_load_dwordx4 %4, %8, %9 offset:128\n\t" "global_load_dwordx4 %5, %8, %9 offset:144\n\t"  \
                   "global_load_dwordx4 %6, %8, %9 offset:192\n\t" "global_load_dwordx4 %7, %8, %9 offset:208"      \
                   : "=&v"(R_[0]), "=&v"(R_[1]), "=&v"(R_[2]), "=&v"(R_[3]), "=&v"(R_[4]), "=&v"(R_[5]), "=&v"(R_[6]), "=&v"(R_[7])  \
                   : "v"(a_voff), "s"(ab_) : "memory");                                            \
    }
#define MMA_NLP_SLABLOAD(C_)                                                                       \
    if (!(dbg & 32)) {                                                                             \
      const _Float16* sb_ = Bh + (int64_t)(C_) * kNlKC;                                            \
      const int64_t sk_ = (int64_t)64 * p.K;                                                       \
      const _Float16 *q0_ = sb_, *q1_ = sb_ + sk_, *q2_ = sb_ + 2 * sk_, *q3_ = sb_ + 3 * sk_;     \
      asm volatile("s_nop 4\n\t"                                                                   \
                   "global_load_dwordx4 %0, %4, %5\n\t" "global_load_dwordx4 %1, %4, %6\n\t"       \
                   "global_load_dwordx4 %2, %4, %7\n\t" "global_load_dwordx4 %3, %4, %8"           \
                   : "=&v"(sr0), "=&v"(sr1), "=&v"(sr2), "=&v"(sr3) : "v"(s_voff), "s"(q0_), "s"(q1_), "s"(q2_), "s"(q3_) : "memory");  \
      if (NT > 4) {                                                                                \
        const _Float16 *q4_ = sb_ + 4 * sk_, *q5_ = sb_ + 5 * sk_, *q6_ = sb_ + 6 * sk_, *q7_ = sb_ + 7 * sk_;      \
        asm volatile("s_nop 4\n\t"                                                                 \
                     "global_load_dwordx4 %0, %4, %5\n\t" "global_load_dwordx4 %1, %4, %6\n\t"     \
                     "global_load_dwordx4 %2, %4, %7\n\t" "global_load_dwordx4 %3, %4, %8"         \
                     : "=&v"(sr4), "=&v"(sr5), "=&v"(sr6), "=&v"(sr7) : "v"(s_voff), "s"(q4_), "s"(q5_), "s"(q6_), "s"(q7_) : "memory"); \
      }                                                                                            \
    }
// wait until at most N_ younger loads are in flight; releases the slab registers and the raw set R_
#define MMA_NLP_WAIT(N_, R_)                                                                       \
    if (!(dbg & 8) || !(dbg & 32)) {                                                               \
      if (NT > 4)                                                                                  \
        asm volatile("s_waitcnt vmcnt(" #N_ ")" : "+v"(sr0), "+v"(sr1), "+v"(sr2), "+v"(sr3), "+v"(sr4), "+v"(sr5), "+v"(sr6), "+v"(sr7),  \
                     "+v"(R_[0]), "+v"(R_[1]), "+v"(R_[2]), "+v"(R_[3]), "+v"(R_[4]), "+v"(R_[5]), "+v"(R_[6]), "+v"(R_[7]) :: "memory"); \
      else                                                                                         \
        asm volatile("s_waitcnt vmcnt(" #N_ ")" : "+v"(sr0), "+v"(sr1), "+v"(sr2), "+v"(sr3),      \
                     "+v"(R_[0]), "+v"(R_[1]), "+v"(R_[2]), "+v"(R_[3]), "+v"(R_[4]), "+v"(R_[5]), "+v"(R_[6]), "+v"(R_[7]) :: "memory"); \
    }
#define MMA_NLP_SS1(I_, B_) *reinterpret_cast<nlp_u4*>(s_dst + (B_) * kSlabB + 64 * (I_) * kNlPitch)
#define MMA_NLP_SLABSTORE(B_)                                                                      \
    MMA_NLP_SS1(0, B_) = sr0; MMA_NLP_SS1(1, B_) = sr1; MMA_NLP_SS1(2, B_) = sr2; MMA_NLP_SS1(3, B_) = sr3;   \
    if (NT > 4) { MMA_NLP_SS1(4, B_) = sr4; MMA_NLP_SS1(5, B_) = sr5; MMA_NLP_SS1(6, B_) = sr6; MMA_NLP_SS1(7, B_) = sr7; }
// value V_ (0..31) of raw chunk R_ -> pieces of AF_: k-step V_ / 8, element V_ % 8 (vector V_ / 4, component V_ % 4)
#define MMA_NLP_SPLIT1(R_, AF_, V_)                                                                \
    if (!(dbg & 16)) {                                                                             \
      const float xv = R_[(V_) >> 2][(V_) & 3] * sc;                                               \
      const _Float16 hi = (_Float16)xv;                                                            \
      AF_[(V_) >> 3][0][(V_) & 7] = hi;                                                            \
      AF_[(V_) >> 3][1][(V_) & 7] = (_Float16)(xv - (float)hi);                                    \
    }
// One K chunk: 4 NT groups of three MFMAs (column tile x k-step); fragments of group g+1 are read while group g multiplies.
//   SLAB_ : stage the NEXT chunk's B slab - requested first, stored last (behind WAITN_), then the raw barrier;
//   PIPE  : the raw set RS_ (chunk C_+1, released by the previous step's wait) is split into AFN_ during the FIRST half of the step, in
//           program order BETWEEN the MFMA groups (a wave issues in order: VALU work hides in an MFMA's shadow only if it sits there),
//           and re-requested (LOADA_: chunk C_+3) at mid-step: a chunk has a step and a half to arrive.  A CU's fair share of the HBM
//           rate moves the 64 KB its eight waves ask for per step in about a step's time - one step of slack was not enough (measured:
//           with the loads issued but never waited for the kernel takes 0.69 ms, waiting for them a step later added 0.67);
//   !PIPE : LOADA_ requests chunk C_+1 before the MFMAs, the caller splits it after the barrier;
//   WAITN_: 8 when this step requested an A chunk (those 8 loads stay in flight), else 0;  RW_: the raw set the wait releases.
#define MMA_NLP_STEP(C_, B_, AFC_, AFN_, RS_, RW_, SLAB_, LOADA_, SPLIT_, WAITN_)                  \
    {                                                                                              \
      if (SLAB_) { MMA_NLP_SLABLOAD((C_) + 1) }                                                    \
      if (LOADA_ && !PIPE) { MMA_NLP_LOADA(RS_, (C_) + 1) }                                        \
      const unsigned char* sb = fb + (B_) * kSlabB;                                                \
      HFrag cur;                                                                                   \
      cur.b1 = *reinterpret_cast<const f16x8*>(sb);                                                \
      cur.b2 = *reinterpret_cast<const f16x8*>(sb + kPieceB);                                      \
      _Pragma("unroll") for (int g = 0; g < 4 * NT; ++g) {                                         \
        const int ct = g >> 2, ks = g & 3;                                                         \
        HFrag nxt = cur;                                                                           \
        if (g + 1 < 4 * NT) {                                                                      \
          const unsigned char* q = sb + ((g + 1) >> 2) * 32 * kNlPitch + ((g + 1) & 3) * 32;       \
          nxt.b1 = *reinterpret_cast<const f16x8*>(q);                                             \
          nxt.b2 = *reinterpret_cast<const f16x8*>(q + kPieceB);                                   \
        }                                                                                          \
        if (PIPE && LOADA_ && g == 2 * NT) { MMA_NLP_LOADA(RS_, (C_) + 3) }                        \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (!(dbg & 2)) {                                                                          \
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AFC_[ks][1], cur.b1, acc[ct], 0, 0, 0);   \
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AFC_[ks][0], cur.b2, acc[ct], 0, 0, 0);   \
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AFC_[ks][0], cur.b1, acc[ct], 0, 0, 0);   \
        } else { asm volatile("" :: "v"(AFC_[ks][0]), "v"(AFC_[ks][1]), "v"(cur.b1), "v"(cur.b2)); } \
        if (PIPE && SPLIT_ && g < 2 * NT) {                                                        \
          constexpr int per = 32 / (2 * NT);                /* 4 values per group (NT = 4) */      \
          _Pragma("unroll") for (int j = 0; j < per; ++j) MMA_NLP_SPLIT1(RS_, AFN_, g * per + j)   \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        cur = nxt;                                                                                 \
      }                                                                                            \
      if (SLAB_) {                                                                                 \
        MMA_NLP_WAIT(WAITN_, RW_)                                                                  \
        MMA_NLP_SLABSTORE(1 - (B_))                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        __builtin_amdgcn_s_barrier();                                                              \
        asm volatile("" ::: "memory");                                                             \
      }                                                                                            \
    }

    MMA_NLP_SLABLOAD(0)
    MMA_NLP_LOADA(rawA, 0)
    if (PIPE) {
      MMA_NLP_LOADA(rawB, 1)
      MMA_NLP_WAIT(8, rawA)                                   // slab 0 and chunk 0 have landed, chunk 1 may still fly
    } else {
      MMA_NLP_WAIT(0, rawA)
    }
    // every wave has left the previous unit's last chunk (buffer 1) behind the barrier below; buffer 0 was last read one step earlier
    MMA_NLP_SLABSTORE(0)
#pragma unroll
    for (int v = 0; v < 32; ++v) MMA_NLP_SPLIT1(rawA, afP, v)
    if (PIPE) {
      MMA_NLP_LOADA(rawA, 2)                                  // chunk 2 into the set chunk 0 has just left
      MMA_NLP_WAIT(8, rawB)                                   // chunk 1 (split by step 0) has landed; chunk 2 flies
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if (PIPE) {
      // step c: multiply chunk c (afP / afQ in turn); split chunk c+1 out of one raw set in the first half and re-request that set with
      // chunk c+3 at mid-step; the OTHER set holds chunk c+2, under way since the middle of the previous step and released by this
      // step's closing wait
      for (int kc = 0; kc < n_kc - 4; kc += 2) {
        MMA_NLP_STEP(kc, 0, afP, afQ, rawB, rawA, true, true, true, 8)
        MMA_NLP_STEP(kc + 1, 1, afQ, afP, rawA, rawB, true, true, true, 8)
      }
      MMA_NLP_STEP(n_kc - 4, 0, afP, afQ, rawB, rawA, true, true, true, 8)
      MMA_NLP_STEP(n_kc - 3, 1, afQ, afP, rawA, rawB, true, false, true, 0)
      MMA_NLP_STEP(n_kc - 2, 0, afP, afQ, rawB, rawA, true, false, true, 0)
      MMA_NLP_STEP(n_kc - 1, 1, afQ, afP, rawA, rawB, false, false, false, 0)
    } else {
      // NT = 8: 128 accumulator registers leave room for ONE raw chunk - requested before the step's MFMAs, split after its barrier
      for (int kc = 0; kc < n_kc - 1; ++kc) {
        MMA_NLP_STEP(kc, kc & 1, afP, afP, rawA, rawA, true, true, false, 0)
#pragma unroll
        for (int v = 0; v < 32; ++v) MMA_NLP_SPLIT1(rawA, afP, v)
      }
      MMA_NLP_STEP(n_kc - 1, (n_kc - 1) & 1, afP, afP, rawA, rawA, false, false, false, 0)
    }
#undef MMA_NLP_STEP
#undef MMA_NLP_SPLIT1
#undef MMA_NLP_SLABSTORE
#undef MMA_NLP_SS1
#undef MMA_NLP_WAIT
#undef MMA_NLP_SLABLOAD
#undef MMA_NLP_LOADA
    // the last chunk was read from buffer 1; the next unit's prologue writes buffer 0 (last read before the previous barrier) and
    // then meets everybody at its own barrier before buffer 1 is written again

    if (rows_here <= 0) continue;
    const __amdgpu_buffer_rsrc_t crow =
        __builtin_amdgcn_make_buffer_rsrc(p.C + row0 * p.ldc, 0, (int)min((int64_t)0x7fffffff, rows_here * p.ldc * 4), 0x00020000);
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 4) {                  // four tiles at a time: 64 old values in flight (the raw / fragment registers are dead)
      float oldv[4][16];
      if (ACC == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            oldv[t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crow, c_off, (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + (t0 + t) * 32) * 4u, 0));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cue = (int)((__float_as_uint(col_unscale[(t0 + t) * 32 + r31]) >> 23) & 0xFF) - 127;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rse = __shfl(sce, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
          const float val = ldexpf(acc[t0 + t][r], cue - rse);
          const uint32_t so = (uint32_t)(((r & 3) + 8 * (r >> 2)) * (int)p.ldc + (t0 + t) * 32) * 4u;
          if (ACC == 1) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(val, crow, c_off, so, 0);
          else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ACC == 2 ? oldv[t][r] + val : val), crow, c_off, so, 0);
        }
      }
    }
  }
}

// split a row-major fp32 matrix (rows, cols) into its three bf16 pieces: out (3, rows, cols)
// B operand of the three-product kernels in ONE launch: column n of w (K,N) -> power-of-two scale that puts max_k |w[k,n]| into
// [2^14, 2^15), pieces hi = fp16(w s), lo = fp16((w s - hi) 2^11) at [piece][n][k] (k contiguous), col_unscale[n] = 1/s.
// (The torch expression of the same thing was twelve small launches per GEMM call.)  One workgroup per column, any strides.
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* w, int64_t stride_k, int64_t stride_n, int K, int N, _Float16* bt2,
                                                          float* col_unscale, int plain_lo) {
  __shared__ uint32_t red[4];
  const int n = blockIdx.x;
  uint32_t m = 0;
  for (int k = threadIdx.x; k < K; k += 256) m = max(m, __float_as_uint(w[k * stride_k + n * stride_n]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = max(max(red[0], red[1]), max(red[2], red[3]));
  const float amax = fmaxf(__uint_as_float(m), 1e-30f);                     // an all-zero column: any scale will do
  const int e = min((int)((__float_as_uint(amax) >> 23) & 0xFF) - 127, 113);  // inf / NaN columns keep a finite scale
  const float sc = __uint_as_float((uint32_t)(14 - e + 127) << 23);
  for (int k = threadIdx.x; k < K; k += 256) {
    const float x = w[k * stride_k + n * stride_n] * sc;
    const _Float16 hi = (_Float16)x;
    bt2[(size_t)n * K + k] = hi;
    bt2[((size_t)N + n) * K + k] = (_Float16)(plain_lo ? x - (float)hi : (x - (float)hi) * 2048.f);
  }
  if (threadIdx.x == 0) col_unscale[n] = __uint_as_float((uint32_t)(e - 14 + 127) << 23);
}

__global__ void split3_kernel(const float* in, int64_t n, __bf16* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const Bf3 t = split3(in[i]);
    out[i] = t.a; out[n + i] = t.b; out[2 * n + i] = t.c;
  }
}


// ---- TN form: C (KA, NC) = X^T G for X (M, KA), G (M, NC), M >> KA, NC: the weight gradients x^T [gP | gQ] -----------
// Both operands are fp32 activations whose reduction index is the ROW, and the MFMA wants 8 consecutive k per lane.
// No transpose is needed for that: a lane simply loads its 8 k values from 8 different rows (each load instruction
// still reads 128 contiguous bytes of one row across 32 lanes).  X fragments go straight to registers (wave w owns the
// X columns [32w, 32w+32)); the G tile (32 rows x 128 columns per chunk) is split once per workgroup and staged
// k-contiguous through a double-buffered LDS slab, so its 6.5 VALU ops per value are paid once, not per wave.
// The reduction over M is cut into `splits` row ranges whose partial tiles are summed by the K8 kernel in a fixed order:
// no atomics, bitwise repeatable.
// Measured (C4: M = 2^20, KA = 128, NC = 1024): 1.70 ms vs 2.1 ms for the split-K fp32 library GEMM; HBM traffic 4.9 GB
// (PMC; G once + X once thanks to the XCD-local ids), matrix pipe 53 % busy.  Ablations: MFMAs + fragment reads alone
// 1.1 ms, loads + split alone 0.9 ms.  Tried and not faster: 8-wave workgroups whose two halves alternate multiply /
// split roles (2.2 ms; LDS-bound once X is staged as well), 64 x 64 wave tiles with both operands through LDS and
// dwordx4 loads (1.8 ms), dwordx2 G loads with interleaved column tiles (half the G load instructions: no change),
// -fno-slp-vectorize.  All three bf16x3 kernels of this file land at ~1.7 ms per 275 GFLOP:
// the per-value split (6.5 VALU ops, 4 issue cycles each, against 24 free issue cycles per 32-cycle MFMA) is what the
// three-piece scheme costs on top of its 6x MFMA count.
struct TnParams {
  const float* X; int64_t ldx; const float* G; int64_t ldg;
  float* part;                 // (splits, KA, NC)
  int64_t M, rows_per_split; int KA, NC, splits;
  // three-product (fp16 x 2) form only - zero / NULL for the six-product kernel called on its own
  const uint16_t* sxh;         // (M,) upper 16 bits of the power-of-two fp32 scale of each X row
  const uint16_t* sgh;         // (M,) the same for the G rows
  const int* state;            // {max t, min t, bad, unscale exponent}: see tn_scale_* below.  state != NULL: the kernel runs only
  int want_bad;                //   if (state[2] != 0) == want_bad (the two forms are launched back to back, one of them leaves at once)
  // batched form (blockIdx.y = b): X + b*xb and G + b*gb (column offsets inside wider rows: the towers of MMAConv), partial tiles
  // (splits, batch, KA, NC); zero for a single product
  int64_t xb, gb, part_ss;
  // [r5] three-product form, X PACKED (tn_pack_x_kernel): per 32-row chunk and 32-column block of X the two k-steps x two fp16 pieces in
  // the MFMA's fragment order (4 KB); NULL = the kernel loads and splits the fp32 rows itself
  const void* xp; int xp_nb;
};

constexpr int kTnKC = 32;                        // rows of X / G per chunk
constexpr int kTnPitch = kTnKC * 2 + 16;         // bytes per (piece, column) LDS row: 32 bf16 + 16 B pad
constexpr int kTnPiece = 128 * kTnPitch;
constexpr int kTnSlab = 3 * kTnPiece;            // 30 KB; double buffered => two workgroups per CU

__device__ __forceinline__ void split3x8(const float (&v)[16], int o, bf16x8& a, bf16x8& b, bf16x8& c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const Bf3 t = split3(v[o + i]);
    a[i] = t.a; b[i] = t.b; c[i] = t.c;
  }
}

// FULLCT: all four 32-column tiles of the block exist and are whole (NC % 128 == 0, KA % 32 == 0): the MFMA phase is then one
// branch-free block and the stores need no guards.  Otherwise any KA <= 128 and any NC: ragged tiles load clamped columns and
// store under a guard.
template <bool FULLCT>
__global__ __launch_bounds__(kBlock, 2) void gemm_x3_tn_kernel(const TnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kTnSlab];
  if (p.state && (p.state[2] != 0) != (p.want_bad != 0)) return;   // the three-product form does this call (uniform: before any barrier)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r31 = lane & 31, h = lane >> 5;
  // Workgroups are dealt round-robin to the 8 XCDs (private L2 each).  The column blocks of one row range all read the same
  // X rows, so they are given ids that land on ONE XCD back to back: X then comes from HBM once, not once per column block.
  const int n_cb = (p.NC + 127) / 128;
  int cb, split;
  if (p.splits % 8 == 0) {
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    cb = slot % n_cb;
    split = (slot / n_cb) * 8 + xcd;
  } else {
    cb = (int)blockIdx.x % n_cb;
    split = (int)blockIdx.x / n_cb;
  }
  const int gcol0 = cb * 128;
  const int n_ct = FULLCT ? 4 : min(4, (p.NC - gcol0 + 31) / 32);     // the last tile may be ragged (stores are guarded)
  const int64_t r0 = (int64_t)split * p.rows_per_split;
  const int64_t r1 = min(p.M, r0 + p.rows_per_split);
  const int n_chunks = r1 > r0 ? (int)((r1 - r0 + kTnKC - 1) / kTnKC) : 0;
  const bool wave_active = wave * 32 < p.KA;
  // staging role: column sn of the block's 128, k-groups skg and skg + 2 (8 rows each)
  const int sn = tid & 127, skg = tid >> 7;
  const int scol = gcol0 + sn < p.NC ? sn : 0;       // columns past NC belong to tiles that are never multiplied
  const int xcol = min(wave * 32 + r31, p.KA - 1);   // columns past KA repeat the last one: they only reach rows that are never stored
  const int ldg_b = (int)p.ldg * 4, ldx_b = (int)p.ldx * 4;                    // row pitches in bytes
  const int rows = (int)(r1 > r0 ? r1 - r0 : 0);
  const int64_t rbase = r1 > r0 ? r0 : 0;                                      // an empty split still gets a valid base
  const int64_t bz = (int64_t)blockIdx.y;                                      // batch index (0 for a single product)
  const auto g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G + bz * p.gb + rbase * p.ldg + gcol0), 0,
                                                        rows ? (rows - 1) * ldg_b + min(128, p.NC - gcol0) * 4 : 0, 0x00020000);
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + bz * p.xb + rbase * p.ldx), 0,
                                                        rows ? (rows - 1) * ldx_b + p.KA * 4 : 0, 0x00020000);
  const int g_voff = skg * 8 * ldg_b + scol * 4;     // per lane: first row of its k-group, its column
  const int x_voff = 8 * h * ldx_b + xcol * 4;

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // two raw (fp32) chunk sets: chunk c+1 waits to be split while chunk c+2 is still on its way from HBM (one MFMA phase,
  // ~0.8 us, is shorter than the loaded memory latency; with a single set every wave sat idle for two thirds of a chunk)
  float graw0[16], xraw0[16], graw1[16], xraw1[16];
  bf16x8 af[2][3];

// Buffer loads: one descriptor per operand covering exactly this split's rows (wave-uniform base and size), a 32-bit
// per-lane offset and no address arithmetic in VGPR pairs; rows past the end of the split are out of range and read as 0
// (hardware range check on voffset), so the ragged last chunk needs no guard.
#define MMA_TN_LOAD(C_, GR_, XR_)                                                              \
  {                                                                                            \
    int gv = g_voff + (C_) * (kTnKC * ldg_b);                                                  \
    int xv = x_voff + (C_) * (kTnKC * ldx_b);                                                  \
    /* opaque: otherwise the 32 loop-invariant sums voff + row*pitch are hoisted into 32 VGPRs (and spilled) */ \
    asm volatile("" : "+v"(gv), "+v"(xv));                                                     \
    _Pragma("unroll") for (int g = 0; g < 2; ++g)                                              \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
        GR_[g * 8 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(g_rsrc, gv + (g * 16 + i) * ldg_b, 0, 0)); \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
        XR_[ks * 8 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, xv + (ks * 16 + i) * ldx_b, 0, 0)); \
  }
// split a raw chunk: G pieces into LDS buffer B_ ([piece][column][k], k contiguous), X pieces into af
#define MMA_TN_PUBLISH(B_, GR_, XR_)                                                           \
  {                                                                                            \
    unsigned char* d = lds + (B_) * kTnSlab + sn * kTnPitch;                                   \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                            \
      bf16x8 a, b, c;                                                                          \
      split3x8(GR_, g * 8, a, b, c);                                                           \
      *reinterpret_cast<bf16x8*>(d + 0 * kTnPiece + (skg + 2 * g) * 16) = a;                   \
      *reinterpret_cast<bf16x8*>(d + 1 * kTnPiece + (skg + 2 * g) * 16) = b;                   \
      *reinterpret_cast<bf16x8*>(d + 2 * kTnPiece + (skg + 2 * g) * 16) = c;                   \
    }                                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) split3x8(XR_, ks * 8, af[ks][0], af[ks][1], af[ks][2]); \
  }
// One chunk.  Its four groups of twelve MFMAs (k-step x pair of column tiles) are interleaved IN PROGRAM ORDER with the split of the
// next chunk's G values (a wave issues in order: VALU work only hides in an MFMA's 32-cycle shadow if it sits between the
// MFMAs), the B fragments of group m+1 are read while group m multiplies, and every group is fenced so the scheduler keeps
// that order.  Then: next chunk's X pieces into af | refill the raw set with chunk C_+3 | raw barrier (__syncthreads()
// would also drain vmcnt, i.e. wait for the prefetches that were just issued).  The publish goes to the other LDS buffer,
// last read before the previous barrier.  Everything is unconditional: chunks past the end of the split are out of the
// descriptors' range and read as zeros (a conditional prefetch also makes the compiler's vmcnt bookkeeping fall back to
// "wait for everything"), and waves beyond KA multiply clamped columns that are never stored.
#define MMA_TN_STEP(C_, B_, GR_, XR_)                                                          \
  {                                                                                            \
    const unsigned char* sb = lds + (B_) * kTnSlab + r31 * kTnPitch + h * 16;                  \
    unsigned char* pd = lds + (1 - (B_)) * kTnSlab + sn * kTnPitch + skg * 16;                 \
    bf16x8 bq[2][2][3], ga, gb, gc;      /* [parity][tile of the pair][piece] */               \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                              \
      _Pragma("unroll") for (int q = 0; q < 3; ++q)                                            \
        bq[0][t][q] = *reinterpret_cast<const bf16x8*>(sb + t * 32 * kTnPitch + q * kTnPiece); \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) {       /* group m: k-step m>>1, column tiles 2*(m&1), 2*(m&1)+1 */ \
      const int ks = m >> 1, c0 = 2 * (m & 1), c1 = c0 + 1, pa = m & 1;                        \
      if (m + 1 < 4) {                                                                         \
        const unsigned char* qn = sb + (2 * ((m + 1) & 1)) * 32 * kTnPitch + ((m + 1) >> 1) * 32; \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                          \
          _Pragma("unroll") for (int q = 0; q < 3; ++q)                                        \
            bq[1 - pa][t][q] = *reinterpret_cast<const bf16x8*>(qn + t * 32 * kTnPitch + q * kTnPiece); \
      }                                                                                        \
      /* the two tiles' chains alternate, so consecutive MFMAs never depend on each other */   \
      const bool t0 = FULLCT || c0 < n_ct, t1 = FULLCT || c1 < n_ct;                           \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], bq[pa][0][2], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], bq[pa][1][2], acc[c1], 0, 0, 0); \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], bq[pa][0][1], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], bq[pa][1][1], acc[c1], 0, 0, 0); \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][2], bq[pa][0][0], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][2], bq[pa][1][0], acc[c1], 0, 0, 0); \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], bq[pa][0][1], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], bq[pa][1][1], acc[c1], 0, 0, 0); \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], bq[pa][0][0], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], bq[pa][1][0], acc[c1], 0, 0, 0); \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], bq[pa][0][0], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], bq[pa][1][0], acc[c1], 0, 0, 0); \
      _Pragma("unroll") for (int j = 4 * m; j < 4 * m + 4; j += 2) { /* values j, j+1 of the next chunk's G set */ \
        const Bf3 u0 = split3(GR_[j]), u1 = split3(GR_[j + 1]);                                \
        ga[j & 7] = u0.a; gb[j & 7] = u0.b; gc[j & 7] = u0.c;                                  \
        ga[(j + 1) & 7] = u1.a; gb[(j + 1) & 7] = u1.b; gc[(j + 1) & 7] = u1.c;                \
      }                                                                                        \
      if (m & 1) { /* a k-group of 8 rows is complete: [piece][column][k] rows, k contiguous */ \
        *reinterpret_cast<bf16x8*>(pd + 0 * kTnPiece + (m >> 1) * 32) = ga;                    \
        *reinterpret_cast<bf16x8*>(pd + 1 * kTnPiece + (m >> 1) * 32) = gb;                    \
        *reinterpret_cast<bf16x8*>(pd + 2 * kTnPiece + (m >> 1) * 32) = gc;                    \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }                                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) split3x8(XR_, ks * 8, af[ks][0], af[ks][1], af[ks][2]); \
    MMA_TN_LOAD((C_) + 3, GR_, XR_)                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_s_barrier();                                                              \
    asm volatile("" ::: "memory");                                                             \
  }

  MMA_TN_LOAD(0, graw0, xraw0)
  MMA_TN_PUBLISH(0, graw0, xraw0)
  MMA_TN_LOAD(1, graw1, xraw1)
  MMA_TN_LOAD(2, graw0, xraw0)
  __syncthreads();
  for (int c = 0; c < n_chunks; c += 2) {                      // an odd count runs one all-zero chunk at the end
    MMA_TN_STEP(c, 0, graw1, xraw1)                            // even chunk: buffer 0; chunk c+1 waits in raw set 1
    MMA_TN_STEP(c + 1, 1, graw0, xraw0)
  }
#undef MMA_TN_STEP
#undef MMA_TN_LOAD
#undef MMA_TN_PUBLISH

  if (wave_active) {   // acc reg r holds X column 32*wave + (r&3) + 8*(r>>2) + 4*h, G column gcol0 + 32*ct + r31
    float* out = p.part + (size_t)split * (p.part_ss ? (size_t)p.part_ss : (size_t)p.KA * p.NC) + (size_t)bz * p.KA * p.NC +
                 (size_t)wave * 32 * p.NC + gcol0 + r31;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      if (ct < n_ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (FULLCT || (wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < p.KA && gcol0 + ct * 32 + r31 < p.NC))
            out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * p.NC + ct * 32] = acc[ct][r];
      }
    }
  }
}

// ---- TN form, three products (round 2): fp16 x 2 pieces, one power-of-two scale per ROW of each operand -------------------------
// The reduction index of x^T g is the row, so a row scale does not factor out of the sum - but a power of two can be MOVED between
// the two operands of a row for free: (x_i 2^a)(g_i 2^-a) = x_i g_i.  With ex_i / eg_i the exponents of the row maxima, t_i = ex_i +
// eg_i the size of row i's products, T = max t_i and r_i = T - t_i, the rows are scaled to
//     x~_i = x_i 2^(14 - ex_i - floor(r_i/2)),     g~_i = g_i 2^(14 - eg_i - ceil(r_i/2)),      x~_i g~_i = x_i g_i 2^(28 - T):
// ONE un-scaling 2^(T-28) for the whole product (applied in the epilogue), the largest rows have their maxima in [2^14, 2^15) in
// BOTH operands, and a row whose products are 2^-r of the largest ones gives up r/2 binades per operand, not r.  Pieces: hi =
// fp16(x~), lo = fp16(x~ - hi) (plain, not pre-scaled: one accumulator set - the 64 registers of a second one do not exist here);
// x~ g~ ~= hi hi + hi lo + lo hi, three MFMAs per k-step and tile instead of six.  Accuracy: 22 bits for every element within
// 2^(17 - r_i/2) of its row maximum, an absolute 2^-25 (scaled) below that, i.e. |error| <= 2^-22 sum|x||g| + (rows) 2^-39 max_i(mx_i
// mg_i).  Rows whose products are more than 2^40 below the largest (they would keep < 19 bits), infinite / NaN / subnormal row
// maxima or scales outside the fp32 exponent range set `bad`, and the six-product kernel does the call instead - decided on the
// device (tn_scale_* kernels: two launches over the (M,) row maxima), no host synchronisation.
// Structure, chunking, XCD mapping and the fixed-order split reduction are gemm_x3_tn_kernel's; the row scales travel as the
// upper halves of their fp32 patterns (8 rows = one 16-byte load) with the raw chunk they belong to.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int kTn2Slab = 2 * kTnPiece;           // 20 KB
constexpr int kTnMaxSpread = 40;

__global__ void tn_state_init_kernel(int* state) {
  if (threadIdx.x < 4) state[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void tn_scale_range_kernel(const float* xmax, const float* gmax, int64_t M, int* state) {
  // a FEW workgroups, grid-stride, one atomic each: 16 000 wavefronts updating the same three words took 0.37 ms at M = 2^20
  __shared__ int red[3][4];
  int hi = 0, lo = 0, bad = 0;                  // t + 1024 and 1024 - t: 0 = "no row yet", so the state starts as plain zeros
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t xb = __float_as_uint(xmax[i]) & 0x7fffffffu, gb = __float_as_uint(gmax[i]) & 0x7fffffffu;
    if (xb != 0 && gb != 0) {
      const int ex = (int)(xb >> 23), eg = (int)(gb >> 23);
      if (ex == 255 || eg == 255 || ex == 0 || eg == 0) bad = 1;
      else { const int t = ex + eg - 254; hi = max(hi, t + 1024); lo = max(lo, 1024 - t); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hi = max(hi, __shfl_xor(hi, o, 64)); lo = max(lo, __shfl_xor(lo, o, 64)); bad |= __shfl_xor(bad, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = hi; red[1][wave] = lo; red[2][wave] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    hi = max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3]));
    lo = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
    bad = red[2][0] | red[2][1] | red[2][2] | red[2][3];
    if (hi) atomicMax(state + 0, hi);
    if (lo) atomicMax(state + 1, lo);
    if (bad) atomicOr(state + 2, 1);
  }
}

// sxh / sgh have M_pad = M rounded up to 32 entries (a chunk's scale loads never straddle the end); the padding gets scale 0
__global__ void tn_scale_rows_kernel(const float* xmax, const float* gmax, int64_t M, int64_t M_pad, int* state, uint16_t* sxh,
                                     uint16_t* sgh) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int T = state[0] ? state[0] - 1024 : 28;
  if (i == 0) state[3] = T - 28;
  const uint32_t xb = i < M ? __float_as_uint(xmax[i]) & 0x7fffffffu : 0u, gb = i < M ? __float_as_uint(gmax[i]) & 0x7fffffffu : 0u;
  uint16_t hx = 0, hg = 0;
  bool bad = false;
  if (xb != 0 && gb != 0) {
    const int ex = (int)(xb >> 23) - 127, eg = (int)(gb >> 23) - 127;
    const int r = T - (ex + eg);
    const int px = 14 - ex - (r >> 1), pg = 14 - eg - (r - (r >> 1));
    bad = r > kTnMaxSpread || px < -126 || px > 127 || pg < -126 || pg > 127;
    if (!bad) { hx = (uint16_t)((px + 127) << 7); hg = (uint16_t)((pg + 127) << 7); }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0 && state[2] == 0) atomicOr(state + 2, 1);     // at most one atomic per wavefront
  if (i < M_pad) { sxh[i] = hx; sgh[i] = hg; }
}

// max |a| of every row of a (M, cols) matrix: 32 lanes per row (only for callers that do not bring the maxima along)
template <bool VEC>
__global__ void row_absmax_kernel(const float* A, int64_t lda, int64_t M, int cols, float* out) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int l = threadIdx.x & 31;
  uint32_t m = 0;                                  // the bit patterns of |a| order like the values, and a NaN (or inf) stays on top
  if (row < M) {
    if (VEC) {
      for (int c = 4 * l; c < cols; c += 128) {
        const uint4 v = *reinterpret_cast<const uint4*>(A + row * lda + c);
        m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
      }
    } else {
      for (int c = l; c < cols; c += 32) m = max(m, __float_as_uint(A[row * lda + c]) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if (row < M && l == 0) out[row] = __uint_as_float(m);
}
static void launch_row_absmax(const float* A, int64_t lda, int64_t M, int cols, float* out, hipStream_t st) {
  const unsigned blocks = (unsigned)((M + 7) / 8);                     // 8 rows per 256-thread block
  if (cols % 4 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0)
    hipLaunchKernelGGL(row_absmax_kernel<true>, dim3(blocks), dim3(256), 0, st, A, lda, M, cols, out);
  else hipLaunchKernelGGL(row_absmax_kernel<false>, dim3(blocks), dim3(256), 0, st, A, lda, M, cols, out);
}

__device__ __forceinline__ void split2s(const float v, const float s, _Float16& hi, _Float16& lo) {
  const float x = v * s;
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
// element i (0..7) of a row-scale vector: the upper halves of eight fp32 powers of two
__device__ __forceinline__ float scale_of(const u32x4& q, const int i) {
  const uint32_t w = q[i >> 1];
  return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
}

// [r5] X of one three-product TN call in fragment order: Xp[chunk of 32 rows][block of 32 columns][k-step 2][piece 2][lane 64] x 16 bytes,
// scaled by the row scales of THIS call (sxh) and split exactly as the kernel's own MMA_TN2_SPLITX does.  Rows past M: zeros; columns past
// KA repeat the last column (they only reach accumulator rows that are never stored).  One wave per (chunk, block).
__global__ __launch_bounds__(256) void tn_pack_x_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int KA, const uint16_t* __restrict__ sxh,
                                                        u32x4* __restrict__ Xp, int nb, int64_t n_items) {
  const int lane = threadIdx.x & 63, r31 = lane & 31, h = lane >> 5;
  for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (int64_t)gridDim.x * 4) {
    const int64_t chunk = it / nb;
    const int w = (int)(it % nb);
    const int col = min(w * 32 + r31, KA - 1);
    u32x4* dst = Xp + (size_t)it * 256 + lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int64_t row0 = chunk * 32 + ks * 16 + 8 * h;
      const u32x4 sc = *reinterpret_cast<const u32x4*>(sxh + row0);             // sxh is padded to whole chunks (scale 0)
      f16x8 a, b;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = row0 + i < M ? X[(row0 + i) * ldx + col] : 0.f;
        _Float16 u, l;
        split2s(v, scale_of(sc, i), u, l);
        a[i] = u; b[i] = l;
      }
      dst[(ks * 2 + 0) * 64] = *reinterpret_cast<const u32x4*>(&a);
      dst[(ks * 2 + 1) * 64] = *reinterpret_cast<const u32x4*>(&b);
    }
  }
}

// NW = 4: 256 threads, X up to 128 columns wide (a wave per 32 of them), two workgroups per CU.  NW = 8 [r4]: 512 threads for X up to 256
// columns wide (hidden width 256, C5) - the eight waves share ONE staged G tile, so G is read once instead of once per 128-column block
// of X (two launches, each reading all 17 GB of [gP|gQ] at the C5 shard shape), and a thread splits 8 of its values per chunk, not 16.
// PX [r5]: X comes packed.  Every 128-column block of G walks all rows of X: at hidden width 256 (C5: 32 column blocks) each of them loaded
// X with 16 dword loads per chunk and lane and split it again - two thirds of the kernel's VALU instructions (SQ_INSTS_VALU 1.18e9 per call
// against 3.0e9 / 6 for the forward).  tn_pack_x_kernel does it once per call (the row scales depend on BOTH operands' maxima, so not once
// per layer); the kernel then takes a chunk's fragments with four coalesced 16-byte loads.  Same pieces, same products: the same bits.
template <bool FULLCT, int NW, bool PX = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void gemm_f16x2_tn_kernel(const TnParams p) {
  constexpr int KG = 8 / NW;                     // 8-row k-groups of the 32-row chunk a thread stages: 2 (groups skg, skg + 2) or 1
  constexpr int GV = 8 * KG;                     // its G values per chunk
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kTn2Slab];
  if (p.state[2] != 0) return;                    // the six-product kernel does this call
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r31 = lane & 31, h = lane >> 5;
  const int n_cb = (p.NC + 127) / 128;
  int cb, split;
  if (p.splits % 8 == 0) {
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    cb = slot % n_cb;
    split = (slot / n_cb) * 8 + xcd;
  } else {
    cb = (int)blockIdx.x % n_cb;
    split = (int)blockIdx.x / n_cb;
  }
  const int gcol0 = cb * 128;
  const int n_ct = FULLCT ? 4 : min(4, (p.NC - gcol0 + 31) / 32);     // the last tile may be ragged (stores are guarded)
  const int64_t r0 = (int64_t)split * p.rows_per_split;
  const int64_t r1 = min(p.M, r0 + p.rows_per_split);
  const int n_chunks = r1 > r0 ? (int)((r1 - r0 + kTnKC - 1) / kTnKC) : 0;
  const bool wave_active = wave * 32 < p.KA;
  const int sn = tid & 127, skg = tid >> 7;
  const int scol = gcol0 + sn < p.NC ? sn : 0;
  const int xcol = min(wave * 32 + r31, p.KA - 1);   // columns past KA repeat the last one: they only reach rows that are never stored
  const int ldg_b = (int)p.ldg * 4, ldx_b = (int)p.ldx * 4;
  const int rows = (int)(r1 > r0 ? r1 - r0 : 0);
  const int64_t rbase = r1 > r0 ? r0 : 0;
  const int64_t bz = (int64_t)blockIdx.y;                                      // batch index (0 for a single product)
  const auto g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G + bz * p.gb + rbase * p.ldg + gcol0), 0,
                                                        rows ? (rows - 1) * ldg_b + min(128, p.NC - gcol0) * 4 : 0, 0x00020000);
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + bz * p.xb + rbase * p.ldx), 0,
                                                        rows ? (rows - 1) * ldx_b + p.KA * 4 : 0, 0x00020000);
  // the scale vectors of this split's rows (rows past the split read as scale 0: their values read as 0 as well)
  const int rows_pad = (rows + kTnKC - 1) / kTnKC * kTnKC;   // the arrays are padded to whole chunks (scale 0), rbase % 32 == 0
  const auto sg_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.sgh + rbase), 0, rows_pad * 2, 0x00020000);
  const auto sx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.sxh + rbase), 0, rows_pad * 2, 0x00020000);
  const int g_voff = skg * 8 * ldg_b + scol * 4;
  const int x_voff = 8 * h * ldx_b + xcol * 4;
  const int sg_voff = skg * 16, sx_voff = h * 16;            // bytes: 8 rows x 2
  // packed X: chunk (rbase / 32 + c), block `wave`: 256 x 16 bytes; chunks past the split's last read as zeros (range check)
  const int xq_stride = p.xp_nb * 4096;
  const auto xq_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(static_cast<const unsigned char*>(p.xp) + (PX ? (size_t)(rbase >> 5) * (size_t)xq_stride : 0)), 0,
      PX ? (rows_pad / kTnKC) * xq_stride : 0, 0x00020000);
  const int xq_voff = ((wave_active ? wave : 0) * 256 + lane) * 16;

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float graw0[GV], xraw0[16], graw1[GV], xraw1[16];
  u32x4 gs0[KG], xs0[2], gs1[KG], xs1[2];                     // row scales of the raw sets
  f16x8 af[2][2];

#define MMA_TN2_LOAD(C_, GR_, XR_, GS_, XS_)                                                   \
  {                                                                                            \
    int gv = g_voff + (C_) * (kTnKC * ldg_b);                                                  \
    int xv = x_voff + (C_) * (kTnKC * ldx_b);                                                  \
    asm volatile("" : "+v"(gv), "+v"(xv));                                                     \
    _Pragma("unroll") for (int g = 0; g < KG; ++g)                                             \
      GS_[g] = __builtin_amdgcn_raw_buffer_load_b128(sg_rsrc, sg_voff + (C_) * (kTnKC * 2) + g * 32, 0, 0); \
    if (!PX) {                                                                                 \
      _Pragma("unroll") for (int g = 0; g < 2; ++g)                                            \
        XS_[g] = __builtin_amdgcn_raw_buffer_load_b128(sx_rsrc, sx_voff + (C_) * (kTnKC * 2) + g * 32, 0, 0); \
    }                                                                                          \
    _Pragma("unroll") for (int g = 0; g < KG; ++g)                                             \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
        GR_[g * 8 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(g_rsrc, gv, (g * 16 + i) * ldg_b, 2 /* nt: G streams through once, X is what the column blocks share in L2 */)); \
    if (PX) {                     /* the 16 registers of the raw set hold the four packed fragments */ \
      _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                          \
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xq_rsrc, xq_voff + f * 1024, (C_) * xq_stride, 0); \
        XR_[4 * f] = __uint_as_float(q[0]); XR_[4 * f + 1] = __uint_as_float(q[1]);            \
        XR_[4 * f + 2] = __uint_as_float(q[2]); XR_[4 * f + 3] = __uint_as_float(q[3]);        \
      }                                                                                        \
    } else {                                                                                   \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                         \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                          \
          XR_[ks * 8 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, xv, (ks * 16 + i) * ldx_b, 0)); \
    }                                                                                          \
  }
#define MMA_TN2_SPLITX(XR_, XS_)                                                               \
  if (PX) {                                                                                    \
    _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                            \
      const u32x4 q = {__float_as_uint(XR_[4 * f]), __float_as_uint(XR_[4 * f + 1]), __float_as_uint(XR_[4 * f + 2]), __float_as_uint(XR_[4 * f + 3])}; \
      af[f >> 1][f & 1] = *reinterpret_cast<const f16x8*>(&q);                                 \
    }                                                                                          \
  } else {                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                          \
        _Float16 a, b;                                                                         \
        split2s(XR_[ks * 8 + i], scale_of(XS_[ks], i), a, b);                                  \
        af[ks][0][i] = a; af[ks][1][i] = b;                                                    \
      }                                                                                        \
  }
#define MMA_TN2_PUBLISH(B_, GR_, XR_, GS_, XS_)                                                \
  {                                                                                            \
    unsigned char* d = lds + (B_) * kTn2Slab + sn * kTnPitch;                                  \
    _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                           \
      f16x8 a, b;                                                                              \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                          \
        _Float16 u, v;                                                                         \
        split2s(GR_[g * 8 + i], scale_of(GS_[g], i), u, v);                                    \
        a[i] = u; b[i] = v;                                                                    \
      }                                                                                        \
      *reinterpret_cast<f16x8*>(d + 0 * kTnPiece + (skg + 2 * g) * 16) = a;                    \
      *reinterpret_cast<f16x8*>(d + 1 * kTnPiece + (skg + 2 * g) * 16) = b;                    \
    }                                                                                          \
    MMA_TN2_SPLITX(XR_, XS_)                                                                   \
  }
// One chunk: four groups of SIX MFMAs (k-step x pair of column tiles x {hi hi, hi lo, lo hi}), each followed in program order by the
// split of four of the next chunk's G values; fragments of group m+1 are read while group m multiplies.
#define MMA_TN2_STEP(C_, B_, GR_, XR_, GS_, XS_)                                               \
  {                                                                                            \
    const unsigned char* sb = lds + (B_) * kTn2Slab + r31 * kTnPitch + h * 16;                 \
    unsigned char* pd = lds + (1 - (B_)) * kTn2Slab + sn * kTnPitch + skg * 16;                \
    f16x8 bq[2][2][2], ga, gb;           /* [parity][tile of the pair][piece] */               \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                              \
      _Pragma("unroll") for (int q = 0; q < 2; ++q)                                            \
        bq[0][t][q] = *reinterpret_cast<const f16x8*>(sb + t * 32 * kTnPitch + q * kTnPiece);  \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                            \
      const int ks = m >> 1, c0 = 2 * (m & 1), c1 = c0 + 1, pa = m & 1;                        \
      if (m + 1 < 4) {                                                                         \
        const unsigned char* qn = sb + (2 * ((m + 1) & 1)) * 32 * kTnPitch + ((m + 1) >> 1) * 32; \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                          \
          _Pragma("unroll") for (int q = 0; q < 2; ++q)                                        \
            bq[1 - pa][t][q] = *reinterpret_cast<const f16x8*>(qn + t * 32 * kTnPitch + q * kTnPiece); \
      }                                                                                        \
      const bool t0 = FULLCT || c0 < n_ct, t1 = FULLCT || c1 < n_ct;                           \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][1], bq[pa][0][0], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][1], bq[pa][1][0], acc[c1], 0, 0, 0); \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], bq[pa][0][1], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], bq[pa][1][1], acc[c1], 0, 0, 0); \
      if (t0) acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], bq[pa][0][0], acc[c0], 0, 0, 0); \
      if (t1) acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], bq[pa][1][0], acc[c1], 0, 0, 0); \
      _Pragma("unroll") for (int j = (GV / 4) * m; j < (GV / 4) * (m + 1); ++j) { /* a quarter of the next chunk's G values */ \
        _Float16 u, v;                                                                         \
        split2s(GR_[j], scale_of(GS_[j >> 3], j & 7), u, v);                                   \
        ga[j & 7] = u; gb[j & 7] = v;                                                          \
      }                                                                                        \
      if (((GV / 4) * (m + 1)) % 8 == 0) {         /* a k-group of 8 rows is complete */       \
        *reinterpret_cast<f16x8*>(pd + 0 * kTnPiece + (((GV / 4) * (m + 1)) / 8 - 1) * 32) = ga; \
        *reinterpret_cast<f16x8*>(pd + 1 * kTnPiece + (((GV / 4) * (m + 1)) / 8 - 1) * 32) = gb; \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }                                                                                          \
    MMA_TN2_SPLITX(XR_, XS_)                                                                   \
    MMA_TN2_LOAD((C_) + 3, GR_, XR_, GS_, XS_)                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_s_barrier();                                                              \
    asm volatile("" ::: "memory");                                                             \
  }

  MMA_TN2_LOAD(0, graw0, xraw0, gs0, xs0)
  MMA_TN2_PUBLISH(0, graw0, xraw0, gs0, xs0)
  MMA_TN2_LOAD(1, graw1, xraw1, gs1, xs1)
  MMA_TN2_LOAD(2, graw0, xraw0, gs0, xs0)
  __syncthreads();
  for (int c = 0; c < n_chunks; c += 2) {
    MMA_TN2_STEP(c, 0, graw1, xraw1, gs1, xs1)
    MMA_TN2_STEP(c + 1, 1, graw0, xraw0, gs0, xs0)
  }
#undef MMA_TN2_STEP
#undef MMA_TN2_LOAD
#undef MMA_TN2_PUBLISH
#undef MMA_TN2_SPLITX

  if (wave_active) {
    const int ue = p.state[3];                   // log2 of the un-scaling (T - 28)
    float* out = p.part + ((size_t)split * p.KA + (size_t)wave * 32) * p.NC + gcol0 + r31;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      if (ct < n_ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (FULLCT || (wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < p.KA && gcol0 + ct * 32 + r31 < p.NC))
            out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * p.NC + ct * 32] = ldexpf(acc[ct][r], ue);
      }
    }
  }
}

static int tn_splits(int64_t M, int NC) {
  const int64_t n_cb = ((int64_t)NC + 127) / 128;
  int64_t s = 512 / n_cb;                               // ~2 workgroups per CU in total
  static const int min_chunks_small = getenv("MMA_TN_MIN_CHUNKS") ? atoi(getenv("MMA_TN_MIN_CHUNKS")) : 2;
  // at least 8 chunks per split - 2 on small problems, where the launch is latency-bound and 16 workgroups walking 11 chunks each
  // (Cora: 2 708 rows) take twice as long as 80 walking 3 (no M + const: M comes from the caller)
  const int64_t max_s = M / ((M >= 65536 ? 8 : min_chunks_small) * kTnKC) + 1;
  if (s > max_s) s = max_s;
  if (s >= 8) s = s / 8 * 8;                            // multiples of 8: the XCD-local id mapping of the kernel
  return (int)(s < 1 ? 1 : s);
}

}  // namespace mma

using namespace mma;

extern "C" int mma_split_bf16x3(const float* in, int64_t n, void* out, void* stream) {
  MMA_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return 0;
  MMA_REQUIRE(in && out, "NULL argument");
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), in, n,
                     static_cast<__bf16*>(out));
  return check_launch("split3_kernel");
}

extern "C" int mma_gemm_bf16x3(const float* A, int64_t lda, const void* Bt3, float* C, int64_t ldc, int64_t M, int32_t N, int32_t K,
                               int32_t accumulate, void* stream) {
  MMA_REQUIRE(M >= 0 && N >= 32 && K >= kKC && N % 32 == 0 && K % kKC == 0, "M=%lld N=%d K=%d: need N %% 32 == 0, K %% 128 == 0",
              (long long)M, N, K);
  MMA_REQUIRE(K == kKC || N <= 128, "either K == 128 or N <= 128 (accumulator tiles live in registers)");
  MMA_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && lda < (1 << 24) && ldc < (1 << 24), "row pitch too small, unaligned or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(A && Bt3 && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt3) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt3), C, ldc, M, N, K, accumulate ? 1 : 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nct = K == kKC ? 0 : N / 32;
  // tall plain K == 128 products with whole 128-column groups: the column-group kernel takes every full 256-row unit
  int64_t first_blk = 0;
  const int n_groups = N / 128;
  if (K == kKC && !accumulate && N % 128 == 0 && n_groups <= kCgSlotsPerXcd && M / kCgRows >= 4 * 8 * (kCgSlotsPerXcd / n_groups)) {
    const int64_t n_units = M / kCgRows;
    hipLaunchKernelGGL(gemm_x3_colgroup_kernel, dim3(256), dim3(kCgThreads), 0, st, p, n_units, n_groups);
    if (int rc = check_launch("gemm_x3_colgroup_kernel")) return rc;
    first_blk = n_units * (kCgRows / 128);
  }
  // N == 128 with a long K (the dL/dx products): the 8-wave kernel takes every full 256-row unit
  if (N == 128 && K > kKC && M / kNlRows >= 256) {
    const int64_t n_units = M / kNlRows;
    const dim3 g((unsigned)(n_units < 256 ? n_units : 256));
    if (accumulate) hipLaunchKernelGGL(gemm_x3_n128_kernel<true>, g, dim3(kNlThreads), 0, st, p, n_units);
    else hipLaunchKernelGGL(gemm_x3_n128_kernel<false>, g, dim3(kNlThreads), 0, st, p, n_units);
    if (int rc = check_launch("gemm_x3_n128_kernel")) return rc;
    first_blk = n_units * (kNlRows / 128);
  }
  const int64_t n_full = M / 128;
  for (int part = 0; part < 2; ++part) {
    const bool full = part == 0;
    const int64_t blk0 = full ? first_blk : n_full, nb = full ? n_full - first_blk : ((M % 128) ? 1 : 0);
    if (nb == 0) continue;
    const dim3 grid((unsigned)(nb < 512 ? nb : 512));      // 2 workgroups per CU (52 KB LDS, ~250 VGPRs each)
#define MMA_X3_LAUNCH(NCT_)                                                                                       \
    if (full && accumulate) hipLaunchKernelGGL((gemm_x3_kernel<NCT_, true, true>), grid, dim3(kBlock), 0, st, p, blk0, nb);   \
    else if (full) hipLaunchKernelGGL((gemm_x3_kernel<NCT_, true, false>), grid, dim3(kBlock), 0, st, p, blk0, nb);           \
    else if (accumulate) hipLaunchKernelGGL((gemm_x3_kernel<NCT_, false, true>), grid, dim3(kBlock), 0, st, p, blk0, nb);     \
    else hipLaunchKernelGGL((gemm_x3_kernel<NCT_, false, false>), grid, dim3(kBlock), 0, st, p, blk0, nb);
    switch (nct) {
      case 0: MMA_X3_LAUNCH(0) break;
      case 1: MMA_X3_LAUNCH(1) break;
      case 2: MMA_X3_LAUNCH(2) break;
      case 3: MMA_X3_LAUNCH(3) break;
      default: MMA_X3_LAUNCH(4) break;
    }
#undef MMA_X3_LAUNCH
  }
  return check_launch("gemm_x3_kernel");
}

extern "C" int mma_split_f16x2(const float* w, int64_t stride_k, int64_t stride_n, int32_t K, int32_t N, void* bt2, float* col_unscale,
                               int32_t plain_lo, void* stream) {
  MMA_REQUIRE(K >= 1 && N >= 1 && K <= (1 << 20) && N <= (1 << 20) && (int64_t)K * N < (1LL << 31), "K=%d N=%d out of range", K, N);
  MMA_REQUIRE(w && bt2 && col_unscale, "NULL argument");
  MMA_REQUIRE(stride_k >= 0 && stride_n >= 0 && (int64_t)(K - 1) * stride_k + (int64_t)(N - 1) * stride_n < (1LL << 31), "strides out of range");
  hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream), w, stride_k, stride_n, K, N,
                     static_cast<_Float16*>(bt2), col_unscale, plain_lo ? 1 : 0);
  return check_launch("split_f16x2_kernel");
}

extern "C" int mma_gemm_f16x2(const float* A, int64_t lda, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                              float* a_row_max, int64_t M, int32_t N, void* stream) {
  MMA_REQUIRE(M >= 0 && N >= 128 && N % 128 == 0 && N / 128 <= kCgSlotsPerXcd, "M=%lld N=%d: need N %% 128 == 0, N <= 4096", (long long)M, N);
  MMA_REQUIRE(lda >= 128 && ldc >= N && lda % 4 == 0 && lda < (1 << 24) && ldc < (1 << 24), "row pitch too small, unaligned or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(A && Bt2 && col_unscale && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt2) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt2), C, ldc, M, N, 128, 0};
#ifdef MMA_EXPERIMENTAL_FWD
  // MMA_FWD_WS=1: the W-stationary kernel (A/B switch, read per call).  Measured SLOWER than the column-group kernel (C4 1.37 vs 1.06 ms,
  // C5 8.4 vs 7.8 ms): its waves meet at a barrier every 64 rows, and a wave's wait for the next rows is a wait for every store it issued
  // before them (one in-order vmcnt) - see DESIGN.md "forward GEMM, round 4"; kept for the next step (loader waves that never store).
  const char* ws_env = getenv("MMA_FWD_WS");
  const bool use_ws = ws_env && ws_env[0] == '1';
  if (ws_env && ws_env[0] == '2' && N % 256 == 0 && (kCgSlotsPerXcd % (N / 256)) == 0) {         // MMA_FWD_WS=2: loader waves
    hipLaunchKernelGGL(gemm_f16x2_lw_kernel, dim3(256), dim3(kLwThreads), 0, static_cast<hipStream_t>(stream), p, col_unscale, (M + 63) / 64,
                       N / 256, a_row_max);
    return check_launch("gemm_f16x2_lw_kernel");
  }
  if (use_ws && N % 256 == 0 && (kCgSlotsPerXcd % (N / 256)) == 0) {
    hipLaunchKernelGGL((gemm_f16x2_ws_kernel<8, 2>), dim3(256), dim3(kCgThreads), 0, static_cast<hipStream_t>(stream), p, col_unscale,
                       (M + 63) / 64, N / 256, a_row_max);
    return check_launch("gemm_f16x2_ws_kernel");
  }
#endif
  const int64_t n_units = (M + kCgRows - 1) / kCgRows;
  const int groups = N / 128;
  if (groups % 2 == 0)        // pairs of column groups per workgroup
    hipLaunchKernelGGL(gemm_f16x2_colgroup_kernel<2>, dim3(256), dim3(kCgThreads), 0, static_cast<hipStream_t>(stream), p, col_unscale,
                       n_units, groups / 2, a_row_max);
  else
    hipLaunchKernelGGL(gemm_f16x2_colgroup_kernel<1>, dim3(256), dim3(kCgThreads), 0, static_cast<hipStream_t>(stream), p, col_unscale,
                       n_units, groups, a_row_max);
  return check_launch("gemm_f16x2_colgroup_kernel");
}

// [r5] K in {64, 96}: the narrow forms of the column-group kernel (Bt2 (2, N, K)); three resident groups where N / 128 is a multiple of 3
extern "C" int mma_gemm_f16x2_k(const float* A, int64_t lda, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                                float* a_row_max, int64_t M, int32_t N, int32_t K, void* stream) {
  if (K == 128) return mma_gemm_f16x2(A, lda, Bt2, col_unscale, C, ldc, a_row_max, M, N, stream);
  MMA_REQUIRE(K == 64 || K == 96, "K=%d unsupported (64, 96 or 128)", K);
  MMA_REQUIRE(M >= 0 && N >= 128 && N % 128 == 0 && N / 128 <= kCgSlotsPerXcd, "M=%lld N=%d: need N %% 128 == 0, N <= 4096", (long long)M, N);
  MMA_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && lda < (1 << 24) && ldc < (1 << 24), "row pitch too small, unaligned or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(A && Bt2 && col_unscale && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt2) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt2), C, ldc, M, N, K, 0};
  const int64_t n_units = (M + kCgRows - 1) / kCgRows;
  const int groups = N / 128;
  int g2 = groups % 3 == 0 ? 3 : (groups % 2 == 0 ? 2 : 1);
  { const char* e = getenv("MMA_F16X2_G2"); if (e && e[0] >= '1' && e[0] <= '3' && groups % (e[0] - '0') == 0) g2 = e[0] - '0'; }      // A/B, read per call
  hipStream_t st = static_cast<hipStream_t>(stream);
#define MMA_CG(GG, KK) hipLaunchKernelGGL((gemm_f16x2_colgroup_kernel<GG, KK>), dim3(256), dim3(kCgThreads), 0, st, p, col_unscale, n_units, groups / GG, a_row_max)
  if (K == 64) { if (g2 == 3) MMA_CG(3, 4); else if (g2 == 2) MMA_CG(2, 4); else MMA_CG(1, 4); }
  else         { if (g2 == 3) MMA_CG(3, 6); else if (g2 == 2) MMA_CG(2, 6); else MMA_CG(1, 6); }
#undef MMA_CG
  return check_launch("gemm_f16x2_colgroup_kernel (narrow)");
}

extern "C" int mma_gemm_f16x2_k256(const float* A, int64_t lda, const float* row_max, const void* Bt2, const float* col_unscale, float* C,
                                   int64_t ldc, int64_t M, int32_t N, void* stream) {
  MMA_REQUIRE(M >= 0 && N >= 128 && N % 128 == 0 && N / 128 <= kCgSlotsPerXcd, "M=%lld N=%d: need N %% 128 == 0, N <= 4096", (long long)M, N);
  MMA_REQUIRE(lda >= 256 && ldc >= N && lda % 4 == 0 && lda < (1 << 24) && ldc < (1 << 24), "row pitch too small, unaligned or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(A && row_max && Bt2 && col_unscale && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt2) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt2), C, ldc, M, N, 256, 0};
  const int64_t n_units = (M + kCgRows - 1) / kCgRows;
  hipLaunchKernelGGL(gemm_f16x2_colgroup_k256_kernel, dim3(256), dim3(kCgThreads), 0, static_cast<hipStream_t>(stream), p, row_max, col_unscale,
                     n_units, N / 128);
  return check_launch("gemm_f16x2_colgroup_k256_kernel");
}

// [r5] the packed A operand of mma_gemm_f16x2_k256p: Ap = ceil(M / 32) units of 32 KB (mma_pack_f16x2_k256_bytes), sce (M,) int32 scale
// exponents, row_max (M,) optional (max |a| per row: the x_row_max of the weight-gradient product) - one pass over A
extern "C" int64_t mma_pack_f16x2_k256_bytes(int64_t M) { return M <= 0 ? 0 : (M + 31) / 32 * (int64_t)(16 * 2 * 64 * 16); }
extern "C" int mma_pack_f16x2_k256(const float* A, int64_t lda, int64_t M, void* Ap, int32_t* sce, float* row_max, void* stream) {
  MMA_REQUIRE(M >= 0 && lda >= 256 && lda % 4 == 0 && lda < (1 << 24), "M=%lld lda=%lld: need lda >= 256, lda %% 4 == 0", (long long)M, (long long)lda);
  if (M == 0) return 0;
  MMA_REQUIRE(A && Ap && sce && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Ap) & 15) == 0, "NULL or misaligned argument");
  const int64_t n_units = (M + 31) / 32;
  hipLaunchKernelGGL(pack_f16x2_k256_kernel, dim3((unsigned)std::min<int64_t>((n_units + 3) / 4, 8 * 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), A, lda, M, static_cast<uint4*>(Ap), sce, row_max);
  return check_launch("pack_f16x2_k256_kernel");
}
extern "C" int mma_gemm_f16x2_k256p(const void* Ap, const int32_t* sce, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                                    int64_t M, int32_t N, void* stream) {
  MMA_REQUIRE(M >= 0 && N >= 128 && N % 128 == 0 && N / 128 <= kCgSlotsPerXcd, "M=%lld N=%d: need N %% 128 == 0, N <= 4096", (long long)M, N);
  MMA_REQUIRE(ldc >= N && ldc < (1 << 24), "row pitch too small or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(Ap && sce && Bt2 && col_unscale && C && (reinterpret_cast<uintptr_t>(Ap) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt2) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{nullptr, 256, static_cast<const __bf16*>(Bt2), C, ldc, M, N, 256, 0};
  const int64_t n_units = (M + kCgRows - 1) / kCgRows;
  hipLaunchKernelGGL(gemm_f16x2_colgroup_k256p_kernel, dim3(256), dim3(kCgThreads), 0, static_cast<hipStream_t>(stream), p,
                     static_cast<const uint4*>(Ap), sce, col_unscale, n_units, N / 128);
  return check_launch("gemm_f16x2_colgroup_k256p_kernel");
}

extern "C" int mma_gemm_f16x2_ws(const float* A, int64_t lda, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                                 float* a_row_max, int64_t M, int32_t N, int32_t K, void* stream) {
  MMA_REQUIRE(M >= 0 && (K == 128 || K == 256) && N >= 256 && N % 256 == 0 && (kCgSlotsPerXcd % (N / 256)) == 0,
              "M=%lld N=%d K=%d: need K in {128, 256}, N %% 256 == 0 and N / 256 dividing %d", (long long)M, N, K, kCgSlotsPerXcd);
  MMA_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && lda < (1 << 24) && ldc < (1 << 24), "row pitch too small, unaligned or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(A && Bt2 && col_unscale && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt2) & 15) == 0,
              "NULL or misaligned argument");
#ifdef MMA_EXPERIMENTAL_FWD
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt2), C, ldc, M, N, K, 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K == 128) hipLaunchKernelGGL((gemm_f16x2_ws_kernel<8, 2>), dim3(256), dim3(kCgThreads), 0, st, p, col_unscale, (M + 63) / 64, N / 256, a_row_max);
  else hipLaunchKernelGGL((gemm_f16x2_ws_kernel<16, 1>), dim3(256), dim3(kCgThreads), 0, st, p, col_unscale, (M + 31) / 32, N / 256, a_row_max);
  return check_launch("gemm_f16x2_ws_kernel");
#else
  (void)a_row_max; (void)stream;
  return fail(3, "mma_gemm_f16x2_ws: built without -DMMA_EXPERIMENTAL_FWD (the W-stationary / loader-wave forward kernels are measurement "
                 "forms, slower than mma_gemm_f16x2 / _k256; `make -C mma_amd/csrc clean all EXTRA=-DMMA_EXPERIMENTAL_FWD` builds them)");
#endif
}

extern "C" int mma_gemm_f16x2_n128(const float* A, int64_t lda, const float* row_max, const void* Bt2, const float* col_unscale, float* C,
                                   int64_t ldc, int64_t M, int32_t K, int32_t accumulate, void* stream) {
  MMA_REQUIRE(M >= 0 && K >= kNlKC && K % kNlKC == 0 && K <= (1 << 20), "M=%lld K=%d: need K %% 64 == 0", (long long)M, K);
  MMA_REQUIRE(lda >= K && ldc >= 128 && lda % 4 == 0 && lda < (1 << 24) && ldc < (1 << 24), "row pitch too small, unaligned or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(A && row_max && Bt2 && col_unscale && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt2) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt2), C, ldc, M, 128, K, accumulate ? 1 : 0};
  const int64_t n_units = (M + kNlRows - 1) / kNlRows;
  const dim3 g((unsigned)(n_units < 256 ? n_units : 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  static const bool acc_atomic = getenv("MMA_DX_ACC") && getenv("MMA_DX_ACC")[0] == 'a';     // MMA_DX_ACC=atomic: round 3's epilogue (A/B switch)
  if (accumulate && acc_atomic) hipLaunchKernelGGL(gemm_f16x2_n128_kernel<1>, g, dim3(kNlThreads), 0, st, p, row_max, col_unscale, n_units);
  else if (accumulate) hipLaunchKernelGGL(gemm_f16x2_n128_kernel<2>, g, dim3(kNlThreads), 0, st, p, row_max, col_unscale, n_units);
  else hipLaunchKernelGGL(gemm_f16x2_n128_kernel<0>, g, dim3(kNlThreads), 0, st, p, row_max, col_unscale, n_units);
  return check_launch("gemm_f16x2_n128_kernel");
}

extern "C" int mma_gemm_f16x2_nlp(const float* A, int64_t lda, const float* row_max, const void* Bt2, const float* col_unscale, float* C,
                                  int64_t ldc, int64_t M, int32_t N, int32_t K, int32_t accumulate, void* stream) {
  MMA_REQUIRE(M >= 0 && (N == 128 || N == 256) && K >= 4 * kNlKC && K % (2 * kNlKC) == 0 && K <= (1 << 20),
              "M=%lld N=%d K=%d: need N in {128, 256}, K %% 128 == 0, K >= 256", (long long)M, N, K);
  MMA_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0 && lda < (1 << 24) && ldc < (1 << 24), "row pitch too small, unaligned or >= 2^24");
  if (M == 0) return 0;
  MMA_REQUIRE(A && row_max && Bt2 && col_unscale && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt2) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt2), C, ldc, M, N, K, accumulate ? 1 : 0};
  const int64_t n_units = (M + kNlRows - 1) / kNlRows;
  const dim3 g((unsigned)(n_units < 256 ? n_units : 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  static const bool acc_atomic = getenv("MMA_DX_ACC") && getenv("MMA_DX_ACC")[0] == 'a';
  const int mode = accumulate ? (acc_atomic ? 1 : 2) : 0;
#define MMA_NLP_LAUNCH(NT_)                                                                                                        \
  if (mode == 0) hipLaunchKernelGGL((gemm_f16x2_nlp_kernel<NT_, 0>), g, dim3(kNlThreads), 0, st, p, row_max, col_unscale, n_units);      \
  else if (mode == 1) hipLaunchKernelGGL((gemm_f16x2_nlp_kernel<NT_, 1>), g, dim3(kNlThreads), 0, st, p, row_max, col_unscale, n_units); \
  else hipLaunchKernelGGL((gemm_f16x2_nlp_kernel<NT_, 2>), g, dim3(kNlThreads), 0, st, p, row_max, col_unscale, n_units);
  if (N == 128) { MMA_NLP_LAUNCH(4) } else { MMA_NLP_LAUNCH(8) }
#undef MMA_NLP_LAUNCH
  return check_launch("gemm_f16x2_nlp_kernel");
}

extern "C" int64_t mma_gemm_bf16x3_tn_workspace_floats(int64_t M, int32_t KA, int32_t NC) {
  if (M <= 0 || KA <= 0 || NC <= 0) return 0;
  const int s = tn_splits(M, NC);
  return s > 1 ? (int64_t)s * KA * NC : 0;
}

extern "C" int mma_col_sum(const float* g, int64_t ldg, int64_t R, int32_t C, float* out, float* ws, int64_t ws_floats, void* stream);

extern "C" int mma_gemm_bf16x3_tn(const float* X, int64_t ldx, const float* G, int64_t ldg, float* C, float* ws, int64_t ws_floats,
                                  int64_t M, int32_t KA, int32_t NC, void* stream) {
  MMA_REQUIRE(M >= 1 && KA >= 1 && KA <= 128 && NC >= 1 && (int64_t)KA * NC < (1LL << 31), "M=%lld KA=%d NC=%d: need 1 <= KA <= 128, NC >= 1",
              (long long)M, KA, NC);
  MMA_REQUIRE(X && G && C && ldx >= KA && ldg >= NC && ldx < (1 << 24) && ldg < (1 << 24), "NULL argument or row pitch out of range");
  MMA_REQUIRE((reinterpret_cast<uintptr_t>(X) & 3) == 0 && (reinterpret_cast<uintptr_t>(G) & 3) == 0, "misaligned argument");
  const int s = tn_splits(M, NC);
  MMA_REQUIRE(s == 1 || (ws && ws_floats >= (int64_t)s * KA * NC), "workspace too small: %lld floats, need %lld",
              (long long)ws_floats, (long long)s * KA * NC);
  int64_t rps = (M + s - 1) / s;
  rps = (rps + kTnKC - 1) / kTnKC * kTnKC;
  MMA_REQUIRE((rps + kTnKC) * (ldx > ldg ? ldx : ldg) * 4 < (1LL << 31), "row range of one split exceeds a 2 GB buffer window");
  TnParams p{X, ldx, G, ldg, s == 1 ? C : ws, M, rps, KA, NC, s};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(((NC + 127) / 128) * s));
  if (NC % 128 == 0 && KA % 32 == 0) hipLaunchKernelGGL(gemm_x3_tn_kernel<true>, grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL(gemm_x3_tn_kernel<false>, grid, dim3(kBlock), 0, st, p);
  if (int rc = check_launch("gemm_x3_tn_kernel")) return rc;
  if (s == 1) return 0;
  return mma_col_sum(ws, (int64_t)KA * NC, s, KA * NC, C, nullptr, 0, stream);      // <= 512 rows: one pass, fixed order
}

// B independent TN products in ONE launch: C[b] (KA,NC) = X_b^T G_b with X_b = X + b*xb, G_b = G + b*gb (column blocks of wider rows).
// The per-tower weight gradients of MMAConv's post-NN (5 products of 48 x 152 over 2 * 10^5 rows at ZINC's shape) were five
// launches + five reductions of 0.08 ms each; the row ranges are cut for the whole batch (splits * column blocks * B workgroups).
static int tn_splits_batched(int64_t M, int NC, int B) {
  const int64_t n_cb = ((int64_t)NC + 127) / 128;
  int64_t s = 512 / (n_cb * B);
  const int64_t max_s = M / (8 * kTnKC) + 1;
  if (s > max_s) s = max_s;
  if (s >= 8) s = s / 8 * 8;
  return (int)(s < 1 ? 1 : s);
}
extern "C" int64_t mma_gemm_bf16x3_tn_batched_workspace_floats(int64_t M, int32_t KA, int32_t NC, int32_t B) {
  if (M <= 0 || KA <= 0 || NC <= 0 || B <= 0) return 0;
  const int s = tn_splits_batched(M, NC, B);
  return s > 1 ? (int64_t)s * B * KA * NC : 0;
}
extern "C" int mma_gemm_bf16x3_tn_batched(const float* X, int64_t ldx, int64_t xb, const float* G, int64_t ldg, int64_t gb, float* C, float* ws,
                                          int64_t ws_floats, int64_t M, int32_t KA, int32_t NC, int32_t B, void* stream) {
  MMA_REQUIRE(M >= 1 && KA >= 1 && KA <= 128 && NC >= 1 && B >= 1 && B <= 65535 && (int64_t)KA * NC * B < (1LL << 31),
              "M=%lld KA=%d NC=%d B=%d: need 1 <= KA <= 128, NC >= 1, 1 <= B <= 65535", (long long)M, KA, NC, B);
  MMA_REQUIRE(X && G && C && xb >= 0 && gb >= 0 && ldx >= xb * (B - 1) + KA && ldg >= gb * (B - 1) + NC && ldx < (1 << 24) && ldg < (1 << 24),
              "NULL argument, or the B column blocks do not fit the row pitch");
  MMA_REQUIRE((reinterpret_cast<uintptr_t>(X) & 3) == 0 && (reinterpret_cast<uintptr_t>(G) & 3) == 0, "misaligned argument");
  const int s = tn_splits_batched(M, NC, B);
  MMA_REQUIRE(s == 1 || (ws && ws_floats >= (int64_t)s * B * KA * NC), "workspace too small: %lld floats, need %lld",
              (long long)ws_floats, (long long)s * B * KA * NC);
  int64_t rps = (M + s - 1) / s;
  rps = (rps + kTnKC - 1) / kTnKC * kTnKC;
  MMA_REQUIRE((rps + kTnKC) * (ldx > ldg ? ldx : ldg) * 4 < (1LL << 31), "row range of one split exceeds a 2 GB buffer window");
  TnParams p{X, ldx, G, ldg, s == 1 ? C : ws, M, rps, KA, NC, s};
  p.xb = xb; p.gb = gb; p.part_ss = (int64_t)B * KA * NC;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(((NC + 127) / 128) * s), (unsigned)B);
  if (NC % 128 == 0 && KA % 32 == 0) hipLaunchKernelGGL(gemm_x3_tn_kernel<true>, grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL(gemm_x3_tn_kernel<false>, grid, dim3(kBlock), 0, st, p);
  if (int rc = check_launch("gemm_x3_tn_kernel (batched)")) return rc;
  if (s == 1) return 0;
  return mma_col_sum(ws, (int64_t)B * KA * NC, s, B * KA * NC, C, nullptr, 0, stream);      // <= 512 rows: one pass, fixed order
}

static int64_t tn_pad32(int64_t M) { return (M + 31) / 32 * 32; }

extern "C" int mma_row_absmax(const float* A, int64_t lda, int64_t M, int32_t cols, float* out, void* stream) {
  MMA_REQUIRE(M >= 0 && cols >= 1 && lda >= cols && lda < (1 << 24), "M=%lld cols=%d lda=%lld out of range", (long long)M, cols, (long long)lda);
  if (M == 0) return 0;
  MMA_REQUIRE(A && out && (reinterpret_cast<uintptr_t>(A) & 3) == 0, "NULL or misaligned argument");
  launch_row_absmax(A, lda, M, cols, out, static_cast<hipStream_t>(stream));
  return check_launch("row_absmax_kernel");
}

// X packed once per call when many 128-column blocks of G walk it (each of them would load and split X again): C5's 32 blocks, not C4's 8
// (measured: see DESIGN.md); MMA_TN_PACKX = 0 / 1 forces it (read per call, A/B)
static bool tn_packx(int64_t M, int KA, int NC) {
  const char* e = getenv("MMA_TN_PACKX");
  if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
  return M >= 65536 && ((int64_t)NC + 127) / 128 >= 16;
}
static int64_t tn_packx_floats(int64_t M, int KA) { return tn_pad32(M) / 32 * (((int64_t)KA + 31) / 32) * 1024; }      // 4 KB per (chunk, block)

extern "C" int64_t mma_gemm_f16x2_tn_workspace_floats(int64_t M, int32_t KA, int32_t NC) {
  if (M <= 0 || KA <= 0 || NC <= 0) return 0;
  const int s = tn_splits(M, NC);
  // partial tiles | row maxima of X, of G | scale halves | state | packed X
  return (s > 1 ? ((int64_t)s * KA * NC + 3) / 4 * 4 : 0) + 3 * tn_pad32(M) + 4 + (tn_packx(M, KA, NC) ? tn_packx_floats(M, KA) : 0);
}

extern "C" int mma_gemm_f16x2_tn(const float* X, int64_t ldx, const float* G, int64_t ldg, const float* x_row_max, const float* g_row_max,
                                 float* C, float* ws, int64_t ws_floats, int64_t M, int32_t KA, int32_t NC, void* stream) {
  MMA_REQUIRE(M >= 1 && M < (1LL << 30) && KA >= 1 && KA <= 256 && NC >= 1 && (int64_t)KA * NC < (1LL << 31),
              "M=%lld KA=%d NC=%d: need M < 2^30, 1 <= KA <= 256, NC >= 1", (long long)M, KA, NC);
  MMA_REQUIRE(X && G && C && ws && ldx >= KA && ldg >= NC && ldx < (1 << 24) && ldg < (1 << 24), "NULL argument or row pitch out of range");
  MMA_REQUIRE((reinterpret_cast<uintptr_t>(X) & 3) == 0 && (reinterpret_cast<uintptr_t>(G) & 3) == 0 &&
              (reinterpret_cast<uintptr_t>(ws) & 15) == 0, "misaligned argument");
  const int s = tn_splits(M, NC);
  const int64_t n_part = s > 1 ? ((int64_t)s * KA * NC + 3) / 4 * 4 : 0, Mp = tn_pad32(M);   // whole 16-byte units: the scale vectors behind it are read as such
  const bool px = tn_packx(M, KA, NC);
  const int64_t n_need = n_part + 3 * Mp + 4 + (px ? tn_packx_floats(M, KA) : 0);
  MMA_REQUIRE(ws_floats >= n_need, "workspace too small: %lld floats, need %lld", (long long)ws_floats, (long long)n_need);
  int64_t rps = (M + s - 1) / s;
  rps = (rps + kTnKC - 1) / kTnKC * kTnKC;
  MMA_REQUIRE((rps + kTnKC) * (ldx > ldg ? ldx : ldg) * 4 < (1LL << 31), "row range of one split exceeds a 2 GB buffer window");
  const int nb = (KA + 31) / 32;
  MMA_REQUIRE(!px || (rps / kTnKC + 1) * (int64_t)nb * 4096 < (1LL << 31), "packed rows of one split exceed a 2 GB buffer window");
  float* xmax = ws + n_part;
  float* gmax = xmax + Mp;
  uint16_t* sxh = reinterpret_cast<uint16_t*>(gmax + Mp);
  uint16_t* sgh = sxh + Mp;
  int* state = reinterpret_cast<int*>(gmax + 2 * Mp);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tn_state_init_kernel, dim3(1), dim3(64), 0, st, state);       // a kernel, not a memset node (see csr_prepare_kernel, [r4])
  if (!x_row_max) { launch_row_absmax(X, ldx, M, KA, xmax, st); x_row_max = xmax; }
  if (!g_row_max) { launch_row_absmax(G, ldg, M, NC, gmax, st); g_row_max = gmax; }
  hipLaunchKernelGGL(tn_scale_range_kernel, dim3((unsigned)std::min<int64_t>((M + 255) / 256, 256)), dim3(256), 0, st, x_row_max, g_row_max, M,
                     state);
  hipLaunchKernelGGL(tn_scale_rows_kernel, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, st, x_row_max, g_row_max, M, Mp, state, sxh, sgh);
  if (int rc = check_launch("tn_scale_rows_kernel")) return rc;
  TnParams p{X, ldx, G, ldg, s == 1 ? C : ws, M, rps, KA, NC, s, sxh, sgh, state, 0};
  const dim3 grid((unsigned)(((NC + 127) / 128) * s));
  const bool full = NC % 128 == 0 && KA % 32 == 0;
  if (px) {
    u32x4* xp = reinterpret_cast<u32x4*>(ws + n_part + 3 * Mp + 4);
    const int64_t n_items = Mp / 32 * nb;
    hipLaunchKernelGGL(tn_pack_x_kernel, dim3((unsigned)std::min<int64_t>((n_items + 3) / 4, 8 * 2048)), dim3(256), 0, st, X, ldx, M, (int)KA, sxh,
                       xp, nb, n_items);
    p.xp = xp; p.xp_nb = nb;
  }
  if (KA > 128) {                                                    // [r4] X up to 256 columns wide: eight waves on one staged G tile
    if (px) { if (full) hipLaunchKernelGGL((gemm_f16x2_tn_kernel<true, 8, true>), grid, dim3(512), 0, st, p);
              else hipLaunchKernelGGL((gemm_f16x2_tn_kernel<false, 8, true>), grid, dim3(512), 0, st, p); }
    else if (full) hipLaunchKernelGGL((gemm_f16x2_tn_kernel<true, 8>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_f16x2_tn_kernel<false, 8>), grid, dim3(512), 0, st, p);
  } else {
    if (px) { if (full) hipLaunchKernelGGL((gemm_f16x2_tn_kernel<true, 4, true>), grid, dim3(kBlock), 0, st, p);
              else hipLaunchKernelGGL((gemm_f16x2_tn_kernel<false, 4, true>), grid, dim3(kBlock), 0, st, p); }
    else if (full) hipLaunchKernelGGL((gemm_f16x2_tn_kernel<true, 4>), grid, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((gemm_f16x2_tn_kernel<false, 4>), grid, dim3(kBlock), 0, st, p);
  }
  p.want_bad = 1;                                                    // the six-product form takes over when the scale kernels said so
  for (int j = 0; j < KA; j += 128) {                                // (it holds 128 columns of X per launch: the same partial tiles)
    TnParams q = p;
    q.X = X + j; q.KA = KA - j < 128 ? KA - j : 128;
    q.part = p.part + (size_t)j * NC; q.part_ss = (int64_t)KA * NC;
    const bool fq = NC % 128 == 0 && q.KA % 32 == 0;
    if (fq) hipLaunchKernelGGL(gemm_x3_tn_kernel<true>, grid, dim3(kBlock), 0, st, q);
    else hipLaunchKernelGGL(gemm_x3_tn_kernel<false>, grid, dim3(kBlock), 0, st, q);
  }
  if (int rc = check_launch("gemm_f16x2_tn_kernel")) return rc;
  if (s == 1) return 0;
  return mma_col_sum(ws, (int64_t)KA * NC, s, KA * NC, C, nullptr, 0, stream);
}
