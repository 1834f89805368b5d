// fp32-accurate tall-skinny GEMM on the bf16 matrix cores ("bf16x3"):  C (M,N) = A (M,K) @ B (K,N),  M >> N, K.
//
// The dense pre/post transforms of the hot path (P = x Wtop, Q = x Wbot and dL/dx = g W^T) have a
// million rows and K, N in {128, 512}.  On gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate (157 TFLOP/s) and
// there is no xf32/TF32 path, while bf16 MFMA is 16x faster per instruction.  Each fp32 value is therefore split
// exactly into three bf16 pieces, a = a1 + a2 + a3 (8+8+8 mantissa bits), and the product is summed from the six
// piece products that matter,   a b ~= a1b3 + a2b2 + a3b1 + a1b2 + a2b1 + a1b1   (dropped terms < 2^-24 |ab|),
// every piece product being exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  The result is as accurate as
// an fp32 GEMM (measured max error 2.4e-7 vs 3.0e-7 for the fp32 library GEMM, both relative to sum|a||b| against
// fp64) at 6/16 of its matrix-core time.  Measured (C4, M = 2^20): 1.06 ms per (M,128)x(128,512) or (M,512)x(512,128)
// product vs 1.3-1.45 ms for rocBLAS fp32; PMC: MFMA pipe 51 % busy at a 1.6 GHz effective clock - the A load/split,
// MFMA and C-store phases of a workgroup still run back to back (ablation: 0.2 + 0.5 + 0.35 ms), which is the next
// thing to fix (LDS-DMA slabs to free 24 VGPRs for a second accumulator so stores drain under the next tile's MFMAs).
//
// Layout: one workgroup = 4 waves = 128 rows of A; a wave keeps its 32 rows x 128 k of A in registers, already split
// (3 x 8 k-steps x 8 bf16 = 96 VGPRs) and walks the 32-column tiles of B; the three bf16 pieces of the B tile
// ([piece][col][k], k contiguous, rows padded by 16 B against LDS bank conflicts) are staged through a double-buffered
// LDS slab shared by the 4 waves, prefetched one tile ahead.  K > 128 is processed in chunks of 128 with the (at most
// four) accumulator tiles kept in registers, so either K == 128 or N <= 128 is required.
#include "common.h"

namespace mma {

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
template <int NCT, bool FULL>
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
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                     \
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;                                         \
          if (FULL || row < p.M) cp[row * p.ldc] = C_[r];                                                    \
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

// split a row-major fp32 matrix (rows, cols) into its three bf16 pieces: out (3, rows, cols)
__global__ void split3_kernel(const float* in, int64_t n, __bf16* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const Bf3 t = split3(in[i]);
    out[i] = t.a; out[n + i] = t.b; out[2 * n + i] = t.c;
  }
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
                               void* stream) {
  MMA_REQUIRE(M >= 0 && N >= 32 && K >= kKC && N % 32 == 0 && K % kKC == 0, "M=%lld N=%d K=%d: need N %% 32 == 0, K %% 128 == 0",
              (long long)M, N, K);
  MMA_REQUIRE(K == kKC || N <= 128, "either K == 128 or N <= 128 (accumulator tiles live in registers)");
  MMA_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0, "row pitch too small or unaligned");
  if (M == 0) return 0;
  MMA_REQUIRE(A && Bt3 && C && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bt3) & 15) == 0,
              "NULL or misaligned argument");
  GemmParams p{A, lda, static_cast<const __bf16*>(Bt3), C, ldc, M, N, K};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nct = K == kKC ? 0 : N / 32;
  const int64_t n_full = M / 128;
  for (int part = 0; part < 2; ++part) {
    const bool full = part == 0;
    const int64_t blk0 = full ? 0 : n_full, nb = full ? n_full : ((M % 128) ? 1 : 0);
    if (nb == 0) continue;
    const dim3 grid((unsigned)(nb < 512 ? nb : 512));      // 2 workgroups per CU (52 KB LDS, ~250 VGPRs each)
#define MMA_X3_LAUNCH(NCT_)                                                                                       \
    if (full) hipLaunchKernelGGL((gemm_x3_kernel<NCT_, true>), grid, dim3(kBlock), 0, st, p, blk0, nb);           \
    else hipLaunchKernelGGL((gemm_x3_kernel<NCT_, false>), grid, dim3(kBlock), 0, st, p, blk0, nb);
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
