// K13 / K14: MMAConv's per-tower post-NN Linear on the aggregates with the degree scalers FACTORED OUT of the aggregate tensor
// (reference graph_regression/mma_conv.py:181-196 builds out = cat_q(agg * prod_{q' <= q} scaler_q'(deg)) of width S*K*F per tower,
// then :132-134 applies post_nns[t] = Linear((K*S+1)*F -> F_out) to cat[x, out]).  The scalers are per-TARGET row factors, so
//     y[n,t,o] = sum_q pre_q(deg_n) * sum_{k,f} agg[n,t,k,f] * Wo[t][o][(q*K+k)*F+f],     pre_q = prod_{q' <= q} scaler_q'
// and neither the (N,T,S*K*F) tensor `out` nor its gradient ever exists: K3 runs with the identity scaler only and leaves the K
// unscaled aggregates (N,T,K*F) - a third of the bytes at ZINC's S = 3 -, K4 takes their gradient.
//   K13  mma_tower_post_fwd : y from agg                                  (replaces the strided-batched library GEMM)
//   K14  mma_tower_post_bwd : gagg[n,t,kf] = sum_q pre_q sum_o gy[n,t,o] Wo[..], and gys[n,t,q,o] = pre_q gy[n,t,o], the left operand
//                             of the weight gradient gW[t][q][o][kf] = sum_n gys[n,t,q,o] agg[n,t,kf] (a TN product on the matrix
//                             cores: mma_gemm_bf16x3_tn per tower)                (replaces K9 tower_bwd_kernel)
// Both are small GEMMs per (64 nodes, tower) with a 16-wide output dimension per scaler - run on the matrix cores in EXACT fp32:
// v_mfma_f32_16x16x4_f32 (one rounding per product, fp32 accumulate: a k-ordered fmaf chain; 64 FLOP/clk/SIMD like the VALU, but
// one VGPR per operand and no broadcast traffic).  A 16-row tile IS one scaler q (rows = its 16 padded outputs), so S*16 rows tile
// exactly; a wavefront owns 64 nodes = four 16-column tiles.
//   forward : D_q[o][node] = sum_kf Wa[kf][q*16+o] * agg[node][kf];  epilogue y[o] = sum_q pre_q[node] D_q[o][node]: the C/D map puts
//             the node on the lane (column) and 4 consecutive o in its 4 registers - the scaler sum never leaves the lane;
//   backward: D[kf][node] = sum_r Wb[r][kf] * gys[node][r], r = (q,o): the B fragments are the lane's own scaled gradients (registers).
// The tower's weights sit in LDS for the whole workgroup (30 KB at ZINC's shape), aggregate rows enter / leave as whole 128-byte row
// segments through a wave-private LDS tile.  (A first version kept lane = node on the VALU with the 16 weights of a (q, kf) as SGPR
// operands from s_load_dwordx16: 0.41 / 1.02 ms at C2L - the 29 KB weight sweep of every wave thrashes the 16 KB scalar cache.)
#include <algorithm>
#include <cstdlib>
#include "common.h"

namespace mma {

constexpr int kPostO = 16;        // outputs per tower, padded (ZINC: 15)
constexpr int kPostMaxS = 5;      // scalers (identity, amplification, attenuation, linear, inverse_linear)
constexpr int kPostTile = 32;     // kf columns per LDS tile round
constexpr int kPostPrePitch = 8;   // floats per node in the table of scaler products K14 leaves for K15
constexpr int kPostPitch = 34;    // tile pitch: (j*34 + k) mod 32 is distinct over the 16 x 2 lanes one ds_read_b32 group covers
typedef float post_f32x4 __attribute__((ext_vector_type(4)));
// Measurement builds only (tools/build_ablation.sh post <bits>: -DMMA_POST_ABL, loaded through MMA_LIB_OVERRIDE): 1 = no MFMAs, 2 = no B loads
// after the first step, 4 = no split (the raw bits as pieces), 16 = no MFMAs in the fp32 forward.  0 in the product library.
#ifndef MMA_POST_ABL
#define MMA_POST_ABL 0
#endif
constexpr int kPostAbl = MMA_POST_ABL;

__device__ __forceinline__ float post_scaler(int code, float deg, float avg_log, float avg_lin) {
  switch (code) {       // mma_conv.py:183-192
    case MMA_SC_AMPLIFICATION: return logf(deg + 1.f) / avg_log;
    case MMA_SC_ATTENUATION: return avg_log / logf(deg + 1.f);
    case MMA_SC_LINEAR: return deg / avg_lin;
    case MMA_SC_INVERSE_LINEAR: return avg_lin / deg;
    default: return 1.f;
  }
}

struct PostParams {
  int64_t N; int T, KF, KFp, S, O;          // KFp: KF rounded up to kPostTile (the padded weight rows / columns are zero)
  int64_t lda, ldy, ldg, ldgs;
  uint32_t scaler_pack; float avg_log, avg_lin;
  int tiles_per_wave;       // 64-node tiles a wavefront walks with the tower's weights staged once
  int vec4;                 // rows of agg / gagg are 16-byte aligned (pitch % 4 == 0, KF % 4 == 0): float4 row segments; else dwords
  const float* bias;        // PLAIN forward only, may be NULL
  const float* addend; int64_t ldadd;   // PLAIN forward only, may be NULL: y = (x W^T + bias) + addend (the post-NN's two halves meet in the epilogue)
  int order;                // workgroup -> (tower, node block) map of post_block(): 0 XCD-grouped [r5], 1 tower-major (round 3), 2 tower-fastest (round 4)
  int64_t nbx;              // node blocks per tower (the grid holds nbx rounded up to a multiple of 8, times T, in order 0)
};

// Workgroup -> (tower, node block).  [r4] TOWER FASTEST: the workgroups that run at the same time read (K13, K15) or write (K14) whole
// rows of agg - T x 608 bytes at C2L - instead of every fifth 608-byte run of them.  With the tower on blockIdx.y the chip walked ONE
// tower's column block of all rows at a time: 20 % of every DRAM page it opened, and no kernel of the family moved more than 3.7 TB/s.
// [r5] XCD-GROUPED: consecutive block ids are dealt round-robin over the 8 XCDs, so under the tower-fastest order the T workgroups of one
// node block sat on T different XCDs - and a tower's 608-byte run of a row starts and ends inside 128-byte lines it shares with its
// neighbours (608 = 4.75 lines; rows are 3040 bytes apart): every L2 fetched those lines again (PMC: 1.39x / 1.41x / 1.82x the algorithmic
// bytes for K13 / K14 / K15).  Now a group of 8 T consecutive ids covers 8 node blocks, id i of the group -> node block i % 8, tower i / 8:
// all towers of a node block have the same id mod 8 = the same XCD (speed only: nothing depends on the placement), a few ids apart.
// false: this workgroup is padding of the last group.
__device__ __forceinline__ bool post_block(const PostParams& p, int& t, int64_t& bx) {
  const unsigned lin = blockIdx.x, T = (unsigned)p.T;
  if (p.order == 1) { const unsigned nbx = gridDim.x / T; t = (int)(lin / nbx); bx = lin % nbx; }
  else if (p.order == 2) { t = (int)(lin % T); bx = lin / T; }
  else { const unsigned g = lin / (8u * T), i = lin - g * 8u * T; t = (int)(i >> 3); bx = (int64_t)g * 8 + (i & 7u); }
  return bx < p.nbx;
}

// running products of the scalers for one node, in the reference's multiplication order ((v f0) f1) ...
template <int S>
__device__ __forceinline__ void post_pre(const PostParams& p, const int32_t* __restrict__ rowptr, int64_t node, bool valid, float (&pre)[S]) {
  const float deg = valid ? (float)max(rowptr[node + 1] - rowptr[node], 1) : 1.f;      // degree(...).clamp_(1), mma_conv.py:178-179
  float run = 1.f;
#pragma unroll
  for (int q = 0; q < S; ++q) {
    run = run * post_scaler((int)((p.scaler_pack >> (4 * q)) & 15u), deg, p.avg_log, p.avg_lin);
    pre[q] = run;
  }
}

// The aggregate tiles are WAVE-private: a wave's LDS accesses execute in order, so its own writes are visible to its own later reads
// without a workgroup barrier; only the compiler has to keep the order.  (With __syncthreads() per tile round the four waves load,
// multiply and store in lockstep - nobody multiplies while anybody loads: 0.35 ms; free-running waves overlap each other's phases.)
__device__ __forceinline__ void post_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void post_stage_weights(float* dst, const float* __restrict__ src, int n_floats) {
  // eight loads in flight per thread, then eight LDS stores: as a plain copy loop every 16-byte piece was a load -> wait -> store round
  // trip (30 of them per thread for ZINC's 30 KB: most of a workgroup's life, PMC: the matrix pipe busy 45 % of the kernel)
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  const int n4 = n_floats / 4;
  for (int i0 = threadIdx.x; i0 < n4; i0 += kBlock * 8) {
    float4 v[8];           // indices past the end are clamped for the load AND the store (every such lane rewrites the last piece with its own value)
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s4[min(i0 + u * kBlock, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) d4[min(i0 + u * kBlock, n4 - 1)] = v[u];
  }
}

// rows n0..n0+63, columns [kf0, kf0+32) of tower t -> registers (coalesced 128-byte row segments; 8 lanes per node row) -> the wave's LDS tile.
// Unconditional loads from clamped addresses, zeroed by selects (a conditional load is an exec branch of its own: eight of them in a row
// kept the scheduler from batching the tile's loads)
template <bool VEC4>
__device__ __forceinline__ void post_tile_fetch(const PostParams& p, const float* __restrict__ agg, int64_t n0, int t, int kf0, int lane,
                                                float4 (&v)[8]) {
  const int lrow = lane >> 3, lq = lane & 7;
  const int col = kf0 + lq * 4;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int64_t row = min(n0 + r * 8 + lrow, p.N - 1);
    const float* base = agg + (size_t)row * p.lda + (size_t)t * p.KF;
    if (VEC4) {                                                // KF % 4 == 0: a quad is inside or outside
      v[r] = *reinterpret_cast<const float4*>(base + min(col, p.KF - 4));
    } else {                                                   // odd widths / pitches (the 75-wide Linear layers): dwords
      v[r].x = base[min(col, p.KF - 1)]; v[r].y = base[min(col + 1, p.KF - 1)];
      v[r].z = base[min(col + 2, p.KF - 1)]; v[r].w = base[min(col + 3, p.KF - 1)];
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const bool rv = n0 + r * 8 + lrow < p.N;
    v[r].x = (rv && col < p.KF) ? v[r].x : 0.f; v[r].y = (rv && col + 1 < p.KF) ? v[r].y : 0.f;
    v[r].z = (rv && col + 2 < p.KF) ? v[r].z : 0.f; v[r].w = (rv && col + 3 < p.KF) ? v[r].w : 0.f;
  }
}
__device__ __forceinline__ void post_tile_put(float* tile, int lane, const float4 (&v)[8]) {
  const int lrow = lane >> 3, lq = lane & 7;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float2* d = reinterpret_cast<float2*>(tile + (r * 8 + lrow) * kPostPitch + lq * 4);   // pitch 34: 8-byte aligned
    d[0] = make_float2(v[r].x, v[r].y); d[1] = make_float2(v[r].z, v[r].w);
  }
}

// the scaler products of every node, once per call: pre[n][q] = prod_{q' <= q} scaler_q'(clamp(deg_n, 1)) - K13, K14 and K15 read the table
// (each re-evaluating them - a logarithm, two divisions and a switch per scaler - put a chain of ~60 branches into every epilogue)
template <int S>
__global__ __launch_bounds__(kBlock) void tower_post_pre_kernel(const PostParams p, const int32_t* __restrict__ rowptr, float* __restrict__ pre_out) {
  const int64_t node = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (node >= p.N) return;
  float pre[S];
  post_pre<S>(p, rowptr, node, true, pre);
#pragma unroll
  for (int q = 0; q < S; ++q) pre_out[(size_t)node * kPostPrePitch + q] = pre[q];
}

// y[n, t*O + o] = sum_q pre_q sum_kf agg[n, t*KF + kf] * Wa[t][kf][q*16 + o]
// PLAIN: a plain skinny Linear, y[n, r] = bias[r] + sum_kf x[n, kf] * Wa[kf][r] for r < O <= S*16 - every 16-row tile is 16 more
// outputs instead of one more scaler (the 75 -> 75 Linear layers around the fused kernels: x-part of the post-NN, `lin`)
template <int S, bool PLAIN, bool VEC4>
__global__ __launch_bounds__(kBlock) void tower_post_fwd_kernel(const PostParams p, const float* __restrict__ agg, const float* __restrict__ pre_tab,
                                                                const float* __restrict__ Wa, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float post_smem[];
  constexpr int R = S * kPostO;                                // weight columns per kf row
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* Wl = post_smem;                                       // (KFp, R)
  float* tile = post_smem + (size_t)p.KFp * R + wave * (kWave * kPostPitch);
  int t; int64_t bx;
  if (!post_block(p, t, bx)) return;                           // (before any barrier: the whole workgroup leaves)
  post_stage_weights(Wl, Wa + (size_t)t * p.KFp * R, p.KFp * R);
  __syncthreads();                                             // weights staged: the only workgroup barrier
  const int j = lane & 15, kq = lane >> 4;                     // MFMA 16x16x4 operand lane: row/column j, k index kq
  for (int rt = 0; rt < p.tiles_per_wave; ++rt) {
  const int64_t nblk = (bx * (kBlock / kWave) + wave) * p.tiles_per_wave + rt;
  const bool tvalid = nblk * kWave < p.N;                      // wave-uniform
  if (!tvalid) break;
  const int64_t n0 = nblk * kWave;
  post_f32x4 acc[4][S];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < S; ++q) acc[nt][q] = post_f32x4{0.f, 0.f, 0.f, 0.f};

  float4 nxt[8];
  post_tile_fetch<VEC4>(p, agg, n0, t, 0, lane, nxt);
  for (int kf0 = 0; kf0 < p.KFp; kf0 += kPostTile) {
    post_tile_put(tile, lane, nxt);
    post_wave_sync();
    if (kf0 + kPostTile < p.KFp) post_tile_fetch<VEC4>(p, agg, n0, t, kf0 + kPostTile, lane, nxt);   // in flight behind the MFMAs
#pragma unroll
    for (int ks = 0; ks < kPostTile / 4; ++ks) {
      const int kf = kf0 + 4 * ks + kq;
      float a[S], b[4];
#pragma unroll
      for (int q = 0; q < S; ++q) a[q] = Wl[kf * R + q * kPostO + j];               // A[i = o][k = kf]
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) b[nt] = tile[(nt * 16 + j) * kPostPitch + 4 * ks + kq];   // B[k = kf][j = node]
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < S; ++q) {
          if (kPostAbl & 16) { asm volatile("" : "+v"(acc[nt][q]) : "v"(a[q]), "v"(b[nt])); }
          else acc[nt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[nt], acc[nt][q], 0, 0, 0);
        }
    }
    post_wave_sync();
  }
  // C/D map of 16x16: column = lane & 15 (the node), row = 4 * (lane >> 4) + reg (the output o): y[o] = sum_q pre_q D_q[o]
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int64_t node = n0 + nt * 16 + j;
    const bool valid = tvalid && node < p.N;
    if (PLAIN) {
      // every load of the tile (bias, addend) ahead of its first store: a load behind a store waits for the store's acknowledgement
      float bvv[S][4], adv[S][4];
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = q * kPostO + 4 * kq + r;
          bvv[q][r] = (valid && col < p.O && p.bias) ? p.bias[col] : 0.f;
          adv[q][r] = (valid && col < p.O && p.addend) ? p.addend[(size_t)node * p.ldadd + col] : 0.f;
        }
      if (valid) {
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = q * kPostO + 4 * kq + r;
            if (col < p.O) {
              const float v = acc[nt][q][r] + bvv[q][r];
              y[(size_t)node * p.ldy + col] = p.addend ? v + adv[q][r] : v;
            }
          }
      }
      continue;
    }
    float pre[S];
#pragma unroll
    for (int q = 0; q < S; ++q) pre[q] = pre_tab[(size_t)min(node, p.N - 1) * kPostPrePitch + q];      // the table of mma_tower_post_pre
    float yv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < S; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) yv[r] = fmaf(pre[q], acc[nt][q][r], yv[r]);
    if (valid) {
      float* yr = y + (size_t)node * p.ldy + (size_t)t * p.O + 4 * kq;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * kq + r < p.O) yr[r] = yv[r];
    }
  }
  }   // tiles of this wave
}

// gagg[n, t*KF + kf] = sum_r gys[n][r] * Wb[t][r][kf],  r = q*16 + o,  gys[n][r] = pre_q gy[n, t*O + o]   (also stored, optional)
// PLAIN: gx[n, kf] = sum_r gy[n, r] * Wb[r][kf], r < O <= S*16 (the dL/dx of the plain skinny Linear)
template <int S, bool PLAIN>
__global__ __launch_bounds__(kBlock) void tower_post_bwd_kernel(const PostParams p, const float* __restrict__ gy, const float* __restrict__ pre_tab,
                                                                const float* __restrict__ Wb, int wb_pitch, float* __restrict__ gagg,
                                                                float* __restrict__ gys) {
  extern __shared__ __attribute__((aligned(16))) float post_smem[];
  constexpr int R = S * kPostO;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* Wl = post_smem;                                       // (R, wb_pitch): pitch = KFp + 16, so that k and k+1 fall on different banks
  float* tile = post_smem + (size_t)R * wb_pitch + wave * (kWave * kPostPitch);
  int t; int64_t bx;
  if (!post_block(p, t, bx)) return;                           // (before any barrier: the whole workgroup leaves)
  post_stage_weights(Wl, Wb + (size_t)t * R * wb_pitch, R * wb_pitch);
  __syncthreads();               // weights staged: the only workgroup barrier
  const int j = lane & 15, kq = lane >> 4;
  // [r5] the raw inputs of a tile (this lane's four gy values and S scaler products per 16-node tile: 28 dwords) are requested ONE TILE
  // AHEAD, by unconditional loads from clamped addresses (zeroed by selects at use): a tile is 5 x 96 MFMAs = 6.4 us, its inputs used to be
  // requested and waited for at its start - 2 us of every 8.4 with nothing to multiply (SQ: 52 % of the wave cycles waiting on instruction
  // dependencies).  PLAIN: R/4 dwords of gy per node tile.
  constexpr int NG = PLAIN ? R / 4 : 4;
  constexpr int NSETS = PLAIN ? 1 : 2;                         // PLAIN (K16: one tile per wave at C2L, 20 dwords per node tile): no second set
  float rg[NSETS][4][NG], rp[NSETS][4][PLAIN ? 1 : S];
  auto fetch = [&](int rt_, float (&g_)[4][NG], float (&p_)[4][PLAIN ? 1 : S]) {
    const int64_t nb_ = (bx * (kBlock / kWave) + wave) * p.tiles_per_wave + rt_;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int64_t node = min(nb_ * kWave + nt * 16 + j, p.N - 1);
      if (PLAIN) {
#pragma unroll
        for (int s = 0; s < NG; ++s) g_[nt][s] = gy[(size_t)node * p.ldg + min(4 * s + kq, p.O - 1)];
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) g_[nt][m] = gy[(size_t)node * p.ldg + (size_t)t * p.O + min(4 * m + kq, p.O - 1)];
#pragma unroll
        for (int q = 0; q < (PLAIN ? 1 : S); ++q) p_[nt][q] = pre_tab[(size_t)node * kPostPrePitch + q];
      }
    }
  };
  if (!PLAIN) fetch(0, rg[0], rp[0]);
#pragma unroll 1
  for (int rt2 = 0; rt2 < p.tiles_per_wave; rt2 += NSETS) {
#pragma unroll
  for (int par = 0; par < NSETS; ++par) {                       // two tiles per trip: the raw sets alternate without register copies
  const int rt = rt2 + par;
  if (rt >= p.tiles_per_wave) break;
  const int64_t nblk = (bx * (kBlock / kWave) + wave) * p.tiles_per_wave + rt;
  const bool tvalid = nblk * kWave < p.N;
  if (!tvalid) break;
  const int64_t n0 = nblk * kWave;
  constexpr int ps = PLAIN ? 0 : 1;                             // set index = par * ps
  if (PLAIN) fetch(rt, rg[0], rp[0]);
  // B fragments: this lane's node (per 16-node tile nt) and its k index kq: gys[node][4*s + kq], s = 0 .. R/4 - 1
  float bf[4][R / 4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int64_t node = n0 + nt * 16 + j;
    const bool valid = tvalid && node < p.N;
    if (PLAIN) {
#pragma unroll
      for (int s = 0; s < R / 4; ++s) bf[nt][s] = (valid && 4 * s + kq < p.O) ? rg[0][nt][s % NG] : 0.f;
      continue;
    }
    // r = 4*s + kq  ->  o = r % 16 = 4*(s % 4) + kq,  q = s / 4: the lane needs gy[o] for o = kq, 4 + kq, 8 + kq, 12 + kq
    float g4[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) g4[m] = (valid && 4 * m + kq < p.O) ? rg[par * ps][nt][m % NG] : 0.f;
#pragma unroll
    for (int s = 0; s < R / 4; ++s) bf[nt][s] = rp[par * ps][nt][(s / 4) % (PLAIN ? 1 : S)] * g4[s % 4];
    if (gys && valid) {          // the left operand of the weight-gradient product; 4-byte stores, 16 lanes of a node tile cover 64 B runs
#pragma unroll
      for (int s = 0; s < R / 4; ++s) gys[(size_t)node * p.ldgs + (size_t)t * R + 4 * s + kq] = bf[nt][s];
    }
  }
  if (!PLAIN && rt + 1 < p.tiles_per_wave) fetch(rt + 1, rg[(1 - par) * ps], rp[(1 - par) * ps]);       // in flight behind this tile's MFMAs
  for (int kf0 = 0; kf0 < p.KFp; kf0 += kPostTile) {
    post_f32x4 acc[2][4];        // two 16-row kf tiles x four node tiles
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[c][nt] = post_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < R / 4; ++s) {
      const float* wr = Wl + (size_t)(4 * s + kq) * wb_pitch + kf0 + j;              // A[i = kf][k = r]
      const float a0 = wr[0], a1 = wr[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf[nt][s], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf[nt][s], acc[1][nt], 0, 0, 0);
      }
    }
    // D: column = the node (lane & 15), row = 4 * kq + reg = the kf inside the 16-row tile  ->  tile[node][kf_local]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(nt * 16 + j) * kPostPitch + c * 16 + 4 * kq + r] = acc[c][nt][r];
    post_wave_sync();
    {
      const int lrow = lane >> 3, lq = lane & 7;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = r * 8 + lrow;
        const int col = kf0 + lq * 4;
        if (tvalid && n0 + row < p.N && col < p.KF) {
          const float* sp = tile + row * kPostPitch + lq * 4;
          float* dst = gagg + (size_t)(n0 + row) * p.lda + (size_t)t * p.KF + col;
          if (p.vec4) {
            *reinterpret_cast<float4*>(dst) = make_float4(sp[0], sp[1], sp[2], sp[3]);
          } else {
            dst[0] = sp[0];
            if (col + 1 < p.KF) dst[1] = sp[1];
            if (col + 2 < p.KF) dst[2] = sp[2];
            if (col + 3 < p.KF) dst[3] = sp[3];
          }
        }
      }
    }
    post_wave_sync();
  }
  }   // the pair
  }   // tiles of this wave
}

// K15: the weight gradient of the factored post-NN, gW[t][q*16 + o][kf] = sum_n pre_q(deg_n) gy[n, t*O + o] agg[n, t*KF + kf], as a TN
// product on the fp32 matrix cores with the NODE as the reduction index (v_mfma_f32_16x16x4_f32: four nodes per instruction).  A wave
// walks its own contiguous node range and keeps S x G accumulator tiles (G = 5 or 10 kf tiles of 16 columns per pass) for the
// whole range; both operands come straight from memory in the MFMA's own lane order - A[i = o][k = node] = pre_q gy (16 lanes read 64
// consecutive bytes of a gy row, the scalers are re-evaluated per node), B[k = node][j = kf] = agg (64-byte runs of four rows) - no
// LDS, no scaled copy of gy (round 3's first form wrote gys (N, T*S*16) and ran five TN launches of the bf16x3 kernel: 0.4-0.5 ms).
// The four waves' tiles of a workgroup are added through LDS, the per-workgroup partial tiles are summed in a fixed order by
// mma_tower_post_gw_reduce / mma_col_sum.
// PLAIN [r5]: the weight AND bias gradient of a plain skinny Linear (K16) on the same kernel - gW[r][k] = sum_n gy[n][r] x[n][k] for r < O
// <= S*16, k < K, every 16-row tile 16 more outputs (A[i][k = node] = gy[node][q*16 + i], no scaler table), and the column k = K of the B
// operand is a column of ONES: gW[r][K] = sum_n gy[n][r] = the bias gradient, from the same pass (K < kfp16 is required for it).
template <int S, int G, bool PLAIN = false>
__global__ __launch_bounds__(kBlock) void tower_post_gw_kernel(const PostParams p, const float* __restrict__ gy, const float* __restrict__ agg,
                                                               const float* __restrict__ pre_tab, float* __restrict__ part, int64_t npw,
                                                               int kfp16) {
  // G: kf tiles per pass (post_gw_tiles): S*G accumulator tiles + two operand groups in flight
  constexpr int GH = PLAIN ? 1 : (G > 5 ? G / 2 : G);          // tiles per LDS reduction round (the LDS holds 4 waves x S x GH tiles; PLAIN: one
                                                               // tile per round - 20 KB at S = 5, several workgroups per CU)
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int t; int64_t bx;
  if (!post_block(p, t, bx)) return;                           // (before any barrier: the whole workgroup leaves)
  const int64_t chunk = bx * (kBlock / kWave) + wave;
  const int64_t nb = chunk * npw, ne = min(p.N, nb + npw);
  const int j = lane & 15, kq = lane >> 4;
  const int n_kft = kfp16 / 16;
  // [r4] the four waves of a workgroup add their tiles through LDS (fixed order (w0 + w1) + (w2 + w3)) and the workgroup writes ONE
  // partial tile: a quarter of the partials to write and to sum (C2L: 63 MB -> 16 MB; the reduction launch 0.052 -> 0.02 ms)
  extern __shared__ __attribute__((aligned(16))) float post_smem[];
  float* out = part + ((size_t)bx * p.T + t) * (size_t)(S * kPostO) * kfp16;
  const int jo = j < p.O ? j : 0;                              // clamped column of gy (the value is zeroed by a select)
  const int64_t nlast = p.N - 1;
  for (int kt0 = 0; kt0 < n_kft; kt0 += G) {
    post_f32x4 acc[S][G];
#pragma unroll
    for (int q = 0; q < S; ++q)
#pragma unroll
      for (int c = 0; c < G; ++c) acc[q][c] = post_f32x4{0.f, 0.f, 0.f, 0.f};
    // The raw operands of a k-step (4 nodes) are requested a whole step AHEAD of the MFMAs that use them, by UNCONDITIONAL loads from
    // clamped addresses (zeroed by selects at use): a conditional load sits in its own exec branch and the first dependent use - the
    // row pointers of the first version, whose scaler products were re-evaluated here - made the wave wait for every load it had just
    // issued (one memory round trip per k-step: 0.43 ms, the matrix pipe idle three quarters of the time).  The scaler products come
    // from the table K14 wrote.
    struct Raw { float g; float pre[S]; float b[G]; bool valid; };
    auto fetch = [&](int64_t n4, Raw& r) {
      const int64_t node = n4 + kq;
      r.valid = node < ne;
      const int64_t nc = node < nlast ? node : nlast;
      if (PLAIN) {                                              // r.pre[q] carries gy[node][q*16 + j] (clamped column, zeroed at use)
        r.g = 1.f;
#pragma unroll
        for (int q = 0; q < S; ++q) r.pre[q] = gy[(size_t)nc * p.ldg + min(q * kPostO + j, p.O - 1)];
      } else {
        r.g = gy[(size_t)nc * p.ldg + (size_t)t * p.O + jo];
#pragma unroll
        for (int q = 0; q < S; ++q) r.pre[q] = pre_tab[(size_t)nc * kPostPrePitch + q];
      }
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const int col = min((kt0 + c) * 16 + j, p.KF - 1);
        r.b[c] = agg[(size_t)nc * p.lda + (size_t)t * p.KF + col];
      }
    };
    auto multiply = [&](const Raw& r) {
      const float g = (r.valid && j < p.O) ? r.g : 0.f;
      float b[G];
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const int col = (kt0 + c) * 16 + j;
        b[c] = (r.valid && col < p.KF) ? r.b[c] : ((PLAIN && r.valid && col == p.KF) ? 1.f : 0.f);      // PLAIN: the ones column behind x
      }
#pragma unroll
      for (int q = 0; q < S; ++q) {
        const float a = PLAIN ? ((r.valid && q * kPostO + j < p.O) ? r.pre[q] : 0.f) : r.pre[q] * g;
#pragma unroll
        for (int c = 0; c < G; ++c) acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], acc[q][c], 0, 0, 0);
      }
    };
    // [r5] FOUR operand sets in a ring, each requested four k-steps (16 nodes) ahead of its MFMAs.  A k-step is S*G MFMAs = 0.4 us at
    // ZINC's shape against a loaded memory latency of 2 us and more, and a SIMD holds one or two of these wavefronts: with round 4's two
    // sets (one step ahead) the matrix pipe sat idle two thirds of the time (0.32 ms against 0.10 ms of MFMAs; one pass or two over the
    // nodes, XCD placement of the towers, 1.8x or 1.0x the algorithmic bytes in the L2 fetch counters - none of it moved the time).
    // No register copies: a copy of a set that is still in flight is a wait for it in the middle of the MFMAs.
    Raw r0, r1, r2, r3;
    fetch(nb, r0); fetch(nb + 4, r1); fetch(nb + 8, r2); fetch(nb + 12, r3);
    for (int64_t n4 = nb; n4 < ne; n4 += 16) {                 // past the range: valid is false, the operands are zeros
      __builtin_amdgcn_sched_barrier(0);                       // the scheduler otherwise sinks the loads to their first use
      multiply(r0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(n4 + 16, r0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(r1);
      __builtin_amdgcn_sched_barrier(0);
      fetch(n4 + 20, r1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(r2);
      __builtin_amdgcn_sched_barrier(0);
      fetch(n4 + 24, r2);
      __builtin_amdgcn_sched_barrier(0);
      multiply(r3);
      __builtin_amdgcn_sched_barrier(0);
      fetch(n4 + 28, r3);
    }
    // D: row = 4 * kq + reg = o, column = lane & 15 = kf inside the tile.  red[wave][(q, c, r)][lane], GH tiles of every q per round
#pragma unroll
    for (int h = 0; h < G; h += GH) {
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int c = 0; c < GH; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) post_smem[((wave * S + q) * GH + c) * 4 * kWave + r * kWave + lane] = acc[q][h + c][r];
      __syncthreads();
      constexpr int kItems = S * GH * 4, kWaveStride = S * GH * 4 * kWave;
      for (int it = wave; it < kItems; it += kBlock / kWave) {
        const int q = it / (GH * 4), c = (it >> 2) % GH, r = it & 3;
        const float* rp = post_smem + it * kWave + lane;
        const float v = (rp[0] + rp[kWaveStride]) + (rp[2 * kWaveStride] + rp[3 * kWaveStride]);
        if (kt0 + h + c < n_kft) out[(size_t)(q * kPostO + 4 * kq + r) * kfp16 + (kt0 + h + c) * 16 + j] = v;
      }
      __syncthreads();                                         // the next round / pass writes the same LDS
    }
  }
}

// ---- [r4] K13 on the bf16 matrix cores: every fp32 operand split exactly into three bf16 pieces, six piece products per k-step
// (gemm_x3.hip's scheme: as accurate as the fp32 product - the forward's largest error against float64 drops from 2.5e-7 to 1.4e-7 of
// sum|a||w| with the small products summed apart - and v_mfma_f32_16x16x32_bf16 runs at 16x the fp32 MFMA's rate, so six products cost
// 6/16 of it).  The k index of an MFMA is a summation index: ANY bijection between the (lane group, element) slots of the two operand
// fragments and the 32 k values of a step is allowed as long as both operands use the same one.  post_kmap gives lane group kg the k
// values {4kg..4kg+3} and {16+4kg..16+4kg+3}: the B fragment is then two float4 loads straight from the lane's own node row - no LDS tile,
// no wave barriers - and the weights, split once per workgroup while they are staged, sit in LDS in fragment order (one conflict-free
// ds_read_b128 per piece).  C2L (tools/post_micro.py): 0.240 -> 0.200 ms.
// What the measurement builds say about the rest: no kernel of this family is bound by its matrix-core time.  This
// kernel without its MFMAs 0.208 ms, without the split 0.21, without MFMAs AND split (loads + skeleton) 0.198, without its loads 0.123,
// without all three 0.075 (weight staging, A-fragment reads, epilogue); the fp32 kernel above without its MFMAs 0.185.  A ring of THREE
// k-steps in flight per wave (inline-asm loads, hand-counted waits, audited by tools/check_asm_waits.py --asm-only) 0.211-0.216: depth
// is not what is missing either; nor is the layout (one tower with contiguous 608-byte rows: 0.199).  The same forms of K14 and K15 were
// written, tested and measured - K14 0.30-0.34 ms against 0.27-0.28 for the fp32 kernel (its 64-byte store runs), K15 0.245 against
// 0.255 - and not kept.  The fp32 kernel stays for shapes whose split weights do not fit the LDS, for K16, and behind MMA_POST_EXACT=1
// (the A/B switch, read per call).
typedef __attribute__((ext_vector_type(8))) __bf16 post_bf16x8;
struct PostBf3 { __bf16 a, b, c; };
__device__ __forceinline__ PostBf3 post_split3(const float v) {
  PostBf3 r;
  r.a = (__bf16)v;
  const float r1 = v - (float)r.a;
  r.b = (__bf16)r1;
  r.c = (__bf16)(r1 - (float)r.b);
  return r;
}
struct PostFrag { post_bf16x8 p1, p2, p3; };
__device__ __forceinline__ void post_frag_set(PostFrag& f, int e, float v) {
  if (kPostAbl & 4) {
    const __bf16 h = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(v) >> 16));
    f.p1[e] = h; f.p2[e] = h; f.p3[e] = h;
    return;
  }
  const PostBf3 t = post_split3(v);
  f.p1[e] = t.a; f.p2[e] = t.b; f.p3[e] = t.c;
}
__device__ __forceinline__ constexpr int post_kmap(int kg, int e) { return e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4); }
// acc += a b from the six piece products that matter, smallest first
#define MMA_POST_X3(ACC, A, B, PROD)                                                                                     \
  if (kPostAbl & 1) { asm volatile("" : "+v"(ACC) : "v"(A.p1), "v"(A.p2), "v"(A.p3), "v"(B.p1), "v"(B.p2), "v"(B.p3)); } else                      \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PROD == 0 ? A.p1 : PROD == 1 ? A.p2 : PROD == 2 ? A.p3 : PROD == 3 ? A.p1 : PROD == 4 ? A.p2 : A.p1, \
                                                PROD == 0 ? B.p3 : PROD == 1 ? B.p2 : PROD == 2 ? B.p1 : PROD == 3 ? B.p2 : PROD == 4 ? B.p1 : B.p1, ACC, 0, 0, 0)

// forward weights of one tower, Wa (KFp, R) fp32 -> LDS [ks][q][piece][lane] fragments: lane (i = lane & 15, kg = lane >> 4), element e holds
// Wa[32 ks + kmap(kg, e)][16 q + i]
template <int S>
__device__ __forceinline__ void post_stage_x3_fwd(post_bf16x8* Wl, const float* __restrict__ Wa_t, int KFp) {
  constexpr int R = S * kPostO;
  constexpr int MU = 4;                                        // units per thread and pass: their 32 loads are ONE round trip, not four
  const int units = (KFp / 32) * S * kWave;
  for (int u0 = threadIdx.x; u0 < units; u0 += kBlock * MU) {
    float w[MU][8];
#pragma unroll
    for (int m = 0; m < MU; ++m) {
      const int u = min(u0 + m * kBlock, units - 1);           // past the end: the last unit again (not stored)
      const int lane = u & (kWave - 1), q = (u >> 6) % S, ks = (u >> 6) / S;
      const int i = lane & 15, kg = lane >> 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) w[m][e] = Wa_t[(size_t)(32 * ks + post_kmap(kg, e)) * R + q * kPostO + i];
    }
#pragma unroll
    for (int m = 0; m < MU; ++m) {
      const int u = u0 + m * kBlock;
      if (u < units) {
        const int lane = u & (kWave - 1), q = (u >> 6) % S, ks = (u >> 6) / S;
        PostFrag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) post_frag_set(f, e, w[m][e]);
        post_bf16x8* d = Wl + (size_t)((ks * S + q) * 3) * kWave + lane;
        d[0] = f.p1; d[kWave] = f.p2; d[2 * kWave] = f.p3;
      }
    }
  }
}

// One k-step of B rows (8 KB per wave) is requested ahead of the step being multiplied; the A fragments of a scaler are read once per
// step and used for the four node tiles.
// PLAIN [r5]: K16's forward on the same kernel - y[n, r] = bias[r] + sum_k x[n, k] Wa[k][r], every 16-row tile 16 more outputs, no scaler
// table (the exact-fp32 MFMA form of it ran 4.6x above its own matrix-core time: 0.092 ms per 75 -> 75 layer at C2L)
template <int S, bool VEC4, bool PLAIN = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void tower_post_fwd_x3_kernel(const PostParams p, const float* __restrict__ agg, const float* __restrict__ pre_tab,
                                                                   const float* __restrict__ Wa, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float post_smem[];
  constexpr int R = S * kPostO;
  post_bf16x8* Wl = reinterpret_cast<post_bf16x8*>(post_smem);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int t; int64_t bx;
  if (!post_block(p, t, bx)) return;                           // (before any barrier: the whole workgroup leaves)
  const int j = lane & 15, kg = lane >> 4;
  const int nks = p.KFp / 32;
  const int64_t tiles = (p.N + kWave - 1) / kWave;
  const int64_t tile0 = (bx * (kBlock / kWave) + wave) * p.tiles_per_wave;
  const int my_tiles = (int)max((int64_t)0, min((int64_t)p.tiles_per_wave, tiles - tile0));
  const int steps = my_tiles * nks;
  float4 raw[4][2];
  auto fetch = [&](int64_t n0, int ks) {
    const int c0 = 32 * ks + 4 * kg, c1 = c0 + 16;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int64_t row = min(n0 + nt * 16 + j, p.N - 1);
      const float* base = agg + (size_t)row * p.lda + (size_t)t * p.KF;
      if (VEC4) {                                              // KF % 4 == 0: a quad is inside or outside
        raw[nt][0] = *reinterpret_cast<const float4*>(base + min(c0, p.KF - 4));
        raw[nt][1] = *reinterpret_cast<const float4*>(base + min(c1, p.KF - 4));
      } else {
        raw[nt][0] = make_float4(base[min(c0, p.KF - 1)], base[min(c0 + 1, p.KF - 1)], base[min(c0 + 2, p.KF - 1)], base[min(c0 + 3, p.KF - 1)]);
        raw[nt][1] = make_float4(base[min(c1, p.KF - 1)], base[min(c1 + 1, p.KF - 1)], base[min(c1 + 2, p.KF - 1)], base[min(c1 + 3, p.KF - 1)]);
      }
    }
  };
  int ks = 0;
  int64_t n0 = tile0 * kWave;
  if (steps > 0) fetch(n0, 0);                                 // the first tile's rows travel while the weights are staged
  post_stage_x3_fwd<S>(Wl, Wa + (size_t)t * p.KFp * R, p.KFp);
  __syncthreads();                                             // weights staged: the only workgroup barrier
  post_f32x4 acc[4][S];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < S; ++q) acc[nt][q] = post_f32x4{0.f, 0.f, 0.f, 0.f};
  float pre[4][S];                                             // the tile's scaler products, requested at its FIRST k-step: no round trip in its epilogue
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < S; ++q) pre[nt][q] = 0.f;
  // PLAIN: this lane's bias values, loaded ONCE, ahead of every store.  Read inside the epilogue (`acc + bias[col]` per element) each of
  // them was a load BEHIND the stores just issued - one in-order vmcnt: the wait for the bias was a wait for their acknowledgement, once per
  // stored element (found in the ISA, round 5: K16's forward 0.074 ms -> see DESIGN.md)
  float bv[S][4];
#pragma unroll
  for (int q = 0; q < S; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = q * kPostO + 4 * kg + r;
      bv[q][r] = (PLAIN && p.bias && col < p.O) ? p.bias[col] : 0.f;
    }
  for (int step = 0; step < steps; ++step) {
    if (ks == 0 && !PLAIN) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < S; ++q) pre[nt][q] = pre_tab[(size_t)min(n0 + nt * 16 + j, p.N - 1) * kPostPrePitch + q];
    }
    // this step's B fragments: columns past KF are zeroed (their weights are zero, but 0 x inf is not); rows past N feed only columns
    // of D nobody stores
    PostFrag b[4];
    {
      const int c0 = 32 * ks + 4 * kg, c1 = c0 + 16;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 lo = raw[nt][0], hi = raw[nt][1];
        post_frag_set(b[nt], 0, c0 < p.KF ? lo.x : 0.f); post_frag_set(b[nt], 1, c0 + 1 < p.KF ? lo.y : 0.f);
        post_frag_set(b[nt], 2, c0 + 2 < p.KF ? lo.z : 0.f); post_frag_set(b[nt], 3, c0 + 3 < p.KF ? lo.w : 0.f);
        post_frag_set(b[nt], 4, c1 < p.KF ? hi.x : 0.f); post_frag_set(b[nt], 5, c1 + 1 < p.KF ? hi.y : 0.f);
        post_frag_set(b[nt], 6, c1 + 2 < p.KF ? hi.z : 0.f); post_frag_set(b[nt], 7, c1 + 3 < p.KF ? hi.w : 0.f);
      }
    }
    const bool last = ks == nks - 1;
    const int ksn = last ? 0 : ks + 1;
    const int64_t n0n = last ? n0 + kWave : n0;
    if (step + 1 < steps && !(kPostAbl & 2)) fetch(n0n, ksn);                     // in flight behind this step's MFMAs
#pragma unroll
    for (int q = 0; q < S; ++q) {
      PostFrag a;
      const post_bf16x8* wl = Wl + (size_t)((ks * S + q) * 3) * kWave + lane;
      a.p1 = wl[0]; a.p2 = wl[kWave]; a.p3 = wl[2 * kWave];
      // the five small piece products (<= 2^-8 of the sixth) are summed on their own and added once: six roundings of the running sum per
      // k-step put the tail of the forward's error at 3.6e-7 sum|a||w| (one element in 3e5 over the 3e-7 bound the fp32 kernels keep)
      post_f32x4 small[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) small[nt] = post_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < 5; ++pr)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { MMA_POST_X3(small[nt], a, b[nt], pr); }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { MMA_POST_X3(acc[nt][q], a, b[nt], 5); }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt][q] += small[nt];
    }
    if (last) {
      // C/D map of 16x16: column = lane & 15 (the node), row = 4 * (lane >> 4) + reg (the output o): y[o] = sum_q pre_q D_q[o]
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int64_t node = n0 + nt * 16 + j;
        const bool valid = node < p.N;
        if (PLAIN) {
          if (valid) {
#pragma unroll
            for (int q = 0; q < S; ++q)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int col = q * kPostO + 4 * kg + r;
                if (col < p.O) y[(size_t)node * p.ldy + col] = acc[nt][q][r] + bv[q][r];
              }
          }
        } else {
          float yv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < S; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = fmaf(pre[nt][q], acc[nt][q][r], yv[r]);
          if (valid) {
            float* yr = y + (size_t)node * p.ldy + (size_t)t * p.O + 4 * kg;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (4 * kg + r < p.O) yr[r] = yv[r];
          }
        }
#pragma unroll
        for (int q = 0; q < S; ++q) acc[nt][q] = post_f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    ks = ksn; n0 = n0n;
  }
}

// K15: kf tiles (16 columns) per pass - S x G accumulator tiles of 4 registers + two operand sets in flight.  [r5] S <= 3 takes 10: ZINC's
// 152 columns (10 tiles) are ONE pass over the nodes instead of two (each pass re-read gy, the scaler table and the lines of agg that
// straddle the two column halves)
static constexpr int post_gw_tiles(int S) { return S <= 3 ? 10 : 5; }
static int64_t post_gw_npw(int64_t N, int T) {
  // node ranges in multiples of 64; [r5] ONE wave per SIMD over the whole grid (1024 SIMDs): K15 holds S x G accumulator tiles + four
  // operand sets - 372 registers at S = 3, one workgroup per CU - so 256 workgroups are exactly one round of the chip.  Measured at C2L
  // (MMA_POST_GW_WAVES sweep): 512 waves 0.366 ms, 768 0.267, 1024 0.249, 1280 0.354 (310 workgroups: a second, badly filled round),
  // 1536 0.335, 2048 0.268 (round 4's choice: two rounds)
  int64_t total = 1024;
  { const char* e = getenv("MMA_POST_GW_WAVES"); if (e && atoi(e) >= 64) total = atoi(e); }      // plan sweep, read per call
  int64_t waves = total / (T < 1 ? 1 : T);
  if (waves < 4) waves = 4;
  int64_t npw = (N + waves - 1) / waves;
  npw = (npw + 63) / 64 * 64;
  return npw < 64 ? 64 : npw;
}

static int post_fill(PostParams* p, int64_t N, int T, int KF, int S, int O, const uint8_t* scaler_host, float avg_log, float avg_lin) {
  MMA_REQUIRE(N >= 0 && N < (1LL << 31) && T >= 1 && T <= 65535 && KF >= 1 && KF <= 512 && S >= 1 && S <= kPostMaxS &&
              O >= 1 && (scaler_host ? O <= kPostO : O <= S * kPostO), "N=%lld T=%d KF=%d S=%d O=%d unsupported (KF <= 512, S <= %d, O <= %d)",
              (long long)N, T, KF, S, O, kPostMaxS, kPostO);
  p->scaler_pack = 0;
  for (int q = 0; scaler_host && q < S; ++q) {
    MMA_REQUIRE(scaler_host[q] <= MMA_SC_INVERSE_LINEAR, "scaler[%d]=%d is not an MMA_SC_* code", q, (int)scaler_host[q]);
    p->scaler_pack |= (uint32_t)scaler_host[q] << (4 * q);
  }
  p->N = N; p->T = T; p->KF = KF; p->KFp = (KF + kPostTile - 1) / kPostTile * kPostTile; p->S = S; p->O = O; p->avg_log = avg_log; p->avg_lin = avg_lin;
  p->tiles_per_wave = 1;
  { const char* e = getenv("MMA_POST_ORDER"); p->order = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 0; }       // read per call (A/B)
  MMA_REQUIRE(((N + kWave - 1) / kWave + 3) / 4 * (int64_t)T < (1LL << 31), "N=%lld x T=%d workgroups exceed the grid", (long long)N, T);
  return 0;
}

// tiles per wave: the tower's weights are staged once per workgroup, so a workgroup should walk many node tiles - but the grid must
// still fill the chip (>= ~4 workgroups per CU over all towers)
static int post_tiles_per_wave(int64_t N, int T) {
  { const char* e = getenv(T == 1 ? "MMA_SKINNY_TPW" : "MMA_POST_TPW_ALL"); if (e && atoi(e) > 0) return atoi(e); }      // plan sweeps, read per call
  // Two workgroups fit a CU (65 KB of LDS each): 512 run at a time.  A workgroup's fixed costs (dispatch, staging 30 KB of weights, the
  // first tile's round trip, the epilogue stores: ~13 us at C2L against 7 us of MFMAs per tile) are paid once per workgroup, so a wave
  // should walk several tiles - but only in a way that keeps the number of workgroup ROUNDS integral: 4000 workgroups at one tile per
  // wave are 8 rounds, 1000 at four tiles are 2 rounds of 4 tiles (the same 8 tile-times, a quarter of the fixed costs); three tiles
  // per wave would be 2.6 rounds, i.e. 9 tile-times.
  const int64_t blocks1 = ((N + kWave - 1) / kWave + kBlock / kWave - 1) / (kBlock / kWave) * (int64_t)T;     // workgroups at one tile per wave
  int best = 1;
  int64_t best_cost = ((blocks1 + 511) / 512);
  for (int r = 2; r <= 8; ++r) {
    const int64_t blocks = (blocks1 + r - 1) / r;
    // never leave CUs without a workgroup.  [r5] A single product (T == 1: K16) may go down to one workgroup per CU: its 800 workgroups at
    // C2L were 1.56 rounds of 512 - two tiles per wave, 400 workgroups in ONE round, pay the weight staging once (0.169 -> 0.152 ms for the
    // two forward layers; for T = 5 the same rule picks 8 tiles per wave, measured slower: 0.183 -> 0.245 ms)
    if (blocks < (T == 1 ? 256 : 512)) break;
    const int64_t cost = ((blocks + 511) / 512) * r;
    if (cost <= best_cost) { best = r; best_cost = cost; }
  }
  return best;
}
static dim3 post_grid(PostParams& p, int T) {       // one-dimensional: post_block() takes it apart; sets p.nbx
  const int64_t tiles = (p.N + kWave - 1) / kWave;
  const int64_t per_block = (int64_t)(kBlock / kWave) * p.tiles_per_wave;
  p.nbx = (tiles + per_block - 1) / per_block;
  const int64_t nbx = p.order == 0 ? (p.nbx + 7) / 8 * 8 : p.nbx;
  return dim3((unsigned)(nbx * T));
}

static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
// LDS of the bf16-piece forward: the split weights in fragment order, nothing else ((KFp/32) x S fragments of 3 x 1 KB)
static unsigned post_x3_lds_bytes(int KFp, int S) { return (unsigned)((KFp / 32) * S * 3 * 1024); }
static bool post_exact() { const char* e = getenv("MMA_POST_EXACT"); return e && e[0] == '1'; }     // the fp32-MFMA kernels (A/B), read per call
static unsigned post_lds_bytes(int KFp, int S, bool bwd) {
  const int R = S * kPostO;
  return (unsigned)(((size_t)(bwd ? R * (KFp + 16) : KFp * R) + (size_t)(kBlock / kWave) * kWave * kPostPitch) * sizeof(float));
}

}  // namespace mma

using namespace mma;

#define MMA_POST_LAUNCH(KERNEL, LDS, ...)                                                                              \
  switch (S) {                                                                                                         \
    case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                       \
    case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                       \
    case 3: hipLaunchKernelGGL((KERNEL<3>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                       \
    case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                       \
    default: hipLaunchKernelGGL((KERNEL<5>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                      \
  }
#define MMA_POST_LAUNCH3(KERNEL, PLAIN, V4, LDS, ...)                                                                  \
  switch (S) {                                                                                                         \
    case 1: hipLaunchKernelGGL((KERNEL<1, PLAIN, V4>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;            \
    case 2: hipLaunchKernelGGL((KERNEL<2, PLAIN, V4>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;            \
    case 3: hipLaunchKernelGGL((KERNEL<3, PLAIN, V4>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;            \
    case 4: hipLaunchKernelGGL((KERNEL<4, PLAIN, V4>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;            \
    default: hipLaunchKernelGGL((KERNEL<5, PLAIN, V4>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;           \
  }
#define MMA_POST_LAUNCH2(KERNEL, PLAIN, LDS, ...)                                                                      \
  switch (S) {                                                                                                         \
    case 1: hipLaunchKernelGGL((KERNEL<1, PLAIN>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                \
    case 2: hipLaunchKernelGGL((KERNEL<2, PLAIN>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                \
    case 3: hipLaunchKernelGGL((KERNEL<3, PLAIN>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                \
    case 4: hipLaunchKernelGGL((KERNEL<4, PLAIN>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;                \
    default: hipLaunchKernelGGL((KERNEL<5, PLAIN>), grid, dim3(kBlock), LDS, st, p, __VA_ARGS__); break;               \
  }

// padded weight shapes the caller prepares: forward Wa (T, KFp, S*16), backward Wb (T, S*16, KFp + 16), KFp = mma_tower_post_kfp(KF)
extern "C" int64_t mma_tower_post_kfp(int32_t KF) { return KF < 1 ? -1 : ((int64_t)KF + kPostTile - 1) / kPostTile * kPostTile; }

// [r4] both padded layouts of the post-NN weight columns in ONE launch (as torch ops: a zero-fill, a strided copy and a transposing copy per
// MMAConv call - three of the ~39 kernels of a ZINC-batch layer step, where every launch is 5 us of 300): Wo (T, O, S*KF) contiguous ->
// Wb[t][q*16+o][c] = Wo[t][o][q*KF + c] for o < O, c < KF (else 0), Wa[t][c][q*16+o] = the same for c < KFp.
__global__ __launch_bounds__(kBlock) void post_weights_kernel(const float* Wo, int T, int O, int S, int KF, int KFp, float* Wa, float* Wb) {
  const int ldb = KFp + 16, R = S * 16;
  const int64_t total = (int64_t)T * R * ldb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldb);
    const int r = (int)((i / ldb) % R);
    const int t = (int)(i / ((int64_t)ldb * R));
    const int q = r >> 4, o = r & 15;
    const float v = (o < O && c < KF) ? Wo[((size_t)t * O + o) * ((size_t)S * KF) + (size_t)q * KF + c] : 0.f;
    Wb[i] = v;
    if (c < KFp) Wa[((size_t)t * KFp + c) * R + r] = v;
  }
}
// the PLAIN (K16) layouts of a Linear weight W (O, K): Wb (S*16, KFp + 16)[r][c] = W[r][c], Wa (KFp, S*16) its transpose, S = ceil(O / 16)
__global__ __launch_bounds__(kBlock) void skinny_weights_kernel(const float* W, int O, int K, int KFp, int R, float* Wa, float* Wb) {
  const int ldb = KFp + 16;
  const int64_t total = (int64_t)R * ldb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldb), r = (int)(i / ldb);
    const float v = (r < O && c < K) ? W[(size_t)r * K + c] : 0.f;
    Wb[i] = v;
    if (c < KFp) Wa[(size_t)c * R + r] = v;
  }
}
extern "C" int mma_skinny_linear_weights(const float* W, int32_t O, int32_t K, float* Wa, float* Wb, void* stream) {
  MMA_REQUIRE(O >= 1 && O <= kPostMaxS * kPostO && K >= 1 && K <= 512, "O=%d K=%d unsupported", O, K);
  MMA_REQUIRE(W && Wa && Wb, "NULL argument");
  const int KFp = (K + kPostTile - 1) / kPostTile * kPostTile, R = (O + kPostO - 1) / kPostO * 16;
  const int64_t total = (int64_t)R * (KFp + 16);
  hipLaunchKernelGGL(skinny_weights_kernel, dim3((unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, kMaxGrid)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), W, (int)O, (int)K, KFp, R, Wa, Wb);
  return check_launch("skinny_weights_kernel");
}

extern "C" int mma_tower_post_weights(const float* Wo, int32_t T, int32_t O, int32_t S, int32_t KF, float* Wa, float* Wb, void* stream) {
  MMA_REQUIRE(T >= 1 && O >= 1 && O <= 16 && S >= 1 && S <= kPostMaxS && KF >= 1 && KF <= 512, "T=%d O=%d S=%d KF=%d unsupported", T, O, S, KF);
  MMA_REQUIRE(Wo && Wa && Wb, "NULL argument");
  const int KFp = (KF + kPostTile - 1) / kPostTile * kPostTile;
  const int64_t total = (int64_t)T * S * 16 * (KFp + 16);
  hipLaunchKernelGGL(post_weights_kernel, dim3((unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, kMaxGrid)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), Wo, (int)T, (int)O, (int)S, (int)KF, KFp, Wa, Wb);
  return check_launch("post_weights_kernel");
}

// the limits post_fill() and the launchers enforce, for the host-side gates (mma_conv.py's `factored`, dense._skinny_ok)
extern "C" int mma_tower_post_fits(int32_t KF, int32_t S) {
  if (KF < 1 || KF > 512 || S < 1 || S > kPostMaxS) return 0;
  const int KFp = (KF + kPostTile - 1) / kPostTile * kPostTile;
  return post_lds_bytes(KFp, S, false) <= 160 * 1024 && post_lds_bytes(KFp, S, true) <= 160 * 1024 ? 1 : 0;
}

extern "C" int mma_tower_post_pre(const int32_t* rowptr, float* pre, int64_t N, int32_t S, const uint8_t* scaler_host, float avg_log, float avg_lin,
                                  void* stream) {
  PostParams p{};
  if (int rc = post_fill(&p, N, 1, 4, S, 1, scaler_host, avg_log, avg_lin)) return rc;
  MMA_REQUIRE(scaler_host != nullptr, "NULL scaler codes");
  if (N == 0) return 0;
  MMA_REQUIRE(rowptr && pre, "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((N + kBlock - 1) / kBlock));
  MMA_POST_LAUNCH(tower_post_pre_kernel, 0, rowptr, pre)
  return check_launch("tower_post_pre_kernel");
}

extern "C" int mma_tower_post_fwd(const float* agg, int64_t lda, const float* pre, const float* Wa, float* y, int64_t ldy,
                                  int64_t N, int32_t T, int32_t KF, int32_t S, int32_t O, const uint8_t* scaler_host, float avg_log,
                                  float avg_lin, void* stream) {
  PostParams p{};
  if (int rc = post_fill(&p, N, T, KF, S, O, scaler_host, avg_log, avg_lin)) return rc;
  if (N == 0) return 0;
  MMA_REQUIRE(scaler_host != nullptr, "NULL scaler codes");
  MMA_REQUIRE(agg && pre && Wa && y && al16(Wa) && lda >= (int64_t)T * KF && ldy >= (int64_t)T * O && (reinterpret_cast<uintptr_t>(agg) & 3) == 0,
              "NULL / misaligned argument or row pitch too small");
  p.lda = lda; p.ldy = ldy; p.vec4 = (KF % 4 == 0 && lda % 4 == 0 && al16(agg)) ? 1 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // 256 * tiles_per_wave nodes (four waves x 64 x tiles) of one tower per workgroup, tower fastest (post_block)
  p.tiles_per_wave = post_tiles_per_wave(N, T);
  if (const char* e = getenv("MMA_POST_TPW")) { if (atoi(e) > 0) p.tiles_per_wave = atoi(e); }      // plan sweep (tools/post_micro.py)
  const dim3 grid = post_grid(p, T);
  if (const unsigned lx = post_x3_lds_bytes(p.KFp, S); lx <= 160 * 1024 && !post_exact()) {
    if (p.vec4) { MMA_POST_LAUNCH2(tower_post_fwd_x3_kernel, true, lx, agg, pre, Wa, y) } else { MMA_POST_LAUNCH2(tower_post_fwd_x3_kernel, false, lx, agg, pre, Wa, y) }
    return check_launch("tower_post_fwd_x3_kernel");
  }
  const unsigned lds = post_lds_bytes(p.KFp, S, false);
  MMA_REQUIRE(lds <= 160 * 1024, "the tower's weights (%u bytes with the tiles) do not fit the LDS", lds);
  if (p.vec4) { MMA_POST_LAUNCH3(tower_post_fwd_kernel, false, true, lds, agg, pre, Wa, y) } else { MMA_POST_LAUNCH3(tower_post_fwd_kernel, false, false, lds, agg, pre, Wa, y) }
  return check_launch("tower_post_fwd_kernel");
}

extern "C" int64_t mma_tower_post_gw_chunks(int64_t N, int32_t T) {
  if (N <= 0 || T <= 0) return 0;
  const int64_t npw = post_gw_npw(N, T);
  return ((N + npw - 1) / npw + kBlock / kWave - 1) / (kBlock / kWave);        // ONE partial tile per workgroup of four waves
}

extern "C" int mma_tower_post_gw(const float* gy, int64_t ldg, const float* agg, int64_t lda, const float* pre, float* part,
                                 int64_t n_chunks, int64_t N, int32_t T, int32_t KF, int32_t S, int32_t O, const uint8_t* scaler_host,
                                 float avg_log, float avg_lin, void* stream) {
  PostParams p{};
  if (int rc = post_fill(&p, N, T, KF, S, O, scaler_host, avg_log, avg_lin)) return rc;
  MMA_REQUIRE(n_chunks == mma_tower_post_gw_chunks(N, T), "n_chunks=%lld, expected mma_tower_post_gw_chunks(N, T)=%lld", (long long)n_chunks,
              (long long)mma_tower_post_gw_chunks(N, T));
  if (N == 0) return 0;
  MMA_REQUIRE(gy && agg && pre && part && lda >= (int64_t)T * KF && ldg >= (int64_t)T * O, "NULL argument or row pitch too small");
  p.lda = lda; p.ldg = ldg;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int kfp16 = (KF + 15) / 16 * 16;
  const int64_t npw = post_gw_npw(N, T);
  p.nbx = n_chunks;
  // K15 keeps the tower-fastest order: under the XCD-grouped one it ran 0.32 -> 0.44 ms at C2L (its ~500 workgroups are all resident at
  // once; only their placement changes - not explained), while K13 / K14 gain 3-5 % from it
  { const char* e = getenv("MMA_POST_ORDER"); if (!(e && e[0] >= '0' && e[0] <= '2')) p.order = 2; }
  const dim3 grid((unsigned)((p.order == 0 ? (n_chunks + 7) / 8 * 8 : n_chunks) * T));
  int G = post_gw_tiles(S);
  { const char* e = getenv("MMA_POST_GW_TILES"); if (e && e[0] == '5') G = 5; }       // read per call (A/B): round 4's two passes at S = 3
  const unsigned lds = (unsigned)((kBlock / kWave) * S * (G > 5 ? G / 2 : G) * 4 * kWave * sizeof(float));
#define MMA_GW(SS, GG) hipLaunchKernelGGL((tower_post_gw_kernel<SS, GG>), grid, dim3(kBlock), lds, st, p, gy, agg, pre, part, npw, kfp16)
  switch (S) {
    case 1: MMA_GW(1, 10); break;
    case 2: MMA_GW(2, 10); break;
    case 3: if (G == 10) MMA_GW(3, 10); else MMA_GW(3, 5); break;
    case 4: MMA_GW(4, 5); break;
    default: MMA_GW(5, 5); break;
  }
#undef MMA_GW
  return check_launch("tower_post_gw_kernel");
}

// [r4] K15's per-workgroup partial tiles (n_chunks, T, S, 16, kfp16) summed in mma_col_sum's order (col_sum_kernel, one row block: four row
// lanes of four accumulators - the same bits as the K8 launch it replaces) and written straight into the weight layout
// gWo (T, O, S*KF): the permuting copy behind the K8 launch was one more launch per layer step.
__global__ __launch_bounds__(kBlock) void post_gw_reduce_kernel(const float* part, int R, int T, int S, int O, int KF, int kfp16, float* gWo) {
  const int64_t total = (int64_t)T * O * S * KF;
  const size_t ld = (size_t)T * S * 16 * kfp16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kf = (int)(i % KF);
    const int q = (int)((i / KF) % S);
    const int o = (int)((i / ((int64_t)KF * S)) % O);
    const int t = (int)(i / ((int64_t)KF * S * O));
    const float* qp = part + (((size_t)t * S + q) * 16 + o) * kfp16 + kf;
    float red[4];
#pragma unroll
    for (int rl = 0; rl < 4; ++rl) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int r = rl;
      for (; r + 12 < R; r += 16) { a0 += qp[(size_t)r * ld]; a1 += qp[(size_t)(r + 4) * ld]; a2 += qp[(size_t)(r + 8) * ld]; a3 += qp[(size_t)(r + 12) * ld]; }
      for (; r < R; r += 4) a0 += qp[(size_t)r * ld];
      red[rl] = (a0 + a1) + (a2 + a3);
    }
    gWo[i] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}
extern "C" int mma_tower_post_gw_reduce(const float* part, int64_t n_chunks, int32_t T, int32_t S, int32_t O, int32_t KF, float* gWo, void* stream) {
  MMA_REQUIRE(n_chunks >= 1 && n_chunks <= 1024 && T >= 1 && S >= 1 && S <= kPostMaxS && O >= 1 && O <= 16 && KF >= 1 && KF <= 512,
              "n_chunks=%lld T=%d S=%d O=%d KF=%d unsupported (n_chunks <= 1024: mma_col_sum's single-pass order)", (long long)n_chunks, T, S, O, KF);
  MMA_REQUIRE(part && gWo, "NULL argument");
  MMA_REQUIRE(n_chunks <= 256 || (int64_t)T * S * 16 * ((KF + 15) / 16 * 16) > 2048,
              "n_chunks=%lld with %lld columns: mma_col_sum sums this shape in its short-matrix order", (long long)n_chunks,
              (long long)T * S * 16 * ((KF + 15) / 16 * 16));
  const int64_t total = (int64_t)T * O * S * KF;
  hipLaunchKernelGGL(post_gw_reduce_kernel, dim3((unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, kMaxGrid)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), part, (int)n_chunks, (int)T, (int)S, (int)O, (int)KF, (KF + 15) / 16 * 16, gWo);
  return check_launch("post_gw_reduce_kernel");
}

extern "C" int mma_tower_post_bwd(const float* gy, int64_t ldg, const float* pre, const float* Wb, float* gagg, int64_t lda,
                                  float* gys, int64_t ldgs, int64_t N, int32_t T, int32_t KF, int32_t S, int32_t O,
                                  const uint8_t* scaler_host, float avg_log, float avg_lin, void* stream) {
  PostParams p{};
  if (int rc = post_fill(&p, N, T, KF, S, O, scaler_host, avg_log, avg_lin)) return rc;
  if (N == 0) return 0;
  MMA_REQUIRE(scaler_host != nullptr, "NULL scaler codes");
  MMA_REQUIRE(gy && pre && Wb && gagg && al16(Wb) && lda >= (int64_t)T * KF && ldg >= (int64_t)T * O && (reinterpret_cast<uintptr_t>(gagg) & 3) == 0,
              "NULL / misaligned argument or row pitch too small");
  MMA_REQUIRE(!gys || ldgs >= (int64_t)T * S * kPostO, "gys needs a pitch >= T*S*16 floats");
  p.lda = lda; p.ldg = ldg; p.ldgs = ldgs; p.vec4 = (KF % 4 == 0 && lda % 4 == 0 && al16(gagg)) ? 1 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int wb_pitch = p.KFp + 16;
  p.tiles_per_wave = post_tiles_per_wave(N, T);
  const dim3 grid = post_grid(p, T);
  const unsigned lds = post_lds_bytes(p.KFp, S, true);
  MMA_REQUIRE(lds <= 160 * 1024, "the tower's weights (%u bytes with the tiles) do not fit the LDS", lds);
  MMA_POST_LAUNCH2(tower_post_bwd_kernel, false, lds, gy, pre, Wb, wb_pitch, gagg, gys)
  return check_launch("tower_post_bwd_kernel");
}

// ---- K16: a plain skinny Linear on the same kernels (PLAIN): y (N,O) = x (N,K) Wa + bias, gx (N,K) = gy (N,O) Wb, O <= 80, K <= 512 --
// The 75 -> 75 Linear layers around MMAConv's fused kernels (the x-part of the post-NN, `lin`: mma_conv.py:99-105,132-136) are four
// library GEMMs per layer step at ~0.11 ms each for 61 MB in / 61 MB out (rocBLAS picks a 16x256 macro tile for a 75-wide output).
// Wa (KFp, S*16) / Wb (S*16, KFp + 16) zero-padded as for mma_tower_post_*, S = ceil(O / 16), KFp = mma_tower_post_kfp(K).
extern "C" int mma_skinny_linear_fwd(const float* x, int64_t ldx, const float* Wa, const float* bias, const float* addend, int64_t ldadd,
                                     float* y, int64_t ldy, int64_t N, int32_t K, int32_t O, void* stream) {
  MMA_REQUIRE(O >= 1 && O <= kPostMaxS * kPostO, "O=%d unsupported (<= %d)", O, kPostMaxS * kPostO);
  const int S = (O + kPostO - 1) / kPostO;
  PostParams p{};
  if (int rc = post_fill(&p, N, 1, K, S, O, nullptr, 1.f, 1.f)) return rc;
  if (N == 0) return 0;
  MMA_REQUIRE(x && Wa && y && al16(Wa) && ldx >= K && ldy >= O && (reinterpret_cast<uintptr_t>(x) & 3) == 0, "NULL / misaligned argument or row pitch too small");
  MMA_REQUIRE(!addend || ldadd >= O, "ldadd=%lld < O", (long long)ldadd);
  p.lda = ldx; p.ldy = ldy; p.bias = bias; p.addend = addend; p.ldadd = ldadd; p.vec4 = (K % 4 == 0 && ldx % 4 == 0 && al16(x)) ? 1 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  p.tiles_per_wave = post_tiles_per_wave(N, 1);
  const dim3 grid = post_grid(p, 1);
  const float* pre = nullptr;
  // [r5] on three bf16 pieces per operand (tower_post_fwd_x3_kernel<.., PLAIN>) where the split weights fit the LDS; MMA_POST_EXACT=1 or an
  // addend: the exact-fp32 kernel
  if (const unsigned lx = post_x3_lds_bytes(p.KFp, S); lx <= 160 * 1024 && !post_exact() && !addend) {
    const char* e = getenv("MMA_SKINNY_X3");                    // read per call (A/B): 0 = the fp32 kernel
    if (!(e && e[0] == '0')) {
#define MMA_SK(SS, V4) hipLaunchKernelGGL((tower_post_fwd_x3_kernel<SS, V4, true>), grid, dim3(kBlock), lx, st, p, x, pre, Wa, y)
      switch (S) {
        case 1: if (p.vec4) MMA_SK(1, true); else MMA_SK(1, false); break;
        case 2: if (p.vec4) MMA_SK(2, true); else MMA_SK(2, false); break;
        case 3: if (p.vec4) MMA_SK(3, true); else MMA_SK(3, false); break;
        case 4: if (p.vec4) MMA_SK(4, true); else MMA_SK(4, false); break;
        default: if (p.vec4) MMA_SK(5, true); else MMA_SK(5, false); break;
      }
#undef MMA_SK
      return check_launch("tower_post_fwd_x3_kernel (plain)");
    }
  }
  const unsigned lds = post_lds_bytes(p.KFp, S, false);
  MMA_REQUIRE(lds <= 160 * 1024, "the weights (%u bytes with the tiles) do not fit the LDS", lds);
  if (p.vec4) { MMA_POST_LAUNCH3(tower_post_fwd_kernel, true, true, lds, x, pre, Wa, y) } else { MMA_POST_LAUNCH3(tower_post_fwd_kernel, true, false, lds, x, pre, Wa, y) }
  return check_launch("tower_post_fwd_kernel (plain)");
}

// [r5] the weight and the bias gradient of the plain skinny Linear from ONE pass over gy and x (K15 in its PLAIN form): the TN library
// GEMM + three column-sum launches it replaces read gy twice and ran 47 + ~40 us per 75 -> 75 layer at C2L.
// part (n_chunks, S*16, kfp16): one partial tile per workgroup, kfp16 = round_up(K + 1, 16) - the ones column sits at k = K.
__global__ __launch_bounds__(kBlock) void skinny_gw_reduce_kernel(const float* __restrict__ part, int R, int O, int K, int S, int kfp16,
                                                                  float* __restrict__ gw, float* __restrict__ gb) {
  // 16 outputs x 16 partial sums per workgroup: lane (sub, i) sums the partials r = sub, sub + 16, ... of output i in order, the 16 sums
  // are added in a fixed tree through LDS - every load of a lane is independent of the others (one round trip, not R of them)
  __shared__ float red[16][17];
  const int i = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const int64_t total = (int64_t)O * (K + 1);
  const int64_t idx = (int64_t)blockIdx.x * 16 + i;
  const bool ok = idx < total;
  const int o = ok ? (int)(idx / (K + 1)) : 0, k = ok ? (int)(idx % (K + 1)) : 0;
  const size_t ld = (size_t)S * kPostO * kfp16;
  const float* qp = part + (size_t)o * kfp16 + k;
  float a = 0.f;
  for (int r = sub; r < R; r += 16) a += qp[(size_t)r * ld];
  red[sub][i] = a;
  __syncthreads();
  if (sub == 0 && ok) {
    float v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = red[m][i];
#pragma unroll
    for (int w = 1; w < 16; w *= 2)
#pragma unroll
      for (int m = 0; m < 16; m += 2 * w) v[m] = v[m] + v[m + w];
    if (k < K) gw[(size_t)o * K + k] = v[0];
    else if (gb) gb[o] = v[0];
  }
}
// node ranges of the plain form (MMA_SKINNY_GW_WAVES: plan sweep, read per call).  Its small LDS round would let several workgroups share
// a CU, but more wavefronts only add partial tiles: C2L, kernel + reduction: 1024 waves 0.049 ms, 2048 0.058, 3072 0.058, 4096 0.075
static int64_t skinny_gw_npw(int64_t N) {
  int64_t waves = 1024;
  { const char* e = getenv("MMA_SKINNY_GW_WAVES"); if (e && atoi(e) >= 64) waves = atoi(e); }
  int64_t npw = (N + waves - 1) / waves;
  npw = (npw + 63) / 64 * 64;
  return npw < 64 ? 64 : npw;
}
static int64_t skinny_gw_chunks(int64_t N) {
  const int64_t npw = skinny_gw_npw(N);
  return ((N + npw - 1) / npw + kBlock / kWave - 1) / (kBlock / kWave);
}
extern "C" int64_t mma_skinny_linear_gw_part(int64_t N, int32_t K, int32_t O) {          // floats of the partial-tile workspace
  if (N <= 0 || K < 1 || K > 512 || O < 1 || O > kPostMaxS * kPostO) return 0;
  const int64_t S = ((int64_t)O + kPostO - 1) / kPostO, kfp16 = ((int64_t)K + 1 + 15) / 16 * 16;
  return skinny_gw_chunks(N) * S * kPostO * kfp16;
}
extern "C" int mma_skinny_linear_gw(const float* gy, int64_t ldg, const float* x, int64_t ldx, float* part, int64_t n_part,
                                    float* gw, float* gb, int64_t N, int32_t K, int32_t O, void* stream) {
  MMA_REQUIRE(O >= 1 && O <= kPostMaxS * kPostO && K >= 1 && K <= 512, "O=%d K=%d unsupported (O <= %d, K <= 512)", O, K, kPostMaxS * kPostO);
  MMA_REQUIRE(N >= 1 && N < (1LL << 31), "N=%lld unsupported", (long long)N);
  MMA_REQUIRE(n_part == mma_skinny_linear_gw_part(N, K, O), "n_part=%lld, expected mma_skinny_linear_gw_part(N, K, O)=%lld", (long long)n_part,
              (long long)mma_skinny_linear_gw_part(N, K, O));
  MMA_REQUIRE(gy && x && part && gw && ldx >= K && ldg >= O, "NULL argument or row pitch too small");
  const int S = (O + kPostO - 1) / kPostO;
  PostParams p{};
  p.N = N; p.T = 1; p.KF = K; p.KFp = (K + kPostTile - 1) / kPostTile * kPostTile; p.S = S; p.O = O; p.avg_log = 1.f; p.avg_lin = 1.f;
  p.lda = ldx; p.ldg = ldg; p.order = 2;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int kfp16 = (K + 1 + 15) / 16 * 16;
  const int64_t npw = skinny_gw_npw(N);
  const int64_t n_chunks = skinny_gw_chunks(N);
  p.nbx = n_chunks;
  const dim3 grid((unsigned)n_chunks);
  const float* pre = nullptr;
  // G = 5 kf tiles per pass for every S here: S x 5 accumulator tiles + four operand sets of S + 5 + 1 registers stay inside one
  // workgroup per CU at S = 5 (the 75 -> 75 layers: 76 columns = 5 tiles, ONE pass over the rows)
  const unsigned lds = (unsigned)((kBlock / kWave) * S * 1 * 4 * kWave * sizeof(float));
#define MMA_GWP(SS) hipLaunchKernelGGL((tower_post_gw_kernel<SS, 5, true>), grid, dim3(kBlock), lds, st, p, gy, x, pre, part, npw, kfp16)
  switch (S) {
    case 1: MMA_GWP(1); break;
    case 2: MMA_GWP(2); break;
    case 3: MMA_GWP(3); break;
    case 4: MMA_GWP(4); break;
    default: MMA_GWP(5); break;
  }
#undef MMA_GWP
  if (int rc = check_launch("tower_post_gw_kernel (plain)")) return rc;
  const int64_t total = (int64_t)O * (K + 1);
  hipLaunchKernelGGL(skinny_gw_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(kBlock), 0, st, part, (int)n_chunks, (int)O, (int)K, S,
                     kfp16, gw, gb);
  return check_launch("skinny_gw_reduce_kernel");
}

extern "C" int mma_skinny_linear_bwd_dx(const float* gy, int64_t ldg, const float* Wb, float* gx, int64_t ldx,
                                        int64_t N, int32_t K, int32_t O, void* stream) {
  MMA_REQUIRE(O >= 1 && O <= kPostMaxS * kPostO, "O=%d unsupported (<= %d)", O, kPostMaxS * kPostO);
  const int S = (O + kPostO - 1) / kPostO;
  PostParams p{};
  if (int rc = post_fill(&p, N, 1, K, S, O, nullptr, 1.f, 1.f)) return rc;
  if (N == 0) return 0;
  MMA_REQUIRE(gy && Wb && gx && al16(Wb) && ldx >= K && ldg >= O && (reinterpret_cast<uintptr_t>(gx) & 3) == 0, "NULL / misaligned argument or row pitch too small");
  p.lda = ldx; p.ldg = ldg; p.vec4 = (K % 4 == 0 && ldx % 4 == 0 && al16(gx)) ? 1 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int wb_pitch = p.KFp + 16;
  p.tiles_per_wave = post_tiles_per_wave(N, 1);
  const dim3 grid = post_grid(p, 1);
  const float* pre = nullptr;
  float* gys = nullptr;
  const unsigned lds = post_lds_bytes(p.KFp, S, true);
  MMA_REQUIRE(lds <= 160 * 1024, "the weights (%u bytes with the tiles) do not fit the LDS", lds);
  MMA_POST_LAUNCH2(tower_post_bwd_kernel, true, lds, gy, pre, Wb, wb_pitch, gx, gys)
  return check_launch("tower_post_bwd_kernel (plain)");
}
