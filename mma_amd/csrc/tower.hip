// K9: backward of the per-tower post-NN Linear of MMAConv on the aggregates (reference graph_regression/mma_conv.py:132-134,
//     post_nns[t](cat[x_t, out_t])): for a (N,T,C), W (T,O,C), upstream gradient gy (N,T,O) with a SMALL O (<= 16; ZINC: 15)
//        ga[n,t,c]  = sum_o gy[n,t,o] * W[t,o,c]          (written once, in the layout K4 reads it)
//        gW[t,o,c]  = sum_n gy[n,t,o] * a[n,t,c]          (fixed-order partial sums, finished by K8)
//     in ONE pass over `a`: as two library batched GEMMs with a 15-wide dimension these cost 0.79 + 0.94 ms on the
//     10 000-molecule batch (rocBLAS reaches ~15 TFLOP/s on them), while the data is 1.9 GB in and 1.9 GB out.
// A thread owns 4 consecutive columns c of one tower and keeps W[t, :, c..c+3] (O x 4 floats) and its gW partial (O x 4)
// in registers; per node it reads 16 B of `a`, the wave reads the node's O gradient values with ONE 64-byte load and hands
// them out with v_readlane (SGPR operands), 2 x O x 4 FMAs, one 16-byte store.  The 4 waves of a workgroup take adjacent
// (tower, column chunk) pieces of the same nodes; the node blocks' partial gW tiles are summed by K8 in a fixed order.
#include "common.h"

namespace mma {

constexpr int kTowerMaxO = 16;

struct TowerParams {
  const float* gy; const float* a; const float* W;
  float* ga; float* part;               // part: (n_blocks, T, O, C)
  int64_t N, nodes_per_block; int T, O, C;
};

__global__ __launch_bounds__(kBlock) void tower_bwd_kernel(const TowerParams p) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunks = (p.C / 4 + kWave - 1) / kWave;
  // the 4 waves of a workgroup take 4 consecutive (tower, 256-column chunk) pieces of the SAME nodes: together they stream
  // 4 KB of contiguous memory per node instead of four 1 KB pieces that are T*C floats apart
  const int piece = (int)blockIdx.y * (kBlock / kWave) + wave;
  if (piece >= p.T * chunks) return;
  const int t = piece / chunks, chunk = piece % chunks;
  const int c = (chunk * kWave + lane) * 4;
  const bool valid = c < p.C;
  const int cc = valid ? c : 0;
  float4 w[kTowerMaxO], acc[kTowerMaxO];
#pragma unroll
  for (int o = 0; o < kTowerMaxO; ++o) {
    w[o] = (o < p.O && valid) ? *reinterpret_cast<const float4*>(p.W + ((size_t)t * p.O + o) * p.C + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t n0 = (int64_t)blockIdx.x * p.nodes_per_block;
  const int64_t n1 = min(p.N, n0 + p.nodes_per_block);
  const size_t row = (size_t)p.T * p.C, grow = (size_t)p.T * p.O;
  const float* ap = p.a + (size_t)t * p.C + cc;
  float* op = p.ga + (size_t)t * p.C + cc;
  const float* gp = p.gy + (size_t)t * p.O + (lane < p.O ? lane : 0);
  // kU nodes per step and one step ahead: 2 x kU 16-byte loads per lane in flight (one node at a time left the kernel
  // latency-bound at 3.2 TB/s).  A node's O gradient values arrive with ONE 64-byte load and are handed out by v_readlane.
  constexpr int kU = 4, kStride = 1;
  float gv[kU];
  float4 av[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t m = n0 + (int64_t)u * kStride;
    gv[u] = 0.f; av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < n1) { gv[u] = gp[m * grow]; av[u] = *reinterpret_cast<const float4*>(ap + m * row); }
  }
  for (int64_t n = n0; n < n1; n += (int64_t)kU * kStride) {
    float gnext[kU];
    float4 anext[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t m = n + (int64_t)(kU + u) * kStride;
      gnext[u] = 0.f; anext[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < n1) { gnext[u] = gp[m * grow]; anext[u] = *reinterpret_cast<const float4*>(ap + m * row); }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t m = n + (int64_t)u * kStride;
      if (m < n1) {                                   // wave-uniform
        const float gl = lane < p.O ? gv[u] : 0.f;
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < kTowerMaxO; ++o) {       // lanes >= O hold 0 and their w rows are 0: the unused steps add exact zeros
          const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), o));   // the builtin is typed int
          out.x = fmaf(g, w[o].x, out.x); out.y = fmaf(g, w[o].y, out.y); out.z = fmaf(g, w[o].z, out.z); out.w = fmaf(g, w[o].w, out.w);
          acc[o].x = fmaf(g, av[u].x, acc[o].x); acc[o].y = fmaf(g, av[u].y, acc[o].y);
          acc[o].z = fmaf(g, av[u].z, acc[o].z); acc[o].w = fmaf(g, av[u].w, acc[o].w);
        }
        if (valid) *reinterpret_cast<float4*>(op + m * row) = out;
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) { gv[u] = gnext[u]; av[u] = anext[u]; }
  }
  if (valid) {       // one partial tile per (node block, piece); K8 sums the node blocks in a fixed order
    float* q = p.part + ((size_t)blockIdx.x * p.T + t) * p.O * p.C + c;
#pragma unroll                 // compile-time indices: a run-time `o` keeps the whole acc array in scratch memory (it did: 272 B/lane)
    for (int o = 0; o < kTowerMaxO; ++o)
      if (o < p.O) *reinterpret_cast<float4*>(q + (size_t)o * p.C) = acc[o];
  }
}

static int64_t tower_blocks(int64_t N) {
  int64_t b = (N + 63) / 64;              // at least 64 nodes per wave
  if (b > 512) b = 512;
  return b < 1 ? 1 : b;
}

}  // namespace mma

using namespace mma;

extern "C" int64_t mma_tower_linear_bwd_blocks(int64_t N) { return N > 0 ? tower_blocks(N) : 0; }

extern "C" int mma_tower_linear_bwd(const float* gy, const float* a, const float* W, float* ga, float* part, int64_t n_blocks,
                                    int64_t N, int32_t T, int32_t O, int32_t C, void* stream) {
  MMA_REQUIRE(N >= 0 && T >= 1 && O >= 1 && O <= kTowerMaxO && C >= 4 && C % 4 == 0, "N=%lld T=%d O=%d C=%d: need O <= %d, C %% 4 == 0",
              (long long)N, T, O, C, kTowerMaxO);
  if (N == 0) return 0;
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  MMA_REQUIRE(gy && a && W && ga && part && al(a) && al(W) && al(ga) && al(part), "NULL or misaligned argument");
  MMA_REQUIRE(n_blocks == tower_blocks(N), "n_blocks=%lld, expected mma_tower_linear_bwd_blocks(N)=%lld", (long long)n_blocks,
              (long long)tower_blocks(N));
  TowerParams p{gy, a, W, ga, part, N, (N + n_blocks - 1) / n_blocks, T, O, C};
  const int chunks = (C / 4 + kWave - 1) / kWave;
  MMA_REQUIRE((int64_t)T * chunks < 65536, "T * column chunks too large");
  hipLaunchKernelGGL(tower_bwd_kernel, dim3((unsigned)n_blocks, (unsigned)((T * chunks + 3) / 4)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("tower_bwd_kernel");
}
