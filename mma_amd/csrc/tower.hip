// K9: backward of the per-tower post-NN Linear of MMAConv on the aggregates (reference graph_regression/mma_conv.py:132-134,
//     post_nns[t](cat[x_t, out_t])): for a (N,T,C), W (T,O,C), upstream gradient gy (N,T,O) with a SMALL O (<= 16; ZINC: 15)
//        ga[n,t,c]  = sum_o gy[n,t,o] * W[t,o,c]          (written once, in the layout K4 reads it)
//        gW[t,o,c]  = sum_n gy[n,t,o] * a[n,t,c]          (fixed-order partial sums, finished by K8)
//     in ONE pass over `a`: as two library batched GEMMs with a 15-wide dimension these cost 0.79 + 0.94 ms on the
//     10 000-molecule batch (rocBLAS reaches ~15 TFLOP/s on them), while the data is 1.9 GB in and 1.9 GB out.
// A thread owns 4 consecutive columns c of one tower and keeps W[t, :, c..c+3] (O x 4 floats) and its gW partial (O x 4)
// in registers; per node it reads 16 B of `a`, the wave reads the node's O gradient values with ONE 64-byte load and hands
// them out with v_readlane (SGPR operands), 2 x O x 4 FMAs, one 16-byte store.  The 4 waves of a workgroup walk different
// nodes of the same (tower, column chunk) and fold their partials in wave order through LDS (deterministic).
#include "common.h"

namespace mma {

constexpr int kTowerMaxO = 16;

struct TowerParams {
  const float* gy; const float* a; const float* W;
  float* ga; float* part;               // part: (n_blocks, T, O, C)
  int64_t N, nodes_per_block; int T, O, C;
};

__global__ __launch_bounds__(kBlock) void tower_bwd_kernel(const TowerParams p) {
  __shared__ float4 fold[kTowerMaxO][kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunks = (p.C / 4 + kWave - 1) / kWave;
  const int t = (int)blockIdx.y / chunks, chunk = (int)blockIdx.y % chunks;
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
  const float* gp = p.gy + (size_t)t * p.O + (lane < p.O ? lane : 0);
  float* op = p.ga + (size_t)t * p.C + cc;
  // one node ahead: its loads are in flight while the current node's FMAs issue
  int64_t n = n0 + wave;
  float gv = 0.f;
  float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < n1) { gv = gp[n * grow]; av = *reinterpret_cast<const float4*>(ap + n * row); }
  for (; n < n1; n += kBlock / kWave) {
    const int64_t nn = n + kBlock / kWave;
    float gnext = 0.f;
    float4 anext = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nn < n1) { gnext = gp[nn * grow]; anext = *reinterpret_cast<const float4*>(ap + nn * row); }
    const float gl = lane < p.O ? gv : 0.f;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int o = 0; o < kTowerMaxO; ++o) {     // lanes >= O hold 0 and their w rows are 0: the unused steps add exact zeros
      const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), o));   // the builtin is typed int
      out.x = fmaf(g, w[o].x, out.x); out.y = fmaf(g, w[o].y, out.y); out.z = fmaf(g, w[o].z, out.z); out.w = fmaf(g, w[o].w, out.w);
      acc[o].x = fmaf(g, av.x, acc[o].x); acc[o].y = fmaf(g, av.y, acc[o].y);
      acc[o].z = fmaf(g, av.z, acc[o].z); acc[o].w = fmaf(g, av.w, acc[o].w);
    }
    if (valid) *reinterpret_cast<float4*>(op + n * row) = out;
    gv = gnext; av = anext;
  }
  // fold the four waves' partials in wave order (fixed => bitwise repeatable), then one partial tile per workgroup
  for (int wv = 0; wv < kBlock / kWave; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int o = 0; o < kTowerMaxO; ++o) {
        if (wv == 0) fold[o][lane] = acc[o];
        else {
          float4 f = fold[o][lane];
          f.x += acc[o].x; f.y += acc[o].y; f.z += acc[o].z; f.w += acc[o].w;
          fold[o][lane] = f;
        }
      }
    }
    __syncthreads();
  }
  if (wave == 0 && valid) {
    float* q = p.part + ((size_t)blockIdx.x * p.T + t) * p.O * p.C + c;
    for (int o = 0; o < p.O; ++o) *reinterpret_cast<float4*>(q + (size_t)o * p.C) = fold[o][lane];
  }
}

static int64_t tower_blocks(int64_t N) {
  int64_t b = (N + 255) / 256;            // at least 64 nodes per wave
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
  hipLaunchKernelGGL(tower_bwd_kernel, dim3((unsigned)n_blocks, (unsigned)(T * chunks)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("tower_bwd_kernel");
}
