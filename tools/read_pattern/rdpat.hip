// read-pattern micro-benchmark: agg (N, T*KF) fp32, workgroup = (tower t, 256 node rows), wave = 64 rows.
// variant 0: per step every lane (j = lane & 15, kg = lane >> 4) loads 2 float4 of row (nt*16 + j) at columns 32 ks + 4 kg and + 16: 4 nt per step, 5 steps (K13's pattern)
// variant 1: full rows: for nt: lane loads its 10 float4 (5 steps x 2) of row nt*16+j back to back
// variant 2: 8 lanes per row x 128 B contiguous, 8 rows per instruction, 8 instructions per k-round of 32 columns (the fp32 kernel's pattern)
// variant 3: rows contiguous: lane l of instruction i reads float4 at byte (i*1024 + l*16) of the tower row (152 floats = 38 float4 per row: lanes 0..37), one row per instruction
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void rd(const float* __restrict__ agg, float* __restrict__ out, int64_t N, int T, int KF, int variant, int tiles_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lin = blockIdx.x;
  const int t = lin % T; const int64_t bx = lin / T;
  const int j = lane & 15, kg = lane >> 4;
  const int64_t lda = (int64_t)T * KF;
  float4 s = make_float4(0, 0, 0, 0);
  for (int rt = 0; rt < tiles_per_wave; ++rt) {
    const int64_t n0 = ((bx * 4 + wave) * tiles_per_wave + rt) * 64;
    if (n0 >= N) break;
    if (variant == 0) {
      for (int ks = 0; ks < 5; ++ks) {
        float4 v[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int64_t row = min(n0 + nt * 16 + j, N - 1);
          const float* b = agg + row * lda + (int64_t)t * KF;
          v[2 * nt] = *(const float4*)(b + min(32 * ks + 4 * kg, KF - 4));
          v[2 * nt + 1] = *(const float4*)(b + min(32 * ks + 16 + 4 * kg, KF - 4));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
      }
    } else if (variant == 1) {
      for (int nt = 0; nt < 4; ++nt) {
        const int64_t row = min(n0 + nt * 16 + j, N - 1);
        const float* b = agg + row * lda + (int64_t)t * KF;
        float4 v[10];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
          v[2 * ks] = *(const float4*)(b + min(32 * ks + 4 * kg, KF - 4));
          v[2 * ks + 1] = *(const float4*)(b + min(32 * ks + 16 + 4 * kg, KF - 4));
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
      }
    } else if (variant == 2) {
      const int lrow = lane >> 3, lq = lane & 7;
      for (int ks = 0; ks < 5; ++ks) {
        float4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int64_t row = min(n0 + r * 8 + lrow, N - 1);
          v[r] = *(const float4*)(agg + row * lda + (int64_t)t * KF + min(32 * ks + 4 * lq, KF - 4));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
      }
    } else {
      for (int r0 = 0; r0 < 64; r0 += 8) {
        float4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int64_t row = min(n0 + r0 + r, N - 1);
          v[r] = *(const float4*)(agg + row * lda + (int64_t)t * KF + min(4 * lane, KF - 4));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
      }
    }
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;      // keep the loads
}
extern "C" int rd_launch(const float* agg, float* out, int64_t N, int T, int KF, int variant, int tiles_per_wave, int block_major, void* stream) {
  const int64_t tiles = (N + 63) / 64;
  const int64_t per = 4 * (int64_t)tiles_per_wave;
  const unsigned blocks = (unsigned)((tiles + per - 1) / per * T);
  hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, agg, out, N, T, KF, variant, tiles_per_wave);
  return (int)hipGetLastError();
}
