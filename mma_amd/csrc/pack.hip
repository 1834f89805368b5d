// K18: block pack / unpack - the weight plumbing of one MMAConv call in ONE launch each way.
// The reference's layer keeps T per-tower pre-NN Linears (F, 3F), T post-NN Linears (F_out, (K*S+1)*F) and their biases as separate
// Parameters (mma_conv.py:96-118).  The fused kernels want them as a handful of padded matrices ([Wi;Wj] (2*T*Fw, F), We (T*Fw, F),
// Wx, Wo (T, F_out, K*S*Fw), the biases).  Built with torch.stack / slices / F.pad / cat that is ~25 tiny launches per forward and
// ~35 per backward (zero-fill + copy per slice gradient, an add per accumulation): at ZINC's batch of 64 molecules, where every
// kernel of the layer is ~5 us, more than half of the replayed step.  Here: a table of 2-D blocks, one workgroup per block.
//   pack:    B[b][off + r*ldb + c] = (r < rows && c < cols) ? A[r*lda + c] : 0      for r < b_rows, c < b_cols  (zero padding)
//   unpack:  A[r*lda + c]     (+)= B[b][off + r*ldb + c]                          for r < rows,   c < cols    (the gradients)
#include "common.h"

namespace mma {

constexpr int kPackFields = 10;     // int64 per block: a (address, or float offset from a_base), lda, rows, cols, b index, b offset, ldb, b_rows, b_cols, flags
constexpr int64_t kPackAbsolute = 1;     // flags: `a` is an absolute address even though a_base is given
constexpr int64_t kPackAccumulate = 2;   //        unpack adds onto A (gradient accumulation into a buffer that outlives the call)
constexpr int kPackBases = 8;

struct PackParams { const int64_t* table; float* a_base; float* b[kPackBases]; };

template <bool UNPACK>
__global__ __launch_bounds__(kBlock) void pack_blocks_kernel(const PackParams p) {
  const int64_t* e = p.table + (int64_t)blockIdx.x * kPackFields;
  const int64_t flags = e[9];
  float* a = (p.a_base && !(flags & kPackAbsolute)) ? p.a_base + e[0] : reinterpret_cast<float*>(static_cast<uintptr_t>(e[0]));
  const int64_t lda = e[1];
  const int rows = (int)e[2], cols = (int)e[3];
  float* b = p.b[e[4]] + e[5];
  const int64_t ldb = e[6];
  const int b_rows = (int)e[7], b_cols = (int)e[8];
  if (UNPACK) {
    const int total = rows * cols;
    for (int i = threadIdx.x; i < total; i += kBlock) {
      const int r = i / cols, c = i - r * cols;
      if (flags & kPackAccumulate) a[r * lda + c] += b[r * ldb + c];
      else a[r * lda + c] = b[r * ldb + c];
    }
  } else {
    const int total = b_rows * b_cols;
    for (int i = threadIdx.x; i < total; i += kBlock) {
      const int r = i / b_cols, c = i - r * b_cols;
      b[r * ldb + c] = (r < rows && c < cols) ? a[r * lda + c] : 0.f;
    }
  }
}

}  // namespace mma

using namespace mma;

extern "C" int mma_pack_blocks(const int64_t* table, int64_t n_blocks, float* a_base, float* b0, float* b1, float* b2, float* b3, float* b4,
                               float* b5, float* b6, float* b7, int32_t unpack, void* stream) {
  MMA_REQUIRE(n_blocks >= 0 && n_blocks < (1LL << 31), "n_blocks=%lld out of range", (long long)n_blocks);
  if (n_blocks == 0) return 0;
  MMA_REQUIRE(table, "NULL table");
  PackParams p{table, a_base, {b0, b1, b2, b3, b4, b5, b6, b7}};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (unpack) hipLaunchKernelGGL(pack_blocks_kernel<true>, dim3((unsigned)n_blocks), dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL(pack_blocks_kernel<false>, dim3((unsigned)n_blocks), dim3(kBlock), 0, st, p);
  return check_launch("pack_blocks_kernel");
}

// ---- K19: the edge encoder folded into the pre-NN's edge block, one launch each way ---------------------------------------------
// MMAConv.message applies pre_nn([x_i | x_j | enc(e)]) with enc = Linear(edge_dim -> F) (mma_conv.py:141-146).  The fused path never
// forms enc(e):  enc(e) We^T = e (We Wenc)^T + We benc,  so per layer call it needs  wz = We Wenc (T*Fw, edge_dim)  and
// bz = We benc (T*Fw)  - 380 x 50 x 75 multiply-adds at ZINC's shape.  As torch ops that is a GEMM + a GEMV forward and a TN GEMM, a
// GEMM, a GEMV, an outer product and an add backward: seven ~5 us launches for microseconds of arithmetic.  Plain fp32 FMA chains in a
// fixed order (lengths F, edge_dim + 1, T*Fw).
namespace mma {

constexpr int kFoldRows = 8;      // rows of We (or gwz) a workgroup keeps in LDS
constexpr int kFoldMaxWidth = 512;

struct FoldParams {
  const float* We; int64_t ldw; const float* Wenc; const float* benc;      // We (TF, F) pitch ldw; Wenc (F, ED) contiguous; benc (F) or NULL
  float* wz; float* bz;                                                    // forward out: (TF, ED), (TF)
  const float* gwz; const float* gbz; float* gWe; float* gWenc; float* gbenc;   // backward: in (TF, ED), (TF); out (TF, F), (F, ED), (F)
  int TF, F, ED, row_blocks;
};

__global__ __launch_bounds__(kBlock) void edge_fold_fwd_kernel(const FoldParams p) {
  __shared__ float rows[kFoldRows][kFoldMaxWidth];
  const int r0 = (int)blockIdx.x * kFoldRows;
  const int nr = min(kFoldRows, p.TF - r0);
  for (int e = threadIdx.x; e < nr * p.F; e += kBlock) { const int r = e / p.F, f = e - r * p.F; rows[r][f] = p.We[(int64_t)(r0 + r) * p.ldw + f]; }
  __syncthreads();
  const int cols = p.ED + (p.bz ? 1 : 0);
  for (int o = threadIdx.x; o < nr * cols; o += kBlock) {
    const int r = o / cols, c = o - r * cols;
    float a = 0.f;
    if (c < p.ED) {
#pragma unroll 5
      for (int f = 0; f < p.F; ++f) a = fmaf(rows[r][f], p.Wenc[(int64_t)f * p.ED + c], a);
      p.wz[(int64_t)(r0 + r) * p.ED + c] = a;
    } else {
#pragma unroll 5
      for (int f = 0; f < p.F; ++f) a = fmaf(rows[r][f], p.benc[f], a);
      p.bz[r0 + r] = a;
    }
  }
}

// blocks [0, row_blocks): gWe rows = gwz Wenc^T + gbz (x) benc; the rest: gWenc = We^T gwz and gbenc = We^T gbz, one output per wavefront
__global__ __launch_bounds__(kBlock) void edge_fold_bwd_kernel(const FoldParams p) {
  __shared__ float rows[kFoldRows][kFoldMaxWidth];
  if ((int)blockIdx.x < p.row_blocks) {
    const int r0 = (int)blockIdx.x * kFoldRows;
    const int nr = min(kFoldRows, p.TF - r0);
    for (int e = threadIdx.x; e < nr * p.ED; e += kBlock) { const int r = e / p.ED, c = e - r * p.ED; rows[r][c] = p.gwz[(int64_t)(r0 + r) * p.ED + c]; }
    __syncthreads();
    for (int o = threadIdx.x; o < nr * p.F; o += kBlock) {
      const int r = o / p.F, f = o - r * p.F;
      const float* w = p.Wenc + (int64_t)f * p.ED;
      float a = 0.f;
#pragma unroll 5
      for (int c = 0; c < p.ED; ++c) a = fmaf(rows[r][c], w[c], a);
      if (p.gbz) a = fmaf(p.gbz[r0 + r], p.benc[f], a);
      p.gWe[(int64_t)(r0 + r) * p.F + f] = a;
    }
    return;
  }
  // one WAVEFRONT per output: the lanes split the T*Fw rows of the reduction (a thread per output walked 380 dependent cache misses)
  const int cols = p.ED + (p.gbenc ? 1 : 0);
  const int lane = threadIdx.x & (kWave - 1);
  const int o = ((int)blockIdx.x - p.row_blocks) * (kBlock / kWave) + (int)(threadIdx.x >> 6);
  if (o >= p.F * cols) return;                       // wave-uniform
  const int f = o / cols, c = o - f * cols;
  float a = 0.f;
  if (c < p.ED) {
    for (int r = lane; r < p.TF; r += kWave) a = fmaf(p.We[(int64_t)r * p.ldw + f], p.gwz[(int64_t)r * p.ED + c], a);
  } else {
    for (int r = lane; r < p.TF; r += kWave) a = fmaf(p.We[(int64_t)r * p.ldw + f], p.gbz[r], a);
  }
  for (int off = kWave / 2; off > 0; off >>= 1) a += __shfl_xor(a, off, kWave);
  if (lane == 0) { if (c < p.ED) p.gWenc[(int64_t)f * p.ED + c] = a; else p.gbenc[f] = a; }
}

}  // namespace mma

static int fold_check(int64_t TF, int32_t F, int32_t ED) {
  MMA_REQUIRE(TF >= 1 && TF < (1 << 24) && F >= 1 && F <= kFoldMaxWidth && ED >= 1 && ED <= kFoldMaxWidth, "TF=%lld F=%d ED=%d unsupported (widths <= %d)",
              (long long)TF, F, ED, kFoldMaxWidth);
  return 0;
}

// [r4] The operands of the zero-padded tall Linear (dense._LinearX3: 75 -> 760 on 2e5 rows, 50 -> 380 on 4e5): out (M_out, width) =
// [x | col | 0 ...] from x (M, fin), fin < width, width % 4 == 0; col (M,) or NULL = a column of ones; rows M .. M_out - 1 are zero.
// A = [x | 1 | 0] carries the bias through the GEMM, [W | b | 0] (fout rows padded to a multiple of 128) is the weight operand of the
// forward (as its transposed view) AND of dL/dx (as its leading columns).  As torch ops: a zero fill of the whole buffer, a strided copy
// and a strided fill per operand, and a second zero-padded copy of W in backward.
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int fin, const float* __restrict__ col,
                                                       const int32_t* __restrict__ row_index, float* __restrict__ out, int64_t ldo, int width,
                                                       int64_t M_out) {
  const int q4 = width / 4;
  const int64_t total = M_out * q4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q4;
    const int c = (int)(i % q4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < M) {
      const float* xr = x + (row_index ? (int64_t)row_index[r] : r) * ldx;     // [r5] out row r = x row row_index[r]: the gather rides along
      const float one = col ? col[r] : 1.f;
      v.x = c < fin ? xr[c] : (c == fin ? one : 0.f);
      v.y = c + 1 < fin ? xr[c + 1] : (c + 1 == fin ? one : 0.f);
      v.z = c + 2 < fin ? xr[c + 2] : (c + 2 == fin ? one : 0.f);
      v.w = c + 3 < fin ? xr[c + 3] : (c + 3 == fin ? one : 0.f);
    }
    *reinterpret_cast<float4*>(out + r * ldo + c) = v;
  }
}

extern "C" int mma_pad_rows(const float* x, int64_t ldx, int64_t M, int32_t fin, const float* col, const int32_t* row_index, float* out,
                            int64_t ldo, int32_t width, int64_t M_out, void* stream) {
  MMA_REQUIRE(M >= 0 && M_out >= M && fin >= 1 && width > fin && width % 4 == 0 && ldx >= fin && ldo >= width && ldo % 4 == 0,
              "M=%lld M_out=%lld fin=%d width=%d ldx=%lld ldo=%lld unsupported (fin < width, width and ldo multiples of 4)", (long long)M,
              (long long)M_out, fin, width, (long long)ldx, (long long)ldo);
  if (M_out == 0) return 0;
  MMA_REQUIRE((x || M == 0) && out && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "NULL or misaligned argument");
  const int64_t total = M_out * (width / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8 * kMaxGrid) blocks = 8 * kMaxGrid;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, M, (int)fin, col, row_index,
                     out, ldo, (int)width, M_out);
  return check_launch("pad_rows_kernel");
}

extern "C" int mma_edge_fold_fwd(const float* We, int64_t ldw, const float* Wenc, const float* benc, float* wz, float* bz, int64_t TF, int32_t F,
                                 int32_t ED, void* stream) {
  if (int rc = fold_check(TF, F, ED)) return rc;
  MMA_REQUIRE(We && Wenc && wz && ldw >= F && ((benc == nullptr) == (bz == nullptr)), "NULL argument (benc and bz come together)");
  FoldParams p{};
  p.We = We; p.ldw = ldw; p.Wenc = Wenc; p.benc = benc; p.wz = wz; p.bz = bz; p.TF = (int)TF; p.F = F; p.ED = ED;
  hipLaunchKernelGGL(edge_fold_fwd_kernel, dim3((unsigned)((TF + kFoldRows - 1) / kFoldRows)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("edge_fold_fwd_kernel");
}

extern "C" int mma_edge_fold_bwd(const float* We, int64_t ldw, const float* Wenc, const float* benc, const float* gwz, const float* gbz, float* gWe,
                                 float* gWenc, float* gbenc, int64_t TF, int32_t F, int32_t ED, void* stream) {
  if (int rc = fold_check(TF, F, ED)) return rc;
  MMA_REQUIRE(We && Wenc && gwz && gWe && gWenc && ldw >= F, "NULL argument");
  MMA_REQUIRE((gbz == nullptr) == (gbenc == nullptr) && (gbz == nullptr || benc != nullptr), "gbz, gbenc and benc come together");
  FoldParams p{};
  p.We = We; p.ldw = ldw; p.Wenc = Wenc; p.benc = benc; p.gwz = gwz; p.gbz = gbz; p.gWe = gWe; p.gWenc = gWenc; p.gbenc = gbenc;
  p.TF = (int)TF; p.F = F; p.ED = ED; p.row_blocks = (int)((TF + kFoldRows - 1) / kFoldRows);
  const int64_t outs = (int64_t)F * (ED + (gbenc ? 1 : 0));
  hipLaunchKernelGGL(edge_fold_bwd_kernel, dim3((unsigned)(p.row_blocks + (outs + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("edge_fold_bwd_kernel");
}
