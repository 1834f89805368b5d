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
