// K10 / K11: the two non-layer pieces of the reference's training step (node_classification/train.py:72-80), fused.
//   K10  log_softmax over the classes + nll_loss over the training rows  (models.py:68 `F.log_softmax(x, dim=1)` +
//        train.py:77 `F.nll_loss(output[idx_train], labels[idx_train])`): forward writes the (N,C) log-probabilities the
//        model returns and the scalar mean loss; backward writes dL/dlogits = (softmax - onehot) / n on the training rows, 0
//        elsewhere - one launch each instead of ~10 element-wise / index / reduction launches.
//   K11  Adam (train.py:69 `optim.Adam(model.parameters(), lr, weight_decay)`), all parameter tensors in ONE launch: the
//        reference's 25 Parameters are 25 x ~8 tiny launches per step with the stock per-tensor loop.
// Both are deterministic (fixed reduction order, no atomics) and capture-safe (the step counter lives in device memory).
#include "common.h"

namespace mma {

// one wavefront per row: C <= 64 * kMaxPerLane classes
constexpr int kMaxPerLane = 16;

__device__ __forceinline__ float wave_max(float v) {
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

__global__ __launch_bounds__(kBlock) void logsoftmax_kernel(const float* x, int64_t ldx, float* logp, int64_t ldo, int64_t N, int C) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave);
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); r < N; r += stride) {
    float v[kMaxPerLane];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
      const int c = lane + i * kWave;
      v[i] = c < C ? x[r * ldx + c] : -INFINITY;
      m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) if (lane + i * kWave < C) s += expf(v[i] - m);
    const float lse = m + logf(wave_sum(s));
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
      const int c = lane + i * kWave;
      if (c < C) logp[r * ldo + c] = v[i] - lse;
    }
  }
}

// loss = -(1/n) sum_i logp[idx[i], labels[idx[i]]]: ONE workgroup, fixed order (strided partial sums, then a tree)
__global__ __launch_bounds__(kBlock) void nll_mean_kernel(const float* logp, int64_t ldo, const int64_t* idx, const int64_t* labels,
                                                          int64_t n, float* loss, int64_t N, int C) {
  __shared__ float part[kBlock];
  float s = 0.f;
  // rows / labels outside [0,N) x [0,C) contribute nothing (never an out-of-bounds access); the Python side has already raised
  // for them, as F.nll_loss does (mma_amd/train_step.py::_check_targets)
  for (int64_t i = threadIdx.x; i < n; i += kBlock) {
    const int64_t r = idx[i];
    if ((uint64_t)r >= (uint64_t)N) continue;
    const int64_t lab = labels[r];
    if ((uint64_t)lab < (uint64_t)C) s -= logp[r * ldo + lab];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = part[0] / (float)n;
}

// gx[idx[i], c] = gloss * (exp(logp) - [c == label]) / n   (the caller zeroed gx; idx rows are unique)
__global__ __launch_bounds__(kBlock) void nll_logsoftmax_bwd_kernel(const float* logp, int64_t ldo, const int64_t* idx, const int64_t* labels,
                                                                    int64_t n, int C, const float* gloss, float* gx, int64_t ldg, int64_t N) {
  const float scale = *gloss / (float)n;
  const int64_t total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C; const int c = (int)(t % C);
    const int64_t r = idx[i];
    if ((uint64_t)r >= (uint64_t)N) continue;
    gx[r * ldg + c] = scale * (expf(logp[r * ldo + c]) - (labels[r] == c ? 1.f : 0.f));
  }
}

// ---- K12: the graph-regression loss, mean |pred - target| (graph_regression/mma.py:156 `(out.squeeze() - data.y).abs().mean()`) ----
// ONE workgroup, fixed order: the batch is a few thousand graphs at most.
__global__ __launch_bounds__(kBlock) void l1_mean_kernel(const float* pred, const float* target, int64_t n, float* loss) {
  __shared__ float part[kBlock];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kBlock) s += fabsf(pred[i] - target[i]);
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = part[0] / (float)n;
}
// d|d|/dd = sign(d) with sign(0) = 0 (torch's abs backward)
__global__ void l1_mean_bwd_kernel(const float* pred, const float* target, int64_t n, const float* gloss, float* gpred) {
  const float scale = *gloss / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - target[i];
    gpred[i] = d > 0.f ? scale : (d < 0.f ? -scale : (d == d ? 0.f : d));
  }
}

// ---- K11 ------------------------------------------------------------------------------------------------------------
struct AdamTensor { float* p; const float* g; float* m; float* v; int64_t n; int64_t chunk0; };   // chunk0: first chunk id of the tensor
constexpr int64_t kAdamChunk = 4096;     // elements per workgroup

constexpr int kAdamByValue = 120;        // gradient pointers that fit the kernel arguments (960 of the 4096 bytes)
struct AdamGrads { const float* g[kAdamByValue]; };

// BY_VALUE: the gradient pointers come with the launch instead of from the table: autograd may hand over FRESH gradient tensors
// every step (optimizer.zero_grad(set_to_none=True) - no zero-fill and no accumulating add per parameter), and a captured graph
// keeps the pointers of its capture, which its private pool makes the pointers of every replay.
template <bool BY_VALUE>
__global__ __launch_bounds__(kBlock) void adam_kernel(const AdamTensor* tab, int n_tensors, const int32_t* chunk_tensor, float* step,
                                                      double lr_d, double b1_d, double b2_d, float eps, float wd, unsigned* ticket,
                                                      const AdamGrads grads) {
  // torch.optim.Adam (no amsgrad, L2 weight decay folded into the gradient), step t = *step + 1 for every workgroup.
  // The scalars are formed in DOUBLE from double hyper-parameters and only then rounded to fp32, as torch does with its Python
  // floats: 1 - 0.999f is 1.3e-5 off 1e-3, which would put every second-moment update and bias correction off by that much.
  const double t = (double)*step + 1.0;
  const float step_size = (float)(lr_d / (1.0 - pow(b1_d, t)));
  const float bc2s = (float)sqrt(1.0 - pow(b2_d, t));
  const float b2 = (float)b2_d, omb1 = (float)(1.0 - b1_d), omb2 = (float)(1.0 - b2_d);
  const int ti = chunk_tensor[blockIdx.x];
  const AdamTensor T = tab[ti];
  const float* gp = BY_VALUE ? grads.g[ti] : T.g;
  const int64_t base = ((int64_t)blockIdx.x - T.chunk0) * kAdamChunk;
  const int64_t end = min(T.n, base + kAdamChunk);
  for (int64_t i = base + threadIdx.x; i < end; i += kBlock) {
    const float pv = T.p[i];
    const float g = gp[i] + wd * pv;                        // grad.add(param, alpha=weight_decay)
    const float m = T.m[i] + omb1 * (g - T.m[i]);           // exp_avg.lerp_(grad, 1 - beta1)
    const float v = b2 * T.v[i] + omb2 * g * g;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    T.m[i] = m; T.v[i] = v;
    T.p[i] = pv - step_size * (m / (sqrtf(v) / bc2s + eps));   // param.addcdiv_(exp_avg, denom, value=-step_size)
  }
  // the step counter moves when every workgroup has read it: each takes a ticket after its read, the last one ticks (the counter
  // lives behind the table and is left at zero) - a second one-thread launch did this before
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1u) { *step = (float)t; *ticket = 0u; }
}

// ---- K17: BatchNorm1d (training mode) + ReLU over the first n_valid rows of a padded batch, one launch each way ------------------
// mma.py:121 `x = F.relu(batch_norm(conv(...)))` inside the graphed Net step (train_step.GraphedNetStep): the batch is padded to a
// static shape, so the statistics run over the first *n_valid rows (a DEVICE scalar: the captured graph replays on batches of any
// size) while every row is normalised.  As torch ops the masked form is ~30 small launches per layer and direction.
// One workgroup owns 4 columns: 64 row lanes x 4 columns, fixed summation order (a strided partial sum per row lane, then a tree); the
// row loops fetch 8 rows per lane before they add (the batches of the graphed step are ~1 500 rows: with 16 columns per workgroup
// and one load in flight per lane the two kernels took 48 / 77 us, a fifth of the whole Net step).
constexpr int kBnCols = 4, kBnRows = kBlock / kBnCols, kBnAhead = 8;

__device__ __forceinline__ float bn_block_sum(float v, float (*red)[kBnCols], int r, int c) {
  __syncthreads();
  red[r][c] = v;
  __syncthreads();
  for (int o = kBnRows / 2; o > 0; o >>= 1) {
    if (r < o) red[r][c] += red[r + o][c];
    __syncthreads();
  }
  return red[0][c];
}

__global__ __launch_bounds__(kBlock) void masked_bn_relu_fwd_kernel(const float* x, int64_t ldx, const int64_t* n_valid, const float* gamma,
                                                                    const float* beta, float* y, int64_t ldy, float* mean_out, float* rstd_out,
                                                                    float* run_mean, float* run_var, int64_t* n_tracked, float momentum,
                                                                    float eps, int64_t N, int C, int relu) {
  __shared__ float red[kBnRows][kBnCols];
  const int c = threadIdx.x % kBnCols, r = threadIdx.x / kBnCols;
  const int col = (int)blockIdx.x * kBnCols + c;
  const bool cv = col < C;
  const int64_t nv = min(max(*n_valid, (int64_t)1), N);
  const float cnt = (float)nv;
  const int xc = cv ? col : 0;                                         // lanes past C read column 0 and add nothing
  float s = 0.f;
  for (int64_t i = r; i < nv; i += kBnRows * kBnAhead) {
    float v[kBnAhead];
#pragma unroll
    for (int u = 0; u < kBnAhead; ++u) { const int64_t j = i + u * kBnRows; v[u] = x[min(j, nv - 1) * ldx + xc]; }
#pragma unroll
    for (int u = 0; u < kBnAhead; ++u) s += (cv && i + u * kBnRows < nv) ? v[u] : 0.f;
  }
  const float mean = bn_block_sum(s, red, r, c) / cnt;
  float q = 0.f;
  for (int64_t i = r; i < nv; i += kBnRows * kBnAhead) {
    float v[kBnAhead];
#pragma unroll
    for (int u = 0; u < kBnAhead; ++u) { const int64_t j = i + u * kBnRows; v[u] = x[min(j, nv - 1) * ldx + xc]; }
#pragma unroll
    for (int u = 0; u < kBnAhead; ++u) { const float d = (cv && i + u * kBnRows < nv) ? v[u] - mean : 0.f; q += d * d; }
  }
  const float var = bn_block_sum(q, red, r, c) / cnt;                  // biased: what normalises (torch BatchNorm training mode)
  const float rstd = rsqrtf(var + eps);
  if (cv) {
    const float g = gamma ? gamma[col] : 1.f, b = beta ? beta[col] : 0.f;
    for (int64_t i = r; i < N; i += kBnRows * kBnAhead) {
      float v[kBnAhead];
#pragma unroll
      for (int u = 0; u < kBnAhead; ++u) { const int64_t j = i + u * kBnRows; v[u] = x[min(j, N - 1) * ldx + col]; }
#pragma unroll
      for (int u = 0; u < kBnAhead; ++u) {
        const int64_t j = i + u * kBnRows;
        const float w = (v[u] - mean) * rstd * g + b;
        if (j < N) y[j * ldy + col] = relu ? fmaxf(w, 0.f) : w;
      }
    }
    if (r == 0) {
      mean_out[col] = mean; rstd_out[col] = rstd;
      if (run_mean) {                                                  // running averages: the UNBIASED variance, like torch
        run_mean[col] = (1.f - momentum) * run_mean[col] + momentum * mean;
        run_var[col] = (1.f - momentum) * run_var[col] + momentum * var * (cnt / fmaxf(cnt - 1.f, 1.f));
      }
    }
  }
  if (n_tracked && blockIdx.x == 0 && threadIdx.x == 0) *n_tracked += 1;
}

// g = gy * [y > 0] (ReLU);  gx = gamma rstd (g - sum(g)/n - xhat sum(g xhat)/n);  ggamma = sum(g xhat), gbeta = sum(g): sums over ALL rows -
// the rows past n_valid belong to the dummy graph the loss never reads, their g is exactly 0
__global__ __launch_bounds__(kBlock) void masked_bn_relu_bwd_kernel(const float* gy, int64_t ldg, const float* y, int64_t ldy, const float* x,
                                                                    int64_t ldx, const float* mean_in, const float* rstd_in, const float* gamma,
                                                                    const int64_t* n_valid, float* gx, int64_t ldgx, float* ggamma, float* gbeta,
                                                                    int64_t N, int C, int relu) {
  __shared__ float red[kBnRows][kBnCols];
  const int c = threadIdx.x % kBnCols, r = threadIdx.x / kBnCols;
  const int col = (int)blockIdx.x * kBnCols + c;
  const bool cv = col < C;
  const int64_t nv = min(max(*n_valid, (int64_t)1), N);
  const float cnt = (float)nv;
  const float mean = cv ? mean_in[col] : 0.f, rstd = cv ? rstd_in[col] : 0.f;
  const int xc = cv ? col : 0;
  float sg = 0.f, sgx = 0.f;
  for (int64_t i = r; i < N; i += kBnRows * kBnAhead) {
    float vy[kBnAhead], vg[kBnAhead], vx[kBnAhead];
#pragma unroll
    for (int u = 0; u < kBnAhead; ++u) {
      const int64_t j = min(i + u * kBnRows, N - 1);
      vy[u] = y[j * ldy + xc]; vg[u] = gy[j * ldg + xc]; vx[u] = x[j * ldx + xc];
    }
#pragma unroll
    for (int u = 0; u < kBnAhead; ++u) {
      if (cv && i + u * kBnRows < N) {
        const float g = (relu && !(vy[u] > 0.f)) ? 0.f : vg[u];
        sg += g; sgx += g * ((vx[u] - mean) * rstd);
      }
    }
  }
  const float tg = bn_block_sum(sg, red, r, c);
  const float tgx = bn_block_sum(sgx, red, r, c);
  if (cv) {
    const float k = (gamma ? gamma[col] : 1.f) * rstd;
    for (int64_t i = r; i < N; i += kBnRows * kBnAhead) {
      float vy[kBnAhead], vg[kBnAhead], vx[kBnAhead];
#pragma unroll
      for (int u = 0; u < kBnAhead; ++u) {
        const int64_t j = min(i + u * kBnRows, N - 1);
        vy[u] = y[j * ldy + col]; vg[u] = gy[j * ldg + col]; vx[u] = x[j * ldx + col];
      }
#pragma unroll
      for (int u = 0; u < kBnAhead; ++u) {
        const int64_t j = i + u * kBnRows;
        const float g = (relu && !(vy[u] > 0.f)) ? 0.f : vg[u];
        const float xh = (vx[u] - mean) * rstd;
        // the statistics depend on the first nv rows only: a padded row's x reaches nothing but its own y
        if (j < N) gx[j * ldgx + col] = j < nv ? k * (g - tg / cnt - xh * (tgx / cnt)) : k * g;
      }
    }
    if (r == 0) { if (ggamma) ggamma[col] = tgx; if (gbeta) gbeta[col] = tg; }
  }
}

// Dropout seeds of a captured step: seeds[0..n) are what the fused kernels read (DropParams::seed_dev), seeds[n..2n) the splitmix64
// states behind them.  One launch replaces torch's captured random_() (a philox kernel plus two fills of its seed / offset tensors
// that torch enqueues ahead of EVERY replay of a graph that holds a generator op).
__global__ void seed_advance_kernel(uint64_t* seeds, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = (seeds[n + i] += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  seeds[i] = z ^ (z >> 31);
}

}  // namespace mma

using namespace mma;

extern "C" int mma_logsoftmax_nll_fwd(const float* x, int64_t ldx, const int64_t* idx, const int64_t* labels, int64_t n_idx,
                                      float* logp, int64_t ldo, float* loss, int64_t N, int32_t C, void* stream) {
  MMA_REQUIRE(N >= 0 && n_idx >= 0 && C >= 1 && C <= kWave * kMaxPerLane && ldx >= C && ldo >= C, "N=%lld n_idx=%lld C=%d unsupported",
              (long long)N, (long long)n_idx, C);
  if (N == 0) return 0;
  MMA_REQUIRE(x && logp, "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t blocks = (N + 3) / 4;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  hipLaunchKernelGGL(logsoftmax_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, ldx, logp, ldo, N, C);
  if (loss) {
    MMA_REQUIRE(n_idx > 0 && idx && labels, "the loss needs a non-empty idx and the labels");
    hipLaunchKernelGGL(nll_mean_kernel, dim3(1), dim3(kBlock), 0, st, logp, ldo, idx, labels, n_idx, loss, N, C);
  }
  return check_launch("logsoftmax_nll_fwd");
}

extern "C" int mma_logsoftmax_nll_bwd(const float* logp, int64_t ldo, const int64_t* idx, const int64_t* labels, int64_t n_idx,
                                      const float* gloss, float* gx, int64_t ldg, int64_t N, int32_t C, void* stream) {
  MMA_REQUIRE(N >= 0 && n_idx >= 0 && C >= 1 && ldo >= C && ldg >= C, "N=%lld n_idx=%lld C=%d unsupported", (long long)N, (long long)n_idx, C);
  if (N == 0) return 0;
  MMA_REQUIRE(logp && gx && gloss && (n_idx == 0 || (idx && labels)), "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemset2DAsync(gx, (size_t)ldg * 4, 0, (size_t)C * 4, (size_t)N, st) != hipSuccess) return fail(2, "memset of the gradient failed");
  if (n_idx == 0) return 0;
  int64_t blocks = (n_idx * C + kBlock - 1) / kBlock;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  hipLaunchKernelGGL(nll_logsoftmax_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, logp, ldo, idx, labels, n_idx, C, gloss, gx, ldg, N);
  return check_launch("logsoftmax_nll_bwd");
}

extern "C" int64_t mma_adam_table_bytes(int64_t n_tensors, int64_t total_chunks) {
  if (n_tensors < 0 || total_chunks < 0) return -1;
  return n_tensors * (int64_t)sizeof(AdamTensor) + total_chunks * 4 + 4;       // records | chunk -> tensor ids | ticket counter (zero)
}
extern "C" int64_t mma_adam_chunks(int64_t n_elements) { return n_elements <= 0 ? 0 : (n_elements + kAdamChunk - 1) / kAdamChunk; }

static int adam_launch(const void* table, int64_t n_tensors, int64_t total_chunks, float* step, double lr, double beta1, double beta2,
                       float eps, float weight_decay, const uint8_t* grad_ptrs_host, void* stream) {
  MMA_REQUIRE(n_tensors >= 0 && total_chunks >= 0 && total_chunks < (1LL << 31), "bad table size");
  if (n_tensors == 0 || total_chunks == 0) return 0;
  MMA_REQUIRE(table && step, "NULL argument");
  MMA_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.f, "invalid Adam hyper-parameters");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const AdamTensor* tab = static_cast<const AdamTensor*>(table);
  const int32_t* chunk_tensor = reinterpret_cast<const int32_t*>(tab + n_tensors);
  unsigned* ticket = reinterpret_cast<unsigned*>(const_cast<int32_t*>(chunk_tensor) + total_chunks);
  AdamGrads grads{};
  if (grad_ptrs_host) {
    MMA_REQUIRE(n_tensors <= kAdamByValue, "n_tensors=%lld: at most %d gradient pointers travel with the launch", (long long)n_tensors, kAdamByValue);
    for (int64_t i = 0; i < n_tensors; ++i) {
      uint64_t a = 0;
      for (int b = 0; b < 8; ++b) a |= (uint64_t)grad_ptrs_host[i * 8 + b] << (8 * b);
      MMA_REQUIRE(a != 0 && (a & 3u) == 0, "gradient %lld: NULL or misaligned pointer", (long long)i);
      grads.g[i] = reinterpret_cast<const float*>(static_cast<uintptr_t>(a));
    }
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)total_chunks), dim3(kBlock), 0, st, tab, (int)n_tensors, chunk_tensor, step, lr, beta1,
                       beta2, eps, weight_decay, ticket, grads);
  } else {
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)total_chunks), dim3(kBlock), 0, st, tab, (int)n_tensors, chunk_tensor, step, lr, beta1,
                       beta2, eps, weight_decay, ticket, grads);
  }
  return check_launch("adam_step");
}

extern "C" int mma_adam_step(const void* table, int64_t n_tensors, int64_t total_chunks, float* step, double lr, double beta1, double beta2,
                             float eps, float weight_decay, void* stream) {
  return adam_launch(table, n_tensors, total_chunks, step, lr, beta1, beta2, eps, weight_decay, nullptr, stream);
}

extern "C" int64_t mma_adam_max_grads_by_value(void) { return kAdamByValue; }

extern "C" int mma_adam_step_grads(const void* table, int64_t n_tensors, int64_t total_chunks, float* step, double lr, double beta1,
                                   double beta2, float eps, float weight_decay, const uint8_t* grad_ptrs_host, void* stream) {
  MMA_REQUIRE(grad_ptrs_host, "NULL gradient pointer list");
  return adam_launch(table, n_tensors, total_chunks, step, lr, beta1, beta2, eps, weight_decay, grad_ptrs_host, stream);
}

extern "C" int mma_l1_loss_fwd(const float* pred, const float* target, int64_t n, float* loss, void* stream) {
  MMA_REQUIRE(n >= 1, "n=%lld: the mean of an empty batch is undefined", (long long)n);
  MMA_REQUIRE(pred && target && loss, "NULL argument");
  hipLaunchKernelGGL(l1_mean_kernel, dim3(1), dim3(kBlock), 0, static_cast<hipStream_t>(stream), pred, target, n, loss);
  return check_launch("l1_mean_kernel");
}

extern "C" int mma_l1_loss_bwd(const float* pred, const float* target, int64_t n, const float* gloss, float* gpred, void* stream) {
  MMA_REQUIRE(n >= 1, "n=%lld: the mean of an empty batch is undefined", (long long)n);
  MMA_REQUIRE(pred && target && gloss && gpred, "NULL argument");
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), pred, target, n, gloss, gpred);
  return check_launch("l1_mean_bwd_kernel");
}

extern "C" int mma_masked_bn_relu_fwd(const float* x, int64_t ldx, const int64_t* n_valid, const float* gamma, const float* beta, float* y,
                                      int64_t ldy, float* mean_out, float* rstd_out, float* running_mean, float* running_var,
                                      int64_t* n_tracked, float momentum, float eps, int64_t N, int32_t C, int32_t relu, void* stream) {
  MMA_REQUIRE(N >= 1 && C >= 1 && ldx >= C && ldy >= C && momentum >= 0.f && momentum <= 1.f && eps >= 0.f, "N=%lld C=%d unsupported", (long long)N, C);
  MMA_REQUIRE(x && n_valid && y && mean_out && rstd_out && ((running_mean == nullptr) == (running_var == nullptr)), "NULL argument");
  hipLaunchKernelGGL(masked_bn_relu_fwd_kernel, dim3((unsigned)((C + kBnCols - 1) / kBnCols)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     x, ldx, n_valid, gamma, beta, y, ldy, mean_out, rstd_out, running_mean, running_var, n_tracked, momentum, eps, N, (int)C, (int)relu);
  return check_launch("masked_bn_relu_fwd_kernel");
}

extern "C" int mma_masked_bn_relu_bwd(const float* gy, int64_t ldg, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* mean,
                                      const float* rstd, const float* gamma, const int64_t* n_valid, float* gx, int64_t ldgx, float* ggamma,
                                      float* gbeta, int64_t N, int32_t C, int32_t relu, void* stream) {
  MMA_REQUIRE(N >= 1 && C >= 1 && ldx >= C && ldy >= C && ldg >= C && ldgx >= C, "N=%lld C=%d unsupported", (long long)N, C);
  MMA_REQUIRE(gy && y && x && mean && rstd && n_valid && gx, "NULL argument");
  hipLaunchKernelGGL(masked_bn_relu_bwd_kernel, dim3((unsigned)((C + kBnCols - 1) / kBnCols)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     gy, ldg, y, ldy, x, ldx, mean, rstd, gamma, n_valid, gx, ldgx, ggamma, gbeta, N, (int)C, (int)relu);
  return check_launch("masked_bn_relu_bwd_kernel");
}

extern "C" int mma_seed_advance(uint64_t* seeds, int32_t n, void* stream) {
  MMA_REQUIRE(n >= 0 && n <= (1 << 24), "n=%d out of range (0 .. 2^24)", n);
  if (n == 0) return 0;
  MMA_REQUIRE(seeds, "NULL argument");
  hipLaunchKernelGGL(seed_advance_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), seeds, (int)n);
  return check_launch("seed_advance_kernel");
}
