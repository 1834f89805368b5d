// K3 / K4 / K6: graph-regression form of MMA (reference graph_regression/mma_conv.py:121-196).
//
// K3 fuses, for one target node per wavefront, what the reference does with ~K+12 launches:
//   message  h_e = drop( U[i] + V[j] + Z[e] )            (= per-tower Linear([x_i || x_j || enc(e_ij)]) + dropout,
//                                                          mma_conv.py:138-157, split as three dense GEMMs)
//   K x torch_scatter.scatter(h, index, reduce)           sum / mean / min(+arg) / max(+arg) / var / std  (:164-172)
//   degree + compounding scalers + final layout           (:176-196)  out[n, t, s*K*F + k*F + f]
// in ONE pass over the node's target-sorted edge segment.  Edges are sorted by target with a STABLE radix sort
// (K6, rocPRIM), so inside a segment the original edge positions ascend; min/max ties therefore resolve to the
// lowest edge position exactly like torch_scatter's sequential CPU kernel, and the sub-row butterfly keeps that
// order by comparing (value, edge id) lexicographically.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <limits.h>

namespace mma {

enum { GR_SUM = 0, GR_MEAN = 1, GR_MIN = 2, GR_MAX = 3, GR_VAR = 4, GR_STD = 5 };
enum { SC_IDENTITY = 0, SC_AMPLIFICATION = 1, SC_ATTENUATION = 2, SC_LINEAR = 3, SC_INVERSE_LINEAR = 4 };

struct GrParams {
  const int32_t* rowptr; const int32_t* src; const int32_t* perm;
  const float* U; const float* V; int64_t lduv; const float* Z; int64_t ldz;   // fused-message mode
  const float* inputs; int64_t ldi;                                              // given-messages mode (E,D)
  float* out; const float* gout;                                                 // (N, T, S*K*F)
  int32_t* amin; int32_t* amax; float* mean; float* var; int64_t ldsave;        // (N,ldsave>=D) saved for backward (may be NULL)
  float* gmsg; int64_t ldg;                                                      // backward: (E,D) by original edge id
  int N, D, T, F, K, S, lpr_log;
  uint8_t aggr[MMA_MAX_K]; uint8_t scaler[8];
  float avg_log, avg_lin;
  DropParams drop;
};

__device__ __forceinline__ float scaler_factor(int code, float deg, float avg_log, float avg_lin) {
  switch (code) {
    case SC_AMPLIFICATION: return logf(deg + 1.f) / avg_log;
    case SC_ATTENUATION: return avg_log / logf(deg + 1.f);
    case SC_LINEAR: return deg / avg_lin;
    case SC_INVERSE_LINEAR: return avg_lin / deg;
    default: return 1.f;
  }
}

// the VEC messages h_e[c..c+VEC) of edge e (fused: drop(U[i] + V[j] + Z[e]); given: inputs[e])
template <int VEC>
__device__ __forceinline__ Vec<VEC> gr_message(const GrParams& p, const DropParams& dp, bool fused, const Vec<VEC>& u, int j,
                                               uint32_t e, int cc) {
  if (!fused) return ldv<VEC>(p.inputs + (size_t)e * p.ldi + cc);
  Vec<VEC> h = ldv<VEC>(p.V + (size_t)j * p.lduv + cc);
#pragma unroll
  for (int i = 0; i < VEC; ++i) h.v[i] += u.v[i];
  if (p.Z) {
    const Vec<VEC> z = ldv<VEC>(p.Z + (size_t)e * p.ldz + cc);
#pragma unroll
    for (int i = 0; i < VEC; ++i) h.v[i] += z.v[i];
  }
  if (dp.mode != MMA_DROP_NONE) {
    float f[VEC];
    drop_factors<VEC>(dp, e, 0, cc, p.D, 0, f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) h.v[i] *= f[i];
  }
  return h;
}

// VEC = 4 (F % 4 == 0 and 16-byte aligned rows everywhere): a lane owns 4 consecutive columns of one tower and moves one
// dwordx4 per row, also for the (N,T,S*K*F) output.  Otherwise (e.g. ZINC's F = 75) VEC = 1: lanes own consecutive
// columns, so all accesses are still coalesced 4-byte ones.
template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_fwd_kernel(const GrParams p) {
  const DropParams dp = drop_resolve(p.drop);
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log, epg = kWave >> p.lpr_log;
  const int sub = lane >> p.lpr_log;
  const int c = ((int)blockIdx.y * lpr + (lane & (lpr - 1))) * VEC;
  const bool valid = c < p.D;
  const int cc = valid ? c : 0;
  const bool fused = p.inputs == nullptr;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave);
  for (int64_t n0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); n0 < p.N; n0 += stride) {
    const int node = __builtin_amdgcn_readfirstlane((int)n0);
    const int ebeg = p.rowptr[node], eend = p.rowptr[node + 1];
    const Vec<VEC> u = fused ? ldv<VEC>(p.U + (size_t)node * p.lduv + cc) : vzero<VEC>();
    float sum[VEC], sq[VEC], mn[VEC], mx[VEC];
    int an[VEC], ax[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sum[i] = 0.f; sq[i] = 0.f; mn[i] = INFINITY; mx[i] = -INFINITY; an[i] = INT_MAX; ax[i] = INT_MAX; }
    for (int base = ebeg; base < eend; base += kWave) {
      const int cnt = min(kWave, eend - base);
      const int myj = (lane < cnt) ? p.src[base + lane] : 0;
      const int mye = (lane < cnt) ? p.perm[base + lane] : 0;
      for (int t0 = 0; t0 < cnt; t0 += epg) {
        const int t = t0 + sub;
        const int j_ = __shfl(myj, t & (kWave - 1), kWave);
        const int e_ = __shfl(mye, t & (kWave - 1), kWave);
        const bool ev = t < cnt;
        const int j = ev ? j_ : node;
        const int e = ev ? e_ : mye;       // any valid edge id of this segment
        const Vec<VEC> h = gr_message<VEC>(p, dp, fused, u, j, (uint32_t)e, cc);
        if (ev) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            sum[i] += h.v[i]; sq[i] += h.v[i] * h.v[i];
            if (h.v[i] < mn[i]) { mn[i] = h.v[i]; an[i] = e; }   // strict: the first (lowest position) extremal edge wins
            if (h.v[i] > mx[i]) { mx[i] = h.v[i]; ax[i] = e; }
          }
        }
      }
    }
    for (int off = kWave / 2; off >= lpr; off >>= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sum[i] += __shfl_xor(sum[i], off, kWave);
        sq[i] += __shfl_xor(sq[i], off, kWave);
        const float omn = __shfl_xor(mn[i], off, kWave); const int oan = __shfl_xor(an[i], off, kWave);
        const float omx = __shfl_xor(mx[i], off, kWave); const int oax = __shfl_xor(ax[i], off, kWave);
        if (omn < mn[i] || (omn == mn[i] && oan < an[i])) { mn[i] = omn; an[i] = oan; }
        if (omx > mx[i] || (omx == mx[i] && oax < ax[i])) { mx[i] = omx; ax[i] = oax; }
      }
    }
    if (sub == 0 && valid) {
      const int cnt = eend - ebeg;
      const float deg = (float)max(cnt, 1);                       // degree(...).clamp_(1), mma_conv.py:178-179
      float fac[8];
      for (int s = 0; s < p.S; ++s) fac[s] = scaler_factor(p.scaler[s], deg, p.avg_log, p.avg_lin);
      // VEC == 4 is only launched when F % 4 == 0: the 4 columns of a lane then sit in one tower, 16-byte aligned in `out`
      const int t = c / p.F, f = c - t * p.F;
      float* o = p.out + ((size_t)node * p.T + t) * ((size_t)p.S * p.K * p.F) + f;
      float mean[VEC], var[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        mean[i] = sum[i] / deg;                                   // scatter mean: sum / clamp(count, 1)
        var[i] = sq[i] / deg - mean[i] * mean[i];                 // mma_conv.py:167-170
      }
      for (int k = 0; k < p.K; ++k) {
        Vec<VEC> run;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          switch (p.aggr[k]) {
            case GR_SUM: run.v[i] = sum[i]; break;
            case GR_MEAN: run.v[i] = mean[i]; break;
            case GR_MIN: run.v[i] = cnt ? mn[i] : 0.f; break;    // empty target -> 0 (torch_scatter)
            case GR_MAX: run.v[i] = cnt ? mx[i] : 0.f; break;
            case GR_VAR: run.v[i] = var[i]; break;
            default: run.v[i] = sqrtf(fmaxf(var[i], 0.f) + 1e-5f); break;
          }
        }
        for (int s = 0; s < p.S; ++s) {                           // compounding (G7)
#pragma unroll
          for (int i = 0; i < VEC; ++i) run.v[i] = run.v[i] * fac[s];
          stv<VEC>(o + (size_t)(s * p.K + k) * p.F, run);
        }
      }
      const size_t so = (size_t)node * p.ldsave + c;
      if (p.amin) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.amin[so + i] = cnt ? an[i] : -1;
      }
      if (p.amax) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.amax[so + i] = cnt ? ax[i] : -1;
      }
      if (p.mean) {
        Vec<VEC> mv, vv;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { mv.v[i] = mean[i]; vv.v[i] = var[i]; }
        stv<VEC>(p.mean + so, mv);
        stv<VEC>(p.var + so, vv);
      }
    }
  }
}

// K4: gradient w.r.t. every edge message, written by original edge id (each edge has exactly one target: no conflicts)
template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_bwd_kernel(const GrParams p) {
  const DropParams dp = drop_resolve(p.drop);
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log, epg = kWave >> p.lpr_log;
  const int sub = lane >> p.lpr_log;
  const int c = ((int)blockIdx.y * lpr + (lane & (lpr - 1))) * VEC;
  const bool valid = c < p.D;
  const int cc = valid ? c : 0;
  const bool fused = p.inputs == nullptr;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave);
  for (int64_t n0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); n0 < p.N; n0 += stride) {
    const int node = __builtin_amdgcn_readfirstlane((int)n0);
    const int ebeg = p.rowptr[node], eend = p.rowptr[node + 1];
    if (ebeg == eend) continue;
    const float deg = (float)(eend - ebeg);
    float fac[8];
    for (int s = 0; s < p.S; ++s) fac[s] = scaler_factor(p.scaler[s], deg, p.avg_log, p.avg_lin);
    float c_all[VEC], c_min[VEC], c_max[VEC], c_var[VEC], mean[VEC];   // coefficients of 1, [e==amin], [e==amax], (h - mean)
    int an[VEC], ax[VEC];
    bool need_h = false;
    {
      const int t = cc / p.F, f = cc - t * p.F;                   // VEC == 4 only with F % 4 == 0 (one tower per lane)
      const float* go = p.gout + ((size_t)node * p.T + t) * ((size_t)p.S * p.K * p.F) + f;
      const size_t so = (size_t)node * p.ldsave + cc;
#pragma unroll
      for (int i = 0; i < VEC; ++i) c_all[i] = c_min[i] = c_max[i] = c_var[i] = 0.f;
      for (int k = 0; k < p.K; ++k) {
        float gb[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) gb[i] = 0.f;
        float run = 1.f;
        for (int s = 0; s < p.S; ++s) {
          run = run * fac[s];
          const Vec<VEC> gv = ldv<VEC>(go + (size_t)(s * p.K + k) * p.F);
#pragma unroll
          for (int i = 0; i < VEC; ++i) gb[i] = fmaf(gv.v[i], run, gb[i]);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          switch (p.aggr[k]) {
            case GR_SUM: c_all[i] += gb[i]; break;
            case GR_MEAN: c_all[i] += gb[i] / deg; break;
            case GR_MIN: c_min[i] += gb[i]; break;
            case GR_MAX: c_max[i] += gb[i]; break;
            case GR_VAR: c_var[i] += gb[i] * 2.f / deg; need_h = true; break;
            default: {
              const float v = p.var[so + i];
              if (v > 0.f) c_var[i] += gb[i] / (2.f * sqrtf(v + 1e-5f)) * 2.f / deg;   // relu'(v) = [v > 0]
              need_h = true;
            } break;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        an[i] = p.amin ? p.amin[so + i] : -1;
        ax[i] = p.amax ? p.amax[so + i] : -1;
        mean[i] = need_h ? p.mean[so + i] : 0.f;
      }
    }
    const Vec<VEC> u = (fused && need_h) ? ldv<VEC>(p.U + (size_t)node * p.lduv + cc) : vzero<VEC>();
    for (int base = ebeg; base < eend; base += kWave) {
      const int cnt = min(kWave, eend - base);
      const int myj = (lane < cnt) ? p.src[base + lane] : 0;
      const int mye = (lane < cnt) ? p.perm[base + lane] : 0;
      for (int t0 = 0; t0 < cnt; t0 += epg) {
        const int tt = t0 + sub;
        const int j_ = __shfl(myj, tt & (kWave - 1), kWave);
        const int e_ = __shfl(mye, tt & (kWave - 1), kWave);
        if (tt < cnt && valid) {
          Vec<VEC> g;
          Vec<VEC> h = vzero<VEC>();
          if (need_h) h = gr_message<VEC>(p, dp, fused, u, j_, (uint32_t)e_, cc);
          float fd[VEC];
          if (fused && dp.mode != MMA_DROP_NONE) drop_factors<VEC>(dp, (uint32_t)e_, 0, cc, p.D, 0, fd);
          else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) fd[i] = 1.f;
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            float gi = c_all[i] + (e_ == an[i] ? c_min[i] : 0.f) + (e_ == ax[i] ? c_max[i] : 0.f);
            if (need_h) gi += c_var[i] * (h.v[i] - mean[i]);
            g.v[i] = gi * fd[i];
          }
          stv<VEC>(p.gmsg + (size_t)e_ * p.ldg + cc, g);
        }
      }
    }
  }
}

// ---- K6: CSR by key (stable), device side ---------------------------------------------------------------
__global__ void csr_prepare_kernel(const int64_t* key, int64_t E, int32_t* key32, int32_t* iota) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    key32[i] = (int32_t)key[i];
    iota[i] = (int32_t)i;
  }
}
// rowptr[n] = first sorted position whose key >= n
__global__ void csr_rowptr_kernel(const int32_t* skey, int64_t E, int64_t N, int32_t* rowptr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= E; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lo = (i == 0) ? 0 : min((int64_t)skey[i - 1] + 1, N + 1);
    const int64_t hi = (i == E) ? N + 1 : min((int64_t)skey[i] + 1, N + 1);   // rows (lo-1 .. hi-1] start at i
    for (int64_t r = lo; r < hi; ++r) rowptr[r] = (int32_t)i;
  }
}
__global__ void csr_gather_kernel(const int64_t* other, const int32_t* perm, int64_t E, int32_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int32_t)other[perm[i]];
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static int sort_bits(int64_t N) { int b = 1; while ((1LL << b) < N) ++b; return b; }

static int fill_codes(const uint8_t* aggr_host, int K, const uint8_t* scaler_host, int S, GrParams* p) {
  MMA_REQUIRE(K >= 1 && K <= MMA_MAX_K && S >= 1 && S <= 8, "K=%d (1..%d) / S=%d (1..8) unsupported", K, MMA_MAX_K, S);
  for (int k = 0; k < K; ++k) {
    MMA_REQUIRE(aggr_host[k] <= GR_STD, "aggregator code %d unknown", (int)aggr_host[k]);
    p->aggr[k] = aggr_host[k];
  }
  for (int s = 0; s < S; ++s) {
    MMA_REQUIRE(scaler_host[s] <= SC_INVERSE_LINEAR, "scaler code %d unknown", (int)scaler_host[s]);
    p->scaler[s] = scaler_host[s];
  }
  p->K = K; p->S = S;
  return 0;
}

// 16-byte row access is possible when every message operand has a pitch that is a multiple of 4 covering D rounded up,
// 16-byte aligned bases, and the saved arrays are padded the same way
static bool gr_vec4(const GrParams& p) {
  if (p.F % 4 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.out ? (const void*)p.out : (const void*)p.gout) & 15) != 0) return false;
  const int Dp = p.D;
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.inputs) {
    if (!(p.ldi % 4 == 0 && p.ldi >= Dp && al(p.inputs))) return false;
  } else {
    if (!(p.lduv % 4 == 0 && p.lduv >= Dp && al(p.U) && al(p.V))) return false;
    if (p.Z && !(p.ldz % 4 == 0 && p.ldz >= Dp && al(p.Z))) return false;
  }
  if ((p.amin || p.amax || p.mean || p.var) && p.ldsave < Dp) return false;
  return true;
}

static dim3 gr_grid(int64_t N, int D, int vec, int* lpr_log) {
  const int per_row = (D + vec - 1) / vec;
  // lanes per row: the power of two in {64,32,16} that wastes the fewest lanes (94 quads -> 3 x 32, not 2 x 64)
  int best = min(ilog2_ceil(per_row), 6);
  if (per_row > 16) {
    int best_waste = 1 << 30;
    for (int lg = 6; lg >= 4; --lg) {
      const int waste = ((per_row + (1 << lg) - 1) >> lg << lg) - per_row;
      if (waste < best_waste) { best_waste = waste; best = lg; }
    }
  }
  *lpr_log = best;
  const int chunks = (per_row + (1 << *lpr_log) - 1) >> *lpr_log;
  int64_t blocks = (N + 3) / 4;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks, (unsigned)chunks);
}

}  // namespace mma

using namespace mma;

extern "C" int64_t mma_csr_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0 || E >= (1LL << 31) || N >= (1LL << 31)) return -1;
  size_t temp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, (int)E, 0, sort_bits(N));
  return (int64_t)(3 * align256((size_t)E * 4) + align256(temp) + 256);
}

extern "C" int mma_build_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm,
                             int32_t* other_sorted, void* workspace, int64_t workspace_bytes, void* stream) {
  MMA_REQUIRE(E >= 0 && N >= 0 && E < (1LL << 31) && N < (1LL << 31), "E=%lld N=%lld out of int32 range", (long long)E, (long long)N);
  MMA_REQUIRE(rowptr != nullptr, "NULL rowptr");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (E == 0) {
    (void)hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 4, st);
    return check_launch("csr memset");
  }
  MMA_REQUIRE(key && perm && workspace, "NULL argument");
  MMA_REQUIRE(workspace_bytes >= mma_csr_workspace_bytes(E, N), "workspace too small: %lld < %lld", (long long)workspace_bytes,
              (long long)mma_csr_workspace_bytes(E, N));
  char* w = static_cast<char*>(workspace);
  w = reinterpret_cast<char*>(align256(reinterpret_cast<size_t>(w)));
  const size_t a = align256((size_t)E * 4);
  int32_t* key32 = reinterpret_cast<int32_t*>(w);
  int32_t* skey = reinterpret_cast<int32_t*>(w + a);
  int32_t* iota = reinterpret_cast<int32_t*>(w + 2 * a);
  void* temp = w + 3 * a;
  size_t temp_bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, key32, skey, iota, perm, (int)E, 0, sort_bits(N));
  const int blocks = (int)min((int64_t)kMaxGrid, (E + kBlock) / kBlock);
  hipLaunchKernelGGL(csr_prepare_kernel, dim3(blocks), dim3(kBlock), 0, st, key, E, key32, iota);
  const hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key32, skey, iota, perm, (int)E, 0, sort_bits(N), st);
  if (e != hipSuccess) return fail(100 + (int)e, "radix sort failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3(blocks), dim3(kBlock), 0, st, skey, E, N, rowptr);
  if (other && other_sorted) hipLaunchKernelGGL(csr_gather_kernel, dim3(blocks), dim3(kBlock), 0, st, other, perm, E, other_sorted);
  return check_launch("csr build");
}

extern "C" int mma_gr_fused_fwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, const float* inputs, int64_t ldi,
    float* out, int32_t* amin, int32_t* amax, float* mean, float* var, int64_t ldsave,
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) && E < (1LL << 31) && T >= 1 && F >= 1, "N=%lld E=%lld T=%d F=%d unsupported",
              (long long)N, (long long)E, T, F);
  GrParams p{};
  if (int rc = fill_codes(aggr_host, K, scaler_host, S, &p)) return rc;
  if (N == 0) return 0;
  const int D = T * F;
  MMA_REQUIRE(rowptr && out, "NULL argument");
  MMA_REQUIRE(E == 0 || (src && perm), "NULL CSR arrays");
  MMA_REQUIRE((inputs != nullptr) != (U != nullptr && V != nullptr), "give either `inputs` or U and V");
  MMA_REQUIRE(inputs ? ldi >= D : (lduv >= D && (!Z || ldz >= D)), "row pitch too small");
  MMA_REQUIRE(drop_mode == MMA_DROP_NONE || drop_mode == MMA_DROP_HASH, "GR dropout: NONE or HASH");
  MMA_REQUIRE(drop_mode == MMA_DROP_NONE || drop_thr < 256, "drop_thr out of range");
  p.rowptr = rowptr; p.src = src; p.perm = perm; p.U = U; p.V = V; p.lduv = lduv; p.Z = Z; p.ldz = ldz;
  p.inputs = inputs; p.ldi = ldi; p.out = out; p.amin = amin; p.amax = amax; p.mean = mean; p.var = var; p.ldsave = ldsave;
  MMA_REQUIRE(!(amin || amax || mean || var) || ldsave >= D, "ldsave=%lld < T*F", (long long)ldsave);
  p.N = (int)N; p.D = D; p.T = T; p.F = F; p.avg_log = avg_log; p.avg_lin = avg_lin;
  p.drop.mode = (drop_mode == MMA_DROP_HASH && drop_thr > 0 && !inputs) ? MMA_DROP_HASH : MMA_DROP_NONE;
  p.drop.thr = drop_thr; p.drop.scale = 256.f / (256.f - (float)drop_thr);
  p.drop.seed_lo = (uint32_t)seed; p.drop.seed_hi = (uint32_t)(seed >> 32); p.drop.seed_dev = seed_dev; p.drop.keep = nullptr; p.drop.E = E; p.drop.edge_base = 0;
  const bool v4 = gr_vec4(p);
  const dim3 grid = gr_grid(N, D, v4 ? 4 : 1, &p.lpr_log);
  if (v4) hipLaunchKernelGGL((gr_fwd_kernel<4>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL((gr_fwd_kernel<1>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("gr_fwd_kernel");
}

extern "C" int mma_gr_fused_bwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, const float* inputs, int64_t ldi,
    const float* gout, const int32_t* amin, const int32_t* amax, const float* mean, const float* var, int64_t ldsave,
    float* gmsg, int64_t ldg,
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) && E < (1LL << 31) && T >= 1 && F >= 1, "N=%lld E=%lld T=%d F=%d unsupported",
              (long long)N, (long long)E, T, F);
  GrParams p{};
  if (int rc = fill_codes(aggr_host, K, scaler_host, S, &p)) return rc;
  if (N == 0 || E == 0) return 0;
  const int D = T * F;
  MMA_REQUIRE(rowptr && src && perm && gout && gmsg && ldg >= D, "NULL argument or pitch too small");
  bool need_min = false, need_max = false, need_stats = false;
  for (int k = 0; k < K; ++k) {
    need_min |= p.aggr[k] == GR_MIN; need_max |= p.aggr[k] == GR_MAX; need_stats |= p.aggr[k] >= GR_VAR;
  }
  MMA_REQUIRE((!need_min || amin) && (!need_max || amax) && (!need_stats || (mean && var)), "saved forward state missing");
  MMA_REQUIRE(!need_stats || (inputs != nullptr) != (U != nullptr && V != nullptr), "var/std backward needs the messages");
  MMA_REQUIRE(drop_mode == MMA_DROP_NONE || drop_mode == MMA_DROP_HASH, "GR dropout: NONE or HASH");
  p.rowptr = rowptr; p.src = src; p.perm = perm; p.U = U; p.V = V; p.lduv = lduv; p.Z = Z; p.ldz = ldz;
  p.inputs = inputs; p.ldi = ldi; p.gout = gout; p.amin = const_cast<int32_t*>(need_min ? amin : nullptr);
  p.amax = const_cast<int32_t*>(need_max ? amax : nullptr); p.mean = const_cast<float*>(mean); p.var = const_cast<float*>(var);
  p.gmsg = gmsg; p.ldg = ldg; p.ldsave = ldsave;
  MMA_REQUIRE(!(need_min || need_max || need_stats) || ldsave >= D, "ldsave=%lld < T*F", (long long)ldsave);
  p.N = (int)N; p.D = D; p.T = T; p.F = F; p.avg_log = avg_log; p.avg_lin = avg_lin;
  p.drop.mode = (drop_mode == MMA_DROP_HASH && drop_thr > 0 && !inputs) ? MMA_DROP_HASH : MMA_DROP_NONE;
  p.drop.thr = drop_thr; p.drop.scale = 256.f / (256.f - (float)drop_thr);
  p.drop.seed_lo = (uint32_t)seed; p.drop.seed_hi = (uint32_t)(seed >> 32); p.drop.seed_dev = seed_dev; p.drop.keep = nullptr; p.drop.E = E; p.drop.edge_base = 0;
  const bool v4 = gr_vec4(p) && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(gmsg) & 15) == 0;
  const dim3 grid = gr_grid(N, D, v4 ? 4 : 1, &p.lpr_log);
  if (v4) hipLaunchKernelGGL((gr_bwd_kernel<4>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL((gr_bwd_kernel<1>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("gr_bwd_kernel");
}
