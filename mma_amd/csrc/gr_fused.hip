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
  int32_t* amin; int32_t* amax; float* mean; float* var;                         // (N,D) saved for backward (may be NULL)
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

__device__ __forceinline__ float gr_message(const GrParams& p, const DropParams& dp, bool fused, float u, int j, uint32_t e, int cc) {
  if (!fused) return p.inputs[(size_t)e * p.ldi + cc];
  float h = u + p.V[(size_t)j * p.lduv + cc];
  if (p.Z) h += p.Z[(size_t)e * p.ldz + cc];
  if (dp.mode != MMA_DROP_NONE) {
    float f[1];
    drop_factors<1>(dp, e, 0, cc, p.D, 0, f);
    h *= f[0];
  }
  return h;
}

__global__ __launch_bounds__(kBlock) void gr_fwd_kernel(const GrParams p) {
  const DropParams dp = drop_resolve(p.drop);
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log, epg = kWave >> p.lpr_log;
  const int sub = lane >> p.lpr_log;
  const int c = (int)blockIdx.y * lpr + (lane & (lpr - 1));
  const bool valid = c < p.D;
  const int cc = valid ? c : 0;
  const bool fused = p.inputs == nullptr;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave);
  for (int64_t n0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); n0 < p.N; n0 += stride) {
    const int node = __builtin_amdgcn_readfirstlane((int)n0);
    const int ebeg = p.rowptr[node], eend = p.rowptr[node + 1];
    const float u = fused ? p.U[(size_t)node * p.lduv + cc] : 0.f;
    float sum = 0.f, sq = 0.f, mn = INFINITY, mx = -INFINITY;
    int an = INT_MAX, ax = INT_MAX;
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
        const float h = gr_message(p, dp, fused, u, j, (uint32_t)e, cc);
        if (ev) {
          sum += h; sq += h * h;
          if (h < mn) { mn = h; an = e; }   // strict: the first (lowest position) extremal edge wins
          if (h > mx) { mx = h; ax = e; }
        }
      }
    }
    for (int off = kWave / 2; off >= lpr; off >>= 1) {
      sum += __shfl_xor(sum, off, kWave);
      sq += __shfl_xor(sq, off, kWave);
      const float omn = __shfl_xor(mn, off, kWave); const int oan = __shfl_xor(an, off, kWave);
      const float omx = __shfl_xor(mx, off, kWave); const int oax = __shfl_xor(ax, off, kWave);
      if (omn < mn || (omn == mn && oan < an)) { mn = omn; an = oan; }
      if (omx > mx || (omx == mx && oax < ax)) { mx = omx; ax = oax; }
    }
    if (sub == 0 && valid) {
      const int cnt = eend - ebeg;
      const float deg = (float)max(cnt, 1);                       // degree(...).clamp_(1), mma_conv.py:178-179
      const float mean = sum / deg;                               // scatter mean: sum / clamp(count, 1)
      const float var = sq / deg - mean * mean;                   // mma_conv.py:167-170
      const int t = c / p.F, f = c - t * p.F;
      float* o = p.out + ((size_t)node * p.T + t) * ((size_t)p.S * p.K * p.F) + f;
      for (int k = 0; k < p.K; ++k) {
        float b;
        switch (p.aggr[k]) {
          case GR_SUM: b = sum; break;
          case GR_MEAN: b = mean; break;
          case GR_MIN: b = cnt ? mn : 0.f; break;                 // empty target -> 0 (torch_scatter)
          case GR_MAX: b = cnt ? mx : 0.f; break;
          case GR_VAR: b = var; break;
          default: b = sqrtf(fmaxf(var, 0.f) + 1e-5f); break;
        }
        float run = b;
        for (int s = 0; s < p.S; ++s) {                           // compounding (G7)
          run = run * scaler_factor(p.scaler[s], deg, p.avg_log, p.avg_lin);
          o[(size_t)(s * p.K + k) * p.F] = run;
        }
      }
      const size_t so = (size_t)node * p.D + c;
      if (p.amin) p.amin[so] = cnt ? an : -1;
      if (p.amax) p.amax[so] = cnt ? ax : -1;
      if (p.mean) p.mean[so] = mean;
      if (p.var) p.var[so] = var;
    }
  }
}

// K4: gradient w.r.t. every edge message, written by original edge id (each edge has exactly one target: no conflicts)
__global__ __launch_bounds__(kBlock) void gr_bwd_kernel(const GrParams p) {
  const DropParams dp = drop_resolve(p.drop);
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log, epg = kWave >> p.lpr_log;
  const int sub = lane >> p.lpr_log;
  const int c = (int)blockIdx.y * lpr + (lane & (lpr - 1));
  const bool valid = c < p.D;
  const int cc = valid ? c : 0;
  const bool fused = p.inputs == nullptr;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave);
  for (int64_t n0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); n0 < p.N; n0 += stride) {
    const int node = __builtin_amdgcn_readfirstlane((int)n0);
    const int ebeg = p.rowptr[node], eend = p.rowptr[node + 1];
    if (ebeg == eend) continue;
    const float deg = (float)(eend - ebeg);
    const int t = cc / p.F, f = cc - t * p.F;
    const float* go = p.gout + ((size_t)node * p.T + t) * ((size_t)p.S * p.K * p.F) + f;
    const size_t so = (size_t)node * p.D + cc;
    float c_all = 0.f, c_min = 0.f, c_max = 0.f, c_var = 0.f;   // coefficients of: 1, [e==amin], [e==amax], (h - mean)
    bool need_h = false;
    for (int k = 0; k < p.K; ++k) {
      float gb = 0.f, run = 1.f;
      for (int s = 0; s < p.S; ++s) {
        run = run * scaler_factor(p.scaler[s], deg, p.avg_log, p.avg_lin);
        gb = fmaf(go[(size_t)(s * p.K + k) * p.F], run, gb);
      }
      switch (p.aggr[k]) {
        case GR_SUM: c_all += gb; break;
        case GR_MEAN: c_all += gb / deg; break;
        case GR_MIN: c_min += gb; break;
        case GR_MAX: c_max += gb; break;
        case GR_VAR: c_var += gb * 2.f / deg; need_h = true; break;
        default: {
          const float v = p.var[so];
          if (v > 0.f) c_var += gb / (2.f * sqrtf(v + 1e-5f)) * 2.f / deg;   // relu'(v) = [v > 0]
          need_h = true;
        } break;
      }
    }
    const int an = p.amin ? p.amin[so] : -1, ax = p.amax ? p.amax[so] : -1;
    const float mean = need_h ? p.mean[so] : 0.f;
    const float u = (fused && need_h) ? p.U[(size_t)node * p.lduv + cc] : 0.f;
    for (int base = ebeg; base < eend; base += kWave) {
      const int cnt = min(kWave, eend - base);
      const int myj = (lane < cnt) ? p.src[base + lane] : 0;
      const int mye = (lane < cnt) ? p.perm[base + lane] : 0;
      for (int t0 = 0; t0 < cnt; t0 += epg) {
        const int tt = t0 + sub;
        const int j_ = __shfl(myj, tt & (kWave - 1), kWave);
        const int e_ = __shfl(mye, tt & (kWave - 1), kWave);
        if (tt < cnt && valid) {
          float g = c_all + (e_ == an ? c_min : 0.f) + (e_ == ax ? c_max : 0.f);
          if (need_h) g += c_var * (gr_message(p, dp, fused, u, j_, (uint32_t)e_, cc) - mean);
          if (fused && dp.mode != MMA_DROP_NONE) {
            float fd[1];
            drop_factors<1>(dp, (uint32_t)e_, 0, cc, p.D, 0, fd);
            g *= fd[0];
          }
          p.gmsg[(size_t)e_ * p.ldg + cc] = g;
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

static dim3 gr_grid(int64_t N, int D, int* lpr_log) {
  *lpr_log = min(ilog2_ceil(D), 6);
  const int chunks = (D + (1 << *lpr_log) - 1) >> *lpr_log;
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
    float* out, int32_t* amin, int32_t* amax, float* mean, float* var,
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
  p.inputs = inputs; p.ldi = ldi; p.out = out; p.amin = amin; p.amax = amax; p.mean = mean; p.var = var;
  p.N = (int)N; p.D = D; p.T = T; p.F = F; p.avg_log = avg_log; p.avg_lin = avg_lin;
  p.drop.mode = (drop_mode == MMA_DROP_HASH && drop_thr > 0 && !inputs) ? MMA_DROP_HASH : MMA_DROP_NONE;
  p.drop.thr = drop_thr; p.drop.scale = 256.f / (256.f - (float)drop_thr);
  p.drop.seed_lo = (uint32_t)seed; p.drop.seed_hi = (uint32_t)(seed >> 32); p.drop.seed_dev = seed_dev; p.drop.keep = nullptr; p.drop.E = E; p.drop.edge_base = 0;
  const dim3 grid = gr_grid(N, D, &p.lpr_log);
  hipLaunchKernelGGL(gr_fwd_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("gr_fwd_kernel");
}

extern "C" int mma_gr_fused_bwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, const float* inputs, int64_t ldi,
    const float* gout, const int32_t* amin, const int32_t* amax, const float* mean, const float* var,
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
  p.gmsg = gmsg; p.ldg = ldg;
  p.N = (int)N; p.D = D; p.T = T; p.F = F; p.avg_log = avg_log; p.avg_lin = avg_lin;
  p.drop.mode = (drop_mode == MMA_DROP_HASH && drop_thr > 0 && !inputs) ? MMA_DROP_HASH : MMA_DROP_NONE;
  p.drop.thr = drop_thr; p.drop.scale = 256.f / (256.f - (float)drop_thr);
  p.drop.seed_lo = (uint32_t)seed; p.drop.seed_hi = (uint32_t)(seed >> 32); p.drop.seed_dev = seed_dev; p.drop.keep = nullptr; p.drop.E = E; p.drop.edge_base = 0;
  const dim3 grid = gr_grid(N, D, &p.lpr_log);
  hipLaunchKernelGGL(gr_bwd_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("gr_bwd_kernel");
}
