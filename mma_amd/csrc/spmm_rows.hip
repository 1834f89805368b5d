// K5 (CSR SpMM over a K-times column-stacked adjacency: reference layers.py:861-865, layers.py:41),
// K7 (halo row pack / unpack-add for the node-sharded multi-GPU path) and
// K8 (column sums of a tall matrix = the bias gradients of the dense transforms around the fused kernels).
#include "common.h"
#include <type_traits>

namespace mma {

struct SpmmParams {
  const int32_t* rowptr; const int32_t* col; const float* val;
  const float* B; int64_t ldb; int64_t rpb; int K;
  const float* bias; float* out; int64_t ldo; int64_t n_rows; int C; int lpr_log;
  uint32_t* rowmax;       // optional (n_rows,): max |out[row,:]| merged into a caller-zeroed array (bits of a non-negative float, atomicMax)
};

// One wave per output row; LPR lanes span the C columns, 64/LPR edges are gathered per step.
template <int VEC>
__global__ __launch_bounds__(kBlock) void spmm_kernel(const SpmmParams p) {
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log;
  const int epg = kWave >> p.lpr_log;
  const int sub = lane >> p.lpr_log;
  const int c = ((int)blockIdx.y * lpr + (lane & (lpr - 1))) * VEC;
  const bool fvalid = c < p.C;
  const int cc = fvalid ? c : 0;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave);
  for (int64_t r0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); r0 < p.n_rows; r0 += stride) {
    const int row = __builtin_amdgcn_readfirstlane((int)r0);
    const int ebeg = p.rowptr[row], eend = p.rowptr[row + 1];
    Vec<VEC> acc = vzero<VEC>();
    for (int base = ebeg; base < eend; base += kWave) {
      const int cnt = min(kWave, eend - base);
      const int myj = (lane < cnt) ? p.col[base + lane] : 0;
      const float myv = (lane < cnt) ? (p.val ? p.val[base + lane] : 1.f) : 0.f;
      for (int t0 = 0; t0 < cnt; t0 += epg) {
        const int t = t0 + sub;
        const int j = __shfl(myj, t & (kWave - 1), kWave);
        // cross-lane reads stay outside divergent control flow: ds_bpermute returns 0 from inactive lanes
        const float vv = __shfl(myv, t & (kWave - 1), kWave);
        const float v = (t < cnt) ? vv : 0.f;
        for (int k = 0; k < p.K; ++k) {
          const Vec<VEC> b = ldv<VEC>(p.B + ((size_t)k * p.rpb + (size_t)j) * p.ldb + cc);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(v, b.v[i], acc.v[i]);
        }
      }
    }
    for (int off = kWave / 2; off >= lpr; off >>= 1)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.v[i] += __shfl_xor(acc.v[i], off, kWave);
    if (sub == 0 && fvalid) {
      if (p.bias) {
        const Vec<VEC> b = ldv<VEC>(p.bias + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.v[i] += b.v[i];
      }
      stv<VEC>(p.out + (size_t)row * p.ldo + c, acc);
    }
    if (p.rowmax) {                                           // wave-uniform
      float m = 0.f;
      if (sub == 0 && fvalid) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) m = fmaxf(m, fabsf(acc.v[i]));
      }
      for (int off = kWave / 2; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
      if (lane == 0 && m > 0.f) atomicMax(p.rowmax + row, __float_as_uint(m));      // one merge per row and column chunk
    }
  }
}


// ---- [r5] segment sum of short rows, block form: out[r, :] = sum_{e in row r} B[col[e], :] (K = 1, unit weights, no bias) ----------------
// The dV segment sum of graph regression's backward (rows = source nodes, ~2 edges each, C = T*F = 380 floats) on the structure of the GR
// block kernels (gr_fused.hip): ONE workgroup per kSegRows rows; row pointers and edge ids staged in LDS; phase A = loads only - a lane
// owns a (row, 4 columns) item, issues its gathers in a fixed batch (four edge slots, clamped, + a rare tail) and leaves the sum in LDS;
// phase B = stores only.  The wave-per-row kernel above spends a dependent round trip on the row pointers, another on the edge ids and a
// third on the gathered rows for every row of two edges: 0.27 ms at C2L for 0.96 GB (3.5 TB/s).  Same bits: a row's members are added
// sequentially in edge order.  Rows above kSegLong edges are left to a column-parallel tail inside the same workgroup (same order).
constexpr int kSegRows = 16;         // rows per workgroup
constexpr int kSegCap = 512;         // edge ids of a block staged in LDS; blocks with more read them from memory
constexpr int kSegLong = 64;
struct SegSumParams {
  const int32_t* rowptr; const int32_t* col; const float* B; int64_t ldb; float* out; int64_t ldo; int n_rows; int C; uint32_t qd, qd_magic;
  uint32_t* rowmax;
};
__device__ __forceinline__ uint32_t seg_udiv(uint32_t n, uint32_t magic) { return magic ? __umulhi(n, magic) : n; }

__global__ __launch_bounds__(kBlock) void segsum_block_kernel(const SegSumParams p) {
  extern __shared__ __attribute__((aligned(16))) float seg_out[];          // (kSegRows, C)
  __shared__ int s_rp[kSegRows + 1];
  __shared__ int s_idx[kSegCap];
  const int tid = threadIdx.x;
  const int n0 = (int)blockIdx.x * kSegRows;
  const int n_here = min(kSegRows, p.n_rows - n0);
  if (n_here <= 0) return;
  const int p0 = p.rowptr[n0], p1 = p.rowptr[n0 + n_here];
  const bool staged = p1 - p0 <= kSegCap;
  if (tid <= n_here) s_rp[tid] = p.rowptr[n0 + tid];
  if (staged) for (int i = tid; i < p1 - p0; i += kBlock) s_idx[i] = p.col[p0 + i];
  __syncthreads();
  const int items = n_here * (int)p.qd;
  // STAGED as a compile-time constant: `staged ? s_idx[..] : p.col[..]` in one loop compiled into FLAT loads from a selected address (a
  // flat load counts in vmcnt AND lgkmcnt and takes the global path for what is an LDS word; the same pattern cost K4 0.05 ms - found in
  // the ISA, round 5)
  auto gather = [&](auto staged_c) {
  constexpr bool STAGED = decltype(staged_c)::value;
  auto col_of = [&](int pos) { return STAGED ? s_idx[pos - p0] : p.col[pos]; };
  for (int it = tid; it < items; it += kBlock) {
    const int dn = (int)seg_udiv((uint32_t)it, p.qd_magic);
    const int c = (it - dn * (int)p.qd) * 4;
    const int b = s_rp[dn], deg = s_rp[dn + 1] - b;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (deg <= kSegLong && deg > 0) {
      int j[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) j[i] = col_of(b + min(i, deg - 1));
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(p.B + (size_t)j[i] * p.ldb + c);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < deg) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
      for (int t = 4; t < deg; ++t) {
        const int jj = col_of(b + t);
        const float4 vv = *reinterpret_cast<const float4*>(p.B + (size_t)jj * p.ldb + c);
        acc.x += vv.x; acc.y += vv.y; acc.z += vv.z; acc.w += vv.w;
      }
    }
    *reinterpret_cast<float4*>(seg_out + (size_t)dn * p.C + c) = acc;
  }
  // long rows (rare): the whole workgroup, one lane per 4 columns, eight gathers in flight, members added in edge order
  for (int dn = 0; dn < n_here; ++dn) {
    const int b = s_rp[dn], deg = s_rp[dn + 1] - b;
    if (deg <= kSegLong) continue;                              // workgroup-uniform
    for (int q = tid; q < (int)p.qd; q += kBlock) {
      const int c = q * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t0 = 0; t0 < deg; t0 += 8) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int jj = col_of(b + min(t0 + i, deg - 1));
          v[i] = *reinterpret_cast<const float4*>(p.B + (size_t)jj * p.ldb + c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (t0 + i < deg) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
      }
      *reinterpret_cast<float4*>(seg_out + (size_t)dn * p.C + c) = acc;
    }
  }
  };
  if (staged) gather(std::true_type{}); else gather(std::false_type{});
  __syncthreads();
  for (int it = tid; it < items; it += kBlock) {
    const int dn = (int)seg_udiv((uint32_t)it, p.qd_magic);
    const int c = (it - dn * (int)p.qd) * 4;
    *reinterpret_cast<float4*>(p.out + (size_t)(n0 + dn) * p.ldo + c) = *reinterpret_cast<const float4*>(seg_out + (size_t)dn * p.C + c);
  }
  // row maxima (exact), once per workgroup from the finished rows in LDS: 32 lanes per row, a five-step butterfly (the first form merged
  // per item - ballots, a butterfly, elected-lane LDS atomics for every 4 columns; see K4)
  if (p.rowmax) {
    const int sub = tid & 31;
    for (int g = 0; g < n_here; g += kBlock / 32) {             // every lane runs every round: the butterfly needs whole wavefronts
      const int dn = g + (tid >> 5);
      float m = 0.f;
      if (dn < n_here)
        for (int q = sub; q < (int)p.qd; q += 32) {
          const float4 v = *reinterpret_cast<const float4*>(seg_out + (size_t)dn * p.C + 4 * q);
          m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0xB1, 0xF, 0xF, true)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x4E, 0xF, 0xF, true)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x141, 0xF, 0xF, true)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x140, 0xF, 0xF, true)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(m), 0x401F)));
      if (sub == 0 && dn < n_here && m > 0.f) atomicMax(p.rowmax + n0 + dn, __float_as_uint(m));
    }
  }
}

// Item-driven variant (K = 1): rows are cut into work items of bounded length, longest first (mma_amd/graph.py),
// so one 26 k-edge hub row no longer serialises the launch on a single wave.  Hub chunks write partials that the
// finalize kernel sums in slot order (deterministic, no atomics).
struct SpmmItemParams {
  const int32_t* col; const float* val; const float* B; int64_t ldb; const float* bias; float* out; int64_t ldo;
  const int4* items; int64_t n_items; float* partial; int C, lpr_log;
};

// items [0, n): one per wavefront; (bx, gx) = this workgroup's index and count among the workgroups that run this body, by = column chunk
template <int VEC>
__device__ __forceinline__ void spmm_wave_items(const SpmmItemParams& p, const int4* items, int64_t n, int bx, int gx, int by) {
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log;
  const int epg = kWave >> p.lpr_log;
  const int sub = lane >> p.lpr_log;
  const int c = (by * lpr + (lane & (lpr - 1))) * VEC;
  const bool fvalid = c < p.C;
  const int cc = fvalid ? c : 0;
  const int64_t stride = (int64_t)gx * (kBlock / kWave);
  for (int64_t it0 = (int64_t)bx * (kBlock / kWave) + (threadIdx.x >> 6); it0 < n; it0 += stride) {
    const int4 item = items[__builtin_amdgcn_readfirstlane((int)it0)];
    const int row = __builtin_amdgcn_readfirstlane(item.x);
    const int ebeg = __builtin_amdgcn_readfirstlane(item.y);
    const int eend = __builtin_amdgcn_readfirstlane(item.z);
    const int slot = __builtin_amdgcn_readfirstlane(item.w);
    Vec<VEC> acc = vzero<VEC>();
    for (int base = ebeg; base < eend; base += kWave) {
      const int cnt = min(kWave, eend - base);
      const int myj = (lane < cnt) ? p.col[base + lane] : 0;
      const float myv = (lane < cnt) ? (p.val ? p.val[base + lane] : 1.f) : 0.f;
      for (int t0 = 0; t0 < cnt; t0 += 2 * epg) {
        Vec<VEC> b[2]; float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = t0 + u * epg + sub;
          const int j = __shfl(myj, t & (kWave - 1), kWave);
          const float vv = __shfl(myv, t & (kWave - 1), kWave);
          v[u] = (t < cnt) ? vv : 0.f;
          b[u] = ldv<VEC>(p.B + (size_t)((t < cnt) ? j : 0) * p.ldb + cc);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(v[u], b[u].v[i], acc.v[i]);
      }
    }
    for (int off = kWave / 2; off >= lpr; off >>= 1)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.v[i] += __shfl_xor(acc.v[i], off, kWave);
    if (sub == 0 && fvalid) {
      if (slot < 0) {
        if (p.bias) {
          const Vec<VEC> bb = ldv<VEC>(p.bias + c);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc.v[i] += bb.v[i];
        }
        stv<VEC>(p.out + (size_t)row * p.ldo + c, acc);
      } else {
        stv<VEC>(p.partial + (size_t)slot * p.C + c, acc);
      }
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void spmm_items_kernel(const SpmmItemParams p) {
  spmm_wave_items<VEC>(p, p.items, p.n_items, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// Short items, one per group of LPR lanes (64/LPR items per wavefront), each group walking its own segment with four gathers
// in flight and no cross-lane traffic at all (a lane owns its VEC columns of the output row).  At C = 16 an item-per-wave
// launch issues ONE 640-byte gather per ~10-edge row and then waits out the memory latency; here a wave keeps 64 rows in
// flight.
template <int VEC>
__device__ __forceinline__ void spmm_group_items(const SpmmItemParams& p, const int4* items, int64_t n, int bx, int gx) {
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log;
  const int gpw = kWave >> p.lpr_log;
  const int grp = lane >> p.lpr_log;
  const int c = (lane & (lpr - 1)) * VEC;               // lpr lanes cover the whole row on this path (host: chunks == 1)
  const bool fvalid = c < p.C;
  const int cc = fvalid ? c : 0;
  const int64_t n_w = (n + gpw - 1) / gpw;
  const int64_t stride = (int64_t)gx * (kBlock / kWave);
  for (int64_t w = (int64_t)bx * (kBlock / kWave) + (threadIdx.x >> 6); w < n_w; w += stride) {
    const int64_t it = w * gpw + grp;
    if (it >= n) continue;
    const int4 item = items[it];
    Vec<VEC> acc = vzero<VEC>();
    int e = item.y;
    for (; e + 4 <= item.z; e += 4) {
      int j[4]; float v[4]; Vec<VEC> b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { j[u] = p.col[e + u]; v[u] = p.val ? p.val[e + u] : 1.f; }
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = ldv<VEC>(p.B + (size_t)j[u] * p.ldb + cc);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(v[u], b[u].v[i], acc.v[i]);
    }
    for (; e < item.z; ++e) {
      const float v = p.val ? p.val[e] : 1.f;
      const Vec<VEC> b = ldv<VEC>(p.B + (size_t)p.col[e] * p.ldb + cc);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(v, b.v[i], acc.v[i]);
    }
    if (fvalid) {
      if (item.w < 0) {
        if (p.bias) {
          const Vec<VEC> bb = ldv<VEC>(p.bias + c);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc.v[i] += bb.v[i];
        }
        stv<VEC>(p.out + (size_t)item.x * p.ldo + c, acc);
      } else {
        stv<VEC>(p.partial + (size_t)item.w * p.C + c, acc);
      }
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void spmm_items_group_kernel(const SpmmItemParams p) {
  spmm_group_items<VEC>(p, p.items, p.n_items, (int)blockIdx.x, (int)gridDim.x);
}

// [r4] both passes in ONE launch (column chunks == 1): workgroups [0, blocks_wave) take the long items, one per wavefront, the rest the
// short ones, one per lane group - the same items, walked the same way, as the two launches were (bit-identical); on the Cora / Pubmed
// structures every launch of the layer is 5-6 us of a 0.13-0.18 ms replay.
template <int VEC>
__global__ __launch_bounds__(kBlock) void spmm_items_both_kernel(const SpmmItemParams p, int64_t n_wave, int blocks_wave) {
  if ((int)blockIdx.x < blocks_wave) spmm_wave_items<VEC>(p, p.items, n_wave, (int)blockIdx.x, blocks_wave, 0);
  else spmm_group_items<VEC>(p, p.items + n_wave, p.n_items - n_wave, (int)blockIdx.x - blocks_wave, (int)gridDim.x - blocks_wave);
}

__global__ __launch_bounds__(kBlock) void spmm_items_finalize_kernel(const SpmmItemParams p, const int4* hubs, int64_t n_hubs) {
  const int64_t total = n_hubs * p.C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % p.C);
    const int4 hub = hubs[idx / p.C];
    float s = 0.f;
    for (int sl = hub.y; sl < hub.z; ++sl) s += p.partial[(size_t)sl * p.C + c];
    if (p.bias) s += p.bias[c];
    p.out[(size_t)hub.x * p.ldo + c] = s;
  }
}

struct RowsParams { const float* src; int64_t lds; const int32_t* idx; int64_t n; float* dst; int64_t ldd; int width; };

template <int VEC, bool UNPACK_ADD>
__global__ __launch_bounds__(kBlock) void rows_kernel(const RowsParams p) {
  const int per_row = (p.width + VEC - 1) / VEC;
  const int64_t total = p.n * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row;
    const int c = (int)(i % per_row) * VEC;
    const int64_t node = p.idx[r];
    if (UNPACK_ADD) {
      const Vec<VEC> a = ldv<VEC>(p.src + r * p.lds + c);
      Vec<VEC> d = ldv<VEC>(p.dst + node * p.ldd + c);
#pragma unroll
      for (int q = 0; q < VEC; ++q) d.v[q] += a.v[q];
      stv<VEC>(p.dst + node * p.ldd + c, d);
    } else {
      stv<VEC>(p.dst + r * p.ldd + c, ldv<VEC>(p.src + node * p.lds + c));
    }
  }
}

// unpack-add of a whole reverse halo exchange in ONE launch: destination row rows[t] receives the sum of the received rows
// pos[segptr[t] .. segptr[t+1]) in that (fixed) order - a row several peers read gets all its contributions from one
// thread, so there is no write conflict between peers and no atomics; bitwise repeatable.
struct RowsCsrParams { const float* src; int64_t lds; const int32_t* rows; const int32_t* segptr; const int32_t* pos; int64_t n;
                       float* dst; int64_t ldd; int width; };

template <int VEC>
__global__ __launch_bounds__(kBlock) void rows_csr_add_kernel(const RowsCsrParams p) {
  const int per_row = (p.width + VEC - 1) / VEC;
  const int64_t total = p.n * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / per_row;
    const int c = (int)(i % per_row) * VEC;
    const int64_t node = p.rows[t];
    Vec<VEC> d = ldv<VEC>(p.dst + node * p.ldd + c);
    for (int q = p.segptr[t]; q < p.segptr[t + 1]; ++q) {
      const Vec<VEC> a = ldv<VEC>(p.src + (int64_t)p.pos[q] * p.lds + c);
#pragma unroll
      for (int x = 0; x < VEC; ++x) d.v[x] += a.v[x];
    }
    stv<VEC>(p.dst + node * p.ldd + c, d);
  }
}

// K8: out[rb, c] = sum of g[r, c] over the rows of row block rb.  A workgroup is 64 columns x 4 row lanes (one wave reads 256
// contiguous bytes of a row), each thread keeps four independent accumulation chains; fixed order => deterministic.
struct ColSumParams { const float* g; int64_t ldg; int64_t R; int C; float* out; int64_t rows_per_block; };

__global__ __launch_bounds__(kBlock) void col_sum_kernel(const ColSumParams p) {
  __shared__ float red[4][kWave];
  const int cl = threadIdx.x & (kWave - 1), rl = threadIdx.x >> 6;
  const int c = (int)blockIdx.x * kWave + cl;
  const int64_t r0 = (int64_t)blockIdx.y * p.rows_per_block;
  const int64_t r1 = min(p.R, r0 + p.rows_per_block);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < p.C) {
    const float* q = p.g + c;
    int64_t r = r0 + rl;
    for (; r + 12 < r1; r += 16) {
      a0 += q[r * p.ldg];
      a1 += q[(r + 4) * p.ldg];
      a2 += q[(r + 8) * p.ldg];
      a3 += q[(r + 12) * p.ldg];
    }
    for (; r < r1; r += 4) a0 += q[r * p.ldg];
  }
  red[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && c < p.C) p.out[(int64_t)blockIdx.y * p.C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// Short matrices (the bias gradients of a ~1 500-row molecule batch): ONE launch - a workgroup owns 4 columns with 64 row lanes, each
// lane fetching 8 rows before it adds, then a tree over the lanes - instead of two passes of the 4-lane kernel above (the launch, not
// the bytes, is the cost there).  Fixed order.
constexpr int kColSumShortCols = 4, kColSumShortRows = kBlock / kColSumShortCols, kColSumShortMaxR = 8192;
__global__ __launch_bounds__(kBlock) void col_sum_short_kernel(const ColSumParams p) {
  __shared__ float red[kColSumShortRows][kColSumShortCols];
  const int cl = threadIdx.x % kColSumShortCols, rl = threadIdx.x / kColSumShortCols;
  const int c = (int)blockIdx.x * kColSumShortCols + cl;
  const bool cv = c < p.C;
  const float* q = p.g + (cv ? c : 0);
  float a = 0.f;
  for (int64_t r = rl; r < p.R; r += kColSumShortRows * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = q[min(r + u * kColSumShortRows, p.R - 1) * p.ldg];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += (r + u * kColSumShortRows < p.R) ? v[u] : 0.f;
  }
  red[rl][cl] = a;
  __syncthreads();
  for (int o = kColSumShortRows / 2; o > 0; o >>= 1) {
    if (rl < o) red[rl][cl] += red[rl + o][cl];
    __syncthreads();
  }
  if (rl == 0 && cv) p.out[c] = red[0][cl];
}

static int64_t col_sum_rows_per_block(int64_t R) {      // ~2*sqrt(R), a power of two in [64, 4096]: both passes stay short
  int64_t rpb = 64;
  while (rpb < 4096 && rpb * rpb < 4 * R) rpb *= 2;
  return rpb;
}
static bool col_sum_short(int64_t R, int C) { return R > 256 && R <= kColSumShortMaxR && C <= 2048; }   // few columns: 16-byte row segments are fine
static int64_t col_sum_row_blocks(int64_t R, int C) {
  if (R <= 1024 || col_sum_short(R, C)) return 1;
  const int64_t rpb = col_sum_rows_per_block(R);
  return (R + rpb - 1) / rpb;
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mma

using namespace mma;

static int csr_spmm_impl(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, int64_t ldb,
                         int64_t rows_per_block, int32_t K, const float* bias, float* out, int64_t ldo,
                         int64_t n_rows, int32_t C, float* row_max, void* stream) {
  MMA_REQUIRE(n_rows >= 0 && n_rows < (1LL << 31) && C >= 1 && K >= 1, "n_rows=%lld C=%d K=%d unsupported", (long long)n_rows, C, K);
  MMA_REQUIRE(ldb >= C && ldo >= C && rows_per_block >= 0, "row pitch too small");
  if (n_rows == 0) return 0;
  MMA_REQUIRE(rowptr && B && out, "NULL argument");
  const bool v4 = (C % 4 == 0) && (ldb % 4 == 0) && (ldo % 4 == 0) && al16(B) && al16(out) && (!bias || al16(bias));
  const int vec = v4 ? 4 : 1;
  const int per_row = (C + vec - 1) / vec;
  hipStream_t st0 = static_cast<hipStream_t>(stream);
  {   // [r5] plain segment sums of vector rows: the block form (see segsum_block_kernel); MMA_SEGSUM_BLOCK=0: the wave-per-row kernel (A/B)
    const char* e = getenv("MMA_SEGSUM_BLOCK");
    const int qd = C / 4;
    if (K == 1 && !val && !bias && v4 && col && C >= 32 && C <= 2048 && (int64_t)kSegRows * C * 4 <= 64 * 1024 && qd < 65536 && !(e && e[0] == '0')) {
      SegSumParams sp{rowptr, col, B, ldb, out, ldo, (int)n_rows, C, (uint32_t)qd, qd <= 1 ? 0u : (uint32_t)((1ULL << 32) / (uint32_t)qd) + 1u,
                      reinterpret_cast<uint32_t*>(row_max)};
      const int64_t blocks = (n_rows + kSegRows - 1) / kSegRows;
      hipLaunchKernelGGL(segsum_block_kernel, dim3((unsigned)blocks), dim3(kBlock), (unsigned)(kSegRows * C * 4), st0, sp);
      return check_launch("segsum_block_kernel");
    }
  }
  SpmmParams p{rowptr, col, val, B, ldb, rows_per_block, K, bias, out, ldo, n_rows, C, 0, reinterpret_cast<uint32_t*>(row_max)};
  p.lpr_log = min(ilog2_ceil(per_row), 6);
  const int chunks = (per_row + (1 << p.lpr_log) - 1) >> p.lpr_log;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  const dim3 grid((unsigned)blocks, (unsigned)chunks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (v4) hipLaunchKernelGGL((spmm_kernel<4>), grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL((spmm_kernel<1>), grid, dim3(kBlock), 0, st, p);
  return check_launch("spmm_kernel");
}

extern "C" int mma_csr_spmm(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, int64_t ldb,
                            int64_t rows_per_block, int32_t K, const float* bias, float* out, int64_t ldo,
                            int64_t n_rows, int32_t C, void* stream) {
  return csr_spmm_impl(rowptr, col, val, B, ldb, rows_per_block, K, bias, out, ldo, n_rows, C, nullptr, stream);
}
extern "C" int mma_csr_spmm_rm(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, int64_t ldb,
                               int64_t rows_per_block, int32_t K, const float* bias, float* out, int64_t ldo,
                               int64_t n_rows, int32_t C, float* row_max, void* stream) {
  return csr_spmm_impl(rowptr, col, val, B, ldb, rows_per_block, K, bias, out, ldo, n_rows, C, row_max, stream);
}

extern "C" int mma_csr_spmm_items(const int32_t* col, const float* val, const float* B, int64_t ldb, const float* bias, float* out,
                                  int64_t ldo, const int32_t* items, int64_t n_items, int64_t n_wave_items, const int32_t* hubs,
                                  int64_t n_hubs, float* partial, int64_t n_slots, int32_t C, void* stream) {
  MMA_REQUIRE(C >= 1 && ldb >= C && ldo >= C && n_items >= 0 && n_items < (1LL << 31) && n_hubs >= 0 && n_slots >= 0 && n_wave_items >= 0,
              "C=%d ldb=%lld ldo=%lld n_items=%lld unsupported", C, (long long)ldb, (long long)ldo, (long long)n_items);
  MMA_REQUIRE(n_slots == 0 || (partial && hubs && n_hubs > 0), "hub slots without partial/hubs buffers");
  if (n_items == 0) return 0;
  MMA_REQUIRE(B && out && items && al16(items) && (!hubs || al16(hubs)), "NULL or misaligned argument");
  const bool v4 = (C % 4 == 0) && (ldb % 4 == 0) && (ldo % 4 == 0) && al16(B) && al16(out) && (!bias || al16(bias)) &&
                  (!partial || al16(partial));
  const int vec = v4 ? 4 : 1;
  const int per_row = (C + vec - 1) / vec;
  SpmmItemParams p{col, val, B, ldb, bias, out, ldo, reinterpret_cast<const int4*>(items), n_items, partial, C, 0};
  p.lpr_log = min(ilog2_ceil(per_row), 6);
  const int chunks = (per_row + (1 << p.lpr_log) - 1) >> p.lpr_log;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // items [0, n_wave_items): one per wavefront; the rest (short rows) one per lpr-lane group when a wave holds several groups
  const int gpw = kWave >> p.lpr_log;
  if (gpw == 1 || chunks != 1 || n_wave_items > n_items) n_wave_items = n_items;
  const char* one_env = getenv("MMA_SPMM_ONE_LAUNCH");                   // 0: the two launches of round 3 (A/B and the equality test)
  if (n_wave_items > 0 && n_items > n_wave_items && chunks == 1 && !(one_env && one_env[0] == '0')) {       // both kinds of item: one launch
    int64_t ba = (n_wave_items + 3) / 4, bb = (n_items - n_wave_items + 4 * gpw - 1) / (4 * gpw);
    if (ba > kMaxGrid) ba = kMaxGrid;
    if (bb > 4 * kMaxGrid) bb = 4 * kMaxGrid;
    if (v4) hipLaunchKernelGGL((spmm_items_both_kernel<4>), dim3((unsigned)(ba + bb)), dim3(kBlock), 0, st, p, n_wave_items, (int)ba);
    else hipLaunchKernelGGL((spmm_items_both_kernel<1>), dim3((unsigned)(ba + bb)), dim3(kBlock), 0, st, p, n_wave_items, (int)ba);
    if (int rc = check_launch("spmm_items_both_kernel")) return rc;
    n_wave_items = n_items = 0;                                         // nothing left for the two single-kind launches below
  }
  if (n_wave_items > 0) {
    p.n_items = n_wave_items;
    int64_t blocks = (n_wave_items + 3) / 4;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    const dim3 grid((unsigned)blocks, (unsigned)chunks);
    if (v4) hipLaunchKernelGGL((spmm_items_kernel<4>), grid, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((spmm_items_kernel<1>), grid, dim3(kBlock), 0, st, p);
    if (int rc = check_launch("spmm_items_kernel")) return rc;
  }
  if (n_items > n_wave_items) {
    p.items = reinterpret_cast<const int4*>(items) + n_wave_items;
    p.n_items = n_items - n_wave_items;
    int64_t blocks = (p.n_items + 4 * gpw - 1) / (4 * gpw);
    if (blocks > 4 * kMaxGrid) blocks = 4 * kMaxGrid;
    if (v4) hipLaunchKernelGGL((spmm_items_group_kernel<4>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((spmm_items_group_kernel<1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p);
    if (int rc = check_launch("spmm_items_group_kernel")) return rc;
  }
  if (n_hubs > 0) {
    int64_t fb = (n_hubs * C + kBlock - 1) / kBlock;
    if (fb > kMaxGrid) fb = kMaxGrid;
    hipLaunchKernelGGL(spmm_items_finalize_kernel, dim3((unsigned)fb), dim3(kBlock), 0, st, p, reinterpret_cast<const int4*>(hubs), n_hubs);
    return check_launch("spmm_items_finalize_kernel");
  }
  return 0;
}

static int rows_call(bool unpack, const float* src, int64_t lds, const int32_t* idx, int64_t n_idx, float* dst, int64_t ldd,
                     int32_t width, void* stream) {
  MMA_REQUIRE(n_idx >= 0 && width >= 1 && lds >= width && ldd >= width, "n_idx=%lld width=%d lds=%lld ldd=%lld unsupported",
              (long long)n_idx, width, (long long)lds, (long long)ldd);
  if (n_idx == 0) return 0;
  MMA_REQUIRE(src && idx && dst, "NULL argument");
  const bool v4 = (width % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && al16(src) && al16(dst);
  RowsParams p{src, lds, idx, n_idx, dst, ldd, width};
  const int per_row = v4 ? width / 4 : width;
  int64_t blocks = (n_idx * per_row + kBlock - 1) / kBlock;
  if (blocks > kMaxGrid * 4) blocks = kMaxGrid * 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks);
  if (v4) {
    if (unpack) hipLaunchKernelGGL((rows_kernel<4, true>), grid, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((rows_kernel<4, false>), grid, dim3(kBlock), 0, st, p);
  } else {
    if (unpack) hipLaunchKernelGGL((rows_kernel<1, true>), grid, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((rows_kernel<1, false>), grid, dim3(kBlock), 0, st, p);
  }
  return check_launch("rows_kernel");
}

extern "C" int mma_unpack_add_rows_csr(const float* src, int64_t lds, const int32_t* rows, const int32_t* segptr, const int32_t* pos,
                                       int64_t n_rows, float* dst, int64_t ldd, int32_t width, void* stream) {
  MMA_REQUIRE(n_rows >= 0 && width >= 1 && lds >= width && ldd >= width, "n_rows=%lld width=%d lds=%lld ldd=%lld unsupported",
              (long long)n_rows, width, (long long)lds, (long long)ldd);
  if (n_rows == 0) return 0;
  MMA_REQUIRE(src && rows && segptr && pos && dst, "NULL argument");
  const bool v4 = (width % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && al16(src) && al16(dst);
  RowsCsrParams p{src, lds, rows, segptr, pos, n_rows, dst, ldd, width};
  const int per_row = v4 ? width / 4 : width;
  int64_t blocks = (n_rows * per_row + kBlock - 1) / kBlock;
  if (blocks > kMaxGrid * 4) blocks = kMaxGrid * 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (v4) hipLaunchKernelGGL((rows_csr_add_kernel<4>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL((rows_csr_add_kernel<1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p);
  return check_launch("rows_csr_add_kernel");
}

extern "C" int mma_pack_rows(const float* src, int64_t lds, const int32_t* idx, int64_t n_idx, float* dst, int64_t ldd,
                             int32_t width, void* stream) {
  return rows_call(false, src, lds, idx, n_idx, dst, ldd, width, stream);
}
extern "C" int mma_unpack_add_rows(const float* src, int64_t lds, const int32_t* idx, int64_t n_idx, float* dst, int64_t ldd,
                                   int32_t width, void* stream) {
  return rows_call(true, src, lds, idx, n_idx, dst, ldd, width, stream);
}

extern "C" int64_t mma_col_sum_workspace_floats(int64_t R, int32_t C) {
  const int64_t nrb = col_sum_row_blocks(R, C);
  return nrb > 1 ? nrb * (int64_t)C : 0;
}

extern "C" int mma_col_sum(const float* g, int64_t ldg, int64_t R, int32_t C, float* out, float* ws, int64_t ws_floats,
                           void* stream) {
  MMA_REQUIRE(R >= 0 && C >= 1 && ldg >= C, "R=%lld C=%d ldg=%lld unsupported", (long long)R, C, (long long)ldg);
  MMA_REQUIRE(out && (g || R == 0), "NULL argument");
  const int64_t nrb = col_sum_row_blocks(R, C);
  MMA_REQUIRE(nrb == 1 || (ws && ws_floats >= nrb * (int64_t)C), "workspace too small: %lld floats, need %lld",
              (long long)ws_floats, (long long)(nrb * (int64_t)C));
  MMA_REQUIRE(nrb < 65536, "R=%lld too large", (long long)R);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned cb = (unsigned)((C + kWave - 1) / kWave);
  if (nrb == 1) {
    ColSumParams p{g, ldg, R, C, out, R > 0 ? R : 1};
    if (col_sum_short(R, C)) {
      hipLaunchKernelGGL(col_sum_short_kernel, dim3((unsigned)((C + kColSumShortCols - 1) / kColSumShortCols)), dim3(kBlock), 0, st, p);
      return check_launch("col_sum_short_kernel");
    }
    hipLaunchKernelGGL(col_sum_kernel, dim3(cb, 1), dim3(kBlock), 0, st, p);
    return check_launch("col_sum_kernel");
  }
  ColSumParams p1{g, ldg, R, C, ws, col_sum_rows_per_block(R)};
  hipLaunchKernelGGL(col_sum_kernel, dim3(cb, (unsigned)nrb), dim3(kBlock), 0, st, p1);
  if (int rc = check_launch("col_sum_kernel")) return rc;
  ColSumParams p2{ws, C, nrb, C, out, nrb};
  hipLaunchKernelGGL(col_sum_kernel, dim3(cb, 1), dim3(kBlock), 0, st, p2);
  return check_launch("col_sum_kernel");
}
