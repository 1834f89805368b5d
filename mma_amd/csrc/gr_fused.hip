// K3 / K4 / K6: graph-regression form of MMA (reference graph_regression/mma_conv.py:121-196).
//
// K3 fuses, for one target node per wavefront, what the reference does with ~K+12 launches:
//   message  h_e = drop( U[i] + V[j] + Z[e] )            (= per-tower Linear([x_i || x_j || enc(e_ij)]) + dropout,
//                                                          mma_conv.py:138-157, split into dense GEMMs)
//   K x torch_scatter.scatter(h, index, reduce)           sum / mean / min(+arg) / max(+arg) / var / std  (:164-172)
//   degree + compounding scalers + final layout           (:176-196)  out[n, t, s*K*F + k*F + f]
// in ONE pass over the node's target-sorted edge segment.  Edges are sorted by target with a STABLE radix sort
// (K6, rocPRIM), so inside a segment the original edge positions ascend; min/max ties therefore resolve to the
// lowest edge position exactly like torch_scatter's sequential CPU kernel, and the sub-row butterfly keeps that
// order by comparing (value, edge id) lexicographically.
#include "common.h"
#include <type_traits>
#include <hipcub/hipcub.hpp>
#include <limits.h>

namespace mma {

enum { GR_SUM = 0, GR_MEAN = 1, GR_MIN = 2, GR_MAX = 3, GR_VAR = 4, GR_STD = 5 };
enum { SC_IDENTITY = 0, SC_AMPLIFICATION = 1, SC_ATTENUATION = 2, SC_LINEAR = 3, SC_INVERSE_LINEAR = 4 };

struct GrParams {
  const int32_t* rowptr; const int32_t* src; const int32_t* perm;
  const float* U; const float* V; int64_t lduv; const float* Z; int64_t ldz;   // fused-message mode
  const float* inputs; int64_t ldi;                                              // given-messages mode (E,D), rows by ORIGINAL edge id
  float* out; const float* gout;                                                 // (N, T, S*K*F)
  // saved for backward (may be NULL): argmin / argmax as the edge's OFFSET inside its target segment - one byte per column
  // for segments of < 256 edges (arg8), an int32 row per longer segment in a side table (row = ceil(rowptr[n] / 256): a
  // segment of >= 256 positions contains exactly one first multiple of 256, so the rows are unique and (E >> 8) + 2 suffice)
  uint8_t* amin8; uint8_t* amax8; int32_t* amin_side; int32_t* amax_side;
  float* mean; float* var; int64_t ldsave;                                      // (N,ldsave>=D)
  float* gmsg; int64_t ldg;                                                      // backward: (E,D) message gradients
  float* gU; int64_t ldgu;                                                       // backward, optional: (N,D) sum of gmsg over each target's segment
  // backward, optional (round 5): max |gmsg[r,:]| per message-gradient row and max |gU[n,:]| per node, as the bit patterns of non-negative
  // floats merged by atomicMax into ZEROED arrays - the row scales of the three-product weight-gradient / dL/dx GEMMs behind K4
  uint32_t* gmsg_rmax; uint32_t* gu_rmax;
  int N, D, T, F, K, S, lpr_log;
  int by_pos;                                        // 1: Z and gmsg rows are indexed by the target-sorted POSITION p, 0: by perm[p]
  int wave_min_deg;                                  // wave-per-node pass: skip segments shorter than this
  int nb; uint32_t qd, qd_magic, f_magic;            // flat kernels: nodes per workgroup, lanes (VEC columns each) per row, 2^32/d magics
  // block kernels: float4s per node / per tower / per aggregate block of `out`, with their magics; LDS byte offsets
  uint32_t tskf4, m_tskf4, skf4, m_skf4, f4, m_f4, m_k;
  uint32_t lds_agg, lds_arg, lds_bytes; int n_coef;
  const uint8_t* zidx;          // categorical edge features: Z is a small table and its row for an edge is zidx[position or edge id]
  const int32_t* long_nodes;                         // optional [count, node ids...] of the segments above kGroupMaxDeg (mma_build_csr)
  int long_cap;                                      // ids the list can hold (mma_gr_long_nodes_len(E) - 1): a count beyond it is not believed
  bool need_sum, need_sq, need_min, need_max, need_mean;   // which running reductions the aggregator list uses
  uint8_t aggr[MMA_MAX_K]; uint8_t scaler[8];
  uint32_t aggr_pack, scaler_pack;                   // the same codes, 4 bits each: a run-time index into a kernarg ARRAY is a vector-memory load
  float avg_log, avg_lin;
  DropParams drop;
};

// floor(n / d) for n, d < 2^16 through magic = floor(2^32 / d) + 1 (host: div_magic); d == 1 has no 32-bit magic: 0 stands for it
__device__ __forceinline__ uint32_t udiv(uint32_t n, uint32_t magic) { return magic ? __umulhi(n, magic) : n; }

__device__ __forceinline__ float scaler_factor(int code, float deg, float avg_log, float avg_lin) {
  switch (code) {
    case SC_AMPLIFICATION: return logf(deg + 1.f) / avg_log;
    case SC_ATTENUATION: return avg_log / logf(deg + 1.f);
    case SC_LINEAR: return deg / avg_lin;
    case SC_INVERSE_LINEAR: return avg_lin / deg;
    default: return 1.f;
  }
}

// row of Z that belongs to the edge at target-sorted position pos / original id e: the edge's own row, or - categorical edge
// features (ZINC bond types: mma.py:88,103 embed 4 types) - the row of its TYPE in a table that stays in L1/L2
__device__ __forceinline__ size_t gr_zrow(const GrParams& p, int pos, int e) {
  const int r = p.by_pos ? pos : e;
  return p.zidx ? (size_t)p.zidx[r] : (size_t)r;
}

// the VEC messages h_e[c..c+VEC) of the edge at target-sorted position pos with original id e and source j
// (fused: drop(U[i] + V[j] + Z[e or pos]); given: inputs[e])
template <int VEC>
__device__ __forceinline__ Vec<VEC> gr_message(const GrParams& p, const DropParams& dp, bool fused, const Vec<VEC>& u, int j,
                                               uint32_t e, uint32_t pos, int cc) {
  if (!fused) return ldv<VEC>(p.inputs + (size_t)e * p.ldi + cc);
  Vec<VEC> h = ldv<VEC>(p.V + (size_t)j * p.lduv + cc);
#pragma unroll
  for (int i = 0; i < VEC; ++i) h.v[i] += u.v[i];
  if (p.Z) {
    const Vec<VEC> z = ldv<VEC>(p.Z + gr_zrow(p, pos, e) * p.ldz + cc);
#pragma unroll
    for (int i = 0; i < VEC; ++i) h.v[i] += z.v[i];
  }
  if (dp.mode != MMA_DROP_NONE) {
    float f[VEC];
    drop_factors<VEC>(dp, e, 0, cc, p.D, 0, f);      // the keep bits are keyed by the ORIGINAL edge id (oracle/dropout_rng.py)
#pragma unroll
    for (int i = 0; i < VEC; ++i) h.v[i] *= f[i];
  }
  return h;
}

template <int VEC> __device__ __forceinline__ void ldi(const int32_t* q, int (&a)[VEC]);
template <> __device__ __forceinline__ void ldi<4>(const int32_t* q, int (&a)[4]) {
  const int4 t = *reinterpret_cast<const int4*>(q);
  a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
}
template <> __device__ __forceinline__ void ldi<1>(const int32_t* q, int (&a)[1]) { a[0] = *q; }
template <int VEC> __device__ __forceinline__ void sti(int32_t* q, const int (&a)[VEC]);
template <> __device__ __forceinline__ void sti<4>(int32_t* q, const int (&a)[4]) {
  *reinterpret_cast<int4*>(q) = make_int4(a[0], a[1], a[2], a[3]);
}
template <> __device__ __forceinline__ void sti<1>(int32_t* q, const int (&a)[1]) { *q = a[0]; }

// saved arg = offset of the extremal edge inside the node's segment [segb, segb + cnt)
__device__ __forceinline__ size_t arg_side_row(int segb) { return (size_t)((segb + 255) >> 8); }
template <int VEC>
__device__ __forceinline__ void arg_store(const GrParams& p, uint8_t* a8, int32_t* side, int node, int segb, int cnt, int c,
                                          const int (&off)[VEC]) {
  if (cnt < 256) {                                   // offsets 0..254; 0xFF = no edge selected (all-NaN column)
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) w |= ((uint32_t)off[i] & 0xFFu) << (8 * i);
    stb<VEC>(a8 + (size_t)node * p.ldsave + c, w);
  } else {
    sti<VEC>(side + arg_side_row(segb) * p.ldsave + c, off);
  }
}
template <int VEC>
__device__ __forceinline__ void arg_load(const GrParams& p, const uint8_t* a8, const int32_t* side, int node, int segb, int cnt, int c,
                                         int (&off)[VEC]) {
  if (cnt < 256) {
    const uint32_t w = ldb<VEC>(a8 + (size_t)node * p.ldsave + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) off[i] = (int)((w >> (8 * i)) & 0xFFu);     // 255 never equals an offset of such a segment
  } else {
    ldi<VEC>(side + arg_side_row(segb) * p.ldsave + c, off);
  }
}

// Three launch shapes share the node-level code below.
//  * flat (short segments, E <= 16 N: molecule batches): a workgroup owns a CONTIGUOUS block of nb nodes and a lane one
//    VEC-column piece of the flat (node, column) array of that block, looping over its own node's few edges alone.  U, out,
//    the saved args, gout, gU and (with by_pos) Z and gmsg of consecutive nodes are then contiguous runs of the block -
//    every store instruction of a wave covers whole lines whatever D = T*F is (ZINC: 380 floats per row, 456 per output
//    tower), and a line is only ever shared between the two workgroups at a block boundary.  The round-1 shape (a wave
//    per 2 nodes, column chunks in blockIdx.y, neighbouring nodes on different XCDs) left most 128-byte lines of `out`
//    partially written by workgroups on different XCDs: 1.34x the algorithmic traffic, 0.34 of the HBM peak.
//    Workgroup ids are dealt so that each XCD walks its own contiguous eighth of the node blocks (neighbour rows V[j] of a
//    molecule then sit in the L2 of the XCD that needs them).
//  * wave-per-node: the 64/lpr sub-rows of a wave split one node's edge segment and meet in a butterfly (any degree);
//    behind the flat kernel it only takes the segments above kGroupMaxDeg edges.
constexpr int kGroupMaxDeg = 64;

struct Seg { int b, e; };        // [b, e): positions of one node's target-sorted edges
struct SegIdx { int j, e; };     // per lane: source node and original edge id at position b + lane

__device__ __forceinline__ Seg seg_load(const GrParams& p, int node) {       // beyond N: empty
  Seg s{0, 0};
  if (node < p.N) { s.b = p.rowptr[node]; s.e = p.rowptr[node + 1]; }
  return s;
}
__device__ __forceinline__ SegIdx idx_load(const GrParams& p, int base, int end, int lane) {
  SegIdx r{0, 0};
  if (base + lane < end) { r.j = p.src[base + lane]; r.e = p.perm[base + lane]; }
  return r;
}

struct GrLane { int lane, sub, lpr, epg, c, cc, t, f; bool valid, fused; };

__device__ __forceinline__ GrLane gr_lane(const GrParams& p, int vec) {
  GrLane l;
  l.lane = threadIdx.x & (kWave - 1);
  l.lpr = 1 << p.lpr_log; l.epg = kWave >> p.lpr_log;
  l.sub = l.lane >> p.lpr_log;
  l.c = ((int)blockIdx.y * l.lpr + (l.lane & (l.lpr - 1))) * vec;
  l.valid = l.c < p.D;
  l.cc = l.valid ? l.c : 0;
  l.t = l.cc / p.F; l.f = l.cc - l.t * p.F;       // VEC == 4 only with F % 4 == 0: the 4 columns of a lane sit in one tower
  l.fused = p.inputs == nullptr;
  return l;
}

// running reductions of one (node, VEC columns): only what the aggregator list needs is updated (wave-uniform flags);
// an / ax = OFFSET of the extremal edge inside the node's segment
template <int VEC>
struct GrAcc {
  float sum[VEC], sq[VEC], mn[VEC], mx[VEC];
  int an[VEC], ax[VEC];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sum[i] = 0.f; sq[i] = 0.f; mn[i] = INFINITY; mx[i] = -INFINITY; an[i] = INT_MAX; ax[i] = INT_MAX; }
  }
  __device__ __forceinline__ void take(const GrParams& p, const Vec<VEC>& h, int off) {
    if (p.need_sum) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) sum[i] += h.v[i];
    }
    if (p.need_sq) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) sq[i] += h.v[i] * h.v[i];
    }
    if (p.need_min) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) if (h.v[i] < mn[i]) { mn[i] = h.v[i]; an[i] = off; }   // strict: the first extremal edge wins
    }
    if (p.need_max) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) if (h.v[i] > mx[i]) { mx[i] = h.v[i]; ax[i] = off; }
    }
  }
};

// degree scalers, K aggregates, final layout and the state saved for backward, for the lanes that own columns
template <int VEC>
__device__ __forceinline__ void gr_fwd_store(const GrParams& p, const GrLane& l, int node, int segb, int cnt, GrAcc<VEC>& a,
                                             const float (&fac)[8]) {
  const float deg = (float)max(cnt, 1);                       // degree(...).clamp_(1), mma_conv.py:178-179
  float* o = p.out + ((size_t)node * p.T + l.t) * ((size_t)p.S * p.K * p.F) + l.f;
  float mean[VEC], var[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { mean[i] = 0.f; var[i] = 0.f; }
  if (p.need_mean) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) mean[i] = a.sum[i] / deg;   // scatter mean: sum / clamp(count, 1)
  }
  if (p.need_sq) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) var[i] = a.sq[i] / deg - mean[i] * mean[i];   // mma_conv.py:167-170
  }
  for (int k = 0; k < p.K; ++k) {
    Vec<VEC> run;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      switch (p.aggr[k]) {
        case GR_SUM: run.v[i] = a.sum[i]; break;
        case GR_MEAN: run.v[i] = mean[i]; break;
        case GR_MIN: run.v[i] = cnt ? a.mn[i] : 0.f; break;  // empty target -> 0 (torch_scatter)
        case GR_MAX: run.v[i] = cnt ? a.mx[i] : 0.f; break;
        case GR_VAR: run.v[i] = var[i]; break;
        default: run.v[i] = sqrtf(fmaxf(var[i], 0.f) + 1e-5f); break;
      }
    }
    for (int q = 0; q < p.S; ++q) {                           // compounding (G7)
#pragma unroll
      for (int i = 0; i < VEC; ++i) run.v[i] = run.v[i] * fac[q];
      stv<VEC>(o + (size_t)(q * p.K + k) * p.F, run);
    }
  }
  if (p.amin8) arg_store<VEC>(p, p.amin8, p.amin_side, node, segb, cnt, l.c, a.an);
  if (p.amax8) arg_store<VEC>(p, p.amax8, p.amax_side, node, segb, cnt, l.c, a.ax);
  if (p.mean) {
    const size_t so = (size_t)node * p.ldsave + l.c;
    Vec<VEC> mv, vv;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { mv.v[i] = mean[i]; vv.v[i] = var[i]; }
    stv<VEC>(p.mean + so, mv);
    stv<VEC>(p.var + so, vv);
  }
}

__device__ __forceinline__ void gr_factors(const GrParams& p, float deg, float (&fac)[8]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) fac[q] = q < p.S ? scaler_factor(p.scaler[q], deg, p.avg_log, p.avg_lin) : 1.f;
}

// the scaler factors of every degree a flat-kernel node can have, once per workgroup (logf and the divisions cost ~100
// VALU instructions, which is a sixth of a molecule node's whole budget)
__device__ __forceinline__ void gr_factor_table(const GrParams& p, float (*tab)[8]) {
  for (int d = threadIdx.x; d <= kGroupMaxDeg; d += blockDim.x) {
    float fac[8];
    gr_factors(p, (float)max(d, 1), fac);
#pragma unroll
    for (int q = 0; q < 8; ++q) tab[d][q] = fac[q];
  }
  __syncthreads();
}

// VEC = 4 (F % 4 == 0 and 16-byte aligned rows everywhere): a lane owns 4 consecutive columns of one tower and moves one
// dwordx4 per row, also for the (N,T,S*K*F) output.  Otherwise VEC = 1: lanes own consecutive columns, so all accesses
// are still coalesced 4-byte ones.  (MMAConv pads ZINC's F = 75 to 76 to stay on the VEC = 4 path.)
template <int VEC>
__device__ __forceinline__ void gr_node_fwd(const GrParams& p, const DropParams& dp, const GrLane& l, int node, Seg s,
                                            SegIdx first, const Vec<VEC>& u) {
  if (s.e - s.b < p.wave_min_deg) return;                      // second pass behind the flat kernel: long segments only
  GrAcc<VEC> a;
  a.init();
  for (int base = s.b; base < s.e; base += kWave) {
    const int cnt = min(kWave, s.e - base);
    SegIdx my = first;
    if (base != s.b) my = idx_load(p, base, s.e, l.lane);
    for (int t0 = 0; t0 < cnt; t0 += 2 * l.epg) {            // two edge steps in flight
      const int ta = t0 + l.sub, tb = ta + l.epg;
      const bool two = t0 + l.epg < cnt;                     // wave-uniform
      // cross-lane reads stay outside divergent control flow: ds_bpermute returns 0 from inactive lanes
      const int ja = __shfl(my.j, ta & (kWave - 1), kWave), ea = __shfl(my.e, ta & (kWave - 1), kWave);
      const int jb = __shfl(my.j, tb & (kWave - 1), kWave), eb = __shfl(my.e, tb & (kWave - 1), kWave);
      const bool va = ta < cnt, vb = tb < cnt;
      // lanes past the segment end read a valid row (this node's own / the segment's first edge) and drop the value
      const Vec<VEC> ha = gr_message<VEC>(p, dp, l.fused, u, va ? ja : node, (uint32_t)(va ? ea : 0), (uint32_t)(va ? base + ta : s.b), l.cc);
      Vec<VEC> hb = vzero<VEC>();
      if (two) hb = gr_message<VEC>(p, dp, l.fused, u, vb ? jb : node, (uint32_t)(vb ? eb : 0), (uint32_t)(vb ? base + tb : s.b), l.cc);
      if (va) a.take(p, ha, base + ta - s.b);
      if (two && vb) a.take(p, hb, base + tb - s.b);
    }
  }
  for (int off = kWave / 2; off >= l.lpr; off >>= 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      a.sum[i] += __shfl_xor(a.sum[i], off, kWave);
      a.sq[i] += __shfl_xor(a.sq[i], off, kWave);
      const float omn = __shfl_xor(a.mn[i], off, kWave); const int oan = __shfl_xor(a.an[i], off, kWave);
      const float omx = __shfl_xor(a.mx[i], off, kWave); const int oax = __shfl_xor(a.ax[i], off, kWave);
      if (omn < a.mn[i] || (omn == a.mn[i] && oan < a.an[i])) { a.mn[i] = omn; a.an[i] = oan; }
      if (omx > a.mx[i] || (omx == a.mx[i] && oax < a.ax[i])) { a.mx[i] = omx; a.ax[i] = oax; }
    }
  }
  if (l.sub != 0 || !l.valid) return;
  float fac[8];
  gr_factors(p, (float)max(s.e - s.b, 1), fac);
  gr_fwd_store<VEC>(p, l, node, s.b, s.e - s.b, a, fac);
}

// second pass behind the flat / block kernels when mma_build_csr listed the long segments: one wave per listed node
template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_fwd_list_kernel(const GrParams p) {
  const int raw_count = p.long_nodes[0];
  const int count = min(raw_count, p.long_cap);          // never walk past the list, whatever the count word holds -
  // - but SAY so (round-4 ADVICE: a clobbered count or id would otherwise drop long segments silently): the word behind the list is an
  // error flag the host reads outside capture (functional.DeviceCSR.check)
  if ((raw_count < 0 || raw_count > p.long_cap) && blockIdx.x == 0 && threadIdx.x == 0) const_cast<int32_t*>(p.long_nodes)[1 + p.long_cap] = 1;
  if (count <= 0) return;
  const DropParams dp = drop_resolve(p.drop);
  const GrLane l = gr_lane(p, VEC);
  const int stride = (int)gridDim.x * (kBlock / kWave);
  for (int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6))); i < count; i += stride) {
    const int n = __builtin_amdgcn_readfirstlane(p.long_nodes[1 + i]);
    if ((unsigned)n >= (unsigned)p.N) {                                         // (wave-uniform) not a node: nothing is read or written for it
      if ((threadIdx.x & (kWave - 1)) == 0) const_cast<int32_t*>(p.long_nodes)[1 + p.long_cap] = 2;
      continue;
    }
    const Seg s = seg_load(p, n);
    const SegIdx i0 = idx_load(p, s.b, s.e, l.lane);
    const Vec<VEC> u0 = l.fused ? ldv<VEC>(p.U + (size_t)n * p.lduv + l.cc) : vzero<VEC>();
    gr_node_fwd<VEC>(p, dp, l, n, s, i0, u0);
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_fwd_kernel(const GrParams p) {
  const DropParams dp = drop_resolve(p.drop);
  const GrLane l = gr_lane(p, VEC);
  const int stride = (int)gridDim.x * (kBlock / kWave);
  int n = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)));
  if (n >= p.N) return;
  Seg s0 = seg_load(p, n), s1 = seg_load(p, n + stride);
  SegIdx i0{0, 0};
  Vec<VEC> u0 = vzero<VEC>();
  if (s0.e - s0.b >= p.wave_min_deg) {
    i0 = idx_load(p, s0.b, s0.e, l.lane);
    if (l.fused) u0 = ldv<VEC>(p.U + (size_t)n * p.lduv + l.cc);
  }
  for (;;) {
    const int n1 = n + stride;                               // the host keeps N + 2*stride inside int32
    const Seg s2 = seg_load(p, n1 + stride);
    const bool live1 = n1 < p.N && s1.e - s1.b >= p.wave_min_deg;     // wave-uniform; a skipped node prefetches nothing
    SegIdx i1{0, 0};
    Vec<VEC> u1 = vzero<VEC>();
    if (live1) {
      i1 = idx_load(p, s1.b, s1.e, l.lane);
      if (l.fused) u1 = ldv<VEC>(p.U + (size_t)n1 * p.lduv + l.cc);
    }
    gr_node_fwd<VEC>(p, dp, l, n, s0, i0, u0);
    if (n1 >= p.N) break;
    n = n1; s0 = s1; s1 = s2; i0 = i1; u0 = u1;
  }
}

// ---- flat kernels -----------------------------------------------------------------------------------------------------
// node block of workgroup-slot pb: the 8 XCDs (blocks b and b+8 share one: round-robin dispatch) each take a contiguous
// eighth of the node blocks.  Speed only - any placement gives the same result.
__device__ __forceinline__ int flat_node_block(int pb, int nblocks) {
  const int per_xcd = (nblocks + 7) >> 3;
  const int lb = (pb & 7) * per_xcd + (pb >> 3);
  return lb < nblocks ? lb : -1;
}

template <int VEC>
__device__ __forceinline__ GrLane flat_lane(const GrParams& p, int q, bool fused) {
  GrLane l;
  l.lane = 0; l.sub = 0; l.lpr = 0; l.epg = 0;
  l.c = q * VEC; l.cc = l.c; l.valid = true; l.fused = fused;
  l.t = (int)udiv((uint32_t)l.c, p.f_magic); l.f = l.c - l.t * p.F;
  return l;
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_fwd_flat_kernel(const GrParams p) {
  __shared__ float fac_tab[kGroupMaxDeg + 1][8];
  gr_factor_table(p, fac_tab);
  const DropParams dp = drop_resolve(p.drop);
  const bool fused = p.inputs == nullptr;
  const int nblocks = (p.N + p.nb - 1) / p.nb;
  const int slots = ((nblocks + 7) >> 3) << 3;
  for (int pb = blockIdx.x; pb < slots; pb += gridDim.x) {      // gridDim.x is a multiple of 8: pb % 8 stays the XCD
    const int lb = flat_node_block(pb, nblocks);
    if (lb < 0) continue;
    const int n0 = lb * p.nb;
    const int items = min(p.nb, p.N - n0) * (int)p.qd;
    for (int it = threadIdx.x; it < items; it += kBlock) {
      const int dn = (int)udiv((uint32_t)it, p.qd_magic);   // it / qd
      const int node = n0 + dn;
      const GrLane l = flat_lane<VEC>(p, it - dn * (int)p.qd, fused);
      const int b = p.rowptr[node], deg = p.rowptr[node + 1] - b;
      if (deg > kGroupMaxDeg) continue;                         // left to the wave-per-node pass
      const Vec<VEC> u = fused ? ldv<VEC>(p.U + (size_t)node * p.lduv + l.c) : vzero<VEC>();
      int j[4], e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        j[i] = 0; e[i] = 0;
        if (i < deg) { j[i] = p.src[b + i]; e[i] = p.perm[b + i]; }
      }
      Vec<VEC> h[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {                             // up to four edges' rows in flight before the first use
        h[i] = vzero<VEC>();
        if (i < deg) h[i] = gr_message<VEC>(p, dp, fused, u, j[i], (uint32_t)e[i], (uint32_t)(b + i), l.c);
      }
      GrAcc<VEC> a;
      a.init();
#pragma unroll
      for (int i = 0; i < 4; ++i) if (i < deg) a.take(p, h[i], i);
      for (int t = 4; t < deg; ++t) {                           // ascending positions: ties keep the lowest edge
        const int jj = p.src[b + t], ee = p.perm[b + t];
        a.take(p, gr_message<VEC>(p, dp, fused, u, jj, (uint32_t)ee, (uint32_t)(b + t), l.c), t);
      }
      float fac[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) fac[q] = fac_tab[deg][q];
      gr_fwd_store<VEC>(p, l, node, b, deg, a, fac);
    }
  }
}

// K4: gradient w.r.t. every edge message (each edge has exactly one target: no conflicts)
template <int VEC>
struct GrCoef {       // d(out)/d(h_e) = c_all + [off == amin] c_min + [off == amax] c_max + c_var (h_e - mean), per column
  float c_all[VEC], c_min[VEC], c_max[VEC], c_var[VEC], mean[VEC];
  int an[VEC], ax[VEC];
  bool need_h;
};

template <int VEC>
__device__ __forceinline__ void gr_bwd_coef(const GrParams& p, const GrLane& l, int node, int segb, int cnt, float deg,
                                            const float (&fac)[8], GrCoef<VEC>& k_) {
  const float* go = p.gout + ((size_t)node * p.T + l.t) * ((size_t)p.S * p.K * p.F) + l.f;
  const size_t so = (size_t)node * p.ldsave + l.cc;
  k_.need_h = false;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { k_.c_all[i] = k_.c_min[i] = k_.c_max[i] = k_.c_var[i] = 0.f; k_.an[i] = k_.ax[i] = -1; k_.mean[i] = 0.f; }
  if (p.amin8) arg_load<VEC>(p, p.amin8, p.amin_side, node, segb, cnt, l.cc, k_.an);
  if (p.amax8) arg_load<VEC>(p, p.amax8, p.amax_side, node, segb, cnt, l.cc, k_.ax);
  for (int k = 0; k < p.K; ++k) {
    float gb[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) gb[i] = 0.f;
    float run = 1.f;
    for (int q = 0; q < p.S; ++q) {
      run = run * fac[q];
      const Vec<VEC> gv = ldv<VEC>(go + (size_t)(q * p.K + k) * p.F);
#pragma unroll
      for (int i = 0; i < VEC; ++i) gb[i] = fmaf(gv.v[i], run, gb[i]);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      switch (p.aggr[k]) {
        case GR_SUM: k_.c_all[i] += gb[i]; break;
        case GR_MEAN: k_.c_all[i] += gb[i] / deg; break;
        case GR_MIN: k_.c_min[i] += gb[i]; break;
        case GR_MAX: k_.c_max[i] += gb[i]; break;
        case GR_VAR: k_.c_var[i] += gb[i] * 2.f / deg; k_.need_h = true; break;
        default: {
          const float v = p.var[so + i];
          if (v > 0.f) k_.c_var[i] += gb[i] / (2.f * sqrtf(v + 1e-5f)) * 2.f / deg;   // relu'(v) = [v > 0]
          k_.need_h = true;
        } break;
      }
    }
  }
  if (k_.need_h) {
    const Vec<VEC> mv = ldv<VEC>(p.mean + so);
#pragma unroll
    for (int i = 0; i < VEC; ++i) k_.mean[i] = mv.v[i];
  }
}

// gradient of the edge at position pos (offset off in its segment, original id e, source j); written to gmsg[pos or e]
template <int VEC>
__device__ __forceinline__ Vec<VEC> gr_bwd_edge(const GrParams& p, const DropParams& dp, const GrLane& l, const GrCoef<VEC>& k_,
                                                const Vec<VEC>& u, int j, int e, int pos, int off) {
  Vec<VEC> g;
  Vec<VEC> h = vzero<VEC>();
  if (k_.need_h) h = gr_message<VEC>(p, dp, l.fused, u, j, (uint32_t)e, (uint32_t)pos, l.cc);
  float fd[VEC];
  if (l.fused && dp.mode != MMA_DROP_NONE) drop_factors<VEC>(dp, (uint32_t)e, 0, l.cc, p.D, 0, fd);
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) fd[i] = 1.f;
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float gi = k_.c_all[i] + (off == k_.an[i] ? k_.c_min[i] : 0.f) + (off == k_.ax[i] ? k_.c_max[i] : 0.f);
    if (k_.need_h) gi += k_.c_var[i] * (h.v[i] - k_.mean[i]);
    g.v[i] = gi * fd[i];
  }
  stv<VEC>(p.gmsg + (size_t)(p.by_pos ? pos : e) * p.ldg + l.cc, g);
  if (p.gmsg_rmax) {                                          // wave-uniform; the generic kernels merge per lane (the block kernel reduces in LDS first)
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) m = fmaxf(m, fabsf(g.v[i]));
    if (m > 0.f) atomicMax(p.gmsg_rmax + (p.by_pos ? pos : e), __float_as_uint(m));
  }
  return g;
}

template <int VEC>
__device__ __forceinline__ void gr_node_bwd(const GrParams& p, const DropParams& dp, const GrLane& l, int node, Seg s,
                                            SegIdx first) {
  if (s.e - s.b < p.wave_min_deg) return;                     // second pass behind the flat kernel: long segments only
  Vec<VEC> su = vzero<VEC>();                                 // dL/dU[node] = sum of the segment's message gradients
  if (s.b != s.e) {
    const float deg = (float)(s.e - s.b);
    float fac[8];
    gr_factors(p, deg, fac);
    GrCoef<VEC> k_;
    gr_bwd_coef<VEC>(p, l, node, s.b, s.e - s.b, deg, fac, k_);
    const Vec<VEC> u = (l.fused && k_.need_h) ? ldv<VEC>(p.U + (size_t)node * p.lduv + l.cc) : vzero<VEC>();
    for (int base = s.b; base < s.e; base += kWave) {
      const int cnt = min(kWave, s.e - base);
      SegIdx my = first;
      if (base != s.b) my = idx_load(p, base, s.e, l.lane);
      for (int t0 = 0; t0 < cnt; t0 += l.epg) {
        const int tt = t0 + l.sub;
        const int j_ = __shfl(my.j, tt & (kWave - 1), kWave);
        const int e_ = __shfl(my.e, tt & (kWave - 1), kWave);
        if (tt < cnt && l.valid) {
          const Vec<VEC> g = gr_bwd_edge<VEC>(p, dp, l, k_, u, j_, e_, base + tt, base + tt - s.b);
#pragma unroll
          for (int i = 0; i < VEC; ++i) su.v[i] += g.v[i];
        }
      }
    }
  }
  if (p.gU) {                                                 // wave-uniform; an empty target gets a zero row
    for (int off = kWave / 2; off >= l.lpr; off >>= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) su.v[i] += __shfl_xor(su.v[i], off, kWave);
    }
    if (l.sub == 0 && l.valid) {
      stv<VEC>(p.gU + (size_t)node * p.ldgu + l.c, su);
      if (p.gu_rmax) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) m = fmaxf(m, fabsf(su.v[i]));
        if (m > 0.f) atomicMax(p.gu_rmax + node, __float_as_uint(m));
      }
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_bwd_list_kernel(const GrParams p) {
  const int raw_count = p.long_nodes[0];
  const int count = min(raw_count, p.long_cap);          // never walk past the list, whatever the count word holds -
  // - but SAY so (round-4 ADVICE: a clobbered count or id would otherwise drop long segments silently): the word behind the list is an
  // error flag the host reads outside capture (functional.DeviceCSR.check)
  if ((raw_count < 0 || raw_count > p.long_cap) && blockIdx.x == 0 && threadIdx.x == 0) const_cast<int32_t*>(p.long_nodes)[1 + p.long_cap] = 1;
  if (count <= 0) return;
  const DropParams dp = drop_resolve(p.drop);
  const GrLane l = gr_lane(p, VEC);
  const int stride = (int)gridDim.x * (kBlock / kWave);
  for (int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6))); i < count; i += stride) {
    const int n = __builtin_amdgcn_readfirstlane(p.long_nodes[1 + i]);
    if ((unsigned)n >= (unsigned)p.N) {
      if ((threadIdx.x & (kWave - 1)) == 0) const_cast<int32_t*>(p.long_nodes)[1 + p.long_cap] = 2;
      continue;
    }
    const Seg s = seg_load(p, n);
    gr_node_bwd<VEC>(p, dp, l, n, s, idx_load(p, s.b, s.e, l.lane));
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_bwd_kernel(const GrParams p) {
  const DropParams dp = drop_resolve(p.drop);
  const GrLane l = gr_lane(p, VEC);
  const int stride = (int)gridDim.x * (kBlock / kWave);
  int n = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)));
  if (n >= p.N) return;
  Seg s0 = seg_load(p, n), s1 = seg_load(p, n + stride);
  SegIdx i0 = idx_load(p, s0.b, s0.e, l.lane);
  for (;;) {
    const int n1 = n + stride;
    const Seg s2 = seg_load(p, n1 + stride);
    SegIdx i1{0, 0};
    if (s1.e - s1.b >= p.wave_min_deg) i1 = idx_load(p, s1.b, s1.e, l.lane);      // wave-uniform
    gr_node_bwd<VEC>(p, dp, l, n, s0, i0);
    if (n1 >= p.N) break;
    n = n1; s0 = s1; s1 = s2; i0 = i1;
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gr_bwd_flat_kernel(const GrParams p) {
  __shared__ float fac_tab[kGroupMaxDeg + 1][8];
  gr_factor_table(p, fac_tab);
  const DropParams dp = drop_resolve(p.drop);
  const bool fused = p.inputs == nullptr;
  const bool need_e = !p.by_pos || (fused && dp.mode != MMA_DROP_NONE) || !fused;   // original edge ids: gmsg rows / dropout key / inputs rows
  const int nblocks = (p.N + p.nb - 1) / p.nb;
  const int slots = ((nblocks + 7) >> 3) << 3;
  for (int pb = blockIdx.x; pb < slots; pb += gridDim.x) {
    const int lb = flat_node_block(pb, nblocks);
    if (lb < 0) continue;
    const int n0 = lb * p.nb;
    const int items = min(p.nb, p.N - n0) * (int)p.qd;
    for (int it = threadIdx.x; it < items; it += kBlock) {
      const int dn = (int)udiv((uint32_t)it, p.qd_magic);
      const int node = n0 + dn;
      const GrLane l = flat_lane<VEC>(p, it - dn * (int)p.qd, fused);
      const int b = p.rowptr[node], deg = p.rowptr[node + 1] - b;
      if (deg > kGroupMaxDeg) continue;
      Vec<VEC> su = vzero<VEC>();                             // dL/dU[node]: this lane walks the whole segment, so it is a local sum
      if (deg > 0) {
        float fac[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) fac[q] = fac_tab[deg][q];
        GrCoef<VEC> k_;
        gr_bwd_coef<VEC>(p, l, node, b, deg, (float)deg, fac, k_);
        const Vec<VEC> u = (fused && k_.need_h) ? ldv<VEC>(p.U + (size_t)node * p.lduv + l.c) : vzero<VEC>();
        for (int t = 0; t < deg; ++t) {
          const int jj = k_.need_h ? p.src[b + t] : 0;
          const int ee = need_e ? p.perm[b + t] : 0;
          const Vec<VEC> g = gr_bwd_edge<VEC>(p, dp, l, k_, u, jj, ee, b + t, t);
#pragma unroll
          for (int cidx = 0; cidx < VEC; ++cidx) su.v[cidx] += g.v[cidx];
        }
      }
      if (p.gU) {
        stv<VEC>(p.gU + (size_t)node * p.ldgu + l.c, su);
        if (p.gu_rmax) {
          float m = 0.f;
#pragma unroll
          for (int cidx = 0; cidx < VEC; ++cidx) m = fmaxf(m, fabsf(su.v[cidx]));
          if (m > 0.f) atomicMax(p.gu_rmax + node, __float_as_uint(m));
        }
      }
    }
  }
}

// ---- block kernels: the fast path of the molecule-batch shape ----------------------------------------------------------
// Same node-block mapping as the flat kernels, ONE workgroup per block of nb nodes, in two phases with LDS between them:
//   A  loads only: a lane owns a (node, 4 columns) item and issues a FIXED batch of loads for it (three edge slots,
//      duplicates clamped to the last edge - molecule degrees are 1..4 -, plus a rare conditional tail; backward: the K*S
//      gradient blocks + the arg bytes) before any arithmetic, and leaves the K aggregates (forward) or the gradient
//      coefficients (backward) in LDS;
//   B  stores only: the block's region of `out` (forward) is ONE contiguous run - every lane writes one float4 of it, read
//      back from LDS and scaled, so each store instruction is 1 KB of consecutive bytes whatever D, F and S*K are;
//      backward writes the block's contiguous message-gradient rows and dL/dU rows.
// Why two phases: CDNA has ONE in-order vmcnt for loads and stores, so an item loop that loads, stores, loads ... makes
// every wait for the next item's rows also wait for the previous item's stores to reach memory (PMC on the flat kernel:
// waves parked 83 % of their cycles, 850 VALU instructions per item mostly on run-time switches).  With the phases split a
// workgroup never waits for a store; other workgroups of the CU fill the pipe meanwhile.  Message form, dropout and the
// set of running reductions are template parameters.
constexpr int kBlkCap = 256;         // edge positions of a block staged in LDS (src, perm); blocks with more read them from memory
constexpr int kBlkMaxKS = 8;         // backward: gradient blocks (K*S) loaded per item in one batch
enum { NEED_SUM = 1, NEED_MIN = 2, NEED_MAX = 4 };


struct BlkLds {
  float (*fac)[8]; float (*pre)[8]; int* rowptr; int* src; int* perm; uint8_t* zt; float* agg; uint8_t* arg;
};
constexpr uint32_t kBlkHead = 2 * (kGroupMaxDeg + 1) * 32 + 32 * 4 + 2 * kBlkCap * 4 + kBlkCap;   // two scaler tables, row pointers, src, perm, edge types
__device__ __forceinline__ BlkLds blk_lds(const GrParams& p, unsigned char* smem) {
  BlkLds L;
  L.fac = reinterpret_cast<float(*)[8]>(smem);
  L.pre = reinterpret_cast<float(*)[8]>(smem + (kGroupMaxDeg + 1) * 32);
  L.rowptr = reinterpret_cast<int*>(smem + 2 * (kGroupMaxDeg + 1) * 32);
  L.src = L.rowptr + 32;
  L.perm = L.src + kBlkCap;
  L.zt = reinterpret_cast<uint8_t*>(L.perm + kBlkCap);
  L.agg = reinterpret_cast<float*>(smem + p.lds_agg);
  L.arg = smem + p.lds_arg;
  return L;
}

// stage the block's row pointers, edge indices and the scaler tables (fac[d][q] and its running product pre[d][q], in the
// reference's multiplication order); returns false when the workgroup has no block
__device__ __forceinline__ bool blk_stage(const GrParams& p, const BlkLds& L, int& n0, int& n_here, int& p0, int& p1, bool& staged,
                                          bool want_types = false) {
  const int nblocks = (p.N + p.nb - 1) / p.nb;
  const int lb = flat_node_block((int)blockIdx.x, nblocks);
  if (lb < 0) return false;
  n0 = lb * p.nb;
  n_here = min(p.nb, p.N - n0);
  p0 = p.rowptr[n0]; p1 = p.rowptr[n0 + n_here];            // wave-uniform addresses: scalar loads
  staged = p1 - p0 <= kBlkCap;
  const int tid = threadIdx.x;
  if (tid <= n_here) L.rowptr[tid] = p.rowptr[n0 + tid];
  if (staged) {
    for (int i = tid; i < p1 - p0; i += kBlock) {
      L.src[i] = p.src[p0 + i]; L.perm[i] = p.perm[p0 + i];
      if (want_types && p.zidx) L.zt[i] = p.zidx[p.by_pos ? p0 + i : L.perm[i]];   // the edge's type next to its indices: no dependent load later
    }
  }
  for (int d = tid; d <= kGroupMaxDeg; d += kBlock) {
    float run = 1.f;
    for (int q = 0; q < p.S; ++q) {
      const float f = scaler_factor((int)((p.scaler_pack >> (4 * q)) & 15u), (float)max(d, 1), p.avg_log, p.avg_lin);
      run = run * f;
      L.fac[d][q] = f; L.pre[d][q] = run;
    }
  }
  __syncthreads();
  return true;
}

// keep multipliers of the GR dropout (HASH mode only; keyed by the original edge id, mask index 0)
__device__ __forceinline__ void blk_drop(const DropParams& d, uint32_t e, int c, float (&f)[4]) {
  const uint32_t r = drop_mix(drop_edge_key(e + d.edge_base, d.seed_lo) ^ drop_col_key((uint32_t)(c >> 2), d.seed_hi));
  if (d.mode == MMA_DROP_HASH16) {      // wave-uniform: a threshold that is no multiple of 256 (mma_conv.py:67 hard-codes 0.5; `dropout` is public)
    drop_unpack16<4>(d, r, drop_low_word(r, drop_mask_mult2(0)), 0, f);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = ((r >> (8 * i)) & 0xFFu) >= d.thr ? d.scale : 0.f;
}

template <int NEEDS>
struct BlkAcc {
  float sum[4], mn[4], mx[4]; int an[4], ax[4];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { sum[i] = 0.f; mn[i] = INFINITY; mx[i] = -INFINITY; an[i] = INT_MAX; ax[i] = INT_MAX; }
  }
  __device__ __forceinline__ void take(const Vec<4>& h, int off, bool on) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (NEEDS & NEED_SUM) sum[i] += on ? h.v[i] : 0.f;
      if (NEEDS & NEED_MIN) { if (on && h.v[i] < mn[i]) { mn[i] = h.v[i]; an[i] = off; } }   // strict: the first extremal edge wins
      if (NEEDS & NEED_MAX) { if (on && h.v[i] > mx[i]) { mx[i] = h.v[i]; ax[i] = off; } }
    }
  }
};

// h = drop((V + U) + Z) from raw rows - the same association as gr_message
template <bool FUSED, bool HASZ, bool DROP>
__device__ __forceinline__ Vec<4> blk_combine(const GrParams& p, const DropParams& dp, const Vec<4>& u, Vec<4> v, const Vec<4>& z,
                                              uint32_t e, int c) {
  if (FUSED) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v.v[i] += u.v[i];
    if (HASZ) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v.v[i] += z.v[i];
    }
    if (DROP) {
      float f[4];
      blk_drop(dp, e, c, f);
#pragma unroll
      for (int i = 0; i < 4; ++i) v.v[i] *= f[i];
    }
  }
  return v;
}

template <bool FUSED, bool HASZ, bool DROP, int NEEDS, bool STAGED>
__device__ __forceinline__ void blk_fwd_items(const GrParams& p, const DropParams& dp, const BlkLds& L, int n0, int n_here, int p0) {
  const int items = n_here * (int)p.qd;
  for (int it = threadIdx.x; it < items; it += kBlock) {
    const int dn = (int)udiv((uint32_t)it, p.qd_magic);
    const int c = (it - dn * (int)p.qd) * 4;
    const int node = n0 + dn;
    const int b = L.rowptr[dn], deg = L.rowptr[dn + 1] - b;
    if (deg > kGroupMaxDeg) continue;                         // a long segment: left to the wave-per-node pass
    // ---- the item's loads, all issued before the first use: U row + three edge slots (slot i = edge min(i, deg-1); an
    // empty segment re-reads the block's first edge and drops it)
    int j[3], e[3], pos[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      pos[i] = deg > 0 ? b + min(i, deg - 1) : p0;
      if (STAGED) { j[i] = L.src[pos[i] - p0]; e[i] = L.perm[pos[i] - p0]; }
      else { j[i] = p.src[pos[i]]; e[i] = p.perm[pos[i]]; }
    }
    const Vec<4> u = FUSED ? ldv<4>(p.U + (size_t)node * p.lduv + c) : vzero<4>();
    Vec<4> v[3], z[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i] = FUSED ? ldv<4>(p.V + (size_t)j[i] * p.lduv + c) : ldv<4>(p.inputs + (size_t)e[i] * p.ldi + c);
      const size_t zr = (STAGED && p.zidx) ? (size_t)L.zt[pos[i] - p0] : gr_zrow(p, pos[i], e[i]);
      z[i] = (FUSED && HASZ) ? ldv<4>(p.Z + zr * p.ldz + c) : vzero<4>();
    }
    BlkAcc<NEEDS> a;
    a.init();
#pragma unroll
    for (int i = 0; i < 3; ++i) a.take(blk_combine<FUSED, HASZ, DROP>(p, dp, u, v[i], z[i], (uint32_t)e[i], c), i, i < deg);
    for (int t = 3; t < deg; ++t) {                           // rare on molecule batches (degree-4 atoms)
      const int pt = b + t;
      const int jj = STAGED ? L.src[pt - p0] : p.src[pt], ee = STAGED ? L.perm[pt - p0] : p.perm[pt];
      const Vec<4> vv = FUSED ? ldv<4>(p.V + (size_t)jj * p.lduv + c) : ldv<4>(p.inputs + (size_t)ee * p.ldi + c);
      const size_t zrr = (STAGED && p.zidx) ? (size_t)L.zt[pt - p0] : gr_zrow(p, pt, ee);
      const Vec<4> zz = (FUSED && HASZ) ? ldv<4>(p.Z + zrr * p.ldz + c) : vzero<4>();
      a.take(blk_combine<FUSED, HASZ, DROP>(p, dp, u, vv, zz, (uint32_t)ee, c), t, true);
    }
    const float fdeg = (float)max(deg, 1);                    // degree(...).clamp_(1), mma_conv.py:178-179
    for (int k = 0; k < p.K; ++k) {
      const int code = (int)((p.aggr_pack >> (4 * k)) & 15u);
      Vec<4> r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float val = 0.f;
        if (NEEDS & NEED_SUM) val = code == GR_SUM ? a.sum[i] : (code == GR_MEAN ? a.sum[i] / fdeg : val);   // scatter mean: sum / clamp(count, 1)
        if (NEEDS & NEED_MIN) val = code == GR_MIN ? (deg ? a.mn[i] : 0.f) : val;                            // empty target -> 0 (torch_scatter)
        if (NEEDS & NEED_MAX) val = code == GR_MAX ? (deg ? a.mx[i] : 0.f) : val;
        r.v[i] = val;
      }
      stv<4>(L.agg + ((size_t)dn * p.K + k) * p.D + c, r);
    }
    if ((NEEDS & NEED_MIN) && p.amin8) {
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) w |= ((uint32_t)a.an[i] & 0xFFu) << (8 * i);
      *reinterpret_cast<uint32_t*>(L.arg + (size_t)dn * p.D + c) = w;
    }
    if ((NEEDS & NEED_MAX) && p.amax8) {
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) w |= ((uint32_t)a.ax[i] & 0xFFu) << (8 * i);
      *reinterpret_cast<uint32_t*>(L.arg + (size_t)(p.nb + dn) * p.D + c) = w;
    }
  }
}

template <bool FUSED, bool HASZ, bool DROP, int NEEDS>
__global__ __launch_bounds__(kBlock) void gr_fwd_block_kernel(const GrParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const BlkLds L = blk_lds(p, smem);
  int n0, n_here, p0, p1; bool staged;
  if (!blk_stage(p, L, n0, n_here, p0, p1, staged, HASZ)) return;
  const DropParams dp = DROP ? drop_resolve(p.drop) : p.drop;
  const int tid = threadIdx.x;

  // ---- phase A: (node, 4 columns) items -> K aggregates + args in LDS
  if (p1 == p0) {                                             // no edge in the whole block: every aggregate is 0
    for (int w = tid; w < n_here * p.K * (int)p.qd; w += kBlock) stv<4>(L.agg + (size_t)w * 4, vzero<4>());
    for (int w = tid; w < (2 * p.nb * p.D) >> 2; w += kBlock) reinterpret_cast<uint32_t*>(L.arg)[w] = 0xFFFFFFFFu;
  } else if (staged) {
    blk_fwd_items<FUSED, HASZ, DROP, NEEDS, true>(p, dp, L, n0, n_here, p0);
  } else {
    blk_fwd_items<FUSED, HASZ, DROP, NEEDS, false>(p, dp, L, n0, n_here, p0);
  }
  __syncthreads();

  // ---- phase B: the block's run of `out`, one float4 per lane and step: out[n, t, (q*K + k)*F + f] = agg_k * prod_{q' <= q} fac_q'
  const int total = n_here * (int)p.tskf4;
  float* ob = p.out + (size_t)n0 * p.T * ((size_t)p.S * p.K * p.F);
  for (int w = tid; w < total; w += kBlock) {
    const uint32_t dn = udiv((uint32_t)w, p.m_tskf4);
    uint32_t r = (uint32_t)w - dn * p.tskf4;
    const uint32_t t = udiv(r, p.m_skf4);
    r -= t * p.skf4;
    const uint32_t qk = udiv(r, p.m_f4);
    const uint32_t fq = r - qk * p.f4;
    const uint32_t q = udiv(qk, p.m_k), k = qk - q * (uint32_t)p.K;
    const int deg = L.rowptr[dn + 1] - L.rowptr[dn];
    if (deg > kGroupMaxDeg) continue;
    Vec<4> v = ldv<4>(L.agg + ((size_t)dn * p.K + k) * p.D + t * p.F + fq * 4);
    for (uint32_t qq = 0; qq <= q; ++qq) {                    // compounding (G7), in the reference's order: ((v f0) f1) ...
      const float f = L.fac[deg][qq];
#pragma unroll
      for (int i = 0; i < 4; ++i) v.v[i] = v.v[i] * f;
    }
    stv<4>(ob + (size_t)w * 4, v);
  }
  // saved args: byte rows of the block, skipping the long segments' rows (written by the wave pass)
  const int dq = p.D >> 2;
  for (int w = tid; w < n_here * dq; w += kBlock) {
    const int dn = (int)udiv((uint32_t)w, p.qd_magic), cq = w - dn * dq;    // qd == D/4 on this path
    if (L.rowptr[dn + 1] - L.rowptr[dn] > kGroupMaxDeg) continue;
    if ((NEEDS & NEED_MIN) && p.amin8)
      *reinterpret_cast<uint32_t*>(p.amin8 + (size_t)(n0 + dn) * p.ldsave + cq * 4) = *reinterpret_cast<const uint32_t*>(L.arg + (size_t)dn * p.D + cq * 4);
    if ((NEEDS & NEED_MAX) && p.amax8)
      *reinterpret_cast<uint32_t*>(p.amax8 + (size_t)(n0 + dn) * p.ldsave + cq * 4) =
          *reinterpret_cast<const uint32_t*>(L.arg + (size_t)(p.nb + dn) * p.D + cq * 4);
  }
}

// backward: phase A reduces the K*S gradient blocks of a (node, 4 columns) item to the coefficients
//   dL/dh_e = c_all + [off == amin] c_min + [off == amax] c_max        (LDS: coef[node][0..n_coef)[D], args as bytes)
// phase B walks the item's edges and stores the message gradients (+ their sum, dL/dU) - stores only.
template <bool FUSED, bool DROP, int NEEDS>
__global__ __launch_bounds__(kBlock) void gr_bwd_block_kernel(const GrParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const BlkLds L = blk_lds(p, smem);
  int n0, n_here, p0, p1; bool staged;
  if (!blk_stage(p, L, n0, n_here, p0, p1, staged)) return;
  const DropParams dp = DROP ? drop_resolve(p.drop) : p.drop;
  const int tid = threadIdx.x;
  constexpr int NC = ((NEEDS & NEED_SUM) ? 1 : 0) + ((NEEDS & NEED_MIN) ? 1 : 0) + ((NEEDS & NEED_MAX) ? 1 : 0);
  constexpr int I_ALL = 0, I_MIN = (NEEDS & NEED_SUM) ? 1 : 0, I_MAX = I_MIN + ((NEEDS & NEED_MIN) ? 1 : 0);
  const int items = n_here * (int)p.qd;
  const int KS = p.K * p.S;                                   // <= kBlkMaxKS on this path
  // row maxima (optional): a row's 4-column items sit in different lanes and waves - merged in LDS (edges of a staged block, the
  // block's nodes), then ONE global merge per row; blocks whose edges do not fit the staging area merge per item in global memory
  __shared__ uint32_t rmax_b[32];                             // per node of the block: the bound of its edges' rows (phase A)
  __shared__ uint32_t rmax_n[32];                             // per node: max |dL/dU row| (phase B, exact)
  const bool want_rmax = p.gmsg_rmax != nullptr;              // both arrays or neither (host check)

  for (int it = tid; it < items; it += kBlock) {
    const int dn = (int)udiv((uint32_t)it, p.qd_magic);
    const int c = (it - dn * (int)p.qd) * 4;
    const int node = n0 + dn;
    const int deg = L.rowptr[dn + 1] - L.rowptr[dn];
    if (deg > kGroupMaxDeg || deg == 0) continue;
    const uint32_t t = udiv((uint32_t)c, p.f_magic);
    const int f = c - (int)t * p.F;
    const float* go = p.gout + ((size_t)node * p.T + t) * ((size_t)p.S * p.K * p.F) + f;
    // one batch of loads: the K*S gradient blocks (slots past K*S re-read the last one and are dropped) + the arg bytes
    // (round 3: blocks past K*S are not loaded at all - with the scalers factored out of `out`, K*S is 2 at ZINC's shape and the six
    // clamped re-reads of the last block were most of this batch)
    Vec<4> gv[kBlkMaxKS];
#pragma unroll
    for (int i = 0; i < kBlkMaxKS; ++i) {
      gv[i] = vzero<4>();
      if (i < KS) gv[i] = ldv_nt<4>(go + (size_t)i * p.F);          // wave-uniform condition
    }
    uint32_t wn = 0xFFFFFFFFu, wx = 0xFFFFFFFFu;
    if ((NEEDS & NEED_MIN) && p.amin8) wn = ldb<4>(p.amin8 + (size_t)node * p.ldsave + c);     // NULL: no min in the aggregator list
    if ((NEEDS & NEED_MAX) && p.amax8) wx = ldb<4>(p.amax8 + (size_t)node * p.ldsave + c);
    float c_all[4] = {0.f, 0.f, 0.f, 0.f}, c_min[4] = {0.f, 0.f, 0.f, 0.f}, c_max[4] = {0.f, 0.f, 0.f, 0.f};
    const float inv_cnt_scale = (float)deg;
#pragma unroll
    for (int i = 0; i < kBlkMaxKS; ++i) {                      // block i = (q, k) = (i / K, i % K), weight = prod_{q' <= q} fac_q'
      if (i < KS) {
        const uint32_t q = udiv((uint32_t)i, p.m_k), k = (uint32_t)i - q * (uint32_t)p.K;
        const float run = L.pre[deg][q];
        const int code = (int)((p.aggr_pack >> (4 * k)) & 15u);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          if (NEEDS & NEED_SUM) {
            if (code == GR_SUM) c_all[x] = fmaf(gv[i].v[x], run, c_all[x]);
            if (code == GR_MEAN) c_all[x] = fmaf(gv[i].v[x], run / inv_cnt_scale, c_all[x]);
          }
          if (NEEDS & NEED_MIN) { if (code == GR_MIN) c_min[x] = fmaf(gv[i].v[x], run, c_min[x]); }
          if (NEEDS & NEED_MAX) { if (code == GR_MAX) c_max[x] = fmaf(gv[i].v[x], run, c_max[x]); }
        }
      }
    }
    float* cf = L.agg + (size_t)dn * NC * p.D + c;
    if (NEEDS & NEED_SUM) { Vec<4> v; for (int i = 0; i < 4; ++i) v.v[i] = c_all[i]; stv<4>(cf + I_ALL * p.D, v); }
    if (NEEDS & NEED_MIN) { Vec<4> v; for (int i = 0; i < 4; ++i) v.v[i] = c_min[i]; stv<4>(cf + I_MIN * p.D, v); }
    if (NEEDS & NEED_MAX) { Vec<4> v; for (int i = 0; i < 4; ++i) v.v[i] = c_max[i]; stv<4>(cf + I_MAX * p.D, v); }
    if (NEEDS & NEED_MIN) *reinterpret_cast<uint32_t*>(L.arg + (size_t)dn * p.D + c) = wn;
    if (NEEDS & NEED_MAX) *reinterpret_cast<uint32_t*>(L.arg + (size_t)(p.nb + dn) * p.D + c) = wx;
  }
  __syncthreads();

  // FAST = the block's edges are staged in LDS and the gradient rows are stored by position (the layer's own call).  One loop for both
  // cases compiled `staged ? L.perm[..] : p.perm[..]` into a FLAT load from a selected address - and a flat load counts in vmcnt AND
  // lgkmcnt, so every iteration of the edge loop began with s_waitcnt vmcnt(0) lgkmcnt(0): a wait for the acknowledgement of the gradient
  // row the iteration before it had just stored (found in the ISA, round 5; K4 0.39 -> see DESIGN.md).  FAST also walks the rows by
  // pointer increments (the 64-bit row x pitch product was three quarter-rate multiplies per edge and item).
  auto phase_b = [&](auto fast_c) {
  constexpr bool FAST = decltype(fast_c)::value;
  for (int it = tid; it < items; it += kBlock) {
    const int dn = (int)udiv((uint32_t)it, p.qd_magic);
    const int c = (it - dn * (int)p.qd) * 4;
    const int node = n0 + dn;
    const int b = L.rowptr[dn], deg = L.rowptr[dn + 1] - b;
    if (deg > kGroupMaxDeg) continue;
    Vec<4> su = vzero<4>();                                   // dL/dU[node]: the sum of the segment's message gradients, in position order
    float m_b = 0.f;                                          // this item's share of the node's message-gradient bound
    if (deg > 0) {
      const float* cf = L.agg + (size_t)dn * NC * p.D + c;
      Vec<4> ca = vzero<4>(), cn = vzero<4>(), cx = vzero<4>();
      uint32_t wn = 0xFFFFFFFFu, wx = 0xFFFFFFFFu;
      if (NEEDS & NEED_SUM) ca = ldv<4>(cf + I_ALL * p.D);
      if (NEEDS & NEED_MIN) { cn = ldv<4>(cf + I_MIN * p.D); wn = *reinterpret_cast<const uint32_t*>(L.arg + (size_t)dn * p.D + c); }
      if (NEEDS & NEED_MAX) { cx = ldv<4>(cf + I_MAX * p.D); wx = *reinterpret_cast<const uint32_t*>(L.arg + (size_t)(p.nb + dn) * p.D + c); }
      if (want_rmax) {
        // every message gradient of this node is c_all + [arg hit] c_min + [arg hit] c_max, times the keep factor: ONE bound per node stands
        // for the row maximum of each of its edges - an upper bound (the three-product GEMMs take bounds; an edge is the arg of about 1/deg
        // of the columns, so the bound is rarely more than a binade loose), at a cost per NODE instead of per edge and lane
#pragma unroll
        for (int x = 0; x < 4; ++x) m_b = fmaxf(m_b, fabsf(ca.v[x]) + fabsf(cn.v[x]) + fabsf(cx.v[x]));
        if (DROP) m_b *= dp.scale;
      }
      float* gp = p.gmsg + (size_t)b * p.ldg + c;              // FAST: row `pos` of the message gradients
      for (int tt = 0; tt < deg; ++tt) {
        const int pos = b + tt;
        int ee = 0;
        if (FAST) { if (DROP) ee = L.perm[pos - p0]; }
        else if (DROP || !p.by_pos) ee = staged ? L.perm[pos - p0] : p.perm[pos];
        float fd[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) blk_drop(dp, (uint32_t)ee, c, fd);
        Vec<4> g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gi = ca.v[i] + ((uint32_t)tt == ((wn >> (8 * i)) & 0xFFu) ? cn.v[i] : 0.f) +
                           ((uint32_t)tt == ((wx >> (8 * i)) & 0xFFu) ? cx.v[i] : 0.f);
          g.v[i] = gi * fd[i];
          su.v[i] += g.v[i];
        }
        if (FAST) { stv_nt<4>(gp, g); gp += p.ldg; }
        else stv_nt<4>(p.gmsg + (size_t)(p.by_pos ? pos : ee) * p.ldg + c, g);
      }
    }
    if (p.gU) stv<4>(p.gU + (size_t)node * p.ldgu + c, su);
    if (want_rmax) {
      // The consumers take only the EXPONENT of a row maximum (power-of-two scales): both maxima travel as the upper 16 bits of their
      // fp32 patterns rounded UP (a bound within 2^-7 of the value, as good as the value), one 16-bit pair per lane, ONE butterfly
      const float m_u = p.gU ? fmaxf(fmaxf(fabsf(su.v[0]), fabsf(su.v[1])), fmaxf(fabsf(su.v[2]), fabsf(su.v[3]))) : 0.f;
      // ... and the item leaves the pair in the LDS word its min-arg bytes came from (its own word, read above: no other item touches it).
      // Nothing is merged here: the first two forms reduced across the wavefront and merged into a word per node for every item - ballots,
      // lane reads, one or two butterflies, branches around elected-lane LDS atomics: ~40 instructions per item (run-time switches, one
      // box, on the kernel that still had the flat load below: K4 0.43 ms with the maxima, 0.385 without, 0.39 without the merges).
      *reinterpret_cast<uint32_t*>(L.arg + (size_t)dn * p.D + c) = up16_nonneg(m_b) | (up16_nonneg(m_u) << 16);
    }
  }
  };
  if (staged && p.by_pos) phase_b(std::true_type{}); else phase_b(std::false_type{});
  if (want_rmax) {
    // the LDS words of all four wavefronts, NOT their global stores: __syncthreads() is also a fence, i.e. s_waitcnt vmcnt(0) - every
    // wavefront would sit until the message-gradient rows it has just stored are acknowledged
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // one pass per block: 32 lanes per node fold its row of pairs (v_pk_max_u16), a five-step butterfly inside the 32, one word per node
    {
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      auto pk = [](uint32_t a, uint32_t b) {
        const us2 r = __builtin_elementwise_max(*reinterpret_cast<const us2*>(&a), *reinterpret_cast<const us2*>(&b));
        return *reinterpret_cast<const uint32_t*>(&r);
      };
      const int sub = tid & 31;
      for (int g = 0; g < n_here; g += kBlock / 32) {           // (every lane runs every round: the butterfly needs whole wavefronts)
        const int dn = g + (tid >> 5);
        uint32_t v = 0u;
        if (dn < n_here && L.rowptr[dn + 1] - L.rowptr[dn] <= kGroupMaxDeg)
          for (int q = sub; q < (int)p.qd; q += 32) v = pk(v, *reinterpret_cast<const uint32_t*>(L.arg + (size_t)dn * p.D + 4 * q));
        v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));
        v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));
        v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true));
        v = pk(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true));
        v = pk(v, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));
        if (sub == 0 && dn < n_here) { rmax_b[dn] = v << 16; rmax_n[dn] = v & 0xFFFF0000u; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // the node's bound goes to each of its edges' rows - a plain store: the positions are this block's alone (a long segment's rows are the
    // list pass's; its bound here is 0 and nothing is stored) - and the dL/dU maximum to the node (merged: the list pass and the dV segment
    // sum write there too).  One lane per POSITION of the block's contiguous run (its node: a search of the staged row pointers): the stores
    // of a wavefront are consecutive words.
    for (int j = tid; j < p1 - p0; j += kBlock) {
      const int pos = p0 + j;
      int lo = 0, hi = n_here;                                 // rowptr[lo] <= pos < rowptr[hi]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.rowptr[mid] <= pos) lo = mid; else hi = mid; }
      const uint32_t v = rmax_b[lo];
      if (L.rowptr[lo + 1] - L.rowptr[lo] <= kGroupMaxDeg && v != 0u)
        p.gmsg_rmax[p.by_pos ? pos : (staged ? L.perm[pos - p0] : p.perm[pos])] = v;
    }
    if (tid < n_here && rmax_n[tid]) atomicMax(p.gu_rmax + n0 + tid, rmax_n[tid]);
  }
}

// ---- K6: CSR by key (stable), device side ---------------------------------------------------------------
// (also zeroes the long-segment list's count: three launches ahead of the kernel that appends to it.  It used to be a 4-byte
// hipMemsetAsync; inside a captured step - where the call becomes a memset NODE - the list kernels of bench.py's C2net graph read a
// count of 0x03030303 on the fifth replay, bytes that had lived at that address before the capture's pool took it over; [r4])
__global__ void csr_prepare_kernel(const int64_t* key, int64_t E, int32_t* key32, int32_t* iota, int32_t* long_count) {
  if (long_count && blockIdx.x == 0 && threadIdx.x == 0) { *long_count = 0; long_count[E / (kGroupMaxDeg + 1) + 2] = 0; }      // count and error flag
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

// nodes whose segment is longer than kGroupMaxDeg: out[0] = count (zeroed before the launch), out[1..] = node ids (any order)
__global__ void csr_zero_kernel(int32_t* a, int64_t n, int32_t* b) {
  if (b && blockIdx.x == 0 && threadIdx.x == 0) *b = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = 0;
}
__global__ void csr_long_nodes_kernel(const int32_t* rowptr, int64_t N, int32_t* out) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
    if (rowptr[n + 1] - rowptr[n] > kGroupMaxDeg) out[1 + atomicAdd(out, 1)] = (int32_t)n;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static int sort_bits(int64_t N) { int b = 1; while ((1LL << b) < N) ++b; return b; }

static int fill_codes(const uint8_t* aggr_host, int K, const uint8_t* scaler_host, int S, GrParams* p) {
  MMA_REQUIRE(K >= 1 && K <= MMA_MAX_K && S >= 1 && S <= 8, "K=%d (1..%d) / S=%d (1..8) unsupported", K, MMA_MAX_K, S);
  MMA_REQUIRE(aggr_host && scaler_host, "NULL aggregator / scaler code list (host memory)");
  for (int k = 0; k < K; ++k) {
    MMA_REQUIRE(aggr_host[k] <= GR_STD, "aggregator code %d unknown", (int)aggr_host[k]);
    p->aggr[k] = aggr_host[k];
  }
  for (int s = 0; s < S; ++s) {
    MMA_REQUIRE(scaler_host[s] <= SC_INVERSE_LINEAR, "scaler code %d unknown", (int)scaler_host[s]);
    p->scaler[s] = scaler_host[s];
  }
  p->K = K; p->S = S;
  for (int k = 0; k < K; ++k) p->aggr_pack |= (uint32_t)p->aggr[k] << (4 * k);
  for (int q = 0; q < S; ++q) p->scaler_pack |= (uint32_t)p->scaler[q] << (4 * q);
  for (int k = 0; k < K; ++k) {
    const int a = p->aggr[k];
    p->need_sum |= a == GR_SUM || a == GR_MEAN || a >= GR_VAR;
    p->need_mean |= a == GR_MEAN || a >= GR_VAR;
    p->need_sq |= a >= GR_VAR;
    p->need_min |= a == GR_MIN;
    p->need_max |= a == GR_MAX;
  }
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
  if ((p.amin8 || p.amax8 || p.mean || p.var) && (p.ldsave < Dp || p.ldsave % 4 != 0)) return false;
  if (!(al(p.amin8) && al(p.amax8) && al(p.amin_side) && al(p.amax_side) && al(p.mean) && al(p.var))) return false;
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

// the flat kernels pay when segments are short on average (molecule batches); they leave segments above kGroupMaxDeg
// edges to a wave-per-node pass.  Sets the flat geometry; false: run everything wave-per-node.
static uint32_t div_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ULL << 32) / d) + 1u; }   // see udiv()
constexpr int kFlatNodes = 16;       // nodes per workgroup: ZINC rows (95 quads) -> 1520 lanes of work, 5.9 sweeps of 256 threads
static bool gr_flat_mode(GrParams* p, int64_t E, int vec) {
  const int qd = (p->D + vec - 1) / vec;
  if (!(E <= 16 * (int64_t)p->N) || qd > 4095 || p->D > 65535) return false;
  p->nb = kFlatNodes; p->qd = (uint32_t)qd;
  p->qd_magic = div_magic((uint32_t)qd);      // exact floor(n / qd) for n, qd < 2^16
  p->f_magic = div_magic((uint32_t)p->F);
  return true;
}
// block kernels (fast path): vec4 operands, aggregators out of sum|mean|min|max only, short segments on average, and an LDS
// budget that leaves several workgroups per CU.  Fills the geometry; returns the NEEDS mask, or 0 when the path does not apply.
static int gr_block_mode(GrParams* p, int64_t E, bool v4, bool backward) {
  if (!v4 || E <= 0 || !(E <= 16 * (int64_t)p->N) || p->mean || p->var || p->D > 16380 || p->F < 4) return 0;
  if (backward && p->K * p->S > kBlkMaxKS) return 0;         // the backward loads all K*S gradient blocks of an item in one batch
  int needs = 0;
  for (int k = 0; k < p->K; ++k) {
    switch (p->aggr[k]) {
      case GR_SUM: case GR_MEAN: needs |= NEED_SUM; break;
      case GR_MIN: needs |= NEED_MIN; break;
      case GR_MAX: needs |= NEED_MAX; break;
      default: return 0;                                   // var / std: generic kernels
    }
  }
  // two instantiations: {min,max} and everything.  The LDS planes below are sized for the INSTANTIATION, not for the list
  needs = (needs & ~(NEED_MIN | NEED_MAX)) != 0 ? (NEED_SUM | NEED_MIN | NEED_MAX) : (NEED_MIN | NEED_MAX);
  const int n_coef = (needs & NEED_SUM ? 1 : 0) + (needs & NEED_MIN ? 1 : 0) + (needs & NEED_MAX ? 1 : 0);
  const uint32_t qd = (uint32_t)p->D / 4, f4 = (uint32_t)p->F / 4, skf4 = (uint32_t)(p->S * p->K) * f4, tskf4 = (uint32_t)p->T * skf4;
  const uint32_t head = kBlkHead;
  const int planes = backward ? n_coef : p->K;
  int nb_max = 16;
  { const char* e = getenv("MMA_GR_NB"); if (e && atoi(e) >= 2 && atoi(e) <= 16) nb_max = atoi(e); }       // plan sweep, read per call
  for (int nb = nb_max; nb >= 2; nb >>= 1) {
    const uint32_t agg = (uint32_t)nb * planes * p->D * 4, arg = 2u * nb * p->D;
    if (head + agg + arg > 40 * 1024 || (uint32_t)nb * tskf4 > 65535u || (uint32_t)nb * qd > 65535u) continue;
    p->nb = nb; p->qd = qd; p->qd_magic = div_magic(qd);
    p->f_magic = div_magic((uint32_t)p->F);
    p->tskf4 = tskf4; p->m_tskf4 = div_magic(tskf4);
    p->skf4 = skf4; p->m_skf4 = div_magic(skf4);
    p->f4 = f4; p->m_f4 = div_magic(f4);
    p->m_k = div_magic((uint32_t)p->K);
    p->lds_agg = head; p->lds_arg = head + agg; p->lds_bytes = head + agg + arg; p->n_coef = n_coef;
    return needs;
  }
  return 0;
}
static dim3 gr_block_grid(const GrParams& p) {
  const int64_t nblocks = ((int64_t)p.N + p.nb - 1) / p.nb;
  return dim3((unsigned)((nblocks + 7) / 8 * 8));            // one workgroup per node block; a multiple of 8 (XCD slots)
}

template <bool FUSED, bool HASZ, bool DROP>
static void launch_fwd_block(int needs, dim3 grid, unsigned lds, hipStream_t st, const GrParams& p) {
  if (needs == (NEED_MIN | NEED_MAX))
    hipLaunchKernelGGL((gr_fwd_block_kernel<FUSED, HASZ, DROP, NEED_MIN | NEED_MAX>), grid, dim3(kBlock), lds, st, p);
  else
    hipLaunchKernelGGL((gr_fwd_block_kernel<FUSED, HASZ, DROP, NEED_SUM | NEED_MIN | NEED_MAX>), grid, dim3(kBlock), lds, st, p);
}
template <bool FUSED, bool DROP>
static void launch_bwd_block(int needs, dim3 grid, unsigned lds, hipStream_t st, const GrParams& p) {
  if (needs == (NEED_MIN | NEED_MAX))
    hipLaunchKernelGGL((gr_bwd_block_kernel<FUSED, DROP, NEED_MIN | NEED_MAX>), grid, dim3(kBlock), lds, st, p);
  else
    hipLaunchKernelGGL((gr_bwd_block_kernel<FUSED, DROP, NEED_SUM | NEED_MIN | NEED_MAX>), grid, dim3(kBlock), lds, st, p);
}

static dim3 gr_flat_grid(int64_t N) {
  int64_t slots = ((N + kFlatNodes - 1) / kFlatNodes + 7) / 8 * 8;
  if (slots > 4 * kMaxGrid) slots = 4 * kMaxGrid;                    // a multiple of 8: grid-stride keeps a slot's XCD
  if (slots < 8) slots = 8;
  return dim3((unsigned)slots);
}

}  // namespace mma

using namespace mma;

extern "C" int64_t mma_csr_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0 || E >= (1LL << 31) || N >= (1LL << 31)) return -1;
  size_t temp = 0;
  // keys are node ids: sorted as UNSIGNED (rocPRIM's signed-key mask is (1 << 31) - 1 in int arithmetic when 31 bits are significant)
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, (int)E, 0, sort_bits(N));
  return (int64_t)(3 * align256((size_t)E * 4) + align256(temp) + 256);
}

// [count | ids (capacity E / 65 + 1) | error flag]: the flag (ABI 34) is zeroed by mma_build_csr and set by K3 / K4's list passes when the
// count word exceeds the capacity (1) or an id is not a node (2) - the cases they refuse to believe
extern "C" int64_t mma_gr_long_nodes_len(int64_t E) { return E < 0 ? -1 : E / (kGroupMaxDeg + 1) + 3; }

extern "C" int mma_build_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm,
                             int32_t* other_sorted, int32_t* long_nodes, void* workspace, int64_t workspace_bytes, void* stream) {
  MMA_REQUIRE(E >= 0 && N >= 0 && E < (1LL << 31) && N < (1LL << 31), "E=%lld N=%lld out of int32 range", (long long)E, (long long)N);
  MMA_REQUIRE(rowptr != nullptr, "NULL rowptr");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (E == 0) {                                                    // every segment empty, an empty list
    hipLaunchKernelGGL(csr_zero_kernel, dim3((unsigned)min((int64_t)kMaxGrid, (N + 1 + kBlock) / kBlock)), dim3(kBlock), 0, st, rowptr, N + 1,
                       long_nodes);
    return check_launch("csr_zero_kernel");
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
  const uint32_t* ukey = reinterpret_cast<const uint32_t*>(key32);
  uint32_t* uskey = reinterpret_cast<uint32_t*>(skey);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, ukey, uskey, iota, perm, (int)E, 0, sort_bits(N));
  const int blocks = (int)min((int64_t)kMaxGrid, (E + kBlock) / kBlock);
  hipLaunchKernelGGL(csr_prepare_kernel, dim3(blocks), dim3(kBlock), 0, st, key, E, key32, iota, long_nodes);   // + list count = 0
  const hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, ukey, uskey, iota, perm, (int)E, 0, sort_bits(N), st);
  if (e != hipSuccess) return fail(100 + (int)e, "radix sort failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3(blocks), dim3(kBlock), 0, st, skey, E, N, rowptr);
  if (other && other_sorted) hipLaunchKernelGGL(csr_gather_kernel, dim3(blocks), dim3(kBlock), 0, st, other, perm, E, other_sorted);
  if (long_nodes && N > 0)
    hipLaunchKernelGGL(csr_long_nodes_kernel, dim3((unsigned)min((int64_t)kMaxGrid, (N + kBlock) / kBlock)), dim3(kBlock), 0, st, rowptr, N,
                       long_nodes);
  return check_launch("csr build");
}

extern "C" int64_t mma_gr_arg_side_rows(int64_t E) { return E < 0 ? -1 : (E >> 8) + 2; }

static int gr_fill_common(GrParams& p, const int32_t* rowptr, const int32_t* src, const int32_t* perm, const float* U, const float* V,
                          int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos, const float* inputs, int64_t ldi,
                          uint8_t* amin8, uint8_t* amax8, int32_t* amin_side, int32_t* amax_side, float* mean, float* var,
                          int64_t ldsave, int64_t N, int64_t E, int32_t T, int32_t F, float avg_log, float avg_lin, int32_t drop_mode,
                          uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev) {
  MMA_REQUIRE(drop_mode == MMA_DROP_NONE || drop_mode == MMA_DROP_HASH, "GR dropout: NONE or HASH");
  MMA_REQUIRE(drop_mode == MMA_DROP_NONE || drop_thr < 65536, "drop_thr out of range (0..65535: P(drop) = thr / 65536)");
  MMA_REQUIRE((amin8 == nullptr) == (amin_side == nullptr) && (amax8 == nullptr) == (amax_side == nullptr),
              "arg8 (N,ldsave) bytes and arg_side (mma_gr_arg_side_rows(E), ldsave) int32 come in pairs");
  p.rowptr = rowptr; p.src = src; p.perm = perm; p.U = U; p.V = V; p.lduv = lduv; p.Z = Z; p.ldz = ldz; p.by_pos = by_pos ? 1 : 0;
  p.inputs = inputs; p.ldi = ldi; p.amin8 = amin8; p.amax8 = amax8; p.amin_side = amin_side; p.amax_side = amax_side;
  p.mean = mean; p.var = var; p.ldsave = ldsave;
  p.N = (int)N; p.D = T * F; p.T = T; p.F = F; p.avg_log = avg_log; p.avg_lin = avg_lin;
  drop_set_threshold(&p.drop, (drop_mode == MMA_DROP_HASH && drop_thr > 0 && !inputs) ? MMA_DROP_HASH : MMA_DROP_NONE, drop_thr);
  p.drop.seed_lo = (uint32_t)seed; p.drop.seed_hi = (uint32_t)(seed >> 32); p.drop.seed_dev = seed_dev; p.drop.keep = nullptr;
  p.drop.E = E; p.drop.edge_base = 0;
  return 0;
}

extern "C" int mma_gr_fused_fwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos, const uint8_t* z_index,
    const float* inputs, int64_t ldi,
    float* out, uint8_t* amin8, uint8_t* amax8, int32_t* amin_side, int32_t* amax_side, float* mean, float* var, int64_t ldsave,
    const int32_t* long_nodes,
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) - (1 << 20) && E < (1LL << 31) && T >= 1 && F >= 1 && (int64_t)T * F < (1 << 24),
              "N=%lld E=%lld T=%d F=%d unsupported", (long long)N, (long long)E, T, F);
  GrParams p{};
  if (int rc = fill_codes(aggr_host, K, scaler_host, S, &p)) return rc;
  p.long_nodes = long_nodes;
  p.long_cap = (int)min((int64_t)0x7fffffff, E / (kGroupMaxDeg + 1) + 1);
  if (N == 0) return 0;
  const int D = T * F;
  MMA_REQUIRE(rowptr && out, "NULL argument");
  MMA_REQUIRE(E == 0 || (src && perm), "NULL CSR arrays");
  if (E == 0 && !inputs && !(U && V)) {      // given-messages form with no message: an (0, D) tensor has no address.  No row is ever
    inputs = out; ldi = D;                   // read; the pointer only selects the form (every target is empty: 0, or sqrt(eps) for std)
  }
  MMA_REQUIRE((inputs != nullptr) != (U != nullptr && V != nullptr), "give either `inputs` or U and V");
  MMA_REQUIRE(inputs ? ldi >= D : (lduv >= D && (!Z || ldz >= D)), "row pitch too small");
  MMA_REQUIRE(!(amin8 || amax8 || mean || var) || ldsave >= D, "ldsave=%lld < T*F", (long long)ldsave);
  if (int rc = gr_fill_common(p, rowptr, src, perm, U, V, lduv, Z, ldz, by_pos, inputs, ldi, amin8, amax8, amin_side, amax_side, mean,
                              var, ldsave, N, E, T, F, avg_log, avg_lin, drop_mode, drop_thr, seed, seed_dev)) return rc;
  MMA_REQUIRE(!z_index || Z, "z_index without a Z table");
  p.zidx = z_index;
  p.out = out;
  const bool v4 = gr_vec4(p);
  const dim3 grid = gr_grid(N, D, v4 ? 4 : 1, &p.lpr_log);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool drop = p.drop.mode != MMA_DROP_NONE;
  if (E == 0) {                                                   // no edge: the plain wave-per-node kernel writes the empty results
  } else if (const int needs = gr_block_mode(&p, E, v4, false)) { // molecule-batch shape: two-phase block kernel
    const dim3 bg = gr_block_grid(p);
    if (inputs) launch_fwd_block<false, false, false>(needs, bg, p.lds_bytes, st, p);
    else if (Z) { if (drop) launch_fwd_block<true, true, true>(needs, bg, p.lds_bytes, st, p); else launch_fwd_block<true, true, false>(needs, bg, p.lds_bytes, st, p); }
    else { if (drop) launch_fwd_block<true, false, true>(needs, bg, p.lds_bytes, st, p); else launch_fwd_block<true, false, false>(needs, bg, p.lds_bytes, st, p); }
    if (int rc = check_launch("gr_fwd_block_kernel")) return rc;
    p.wave_min_deg = kGroupMaxDeg + 1;
  } else if (gr_flat_mode(&p, E, v4 ? 4 : 1)) {      // short segments, generic form: flat kernel
    const dim3 fg = gr_flat_grid(N);
    if (v4) hipLaunchKernelGGL((gr_fwd_flat_kernel<4>), fg, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((gr_fwd_flat_kernel<1>), fg, dim3(kBlock), 0, st, p);
    if (int rc = check_launch("gr_fwd_flat_kernel")) return rc;
    p.wave_min_deg = kGroupMaxDeg + 1;
  }
  // the wave-per-node pass: everything, or - behind a block / flat kernel - the segments above kGroupMaxDeg edges only, taken
  // from the list mma_build_csr made when the caller has one (an empty list costs one trivial launch instead of a scan of N)
  if (p.wave_min_deg > 0 && long_nodes) {
    const dim3 lg(128, grid.y);
    if (v4) hipLaunchKernelGGL((gr_fwd_list_kernel<4>), lg, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((gr_fwd_list_kernel<1>), lg, dim3(kBlock), 0, st, p);
    return check_launch("gr_fwd_list_kernel");
  }
  if (v4) hipLaunchKernelGGL((gr_fwd_kernel<4>), grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL((gr_fwd_kernel<1>), grid, dim3(kBlock), 0, st, p);
  return check_launch("gr_fwd_kernel");
}

extern "C" int mma_gr_fused_bwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos, const uint8_t* z_index,
    const float* inputs, int64_t ldi,
    const float* gout, const uint8_t* amin8, const uint8_t* amax8, const int32_t* amin_side, const int32_t* amax_side,
    const float* mean, const float* var, int64_t ldsave, const int32_t* long_nodes, float* gmsg, int64_t ldg, float* gU, int64_t ldgu,
    float* gmsg_row_max, float* gu_row_max,
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) - (1 << 20) && E < (1LL << 31) && T >= 1 && F >= 1 && (int64_t)T * F < (1 << 24),
              "N=%lld E=%lld T=%d F=%d unsupported", (long long)N, (long long)E, T, F);
  GrParams p{};
  if (int rc = fill_codes(aggr_host, K, scaler_host, S, &p)) return rc;
  p.long_nodes = long_nodes;
  p.long_cap = (int)min((int64_t)0x7fffffff, E / (kGroupMaxDeg + 1) + 1);
  MMA_REQUIRE((gmsg_row_max == nullptr) == (gu_row_max == nullptr) && (!gu_row_max || gU), "the row maxima come as a pair, with gU");
  if (N == 0 || E == 0) return 0;
  const int D = T * F;
  MMA_REQUIRE(rowptr && src && perm && gout && gmsg && ldg >= D, "NULL argument or pitch too small");
  p.gmsg_rmax = reinterpret_cast<uint32_t*>(gmsg_row_max); p.gu_rmax = reinterpret_cast<uint32_t*>(gu_row_max);
  bool need_min = false, need_max = false, need_stats = false;
  for (int k = 0; k < K; ++k) {
    need_min |= p.aggr[k] == GR_MIN; need_max |= p.aggr[k] == GR_MAX; need_stats |= p.aggr[k] >= GR_VAR;
  }
  MMA_REQUIRE((!need_min || amin8) && (!need_max || amax8) && (!need_stats || (mean && var)), "saved forward state missing");
  MMA_REQUIRE(!need_stats || (inputs != nullptr) != (U != nullptr && V != nullptr), "var/std backward needs the messages");
  MMA_REQUIRE(!gU || ldgu >= D, "ldgu=%lld < T*F", (long long)ldgu);
  MMA_REQUIRE(!(need_min || need_max || need_stats) || ldsave >= D, "ldsave=%lld < T*F", (long long)ldsave);
  if (int rc = gr_fill_common(p, rowptr, src, perm, U, V, lduv, Z, ldz, by_pos, inputs, ldi,
                              const_cast<uint8_t*>(need_min ? amin8 : nullptr), const_cast<uint8_t*>(need_max ? amax8 : nullptr),
                              const_cast<int32_t*>(need_min ? amin_side : nullptr), const_cast<int32_t*>(need_max ? amax_side : nullptr),
                              const_cast<float*>(mean), const_cast<float*>(var), ldsave, N, E, T, F, avg_log, avg_lin, drop_mode,
                              drop_thr, seed, seed_dev)) return rc;
  MMA_REQUIRE(!z_index || Z, "z_index without a Z table");
  p.zidx = z_index;
  p.gout = gout; p.gmsg = gmsg; p.ldg = ldg; p.gU = gU; p.ldgu = ldgu;
  const bool v4 = gr_vec4(p) && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(gmsg) & 15) == 0 &&
                  (!gU || (ldgu % 4 == 0 && (reinterpret_cast<uintptr_t>(gU) & 15) == 0));
  const dim3 grid = gr_grid(N, D, v4 ? 4 : 1, &p.lpr_log);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool drop = p.drop.mode != MMA_DROP_NONE;
  if (const int needs = need_stats ? 0 : gr_block_mode(&p, E, v4, true)) {
    const dim3 bg = gr_block_grid(p);
    if (inputs) launch_bwd_block<false, false>(needs, bg, p.lds_bytes, st, p);
    else if (drop) launch_bwd_block<true, true>(needs, bg, p.lds_bytes, st, p);
    else launch_bwd_block<true, false>(needs, bg, p.lds_bytes, st, p);
    if (int rc = check_launch("gr_bwd_block_kernel")) return rc;
    p.wave_min_deg = kGroupMaxDeg + 1;
  } else if (gr_flat_mode(&p, E, v4 ? 4 : 1)) {
    const dim3 fg = gr_flat_grid(N);
    if (v4) hipLaunchKernelGGL((gr_bwd_flat_kernel<4>), fg, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((gr_bwd_flat_kernel<1>), fg, dim3(kBlock), 0, st, p);
    if (int rc = check_launch("gr_bwd_flat_kernel")) return rc;
    p.wave_min_deg = kGroupMaxDeg + 1;
  }
  if (p.wave_min_deg > 0 && long_nodes) {
    const dim3 lg(128, grid.y);
    if (v4) hipLaunchKernelGGL((gr_bwd_list_kernel<4>), lg, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((gr_bwd_list_kernel<1>), lg, dim3(kBlock), 0, st, p);
    return check_launch("gr_bwd_list_kernel");
  }
  if (v4) hipLaunchKernelGGL((gr_bwd_kernel<4>), grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL((gr_bwd_kernel<1>), grid, dim3(kBlock), 0, st, p);
  return check_launch("gr_bwd_kernel");
}
