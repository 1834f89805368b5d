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
  int N, D, T, F, K, S, lpr_log;
  int by_pos;                                        // 1: Z and gmsg rows are indexed by the target-sorted POSITION p, 0: by perm[p]
  int wave_min_deg;                                  // wave-per-node pass: skip segments shorter than this
  int nb; uint32_t qd, qd_magic, f_magic;            // flat kernels: nodes per workgroup, lanes (VEC columns each) per row, 2^32/d magics
  bool need_sum, need_sq, need_min, need_max, need_mean;   // which running reductions the aggregator list uses
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
    const Vec<VEC> z = ldv<VEC>(p.Z + (size_t)(p.by_pos ? pos : e) * p.ldz + cc);
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
  l.t = (int)__umulhi((uint32_t)l.c, p.f_magic); l.f = l.c - l.t * p.F;
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
      const int dn = (int)__umulhi((uint32_t)it, p.qd_magic);   // it / qd
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
    if (l.sub == 0 && l.valid) stv<VEC>(p.gU + (size_t)node * p.ldgu + l.c, su);
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
      const int dn = (int)__umulhi((uint32_t)it, p.qd_magic);
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
      if (p.gU) stv<VEC>(p.gU + (size_t)node * p.ldgu + l.c, su);
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
constexpr int kFlatNodes = 16;       // nodes per workgroup: ZINC rows (95 quads) -> 1520 lanes of work, 5.9 sweeps of 256 threads
static bool gr_flat_mode(GrParams* p, int64_t E, int vec) {
  const int qd = (p->D + vec - 1) / vec;
  if (!(E <= 16 * (int64_t)p->N) || qd > 4095 || p->D > 65535) return false;
  p->nb = kFlatNodes; p->qd = (uint32_t)qd;
  p->qd_magic = (uint32_t)((1ULL << 32) / (uint32_t)qd) + 1u;      // exact floor(n / qd) for n, qd < 2^16
  p->f_magic = (uint32_t)((1ULL << 32) / (uint32_t)p->F) + 1u;
  return true;
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

extern "C" int64_t mma_gr_arg_side_rows(int64_t E) { return E < 0 ? -1 : (E >> 8) + 2; }

static int gr_fill_common(GrParams& p, const int32_t* rowptr, const int32_t* src, const int32_t* perm, const float* U, const float* V,
                          int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos, const float* inputs, int64_t ldi,
                          uint8_t* amin8, uint8_t* amax8, int32_t* amin_side, int32_t* amax_side, float* mean, float* var,
                          int64_t ldsave, int64_t N, int64_t E, int32_t T, int32_t F, float avg_log, float avg_lin, int32_t drop_mode,
                          uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev) {
  MMA_REQUIRE(drop_mode == MMA_DROP_NONE || drop_mode == MMA_DROP_HASH, "GR dropout: NONE or HASH");
  MMA_REQUIRE(drop_mode == MMA_DROP_NONE || drop_thr < 256, "drop_thr out of range");
  MMA_REQUIRE((amin8 == nullptr) == (amin_side == nullptr) && (amax8 == nullptr) == (amax_side == nullptr),
              "arg8 (N,ldsave) bytes and arg_side (mma_gr_arg_side_rows(E), ldsave) int32 come in pairs");
  p.rowptr = rowptr; p.src = src; p.perm = perm; p.U = U; p.V = V; p.lduv = lduv; p.Z = Z; p.ldz = ldz; p.by_pos = by_pos ? 1 : 0;
  p.inputs = inputs; p.ldi = ldi; p.amin8 = amin8; p.amax8 = amax8; p.amin_side = amin_side; p.amax_side = amax_side;
  p.mean = mean; p.var = var; p.ldsave = ldsave;
  p.N = (int)N; p.D = T * F; p.T = T; p.F = F; p.avg_log = avg_log; p.avg_lin = avg_lin;
  p.drop.mode = (drop_mode == MMA_DROP_HASH && drop_thr > 0 && !inputs) ? MMA_DROP_HASH : MMA_DROP_NONE;
  p.drop.thr = drop_thr; p.drop.scale = 256.f / (256.f - (float)drop_thr);
  p.drop.seed_lo = (uint32_t)seed; p.drop.seed_hi = (uint32_t)(seed >> 32); p.drop.seed_dev = seed_dev; p.drop.keep = nullptr;
  p.drop.E = E; p.drop.edge_base = 0;
  return 0;
}

extern "C" int mma_gr_fused_fwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos, const float* inputs, int64_t ldi,
    float* out, uint8_t* amin8, uint8_t* amax8, int32_t* amin_side, int32_t* amax_side, float* mean, float* var, int64_t ldsave,
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) - (1 << 20) && E < (1LL << 31) && T >= 1 && F >= 1,
              "N=%lld E=%lld T=%d F=%d unsupported", (long long)N, (long long)E, T, F);
  GrParams p{};
  if (int rc = fill_codes(aggr_host, K, scaler_host, S, &p)) return rc;
  if (N == 0) return 0;
  const int D = T * F;
  MMA_REQUIRE(rowptr && out, "NULL argument");
  MMA_REQUIRE(E == 0 || (src && perm), "NULL CSR arrays");
  MMA_REQUIRE((inputs != nullptr) != (U != nullptr && V != nullptr), "give either `inputs` or U and V");
  MMA_REQUIRE(inputs ? ldi >= D : (lduv >= D && (!Z || ldz >= D)), "row pitch too small");
  MMA_REQUIRE(!(amin8 || amax8 || mean || var) || ldsave >= D, "ldsave=%lld < T*F", (long long)ldsave);
  if (int rc = gr_fill_common(p, rowptr, src, perm, U, V, lduv, Z, ldz, by_pos, inputs, ldi, amin8, amax8, amin_side, amax_side, mean,
                              var, ldsave, N, E, T, F, avg_log, avg_lin, drop_mode, drop_thr, seed, seed_dev)) return rc;
  p.out = out;
  const bool v4 = gr_vec4(p);
  const dim3 grid = gr_grid(N, D, v4 ? 4 : 1, &p.lpr_log);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gr_flat_mode(&p, E, v4 ? 4 : 1)) {      // short segments: flat kernel, then the wave-per-node pass for the few long ones
    const dim3 fg = gr_flat_grid(N);
    if (v4) hipLaunchKernelGGL((gr_fwd_flat_kernel<4>), fg, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((gr_fwd_flat_kernel<1>), fg, dim3(kBlock), 0, st, p);
    if (int rc = check_launch("gr_fwd_flat_kernel")) return rc;
    p.wave_min_deg = kGroupMaxDeg + 1;
  }
  if (v4) hipLaunchKernelGGL((gr_fwd_kernel<4>), grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL((gr_fwd_kernel<1>), grid, dim3(kBlock), 0, st, p);
  return check_launch("gr_fwd_kernel");
}

extern "C" int mma_gr_fused_bwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos, const float* inputs, int64_t ldi,
    const float* gout, const uint8_t* amin8, const uint8_t* amax8, const int32_t* amin_side, const int32_t* amax_side,
    const float* mean, const float* var, int64_t ldsave, float* gmsg, int64_t ldg, float* gU, int64_t ldgu,
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) - (1 << 20) && E < (1LL << 31) && T >= 1 && F >= 1,
              "N=%lld E=%lld T=%d F=%d unsupported", (long long)N, (long long)E, T, F);
  GrParams p{};
  if (int rc = fill_codes(aggr_host, K, scaler_host, S, &p)) return rc;
  if (N == 0 || E == 0) return 0;
  const int D = T * F;
  MMA_REQUIRE(rowptr && src && perm && gout && gmsg && ldg >= D, "NULL argument or pitch too small");
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
  p.gout = gout; p.gmsg = gmsg; p.ldg = ldg; p.gU = gU; p.ldgu = ldgu;
  const bool v4 = gr_vec4(p) && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(gmsg) & 15) == 0 &&
                  (!gU || (ldgu % 4 == 0 && (reinterpret_cast<uintptr_t>(gU) & 15) == 0));
  const dim3 grid = gr_grid(N, D, v4 ? 4 : 1, &p.lpr_log);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gr_flat_mode(&p, E, v4 ? 4 : 1)) {
    const dim3 fg = gr_flat_grid(N);
    if (v4) hipLaunchKernelGGL((gr_bwd_flat_kernel<4>), fg, dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((gr_bwd_flat_kernel<1>), fg, dim3(kBlock), 0, st, p);
    if (int rc = check_launch("gr_bwd_flat_kernel")) return rc;
    p.wave_min_deg = kGroupMaxDeg + 1;
  }
  if (v4) hipLaunchKernelGGL((gr_bwd_kernel<4>), grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL((gr_bwd_kernel<1>), grid, dim3(kBlock), 0, st, p);
  return check_launch("gr_bwd_kernel");
}
