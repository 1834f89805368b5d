// K1 / K2a / K2b: fused masked-message + multi-aggregator segmented reduce for the node-classification
// form of MMA (reference node_classification/layers.py:201-728), hand-written for gfx950.
//
// Mapping (wave64): one wavefront owns one work item (= one target node's CSR segment, or one chunk of
// a hub's segment).  The 64 lanes are split into EPG = 64/LPR sub-rows of LPR lanes; a sub-row holds one
// neighbour row at a time, each lane VEC (=4, one dwordx4) consecutive features, so one wave-instruction
// gathers EPG whole rows with fully coalesced 16-byte accesses.  The node's own row (x_i, P_k[i]) stays in
// registers for the whole segment; all K masks share the single gather of x_j.  Edge indices are read
// 64 at a time (one coalesced load) and handed to the sub-rows with ds_bpermute.  Two edge steps are
// kept in flight per lane.  Sub-row partial sums meet in a butterfly at the end of the segment; there
// are no atomics anywhere, so results are bitwise reproducible.
#include <cstdlib>
#include "common.h"

#ifndef NC_FWD_UNROLL
#define NC_FWD_UNROLL(K) ((K) >= 3 ? 1 : 2)   // measured (C4, K=4): 2 -> 1 takes 183 -> ~160 VGPRs, 2 -> 3 waves/SIMD, 5.4 -> 4.8 ms
#endif
#ifndef NC_BWD_UNROLL
#define NC_BWD_UNROLL(K) ((K) >= 3 ? 1 : 2)
#endif
#ifndef MMA_MIN_WAVES
#define MMA_MIN_WAVES 2   // register cap of the fused kernels in waves per SIMD (3 = 168 VGPRs spills and measured 5-25 % slower)
#endif
namespace mma {

struct NcFwdParams {
  const float* x; int64_t ldx;
  const float* P; int64_t ldp; const float* Q; int64_t ldq;
  const int32_t* rowptr; const int32_t* col;
  const int4* items; int64_t n_items;
  float* partial; int64_t pstride;   // floats per slot = 2*K_total*H
  float* m; int64_t m_kstride;       // (K,N,H) per-mask outputs, may be NULL
  float* msum; int64_t ldms;         // (N,H) sum over the K masks, may be NULL
  float* T; uint8_t* sel; int64_t ldt;
  // shared-gradient backward (round 3): K1 itself leaves the packed code row K2b gathers per edge,
  // crow[i] = [ 1/d_i, 0, 0, 0 | one byte per element for every max/min/softmax-type mask = 2 * dm/ds (0, 1, 2; 255: NaN) ]
  float* crow; int64_t ldc; uint32_t sel_slots;      // 4 bits per mask: its code slot in the row, 0xF = none (sum / mean)
  int H, HQ, K_total, k_base, lpr_log;
  uint32_t kinds, acts;              // 4 bits / 1 bit per mask, indexed by absolute k
  DropParams drop;
  // one-launch form (nc_fwd_small_kernel): the hub list and the ticket counter of the chunk partials
  const int4* hubs; int64_t n_hubs; unsigned* sync; unsigned n_slots;
};

// row * pitch as ONE v_mad_u64_u32: rows and pitches are < 2^31 (checked on the host), so the 64-bit product needs neither the
// sign extension nor the two extra quarter-rate v_mul_lo_u32 the int * int64 form compiles to (3 multiplies per gathered row)
__device__ __forceinline__ size_t row_off(int row, int64_t ld) { return (size_t)((uint64_t)(uint32_t)row * (uint64_t)(uint32_t)ld); }

__device__ __forceinline__ int kind_of(uint32_t kinds, int k) { return (kinds >> (4 * k)) & 0xF; }
__device__ __forceinline__ uint32_t sel_slot_of(uint32_t slots, int k) { return (slots >> (4 * k)) & 0xFu; }

// combine + selection code for one element (layers.py:221,326-329,452,562,676-682,716-720)
__device__ __forceinline__ float nc_combine(int kind, float xi, float s, float deg, uint32_t& code) {
  code = 1;
  switch (kind) {
    case MMA_KIND_SUM: return xi + s;
    case MMA_KIND_MEAN: return (xi + s) / deg;   // deg clamped to >= 1 by the caller (SURVEY Q12 extension)
    case MMA_KIND_MAX: code = s > xi ? 1u : (s == xi ? 2u : 0u); return s > xi ? s : xi;
    case MMA_KIND_MIN: code = s < xi ? 1u : (s == xi ? 2u : 0u); return s < xi ? s : xi;
    default: {  // softmax over a singleton dimension: (e / e) * s, NaN once exp over/underflows
      const float e = expf(kind == MMA_KIND_SOFTMAX ? s : -s);
      const float r = (e / e) * s;
      // code 3 = the reference's GRADIENT is NaN.  That is the case when the value is, and also in the band below it where e is
      // so small that 1/e overflows while e/e is still 1 (|s| in ~87.3..104): autograd's division backward forms
      // g s / e - g s ((e / e) / e) = inf - inf there (found by the generated cases; fp64 and exact arithmetic give g).
      code = (r != r || (1.0f / e) > 3.402823466e38f) ? 3u : 1u;
      return r;
    }
  }
}

template <int VEC, bool SAVE>
__device__ __forceinline__ Vec<VEC> nc_fwd_write(const NcFwdParams& p, int node, int k_abs, int c, const Vec<VEC>& xi,
                                                 const Vec<VEC>& s, const Vec<VEC>& t, float deg) {
  Vec<VEC> mo;
  uint32_t codes = 0;
  const int kind = kind_of(p.kinds, k_abs);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    uint32_t code;
    mo.v[i] = nc_combine(kind, xi.v[i], s.v[i], deg, code);
    codes |= code << (8 * i);
  }
  if (p.m) stv<VEC>(p.m + (size_t)k_abs * p.m_kstride + (size_t)node * p.H + c, mo);
  if (SAVE) {
    const size_t o = (size_t)node * p.ldt + (size_t)k_abs * p.H + c;
    stv_nt<VEC>(p.T + o, t);
    if (p.sel) stb_nt<VEC>(p.sel + o, codes);
    if (p.crow && sel_slot_of(p.sel_slots, k_abs) != 0xFu) {
      // code -> 2 * dm/ds: 1 (s selected) -> 2, 2 (tie) -> 1, 0 (x_i selected) -> 0, 3 (NaN gradient) -> 255
      uint32_t tf = 0;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const uint32_t code = (codes >> (8 * i)) & 0xFFu;
        tf |= (code == 1u ? 2u : (code == 2u ? 1u : (code == 3u ? 255u : 0u))) << (8 * i);
      }
      stb<VEC>(reinterpret_cast<uint8_t*>(p.crow + (size_t)node * p.ldc + 4 + (size_t)sel_slot_of(p.sel_slots, k_abs) * p.HQ) + c, tf);   // re-read per edge by K2b: plain store
    }
  }
  return mo;
}

// sum over the masks of one launch (K-slices after the first accumulate onto the stored row)
template <int VEC>
__device__ __forceinline__ void nc_msum_store(const NcFwdParams& p, int node, int c, Vec<VEC> ms, bool accumulate) {
  float* o = p.msum + (size_t)node * p.ldms + c;
  if (accumulate) {
    const Vec<VEC> prev = ldv<VEC>(o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) ms.v[i] += prev.v[i];
  }
  if (accumulate) stv<VEC>(o, ms); else stv_nt<VEC>(o, ms);
}

// MULTI = false: one item per wavefront (EPG = 64/LPR neighbour rows per step) - long segments.
// MULTI = true : one item per group of G = LPR lanes, 64/G items per wavefront - short segments, where the per-item
//                latency chain (item -> indices -> rows -> store) dominates and more items in flight is what pays.
// DM: dropout mode as a TEMPLATE parameter (MMA_DROP_NONE / HASH / EXPLICIT) - as a run-time field every mask of every edge step
// carried a scalar branch between the hash and the explicit-mask code, which cut the step into basic blocks
template <int VEC, bool SAVE>
__device__ __forceinline__ void nc_fwd_finalize_body(const NcFwdParams& p, const int4* hubs, int64_t n_hubs, const int64_t first, const int64_t step);

// ONE (the one-launch form, nc_fwd_small_kernel): a wavefront that has written a hub chunk's partial takes a ticket on p.sync after a
// release fence at device scope (the L2s of the 8 XCDs are not coherent with each other: the fence writes the partial back), and the
// wavefront that draws the LAST of the n_slots tickets sums the partials of every hub (slot order: same bits as the finalize launch)
// and leaves the counter at zero.  The chunk items head the longest-first list, so this happens while the short items still run.
template <int K, int VEC, bool SAVE, int DM, bool MULTI, bool ONE = false>
__device__ __forceinline__ void nc_fwd_body(const NcFwdParams& p, const int bx, const int nbx) {
  const DropParams dp = (DM == MMA_DROP_HASH || DM == MMA_DROP_HASH16) ? drop_resolve(p.drop) : p.drop;
  uint32_t mult[K], mult2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { mult[k] = drop_mask_mult(p.k_base + k); mult2[k] = drop_mask_mult2(p.k_base + k); }
  // edge steps in flight per lane: 2 (2*(K+1) row loads before the first use); 1 for K = 8, where two would need all
  // 256 VGPRs and leave a single wave per SIMD
  constexpr int U = NC_FWD_UNROLL(K);
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log;
  const int G = MULTI ? lpr : kWave;            // lanes per item
  const int epg = G >> p.lpr_log;               // neighbour rows a group gathers per step
  const int gpw = kWave / G;                    // items per wavefront
  const int grp = MULTI ? lane / G : 0;
  const int gl = lane & (G - 1);
  const int gbase = grp * G;
  const int sub = gl >> p.lpr_log;
  const int c = ((int)blockIdx.y * lpr + (lane & (lpr - 1))) * VEC;
  const bool fvalid = c < p.H;
  const int cc = fvalid ? c : 0;  // masked lanes read column 0 (valid memory), results are discarded
  const int waves_per_block = kBlock / kWave;
  const int64_t stride = (int64_t)nbx * waves_per_block;
  const int64_t n_witems = (p.n_items + gpw - 1) / gpw;

  for (int64_t it0 = (int64_t)bx * waves_per_block + (threadIdx.x >> 6); it0 < n_witems; it0 += stride) {
    int node, ebeg, eend, slot;
    bool ivalid = true;
    if (MULTI) {
      const int64_t idx = it0 * gpw + grp;
      ivalid = idx < p.n_items;
      const int4 item = p.items[ivalid ? idx : 0];
      node = item.x; ebeg = item.y; eend = ivalid ? item.z : item.y; slot = item.w;
    } else {
      const int4 item = p.items[__builtin_amdgcn_readfirstlane((int)it0)];
      node = __builtin_amdgcn_readfirstlane(item.x);
      ebeg = __builtin_amdgcn_readfirstlane(item.y);
      eend = __builtin_amdgcn_readfirstlane(item.z);
      slot = __builtin_amdgcn_readfirstlane(item.w);
    }
    const int len = eend - ebeg;
    int maxlen = len;
    if (MULTI) {
      for (int off = G; off < kWave; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, kWave));
      maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    }

    const Vec<VEC> xi = ldv<VEC>(p.x + row_off(node, p.ldx) + cc);
    Vec<VEC> pk[K], acc[K], tac[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pk[k] = ldv_nt<VEC>(p.P + row_off(node, p.ldp) + (size_t)(p.k_base + k) * p.H + cc);
      acc[k] = vzero<VEC>();
      tac[k] = vzero<VEC>();
    }

    for (int base = 0; base < maxlen; base += G) {
      const int cnt = min(G, max(len - base, 0));      // edges of MY item in this index chunk
      const int ucnt = min(G, maxlen - base);          // wave-uniform trip bound
      const int myj = (gl < cnt) ? p.col[ebeg + base + gl] : 0;
      // U edge steps (U*EPG rows per group) in flight per iteration
      for (int t0 = 0; t0 < ucnt; t0 += U * epg) {
        int tt[U]; bool ev[U]; Vec<VEC> xj[U]; Vec<VEC> qv[U][K];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          tt[u] = t0 + u * epg + sub;
          ev[u] = tt[u] < cnt;
          const int j = __shfl(myj, gbase + (tt[u] & (G - 1)), kWave);
          const int jj = ev[u] ? j : node;   // inactive sub-rows re-read the own row (cached) and are zeroed by SELECTS below
          xj[u] = ldv<VEC>(p.x + row_off(jj, p.ldx) + cc);
          const float* qrow = p.Q + row_off(jj, p.ldq) + cc;
#pragma unroll
          for (int k = 0; k < K; ++k)
            qv[u][k] = ldv<VEC>(qrow + (size_t)(p.k_base + k) * p.H);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // An inactive sub-row (past the end of its item) contributes exactly 0: its neighbour row is replaced by 0 with a
          // SELECT (a multiplication by 0 would turn an inf of the row it re-read into NaN).  Its logits need no such
          // care here: it re-read this item's OWN rows (x_i, Q[i]), so a non-finite logit means node i's own inputs are
          // non-finite and its outputs are NaN/inf in the reference too - nothing foreign can leak in.  (A select on z per
          // mask and element was measured at +5.7 % of the kernel: 4.62 -> 4.89 ms at C4.)
#pragma unroll
          for (int i = 0; i < VEC; ++i) xj[u].v[i] = ev[u] ? xj[u].v[i] : 0.f;
          // inactive: edge 0 (any valid position; EXPLICIT mode reads keep[(k*E + e)*H + c], and ebeg may equal E)
          const uint32_t eu = (uint32_t)(ev[u] ? ebeg + base + tt[u] : 0);
          const uint32_t hw = (DM == MMA_DROP_HASH || DM == MMA_DROP_HASH16) ? drop_base_word(dp, eu, cc >> 2) : 0u;      // ONE full hash for the K masks
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const bool raw = (p.acts >> (p.k_base + k)) & 1u;
            float f[VEC];
            if (DM == MMA_DROP_HASH) {
              drop_unpack<VEC>(dp, drop_mask_word(hw, k == 0 ? p.k_base : 1, mult[k]), cc, f);     // k > 0: never the base word (compile time)
            } else if (DM == MMA_DROP_HASH16) {
              drop_unpack16<VEC>(dp, drop_mask_word(hw, k == 0 ? p.k_base : 1, mult[k]), drop_low_word(hw, mult2[k]), cc, f);
            } else if (DM == MMA_DROP_EXPLICIT) {
              drop_explicit<VEC>(dp, eu, p.k_base + k, cc, p.H, f);
            } else {
#pragma unroll
              for (int i = 0; i < VEC; ++i) f[i] = 1.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const float z = pk[k].v[i] + qv[u][k].v[i];
              float a, da;
              if (raw) { a = z; da = 1.f; }
              else { a = sigmoid_fast(z); da = a - a * a; }
              const float w = f[i] * xj[u].v[i];
              acc[k].v[i] = fmaf(a, w, acc[k].v[i]);
              if (SAVE) tac[k].v[i] = fmaf(da, w, tac[k].v[i]);
            }
          }
        }
      }
    }

    // butterfly over the sub-rows of a group (lanes with equal feature column)
    for (int off = G / 2; off >= lpr; off >>= 1) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          acc[k].v[i] += __shfl_xor(acc[k].v[i], off, kWave);
          if (SAVE) tac[k].v[i] += __shfl_xor(tac[k].v[i], off, kWave);
        }
    }

    if (sub == 0 && fvalid && ivalid) {
      if (slot < 0) {
        const float deg = (float)max(p.rowptr[node + 1] - p.rowptr[node], 1);
        Vec<VEC> ms = vzero<VEC>();
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const Vec<VEC> mo = nc_fwd_write<VEC, SAVE>(p, node, p.k_base + k, c, xi, acc[k], tac[k], deg);
#pragma unroll
          for (int i = 0; i < VEC; ++i) ms.v[i] += mo.v[i];
        }
        if (p.msum) nc_msum_store<VEC>(p, node, c, ms, p.k_base > 0);
        if (SAVE && p.crow && c == 0 && p.k_base == 0) p.crow[(size_t)node * p.ldc] = 1.f / deg;
      } else {
        float* ps = p.partial + (size_t)slot * p.pstride;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          stv<VEC>(ps + (size_t)(p.k_base + k) * p.H + c, acc[k]);
          if (SAVE) stv<VEC>(ps + (size_t)(p.K_total + p.k_base + k) * p.H + c, tac[k]);
        }
      }
    }
    if (ONE) {
      const bool wrote = ivalid && slot >= 0;
      if (__builtin_amdgcn_ballot_w64(wrote) != 0) {            // wave-uniform from here on
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned ticket = 0u;
        if (wrote && gl == 0) ticket = atomicAdd(p.sync, 1u) + 1u;
        if (__builtin_amdgcn_ballot_w64(ticket == p.n_slots) != 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          nc_fwd_finalize_body<VEC, SAVE>(p, p.hubs, p.n_hubs, lane, kWave);
          if (lane == 0) *p.sync = 0u;
        }
      }
    }
  }
}

template <int K, int VEC, bool SAVE, int DM, bool MULTI>
__global__ __launch_bounds__(kBlock, (K <= 4 ? MMA_MIN_WAVES : 1)) void nc_fwd_kernel(const NcFwdParams p) {
  nc_fwd_body<K, VEC, SAVE, DM, MULTI>(p, (int)blockIdx.x, (int)gridDim.x);
}

// hub nodes: sum the chunk partials in slot order, then the same epilogue
template <int VEC, bool SAVE>
__device__ __forceinline__ void nc_fwd_finalize_body(const NcFwdParams& p, const int4* hubs, int64_t n_hubs, const int64_t first, const int64_t step) {
  const int per_row = (p.H + VEC - 1) / VEC;
  const int64_t total = n_hubs * per_row;
  for (int64_t idx = first; idx < total; idx += step) {
    const int c = (int)(idx % per_row) * VEC;
    const int4 hub = hubs[idx / per_row];
    const int node = hub.x;
    const Vec<VEC> xi = ldv<VEC>(p.x + (size_t)node * p.ldx + c);
    const float deg = (float)max(p.rowptr[node + 1] - p.rowptr[node], 1);
    Vec<VEC> ms = vzero<VEC>();
    for (int k = 0; k < p.K_total; ++k) {
      Vec<VEC> s = vzero<VEC>(), t = vzero<VEC>();
      // the partials of a hub are summed in slot order (fixed: bitwise repeatable), but FETCHED eight slots at a time: one load,
      // one wait, one add per slot left the largest hub's 52 slots x K masks as ~200 exposed memory latencies (0.10 ms per call)
      constexpr int kAhead = 8;
      for (int sl0 = hub.y; sl0 < hub.z; sl0 += kAhead) {
        Vec<VEC> a[kAhead], b[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
          const float* ps = p.partial + (size_t)min(sl0 + u, hub.z - 1) * p.pstride;     // past the end: re-read the last one, dropped below
          a[u] = ldv<VEC>(ps + (size_t)k * p.H + c);
          if (SAVE) b[u] = ldv<VEC>(ps + (size_t)(p.K_total + k) * p.H + c);
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
          if (sl0 + u < hub.z) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) s.v[i] += a[u].v[i];
            if (SAVE) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) t.v[i] += b[u].v[i];
            }
          }
        }
      }
      const Vec<VEC> mo = nc_fwd_write<VEC, SAVE>(p, node, k, c, xi, s, t, deg);
#pragma unroll
      for (int i = 0; i < VEC; ++i) ms.v[i] += mo.v[i];
    }
    if (p.msum) nc_msum_store<VEC>(p, node, c, ms, false);
    if (SAVE && p.crow && c == 0) p.crow[(size_t)node * p.ldc] = 1.f / deg;
  }
}
template <int VEC, bool SAVE>
__global__ __launch_bounds__(kBlock) void nc_fwd_finalize_kernel(const NcFwdParams p, const int4* hubs, int64_t n_hubs) {
  nc_fwd_finalize_body<VEC, SAVE>(p, hubs, n_hubs, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// Small graphs (Cora, PubMed: every kernel of the layer is a few microseconds, the launches ARE the step): the wave items, the
// grouped items and the hub sums of one call in ONE launch.  Blocks [0, blocks_a) walk the wave items, the rest the grouped items;
// the hub sums are done by the wavefront that stores the last chunk partial (ONE in nc_fwd_body).
struct NcSmallPlan { int64_t n_wave_items; int blocks_a; };

template <int K, int VEC, bool SAVE, int DM>
__global__ __launch_bounds__(kBlock, (K <= 4 ? MMA_MIN_WAVES : 1)) void nc_fwd_small_kernel(const NcFwdParams p, const NcSmallPlan sp) {
  if ((int)blockIdx.x < sp.blocks_a) {
    NcFwdParams q = p;
    q.n_items = sp.n_wave_items;
    nc_fwd_body<K, VEC, SAVE, DM, false, true>(q, (int)blockIdx.x, sp.blocks_a);
  } else {
    NcFwdParams q = p;
    q.items = p.items + sp.n_wave_items; q.n_items = p.n_items - sp.n_wave_items;
    nc_fwd_body<K, VEC, SAVE, DM, true, true>(q, (int)blockIdx.x - sp.blocks_a, (int)gridDim.x - sp.blocks_a);
  }
}

// ------------------------------------------------------------------------------------------------------
// K2a: node-level backward of the combine
struct NcBwdNodeParams {
  const float* g; int64_t g_kstride, ldgr;   // g[k*g_kstride + node*ldgr + c]; g_kstride == 0: one (N,H) gradient shared by all masks
  const uint8_t* sel; const float* T; int64_t ldt; const int32_t* rowptr;
  float* gs; int64_t ldgs; float* gP; int64_t ldgp; float* gxs; int64_t ldgx;
  int64_t N; int H, K; uint32_t kinds;
  // shared-gradient form (round 3): the selection state comes from the packed code rows K1 wrote (see NcFwdParams::crow) instead
  // of `sel`; nothing is written for K2b any more (round 2 wrote one packed [g | 1/d | codes] row per target here: 0.8 GB at C4)
  const float* crow; int64_t ldc; int HQ; uint32_t sel_slots;
  uint32_t* rowmax;        // optional: max |gP| per node (bits of a non-negative float), merged with K2b's max |gQ| by atomicMax
};

// dm/ds and dm/dx_i of one element from the packed byte tf = 2 * dm/ds of a max/min/softmax-type mask (255: NaN gradient), or
// from the kind alone for sum / mean
__device__ __forceinline__ void combine_grad_tf(int kind, uint32_t tf, float inv_deg, float& fs, float& fx) {
  switch (kind) {
    case MMA_KIND_SUM: fs = 1.f; fx = 1.f; break;
    case MMA_KIND_MEAN: fs = inv_deg; fx = inv_deg; break;
    case MMA_KIND_MAX:
    case MMA_KIND_MIN: fs = 0.5f * (float)tf; fx = 1.f - fs; break;
    default: fs = tf == 255u ? __builtin_nanf("") : 1.f; fx = 0.f; break;
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void nc_bwd_node_kernel(const NcBwdNodeParams p) {
  const int per_row = (p.H + VEC - 1) / VEC;
  const int64_t total = p.N * per_row;
  // One (node, VEC columns) item per thread and pass; ALL of an item's loads (g, T_k, sel_k for every mask) are issued before
  // its first store: vmcnt is one in-order counter for loads and stores, so the round-1 form (load T_k, store gP_k, load
  // T_k+1, ... in a grid-stride loop) made every load wait for the stores in front of it (5.65 TB/s on a pure stream).
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t node = idx / per_row;
    const int c = (int)(idx % per_row) * VEC;
    const float deg = (float)max(p.rowptr[node + 1] - p.rowptr[node], 1);
    const float inv_deg = 1.f / deg;
    Vec<VEC> gk[MMA_MAX_K], tk[MMA_MAX_K];
    uint32_t ck[MMA_MAX_K];
    const bool shared_g = p.g_kstride == 0;
#pragma unroll
    for (int k = 0; k < MMA_MAX_K; ++k) {
      if (k < p.K) {
        const size_t o = (size_t)node * p.ldt + (size_t)k * p.H + c;
        if (k == 0 || !shared_g) gk[k] = ldv_nt<VEC>(p.g + (size_t)k * p.g_kstride + (size_t)node * p.ldgr + c);
        tk[k] = ldv_nt<VEC>(p.T + o);
        if (p.crow) {
          ck[k] = sel_slot_of(p.sel_slots, k) != 0xFu
              ? ldb<VEC>(reinterpret_cast<const uint8_t*>(p.crow + (size_t)node * p.ldc + 4 + (size_t)sel_slot_of(p.sel_slots, k) * p.HQ) + c) : 0u;
        } else {
          ck[k] = ldb_nt<VEC>(p.sel + o);
        }
      }
    }
    Vec<VEC> gx = vzero<VEC>();
    float mxp = 0.f;
#pragma unroll
    for (int k = 0; k < MMA_MAX_K; ++k) {
      if (k < p.K) {
        const int kind = kind_of(p.kinds, k);
        const Vec<VEC> g = shared_g ? gk[0] : gk[k];
        const Vec<VEC> t = tk[k];
        const uint32_t codes = ck[k];
        Vec<VEC> gsv, gpv;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          uint32_t tf = (codes >> (8 * i)) & 0xFFu;
          if (!p.crow) tf = tf == 1u ? 2u : (tf == 2u ? 1u : (tf == 3u ? 255u : 0u));     // sel code -> 2 * dm/ds
          float fs, fx;  // d m / d s, d m / d x_i
          combine_grad_tf(kind, tf, inv_deg, fs, fx);
          gsv.v[i] = g.v[i] * fs;
          gpv.v[i] = gsv.v[i] * t.v[i];
          gx.v[i] = fmaf(g.v[i], fx, gx.v[i]);
          mxp = fmaxf(mxp, fabsf(gpv.v[i]));
        }
        if (p.gs) stv<VEC>(p.gs + (size_t)node * p.ldgs + (size_t)k * p.H + c, gsv);
        stv_nt<VEC>(p.gP + (size_t)node * p.ldgp + (size_t)k * p.H + c, gpv);
      }
    }
    stv_nt<VEC>(p.gxs + (size_t)node * p.ldgx + c, gx);
    if (p.rowmax) {          // the row maximum of gP for the three-product dL/dx GEMM (its A rows are scaled by a power of two)
      if ((per_row & (per_row - 1)) == 0 && per_row <= kWave) {     // a node's threads are an aligned lane group: one atomic per node
        for (int off = per_row >> 1; off > 0; off >>= 1) mxp = fmaxf(mxp, __shfl_xor(mxp, off, kWave));
        if ((idx & (per_row - 1)) == 0 && mxp > 0.f) atomicMax(p.rowmax + node, __float_as_uint(mxp));
      } else if (mxp > 0.f) {
        atomicMax(p.rowmax + node, __float_as_uint(mxp));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// K2b: edge-level backward over the transposed CSR (grouped by source j)
struct NcBwdParams {
  const float* x; int64_t ldx;
  const float* P; int64_t ldp; const float* Q; int64_t ldq;
  const float* gs; int64_t ldg; const float* gxs; int64_t ldgx;
  // SHARED mode (gs == NULL): all masks share one upstream gradient g (n_targets, ldgg); gs_k[i] = g[i] * f(kind_k, code_k[i], 1/d_i)
  // is rebuilt per edge from g[i] and the packed code row K1 wrote, crow[i] = [ 1/d_i,0,0,0 | codes per sel-kind (HQ words) ]
  const float* g; int64_t ldgg; const float* crow; int64_t ldc; uint32_t sel_slots; uint32_t kinds;
  // EPI (T != NULL, SHARED only): the node-level backward of the combine (K2a) runs in the per-source epilogue - source j < n_targets
  // is also target j: gP[j] = g[j] * dm/ds * T[j] and the direct term sum_k g[j] * dm/dx_i are formed here, gxs is not used
  const float* T; int64_t ldt; float* gP; int64_t ldgp; int64_t n_targets;
  const int32_t* t_col; const int32_t* t_eid;
  const int4* items; int64_t n_items;
  float* partial; int64_t pstride;   // floats per slot = (K_total+1)*H
  float* gQ; int64_t ldgq; float* gx; int64_t ldgxo;
  int H, HQ, K_total, k_base, lpr_log;
  uint32_t acts;
  DropParams drop;
  int first_pass;  // k_base == 0: gx starts from gxs (or the epilogue's direct term); later K-slices accumulate onto gx
  uint32_t* rowmax;  // optional: max |gQ| (EPI: and |gP|) per source row (see NcBwdNodeParams::rowmax)
  const int4* hubs; int64_t n_hubs; unsigned* sync; unsigned n_slots;      // one-launch form, as in NcFwdParams
};

// the K2a work of one (node, VEC columns) for the masks [k0, k0 + nk): stores gP, returns the direct term sum_k g * dm/dx_i and
// the running max |gP|.  Shared by the K2b epilogue and the hub finalize kernel.
template <int VEC>
__device__ __forceinline__ Vec<VEC> nc_bwd_epilogue(const NcBwdParams& p, int node, int c, int k0, int nk, float& mx) {
  const Vec<VEC> g = ldv_nt<VEC>(p.g + row_off(node, p.ldgg) + c);
  const float* crow = p.crow + row_off(node, p.ldc);
  const float inv_deg = crow[0];
  Vec<VEC> gxd = vzero<VEC>();
  Vec<VEC> tk[MMA_MAX_K]; uint32_t ck[MMA_MAX_K];
#pragma unroll
  for (int k = 0; k < MMA_MAX_K; ++k) {
    if (k < nk) {
      tk[k] = ldv_nt<VEC>(p.T + row_off(node, p.ldt) + (size_t)(k0 + k) * p.H + c);
      const uint32_t sslot = sel_slot_of(p.sel_slots, k0 + k);
      ck[k] = sslot != 0xFu ? ldb<VEC>(reinterpret_cast<const uint8_t*>(crow + 4 + (size_t)sslot * p.HQ) + c) : 0u;
    }
  }
#pragma unroll
  for (int k = 0; k < MMA_MAX_K; ++k) {
    if (k < nk) {
      const int kind = kind_of(p.kinds, k0 + k);
      Vec<VEC> gpv;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float fs, fx;
        combine_grad_tf(kind, (ck[k] >> (8 * i)) & 0xFFu, inv_deg, fs, fx);
        gpv.v[i] = (g.v[i] * fs) * tk[k].v[i];
        gxd.v[i] = fmaf(g.v[i], fx, gxd.v[i]);
        mx = fmaxf(mx, fabsf(gpv.v[i]));
      }
      stv_nt<VEC>(p.gP + row_off(node, p.ldgp) + (size_t)(k0 + k) * p.H + c, gpv);
    }
  }
  return gxd;
}

template <int VEC>
__device__ __forceinline__ void nc_bwd_finalize_body(const NcBwdParams& p, const int4* hubs, int64_t n_hubs, const int64_t first, const int64_t step);

template <int K, int VEC, int DM, bool SHARED, bool MULTI, bool EPI, bool ONE = false>
__device__ __forceinline__ void nc_bwd_body(const NcBwdParams& p, const int bx, const int nbx) {
  constexpr bool DROP = DM != MMA_DROP_NONE;
  const DropParams dp = (DM == MMA_DROP_HASH || DM == MMA_DROP_HASH16) ? drop_resolve(p.drop) : p.drop;
  uint32_t mult[K], mult2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { mult[k] = drop_mask_mult(p.k_base + k); mult2[k] = drop_mask_mult2(p.k_base + k); }
  constexpr int U = NC_BWD_UNROLL(K);           // edge steps in flight per lane (see nc_fwd_kernel)
  const int lane = threadIdx.x & (kWave - 1);
  const int lpr = 1 << p.lpr_log;
  const int G = MULTI ? lpr : kWave;
  const int epg = G >> p.lpr_log;
  const int gpw = kWave / G;
  const int grp = MULTI ? lane / G : 0;
  const int gl = lane & (G - 1);
  const int gbase = grp * G;
  const int sub = gl >> p.lpr_log;
  const int c = ((int)blockIdx.y * lpr + (lane & (lpr - 1))) * VEC;
  const bool fvalid = c < p.H;
  const int cc = fvalid ? c : 0;
  const int waves_per_block = kBlock / kWave;
  const int64_t stride = (int64_t)nbx * waves_per_block;
  const int64_t n_witems = (p.n_items + gpw - 1) / gpw;

  for (int64_t it0 = (int64_t)bx * waves_per_block + (threadIdx.x >> 6); it0 < n_witems; it0 += stride) {
    int node, ebeg, eend, slot;     // node = the SOURCE j
    bool ivalid = true;
    if (MULTI) {
      const int64_t idx = it0 * gpw + grp;
      ivalid = idx < p.n_items;
      const int4 item = p.items[ivalid ? idx : 0];
      node = item.x; ebeg = item.y; eend = ivalid ? item.z : item.y; slot = item.w;
    } else {
      const int4 item = p.items[__builtin_amdgcn_readfirstlane((int)it0)];
      node = __builtin_amdgcn_readfirstlane(item.x);
      ebeg = __builtin_amdgcn_readfirstlane(item.y);
      eend = __builtin_amdgcn_readfirstlane(item.z);
      slot = __builtin_amdgcn_readfirstlane(item.w);
    }
    const int len = eend - ebeg;
    int maxlen = len;
    if (MULTI) {
      for (int off = G; off < kWave; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, kWave));
      maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    }

    const Vec<VEC> xj = ldv_nt<VEC>(p.x + row_off(node, p.ldx) + cc);
    Vec<VEC> qk[K], aq[K];
    Vec<VEC> ax = vzero<VEC>();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      qk[k] = ldv_nt<VEC>(p.Q + row_off(node, p.ldq) + (size_t)(p.k_base + k) * p.H + cc);
      aq[k] = vzero<VEC>();
    }

    for (int base = 0; base < maxlen; base += G) {
      const int cnt = min(G, max(len - base, 0));
      const int ucnt = min(G, maxlen - base);
      // lane 0 of a group with no edge in this chunk (but edges in an earlier one) offers the item's FIRST target: see the load phase
      const int myi = (gl < cnt) ? p.t_col[ebeg + base + gl] : ((gl == 0 && len > 0) ? p.t_col[ebeg] : 0);
      const int mye = (DROP && gl < cnt) ? p.t_eid[ebeg + base + gl] : 0;
      for (int t0 = 0; t0 < ucnt; t0 += U * epg) {
        // load phase: raw operands only (no arithmetic on loaded values, so all edge steps stay in flight)
        int tt[U]; bool ev[U]; uint32_t eid[U]; Vec<VEC> gv[U][SHARED ? 1 : K]; Vec<VEC> pv[U][K];
        uint32_t codes[U][K]; float idg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          tt[u] = t0 + u * epg + sub;
          ev[u] = tt[u] < cnt;
          // A sub-row past the end of its item repeats the item's LAST valid edge of this index chunk (rows that are in flight
          // or cached anyway) and contributes exactly 0 through ONE select per element of g below - nothing FOREIGN is read, so
          // a non-finite value can only reach an item that holds it itself (round 2 sent such lanes to target row 0 and replaced
          // everything they loaded: 30 selects per edge step).  A group with no edge in this chunk (cnt == 0) repeats its item's first
          // edge; an item with no edge at all (len == 0) reads target 0 and its sums are discarded below.
          const int tl = min(tt[u], max(cnt - 1, 0));
          const int ii = __shfl(myi, gbase + (tl & (G - 1)), kWave);
          eid[u] = DROP ? (uint32_t)__shfl(mye, gbase + (tl & (G - 1)), kWave) : 0u;
          if (SHARED) {
            gv[u][0] = ldv<VEC>(p.g + row_off(ii, p.ldgg) + cc);
            const float* crow = p.crow + row_off(ii, p.ldc);
            idg[u] = crow[0];
#pragma unroll
            for (int k = 0; k < K; ++k) {
              // sum/mean never look at the code: a constant "s selected" (2 = twice the factor 1.0) keeps the arithmetic
              // below branch-free
              const uint32_t sslot = sel_slot_of(p.sel_slots, p.k_base + k);
              codes[u][k] = (sslot != 0xFu) ? ldb<VEC>(reinterpret_cast<const uint8_t*>(crow + 4 + (size_t)sslot * p.HQ) + cc)
                                             : 0x02020202u;
            }
          }
          const float* prow = p.P + row_off(ii, p.ldp) + cc;
          const float* grow = SHARED ? nullptr : p.gs + row_off(ii, p.ldg) + cc;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const size_t o = (size_t)(p.k_base + k) * p.H;
            if (!SHARED) gv[u][k] = ldv<VEC>(grow + o);
            pv[u][k] = ldv<VEC>(prow + o);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = ev[u];
#pragma unroll
          for (int kk = 0; kk < (SHARED ? 1 : K); ++kk)
#pragma unroll
            for (int i = 0; i < VEC; ++i) gv[u][kk].v[i] = live ? gv[u][kk].v[i] : 0.f;
          const uint32_t hw = (DM == MMA_DROP_HASH || DM == MMA_DROP_HASH16) ? drop_base_word(dp, eid[u], cc >> 2) : 0u;      // ONE full hash for the K masks
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const bool raw = (p.acts >> (p.k_base + k)) & 1u;
            const int kind = SHARED ? kind_of(p.kinds, p.k_base + k) : 0;
            const float kscale = 0.5f * (kind == MMA_KIND_MEAN ? idg[u] : 1.f);   // the code byte holds TWICE dm/ds
            float f[VEC];
            if (DM == MMA_DROP_HASH) {
              drop_unpack<VEC>(dp, drop_mask_word(hw, k == 0 ? p.k_base : 1, mult[k]), cc, f);     // k > 0: never the base word (compile time)
            } else if (DM == MMA_DROP_HASH16) {
              drop_unpack16<VEC>(dp, drop_mask_word(hw, k == 0 ? p.k_base : 1, mult[k]), drop_low_word(hw, mult2[k]), cc, f);
            } else if (DM == MMA_DROP_EXPLICIT) {
              drop_explicit<VEC>(dp, eid[u], p.k_base + k, cc, p.H, f);
            } else {
#pragma unroll
              for (int i = 0; i < VEC; ++i) f[i] = 1.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const float z = pv[u][k].v[i] + qk[k].v[i];
              float a, da;
              if (raw) { a = z; da = 1.f; }
              else { a = sigmoid_fast(z); da = a - a * a; }
              float gsv;
              if (SHARED) {
                const uint32_t tf = (codes[u][k] >> (8 * i)) & 0xFFu;      // v_cvt_f32_ubyte<i>
                float cf = (float)tf;
                if (kind >= MMA_KIND_SOFTMAX && tf == 255u) cf = __builtin_nanf("");   // exp overflow in the forward combine
                gsv = gv[u][0].v[i] * (cf * kscale);
              } else {
                gsv = gv[u][k].v[i];
              }
              const float w = f[i] * gsv;
              aq[k].v[i] = fmaf(da, w, aq[k].v[i]);
              ax.v[i] = fmaf(a, w, ax.v[i]);
            }
          }
        }
      }
    }

    // EPI: the node-level loads of the epilogue go out before the butterfly (they do not depend on the sums)
    const bool writer = sub == 0 && fvalid && ivalid;
    const bool tgt = EPI && writer && slot < 0 && (int64_t)node < p.n_targets;
    Vec<VEC> gxd = vzero<VEC>();
    float mxp = 0.f;
    if (EPI) { if (tgt) gxd = nc_bwd_epilogue<VEC>(p, node, c, p.k_base, K, mxp); }

    for (int off = G / 2; off >= lpr; off >>= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        ax.v[i] += __shfl_xor(ax.v[i], off, kWave);
#pragma unroll
        for (int k = 0; k < K; ++k) aq[k].v[i] += __shfl_xor(aq[k].v[i], off, kWave);
      }
    }
    if (len == 0) {            // nothing was accumulated for this item (a chunk trip on behalf of other groups may have run)
      ax = vzero<VEC>();
#pragma unroll
      for (int k = 0; k < K; ++k) aq[k] = vzero<VEC>();
    }

    if (p.rowmax) {           // wave-uniform
      float mxq = mxp;
      if (writer && slot < 0) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < VEC; ++i) mxq = fmaxf(mxq, fabsf(xj.v[i] * aq[k].v[i]));
      }
      for (int off = lpr >> 1; off > 0; off >>= 1) mxq = fmaxf(mxq, __shfl_xor(mxq, off, kWave));
      if (sub == 0 && (lane & (lpr - 1)) == 0 && ivalid && slot < 0 && mxq > 0.f) atomicMax(p.rowmax + node, __float_as_uint(mxq));
    }
    if (writer) {
      if (slot < 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          Vec<VEC> o;
#pragma unroll
          for (int i = 0; i < VEC; ++i) o.v[i] = xj.v[i] * aq[k].v[i];
          stv_nt<VEC>(p.gQ + row_off(node, p.ldgq) + (size_t)(p.k_base + k) * p.H + c, o);
        }
        Vec<VEC> g0;
        if (EPI) {
          g0 = gxd;              // zero for a halo source (no target role)
          if (!p.first_pass) {
            const Vec<VEC> prev = ldv<VEC>(p.gx + row_off(node, p.ldgxo) + c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) g0.v[i] += prev.v[i];
          }
        } else {
          g0 = p.first_pass ? ldv_nt<VEC>(p.gxs + row_off(node, p.ldgx) + c) : ldv<VEC>(p.gx + row_off(node, p.ldgxo) + c);
        }
        Vec<VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = g0.v[i] + ax.v[i];
        stv_nt<VEC>(p.gx + row_off(node, p.ldgxo) + c, o);
      } else {
        float* ps = p.partial + (size_t)slot * p.pstride;
#pragma unroll
        for (int k = 0; k < K; ++k) stv<VEC>(ps + (size_t)(p.k_base + k) * p.H + c, aq[k]);
        // the x-gradient partial of this K-slice; slices are summed by the finalize kernel
        float* px = ps + (size_t)p.K_total * p.H + c;
        if (p.first_pass) stv<VEC>(px, ax);
        else {
          const Vec<VEC> prev = ldv<VEC>(px);
          Vec<VEC> o;
#pragma unroll
          for (int i = 0; i < VEC; ++i) o.v[i] = prev.v[i] + ax.v[i];
          stv<VEC>(px, o);
        }
      }
    }
    if (ONE) {                                                   // see nc_fwd_body
      const bool wrote = ivalid && slot >= 0;
      if (__builtin_amdgcn_ballot_w64(wrote) != 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned ticket = 0u;
        if (wrote && gl == 0) ticket = atomicAdd(p.sync, 1u) + 1u;
        if (__builtin_amdgcn_ballot_w64(ticket == p.n_slots) != 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          nc_bwd_finalize_body<VEC>(p, p.hubs, p.n_hubs, lane, kWave);
          if (lane == 0) *p.sync = 0u;
        }
      }
    }
  }
}

template <int K, int VEC, int DM, bool SHARED, bool MULTI, bool EPI>
__global__ __launch_bounds__(kBlock, (K <= 4 ? MMA_MIN_WAVES : 1)) void nc_bwd_kernel(const NcBwdParams p) {
  nc_bwd_body<K, VEC, DM, SHARED, MULTI, EPI>(p, (int)blockIdx.x, (int)gridDim.x);
}

template <int VEC>
__device__ __forceinline__ void nc_bwd_finalize_body(const NcBwdParams& p, const int4* hubs, int64_t n_hubs, const int64_t first, const int64_t step) {
  const int per_row = (p.H + VEC - 1) / VEC;
  const bool epi = p.T != nullptr;
  const int64_t total = n_hubs * (p.K_total + 1) * per_row;
  for (int64_t idx = first; idx < total; idx += step) {
    const int c = (int)(idx % per_row) * VEC;
    const int k = (int)((idx / per_row) % (p.K_total + 1));
    const int4 hub = hubs[idx / ((int64_t)per_row * (p.K_total + 1))];
    Vec<VEC> s = vzero<VEC>();
    constexpr int kAhead = 8;                       // slot order kept, eight fetches in flight (see nc_fwd_finalize_kernel)
    for (int sl0 = hub.y; sl0 < hub.z; sl0 += kAhead) {
      Vec<VEC> a[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) a[u] = ldv<VEC>(p.partial + (size_t)min(sl0 + u, hub.z - 1) * p.pstride + (size_t)k * p.H + c);
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        if (sl0 + u < hub.z) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) s.v[i] += a[u].v[i];
        }
      }
    }
    const int node = hub.x;
    const bool tgt = epi && (int64_t)node < p.n_targets;
    Vec<VEC> o;
    if (k < p.K_total) {
      const Vec<VEC> xj = ldv<VEC>(p.x + (size_t)node * p.ldx + c);
      float mxq = 0.f;
      if (tgt) nc_bwd_epilogue<VEC>(p, node, c, k, 1, mxq);            // gP of this mask (the direct term is formed by the k == K thread)
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.v[i] = xj.v[i] * s.v[i];
      stv_nt<VEC>(p.gQ + (size_t)node * p.ldgq + (size_t)k * p.H + c, o);
      if (p.rowmax) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) mxq = fmaxf(mxq, fabsf(o.v[i]));
        if (mxq > 0.f) atomicMax(p.rowmax + node, __float_as_uint(mxq));
      }
    } else {
      Vec<VEC> g0 = vzero<VEC>();
      if (tgt) {
        // sum_k g * dm/dx_i over ALL masks: the epilogue's direct term without its stores
        const Vec<VEC> g = ldv<VEC>(p.g + row_off(node, p.ldgg) + c);
        const float* crow = p.crow + row_off(node, p.ldc);
        const float inv_deg = crow[0];
        for (int kk = 0; kk < p.K_total; ++kk) {
          const uint32_t sslot = sel_slot_of(p.sel_slots, kk);
          const uint32_t ck = sslot != 0xFu ? ldb<VEC>(reinterpret_cast<const uint8_t*>(crow + 4 + (size_t)sslot * p.HQ) + c) : 0u;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            float fs, fx;
            combine_grad_tf(kind_of(p.kinds, kk), (ck >> (8 * i)) & 0xFFu, inv_deg, fs, fx);
            g0.v[i] = fmaf(g.v[i], fx, g0.v[i]);
          }
        }
      } else if (!epi) {
        g0 = ldv<VEC>(p.gxs + (size_t)node * p.ldgx + c);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) o.v[i] = g0.v[i] + s.v[i];
      stv<VEC>(p.gx + (size_t)node * p.ldgxo + c, o);
    }
  }
}
template <int VEC>
__global__ __launch_bounds__(kBlock) void nc_bwd_finalize_kernel(const NcBwdParams p, const int4* hubs, int64_t n_hubs) {
  nc_bwd_finalize_body<VEC>(p, hubs, n_hubs, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// one launch for a small graph's backward (see nc_fwd_small_kernel); shared-gradient form only
template <int K, int VEC, int DM, bool EPI>
__global__ __launch_bounds__(kBlock, (K <= 4 ? MMA_MIN_WAVES : 1)) void nc_bwd_small_kernel(const NcBwdParams p, const NcSmallPlan sp) {
  if ((int)blockIdx.x < sp.blocks_a) {
    NcBwdParams q = p;
    q.n_items = sp.n_wave_items;
    nc_bwd_body<K, VEC, DM, true, false, EPI, true>(q, (int)blockIdx.x, sp.blocks_a);
  } else {
    NcBwdParams q = p;
    q.items = p.items + sp.n_wave_items; q.n_items = p.n_items - sp.n_wave_items;
    nc_bwd_body<K, VEC, DM, true, true, EPI, true>(q, (int)blockIdx.x - sp.blocks_a, (int)gridDim.x - sp.blocks_a);
  }
}

// ------------------------------------------------------------------------------------------------------
// host side
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int pack_codes(const uint8_t* kind_host, const uint8_t* act_host, int K, uint32_t* kinds, uint32_t* acts) {
  *kinds = 0; *acts = 0;
  for (int k = 0; k < K; ++k) {
    if (kind_host) {
      if (kind_host[k] > MMA_KIND_SOFTMIN) return fail(1, "kind[%d]=%d is not an MMA_KIND_* code", k, (int)kind_host[k]);
      *kinds |= (uint32_t)kind_host[k] << (4 * k);
    }
    if (act_host) {
      if (act_host[k] > MMA_ACT_RAW) return fail(1, "act[%d]=%d is not an MMA_ACT_* code", k, (int)act_host[k]);
      *acts |= (uint32_t)act_host[k] << k;
    }
  }
  return 0;
}

// sel-kinds (max/min/softmax/softmin) get consecutive code slots in the packed code row (4 bits per mask, 0xF = none); returns the
// row length in floats: [ 1/d_i, 0, 0, 0 | n_sel * ceil(H/4) words of codes ], rounded up to a multiple of 4
static int64_t fill_sel_slots(const uint8_t* kind_host, int K, int H, uint32_t* slots) {
  int n = 0;
  *slots = 0xFFFFFFFFu;
  for (int k = 0; k < K; ++k)
    if (kind_host[k] >= MMA_KIND_MAX) { *slots = (*slots & ~(0xFu << (4 * k))) | ((uint32_t)n << (4 * k)); ++n; }
  return ((4 + (int64_t)n * (((int64_t)H + 3) / 4)) + 3) & ~(int64_t)3;      // 64-bit: H comes straight from the caller
}

static int make_drop(int32_t mode, uint32_t thr, uint64_t seed, const uint64_t* seed_dev, int64_t edge_base, const uint8_t* keep,
                     int64_t E, DropParams* d) {
  d->seed_dev = seed_dev;
  MMA_REQUIRE(edge_base >= 0 && edge_base + E < (1LL << 32), "drop_edge_base %lld out of range", (long long)edge_base);
  d->edge_base = (uint32_t)edge_base;
  MMA_REQUIRE(mode >= MMA_DROP_NONE && mode <= MMA_DROP_EXPLICIT, "drop_mode %d unknown", mode);
  MMA_REQUIRE(mode == MMA_DROP_NONE || thr < 65536, "drop_thr %u out of range (0..65535: P(drop) = thr / 65536)", thr);
  MMA_REQUIRE(mode != MMA_DROP_EXPLICIT || keep != nullptr, "drop_mode EXPLICIT needs a keep mask");
  drop_set_threshold(d, mode, thr);
  d->seed_lo = (uint32_t)seed; d->seed_hi = (uint32_t)(seed >> 32); d->keep = keep; d->E = E;
  return 0;
}

struct Geometry { int vec, lpr_log, chunks; };
// lanes per row: next power of two >= ceil(H/vec), at most one wave; wider rows take gridDim.y chunks
static Geometry geometry(int H, bool vec4_ok) {
  Geometry g;
  g.vec = vec4_ok ? 4 : 1;
  const int per_row = (H + g.vec - 1) / g.vec;
  g.lpr_log = min(ilog2_ceil(per_row), 6);
  g.chunks = (per_row + (1 << g.lpr_log) - 1) >> g.lpr_log;
  return g;
}

static dim3 item_grid(int64_t n_items, int chunks, int items_per_wave = 1) {
  const int64_t per_block = (int64_t)(kBlock / kWave) * items_per_wave;
  int64_t blocks = (n_items + per_block - 1) / per_block;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks, (unsigned)chunks, 1);
}

template <int K, int VEC, bool MULTI>
static void launch_fwd(const NcFwdParams& p, dim3 grid, bool save, int dm, hipStream_t st) {
#define MMA_FWD(SAVE, DM) hipLaunchKernelGGL((nc_fwd_kernel<K, VEC, SAVE, DM, MULTI>), grid, dim3(kBlock), 0, st, p)
  if (save) {
    if (dm == MMA_DROP_HASH) MMA_FWD(true, MMA_DROP_HASH); else if (dm == MMA_DROP_HASH16) MMA_FWD(true, MMA_DROP_HASH16); else if (dm == MMA_DROP_EXPLICIT) MMA_FWD(true, MMA_DROP_EXPLICIT); else MMA_FWD(true, MMA_DROP_NONE);
  } else {
    if (dm == MMA_DROP_HASH) MMA_FWD(false, MMA_DROP_HASH); else if (dm == MMA_DROP_HASH16) MMA_FWD(false, MMA_DROP_HASH16); else if (dm == MMA_DROP_EXPLICIT) MMA_FWD(false, MMA_DROP_EXPLICIT); else MMA_FWD(false, MMA_DROP_NONE);
  }
#undef MMA_FWD
}
template <int VEC, bool MULTI>
static void launch_fwd_k(int Ks, const NcFwdParams& p, dim3 grid, bool save, int dm, hipStream_t st) {
  switch (Ks) {
    case 1: launch_fwd<1, VEC, MULTI>(p, grid, save, dm, st); break;
    case 2: launch_fwd<2, VEC, MULTI>(p, grid, save, dm, st); break;
    case 3: launch_fwd<3, VEC, MULTI>(p, grid, save, dm, st); break;
    case 4: launch_fwd<4, VEC, MULTI>(p, grid, save, dm, st); break;
    default: launch_fwd<8, VEC, MULTI>(p, grid, save, dm, st); break;
  }
}
template <int K, int VEC, bool MULTI>
static void launch_bwd(const NcBwdParams& p, dim3 grid, int dm, hipStream_t st) {
  const bool shared = p.gs == nullptr;
  const bool epi = p.T != nullptr;
#define MMA_BWD(DM, SHARED, EPI) hipLaunchKernelGGL((nc_bwd_kernel<K, VEC, DM, SHARED, MULTI, EPI>), grid, dim3(kBlock), 0, st, p)
#define MMA_BWD_DM(SHARED, EPI) \
  do { if (dm == MMA_DROP_HASH) MMA_BWD(MMA_DROP_HASH, SHARED, EPI); else if (dm == MMA_DROP_HASH16) MMA_BWD(MMA_DROP_HASH16, SHARED, EPI); else if (dm == MMA_DROP_EXPLICIT) MMA_BWD(MMA_DROP_EXPLICIT, SHARED, EPI); \
       else MMA_BWD(MMA_DROP_NONE, SHARED, EPI); } while (0)
  if (shared) { if (epi) MMA_BWD_DM(true, true); else MMA_BWD_DM(true, false); }
  else MMA_BWD_DM(false, false);
#undef MMA_BWD_DM
#undef MMA_BWD
}
template <int VEC, bool MULTI>
static void launch_bwd_k(int Ks, const NcBwdParams& p, dim3 grid, int dm, hipStream_t st) {
  switch (Ks) {
    case 1: launch_bwd<1, VEC, MULTI>(p, grid, dm, st); break;
    case 2: launch_bwd<2, VEC, MULTI>(p, grid, dm, st); break;
    case 3: launch_bwd<3, VEC, MULTI>(p, grid, dm, st); break;
    case 4: launch_bwd<4, VEC, MULTI>(p, grid, dm, st); break;
    default: launch_bwd<8, VEC, MULTI>(p, grid, dm, st); break;
  }
}

// the one-launch form (VEC = 4, hash / no dropout, one K-slice)
template <int K>
static void launch_fwd_small(const NcFwdParams& p, const NcSmallPlan& sp, dim3 grid, bool save, int dm, hipStream_t st) {
#define MMA_FWD_S(SAVE, DM) hipLaunchKernelGGL((nc_fwd_small_kernel<K, 4, SAVE, DM>), grid, dim3(kBlock), 0, st, p, sp)
  if (save) { if (dm == MMA_DROP_HASH) MMA_FWD_S(true, MMA_DROP_HASH); else if (dm == MMA_DROP_HASH16) MMA_FWD_S(true, MMA_DROP_HASH16); else MMA_FWD_S(true, MMA_DROP_NONE); }
  else { if (dm == MMA_DROP_HASH) MMA_FWD_S(false, MMA_DROP_HASH); else if (dm == MMA_DROP_HASH16) MMA_FWD_S(false, MMA_DROP_HASH16); else MMA_FWD_S(false, MMA_DROP_NONE); }
#undef MMA_FWD_S
}
template <int K>
static void launch_bwd_small(const NcBwdParams& p, const NcSmallPlan& sp, dim3 grid, int dm, hipStream_t st) {
  const bool epi = p.T != nullptr;
#define MMA_BWD_S(DM, EPI) hipLaunchKernelGGL((nc_bwd_small_kernel<K, 4, DM, EPI>), grid, dim3(kBlock), 0, st, p, sp)
  if (epi) { if (dm == MMA_DROP_HASH) MMA_BWD_S(MMA_DROP_HASH, true); else if (dm == MMA_DROP_HASH16) MMA_BWD_S(MMA_DROP_HASH16, true); else MMA_BWD_S(MMA_DROP_NONE, true); }
  else { if (dm == MMA_DROP_HASH) MMA_BWD_S(MMA_DROP_HASH, false); else if (dm == MMA_DROP_HASH16) MMA_BWD_S(MMA_DROP_HASH16, false); else MMA_BWD_S(MMA_DROP_NONE, false); }
#undef MMA_BWD_S
}
// Is the one-launch form possible?  Fills the plan (grid = blocks_a + blocks_b) when it is.  The hub sums are done by ONE wavefront
// (the one that stores the last chunk partial), so they must be few: `finalize_elems` = (hubs) x (vectors per row) [x (K+1)].
static bool small_plan(int32_t* sync, int K, const Geometry& g, int dm, int64_t n_items, int64_t n_wave_items, int64_t n_hubs, int64_t n_slots,
                       int64_t finalize_elems, NcSmallPlan* sp, dim3* grid) {
  const bool one_slice = K <= 4 || K == 8;
  if (!sync || !one_slice || g.vec != 4 || g.chunks != 1 || dm == MMA_DROP_EXPLICIT || finalize_elems > 4096 || n_slots >= (1LL << 31)) return false;
  const int64_t n_group = n_items - n_wave_items;
  if (n_hubs == 0 && (n_wave_items == 0 || n_group == 0)) return false;       // already a single launch
  const int ipw = kWave >> g.lpr_log;
  const unsigned ba = n_wave_items > 0 ? item_grid(n_wave_items, 1, 1).x : 0u;
  const unsigned bb = n_group > 0 ? item_grid(n_group, 1, ipw).x : 0u;
  *sp = NcSmallPlan{n_wave_items, (int)ba};
  *grid = dim3(ba + bb, 1, 1);
  return true;
}

// K in 1..8 is issued as slices the kernels are instantiated for: 8 | 4+{1,2,3} | {1,2,3,4}
static int next_slice(int remaining) { return remaining >= 8 ? 8 : (remaining >= 4 ? 4 : remaining); }

static int64_t elementwise_grid(int64_t total) {
  int64_t b = (total + kBlock - 1) / kBlock;
  return b < 1 ? 1 : (b > kMaxGrid * 4 ? kMaxGrid * 4 : b);
}

}  // namespace mma

using namespace mma;

extern "C" int mma_nc_fused_fwd(
    const float* x, int64_t ldx, const float* P, int64_t ldp, const float* Q, int64_t ldq,
    const int32_t* rowptr, const int32_t* col,
    const int32_t* items, int64_t n_items, int64_t n_wave_items, const int32_t* hubs, int64_t n_hubs,
    float* partial, int64_t n_slots, float* m, float* m_sum, int64_t ldms, float* T, uint8_t* sel, int64_t ldt,
    float* crow, int64_t ldc,
    int64_t N, int64_t E, int32_t H, int32_t K, const uint8_t* kind_host, const uint8_t* act_host,
    int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, int64_t drop_edge_base, const uint8_t* keep,
    int32_t* sync, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) && E < (1LL << 31), "N=%lld E=%lld out of int32 range", (long long)N, (long long)E);
  MMA_REQUIRE(H >= 1 && K >= 1 && K <= MMA_MAX_K, "H=%d K=%d unsupported (1<=K<=%d)", H, K, MMA_MAX_K);
  MMA_REQUIRE(ldx >= H && ldp >= (int64_t)K * H && ldq >= (int64_t)K * H, "row pitch too small: ldx=%lld ldp=%lld ldq=%lld",
              (long long)ldx, (long long)ldp, (long long)ldq);
  MMA_REQUIRE(T != nullptr || (sel == nullptr && crow == nullptr), "sel / crow are saved next to T: give T too");
  MMA_REQUIRE(T == nullptr || sel != nullptr || crow != nullptr, "T needs the selection state beside it: sel (N,K*H), crow (N,ldc), or both");
  MMA_REQUIRE(T == nullptr || ldt >= (int64_t)K * H, "ldt=%lld too small", (long long)ldt);
  MMA_REQUIRE(ldx < (1LL << 31) && ldp < (1LL << 31) && ldq < (1LL << 31) && ldt < (1LL << 31) && ldc < (1LL << 31), "row pitch out of range");
  MMA_REQUIRE(n_items >= 0 && n_hubs >= 0 && n_slots >= 0 && n_items < (1LL << 31) && n_wave_items >= 0, "negative or oversize item counts");
  MMA_REQUIRE(n_slots == 0 || (partial != nullptr && hubs != nullptr && n_hubs > 0), "hub slots without partial/hubs buffers");
  if (N == 0 || n_items == 0) return 0;
  MMA_REQUIRE(x && P && Q && rowptr && items && kind_host && act_host, "NULL argument");
  MMA_REQUIRE(m != nullptr || m_sum != nullptr, "give m (K,N,H), m_sum (N,H), or both");
  MMA_REQUIRE(m_sum == nullptr || ldms >= H, "ldms=%lld too small", (long long)ldms);
  MMA_REQUIRE(E == 0 || col != nullptr, "NULL col");
  MMA_REQUIRE(aligned16(items) && (hubs == nullptr || aligned16(hubs)), "items/hubs must be 16-byte aligned int32 quadruples");
  uint32_t kinds, acts;
  if (int rc = pack_codes(kind_host, act_host, K, &kinds, &acts)) return rc;
  NcFwdParams p{};
  if (int rc = make_drop(drop_mode, drop_thr, seed, seed_dev, drop_edge_base, keep, E, &p.drop)) return rc;
  const int dm = (drop_mode == MMA_DROP_HASH && drop_thr == 0) ? MMA_DROP_NONE : p.drop.mode;          // HASH or HASH16 by the threshold
  const bool save = T != nullptr;
  if (crow) {
    const int64_t crow_len = fill_sel_slots(kind_host, K, H, &p.sel_slots);
    MMA_REQUIRE(ldc >= crow_len && ldc % 4 == 0 && aligned16(crow), "crow needs a 16-byte aligned pitch >= %lld floats (mma_nc_crow_floats)",
                (long long)crow_len);
  }
  const bool v4 = (H % 4 == 0) && (ldx % 4 == 0) && (ldp % 4 == 0) && (ldq % 4 == 0) && (!save || ldt % 4 == 0) && aligned16(x) &&
                  aligned16(P) && aligned16(Q) && (!m || aligned16(m)) && (!m_sum || (aligned16(m_sum) && ldms % 4 == 0)) &&
                  (!save || (aligned16(T) && (!sel || aligned16(sel)))) && (partial == nullptr || aligned16(partial));
  const Geometry g = geometry(H, v4);
  p.x = x; p.ldx = ldx; p.P = P; p.ldp = ldp; p.Q = Q; p.ldq = ldq; p.rowptr = rowptr; p.col = col;
  // MEASUREMENT ONLY (round-3 VERDICT item 6, "P inside K1": what would K1 gain if it did not have to read P?): MMA_NC_ABLATE_P=1 makes
  // every target read P's row 0 (a cache hit) - the results are WRONG, only the kernel's time means anything.  DESIGN.md 9 has the number.
  static const bool ablate_p = getenv("MMA_NC_ABLATE_P") && atoi(getenv("MMA_NC_ABLATE_P")) != 0;
  if (ablate_p) p.ldp = 0;
  p.items = reinterpret_cast<const int4*>(items); p.n_items = n_items;
  p.partial = partial; p.pstride = 2LL * K * H;
  p.m = m; p.m_kstride = N * (int64_t)H; p.msum = m_sum; p.ldms = ldms; p.T = T; p.sel = sel; p.ldt = ldt;
  p.crow = crow; p.ldc = ldc;
  p.H = H; p.HQ = (H + 3) / 4; p.K_total = K; p.lpr_log = g.lpr_log; p.kinds = kinds; p.acts = acts;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // items [0, n_wave_items): one per wavefront; items [n_wave_items, n_items): one per LPR-lane group (short segments)
  const int ipw = kWave >> g.lpr_log;
  if (ipw == 1 || n_wave_items > n_items) n_wave_items = n_items;
  const int4* all_items = p.items;
  {
    NcSmallPlan sp; dim3 sgrid;
    if (small_plan(sync, K, g, dm, n_items, n_wave_items, n_hubs, n_slots, n_hubs * ((H + 3) / 4), &sp, &sgrid)) {
      p.k_base = 0;
      p.hubs = reinterpret_cast<const int4*>(hubs); p.n_hubs = n_hubs; p.sync = reinterpret_cast<unsigned*>(sync); p.n_slots = (unsigned)n_slots;
      switch (K) {
        case 1: launch_fwd_small<1>(p, sp, sgrid, save, dm, st); break;
        case 2: launch_fwd_small<2>(p, sp, sgrid, save, dm, st); break;
        case 3: launch_fwd_small<3>(p, sp, sgrid, save, dm, st); break;
        case 4: launch_fwd_small<4>(p, sp, sgrid, save, dm, st); break;
        default: launch_fwd_small<8>(p, sp, sgrid, save, dm, st); break;
      }
      return check_launch("nc_fwd_small_kernel");
    }
  }
  for (int part = 0; part < 2; ++part) {
    const int64_t cnt = part == 0 ? n_wave_items : n_items - n_wave_items;
    if (cnt <= 0) continue;
    p.items = all_items + (part == 0 ? 0 : n_wave_items);
    p.n_items = cnt;
    const dim3 grid = item_grid(cnt, g.chunks, part == 0 ? 1 : ipw);
    for (int k0 = 0; k0 < K;) {
      const int ks = next_slice(K - k0);
      p.k_base = k0;
      if (g.vec == 4) { if (part == 0) launch_fwd_k<4, false>(ks, p, grid, save, dm, st); else launch_fwd_k<4, true>(ks, p, grid, save, dm, st); }
      else { if (part == 0) launch_fwd_k<1, false>(ks, p, grid, save, dm, st); else launch_fwd_k<1, true>(ks, p, grid, save, dm, st); }
      k0 += ks;
    }
  }
  if (int rc = check_launch("nc_fwd_kernel")) return rc;
  if (n_hubs > 0) {
    const int per_row = (H + g.vec - 1) / g.vec;
    const dim3 fg((unsigned)elementwise_grid(n_hubs * per_row));
    const int4* hb = reinterpret_cast<const int4*>(hubs);
    if (g.vec == 4) {
      if (save) hipLaunchKernelGGL((nc_fwd_finalize_kernel<4, true>), fg, dim3(kBlock), 0, st, p, hb, n_hubs);
      else hipLaunchKernelGGL((nc_fwd_finalize_kernel<4, false>), fg, dim3(kBlock), 0, st, p, hb, n_hubs);
    } else {
      if (save) hipLaunchKernelGGL((nc_fwd_finalize_kernel<1, true>), fg, dim3(kBlock), 0, st, p, hb, n_hubs);
      else hipLaunchKernelGGL((nc_fwd_finalize_kernel<1, false>), fg, dim3(kBlock), 0, st, p, hb, n_hubs);
    }
    if (int rc = check_launch("nc_fwd_finalize_kernel")) return rc;
  }
  return 0;
}

extern "C" int64_t mma_nc_crow_floats(int32_t H, int32_t K, const uint8_t* kind_host) {
  if (H < 1 || K < 1 || K > MMA_MAX_K || !kind_host) return -1;
  for (int k = 0; k < K; ++k) if (kind_host[k] > MMA_KIND_SOFTMIN) return -1;
  uint32_t slots;
  return fill_sel_slots(kind_host, K, H, &slots);
}

extern "C" int mma_nc_bwd_node(
    const float* g, int64_t g_kstride, int64_t ldgr, const uint8_t* sel, const float* T, int64_t ldt, const int32_t* rowptr,
    float* gs, int64_t ldgs, const float* crow, int64_t ldc, float* gP, int64_t ldgp, float* gxs, int64_t ldgx, float* row_max,
    int64_t N, int32_t H, int32_t K, const uint8_t* kind_host, void* stream) {
  MMA_REQUIRE(N >= 0 && N < (1LL << 31) && H >= 1 && K >= 1 && K <= MMA_MAX_K, "N=%lld H=%d K=%d unsupported", (long long)N, H, K);
  MMA_REQUIRE(ldt >= (int64_t)K * H && (!gs || ldgs >= (int64_t)K * H) && ldgp >= (int64_t)K * H && ldgx >= H && ldgr >= H &&
              g_kstride >= 0, "row pitch too small");
  if (N == 0) return 0;
  MMA_REQUIRE(g && (sel || crow) && T && rowptr && gP && gxs && kind_host, "NULL argument (the selection state is sel or crow)");
  uint32_t kinds, acts;
  if (int rc = pack_codes(kind_host, nullptr, K, &kinds, &acts)) return rc;
  NcBwdNodeParams p{g, g_kstride, ldgr, sel, T, ldt, rowptr, gs, ldgs, gP, ldgp, gxs, ldgx, N, H, K, kinds};
  p.rowmax = reinterpret_cast<uint32_t*>(row_max);
  p.crow = crow; p.ldc = ldc; p.HQ = (H + 3) / 4;
  const int64_t crow_len = fill_sel_slots(kind_host, K, H, &p.sel_slots);
  MMA_REQUIRE(!crow || (ldc >= crow_len && ldc % 4 == 0 && aligned16(crow)),
              "crow needs a 16-byte aligned pitch >= %lld floats (mma_nc_crow_floats)", (long long)crow_len);
  const bool v4 = (H % 4 == 0) && (ldt % 4 == 0) && (!gs || ldgs % 4 == 0) && (ldgp % 4 == 0) && (ldgx % 4 == 0) &&
                  (ldgr % 4 == 0) && (g_kstride % 4 == 0) && aligned16(g) && (crow || aligned16(sel)) &&
                  aligned16(T) && (!gs || aligned16(gs)) && aligned16(gP) && aligned16(gxs);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int per_row = v4 ? H / 4 : H;
  int64_t nblk = (N * per_row + kBlock - 1) / kBlock;          // one item per thread (see the kernel): no grid-stride in practice
  if (nblk > (1LL << 30)) nblk = 1LL << 30;
  const dim3 grid((unsigned)nblk);
  if (v4) hipLaunchKernelGGL((nc_bwd_node_kernel<4>), grid, dim3(kBlock), 0, st, p);
  else hipLaunchKernelGGL((nc_bwd_node_kernel<1>), grid, dim3(kBlock), 0, st, p);
  return check_launch("nc_bwd_node_kernel");
}

extern "C" int mma_nc_fused_bwd(
    const float* x, int64_t ldx, const float* P, int64_t ldp, const float* Q, int64_t ldq,
    const float* gs, int64_t ldg, const float* g, int64_t ldgg, const float* crow, int64_t ldc, const uint8_t* kind_host,
    const float* gxs, int64_t ldgx, const float* T, int64_t ldt, float* gP, int64_t ldgp, int64_t n_targets,
    const int32_t* t_col, const int32_t* t_eid,
    const int32_t* items, int64_t n_items, int64_t n_wave_items, const int32_t* hubs, int64_t n_hubs,
    float* partial, int64_t n_slots, float* gQ, int64_t ldgq, float* gx, int64_t ldgxo, float* row_max,
    int64_t N, int64_t E, int32_t H, int32_t K, const uint8_t* act_host,
    int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, int64_t drop_edge_base, const uint8_t* keep,
    int32_t* sync, void* stream) {
  MMA_REQUIRE(N >= 0 && E >= 0 && N < (1LL << 31) && E < (1LL << 31), "N=%lld E=%lld out of int32 range", (long long)N, (long long)E);
  MMA_REQUIRE(H >= 1 && K >= 1 && K <= MMA_MAX_K, "H=%d K=%d unsupported (1<=K<=%d)", H, K, MMA_MAX_K);
  MMA_REQUIRE(ldx >= H && ldp >= (int64_t)K * H && ldq >= (int64_t)K * H && (!gs || ldg >= (int64_t)K * H) && ldgq >= (int64_t)K * H &&
              ldgxo >= H, "row pitch too small");
  MMA_REQUIRE(ldx < (1LL << 31) && ldp < (1LL << 31) && ldq < (1LL << 31) && ldg < (1LL << 31) && ldgg < (1LL << 31) && ldc < (1LL << 31) &&
              ldt < (1LL << 31) && ldgp < (1LL << 31) && ldgq < (1LL << 31) && ldgx < (1LL << 31) && ldgxo < (1LL << 31), "row pitch out of range");
  MMA_REQUIRE(n_items >= 0 && n_hubs >= 0 && n_slots >= 0 && n_items < (1LL << 31) && n_wave_items >= 0, "negative or oversize item counts");
  MMA_REQUIRE(n_slots == 0 || (partial != nullptr && hubs != nullptr && n_hubs > 0), "hub slots without partial/hubs buffers");
  if (N == 0 || n_items == 0) return 0;       // an empty shard: nothing to do (its zero-row buffers are NULL)
  const bool shared = gs == nullptr;
  const bool epi = T != nullptr;
  MMA_REQUIRE(!shared || (g && crow && kind_host && ldgg >= H),
              "give gs (n_targets,K*H), or the shared-gradient form: g (n_targets,H), the code rows crow K1 wrote, and kinds");
  MMA_REQUIRE(!epi || (shared && gP && ldt >= (int64_t)K * H && ldgp >= (int64_t)K * H && n_targets >= 0 && n_targets <= N),
              "the fused node-level epilogue (T given) needs the shared-gradient form, gP and 0 <= n_targets <= N");
  MMA_REQUIRE(epi || (gxs && ldgx >= H), "gxs (N,H) from mma_nc_bwd_node is needed unless the epilogue is fused (T given)");
  MMA_REQUIRE(x && P && Q && items && gQ && gx && act_host, "NULL argument");
  MMA_REQUIRE(E == 0 || (t_col != nullptr && t_eid != nullptr), "NULL transposed CSR");
  MMA_REQUIRE(aligned16(items) && (hubs == nullptr || aligned16(hubs)), "items/hubs must be 16-byte aligned int32 quadruples");
  uint32_t kinds, acts;
  if (int rc = pack_codes(shared ? kind_host : nullptr, act_host, K, &kinds, &acts)) return rc;
  NcBwdParams p{};
  if (int rc = make_drop(drop_mode, drop_thr, seed, seed_dev, drop_edge_base, keep, E, &p.drop)) return rc;
  const int dm = (drop_mode == MMA_DROP_HASH && drop_thr == 0) ? MMA_DROP_NONE : p.drop.mode;          // HASH or HASH16 by the threshold
  const bool v4 = (H % 4 == 0) && (ldx % 4 == 0) && (ldp % 4 == 0) && (ldq % 4 == 0) && (epi || (ldgx % 4 == 0 && aligned16(gxs))) && (ldgq % 4 == 0) &&
                  (shared ? (ldgg % 4 == 0 && aligned16(g) && ldc % 4 == 0 && aligned16(crow)) : (ldg % 4 == 0 && aligned16(gs))) &&
                  (!epi || (ldt % 4 == 0 && aligned16(T) && ldgp % 4 == 0 && aligned16(gP))) &&
                  (ldgxo % 4 == 0) && aligned16(x) && aligned16(P) && aligned16(Q) &&
                  aligned16(gQ) && aligned16(gx) && (partial == nullptr || aligned16(partial));
  const Geometry geo = geometry(H, v4);
  p.x = x; p.ldx = ldx; p.P = P; p.ldp = ldp; p.Q = Q; p.ldq = ldq; p.gs = gs; p.ldg = ldg; p.gxs = gxs; p.ldgx = ldgx;
  p.g = g; p.ldgg = ldgg; p.crow = crow; p.ldc = ldc; p.kinds = kinds;
  p.T = T; p.ldt = ldt; p.gP = gP; p.ldgp = ldgp; p.n_targets = n_targets;
  if (shared) {
    const int64_t crow_len = fill_sel_slots(kind_host, K, H, &p.sel_slots);
    MMA_REQUIRE(ldc >= crow_len, "ldc=%lld < %lld", (long long)ldc, (long long)crow_len);
  }
  p.t_col = t_col; p.t_eid = t_eid; p.items = reinterpret_cast<const int4*>(items); p.n_items = n_items;
  p.partial = partial; p.pstride = (int64_t)(K + 1) * H; p.gQ = gQ; p.ldgq = ldgq; p.gx = gx; p.ldgxo = ldgxo;
  p.rowmax = reinterpret_cast<uint32_t*>(row_max);
  p.H = H; p.HQ = (H + 3) / 4; p.K_total = K; p.lpr_log = geo.lpr_log; p.acts = acts;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ipw = kWave >> geo.lpr_log;
  if (ipw == 1 || n_wave_items > n_items) n_wave_items = n_items;
  const int4* all_items = p.items;
  {
    NcSmallPlan sp; dim3 sgrid;
    if (shared && small_plan(sync, K, geo, dm, n_items, n_wave_items, n_hubs, n_slots, n_hubs * (K + 1) * ((H + 3) / 4), &sp, &sgrid)) {
      p.k_base = 0; p.first_pass = 1;
      p.hubs = reinterpret_cast<const int4*>(hubs); p.n_hubs = n_hubs; p.sync = reinterpret_cast<unsigned*>(sync); p.n_slots = (unsigned)n_slots;
      switch (K) {
        case 1: launch_bwd_small<1>(p, sp, sgrid, dm, st); break;
        case 2: launch_bwd_small<2>(p, sp, sgrid, dm, st); break;
        case 3: launch_bwd_small<3>(p, sp, sgrid, dm, st); break;
        case 4: launch_bwd_small<4>(p, sp, sgrid, dm, st); break;
        default: launch_bwd_small<8>(p, sp, sgrid, dm, st); break;
      }
      return check_launch("nc_bwd_small_kernel");
    }
  }
  for (int part = 0; part < 2; ++part) {
    const int64_t cnt = part == 0 ? n_wave_items : n_items - n_wave_items;
    if (cnt <= 0) continue;
    p.items = all_items + (part == 0 ? 0 : n_wave_items);
    p.n_items = cnt;
    const dim3 grid = item_grid(cnt, geo.chunks, part == 0 ? 1 : ipw);
    for (int k0 = 0; k0 < K;) {
      const int ks = next_slice(K - k0);
      p.k_base = k0; p.first_pass = (k0 == 0);
      if (geo.vec == 4) { if (part == 0) launch_bwd_k<4, false>(ks, p, grid, dm, st); else launch_bwd_k<4, true>(ks, p, grid, dm, st); }
      else { if (part == 0) launch_bwd_k<1, false>(ks, p, grid, dm, st); else launch_bwd_k<1, true>(ks, p, grid, dm, st); }
      k0 += ks;
    }
  }
  if (int rc = check_launch("nc_bwd_kernel")) return rc;
  if (n_hubs > 0) {
    const int per_row = (H + geo.vec - 1) / geo.vec;
    const dim3 fg((unsigned)elementwise_grid(n_hubs * (K + 1) * per_row));
    const int4* hb = reinterpret_cast<const int4*>(hubs);
    if (geo.vec == 4) hipLaunchKernelGGL((nc_bwd_finalize_kernel<4>), fg, dim3(kBlock), 0, st, p, hb, n_hubs);
    else hipLaunchKernelGGL((nc_bwd_finalize_kernel<1>), fg, dim3(kBlock), 0, st, p, hb, n_hubs);
    if (int rc = check_launch("nc_bwd_finalize_kernel")) return rc;
  }
  return 0;
}
