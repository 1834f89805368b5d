/* mma_amd.h - C ABI of libmma_amd.so: MI355X (gfx950) kernels for the MMA message-passing hot path.
 *
 * The reference (asarigun/mma) is pure Python with no FFI layer; its hot path sits behind two
 * nn.Module surfaces.  These entry points are what a binding for that path binds:
 * each one replaces the tensor-op sequence cited next to it (file:line under /root/reference).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the caller owns all memory;
 *   - no hidden allocation, no host synchronisation: work is enqueued on `stream` (a hipStream_t
 *     passed as void*, NULL = the null stream) and the call returns immediately;
 *   - returns 0 on success; otherwise a nonzero code, and mma_last_error() describes it
 *     (argument checks run on the host BEFORE any launch: a bad shape never reaches the GPU);
 *   - all floating point data is fp32, all indices int32, row-major; `ld*` are row pitches in elements;
 *   - stateless and re-entrant (mma_last_error is thread-local).
 */
#ifndef MMA_AMD_H
#define MMA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMA_ABI_VERSION 35
#define MMA_MAX_K 8          /* masks fused per launch; more are issued as several launches */

/* combine kinds of the node-classification aggregators (layers.py:201-728) */
enum {
  MMA_KIND_SUM = 0,      /* m = x_i + s            learnable_sum*   layers.py:221 */
  MMA_KIND_MEAN = 1,     /* m = (x_i + s) / d_i    learnable_mean*  layers.py:326-329 */
  MMA_KIND_MAX = 2,      /* m = max(x_i, s)        learnable_max*   layers.py:452 */
  MMA_KIND_MIN = 3,      /* m = min(x_i, s)        learnable_min*   layers.py:562 */
  MMA_KIND_SOFTMAX = 4,  /* m = (e/e)*s, e=exp(s)  learnable_softmax layers.py:676-682 */
  MMA_KIND_SOFTMIN = 5   /* m = (e/e)*s, e=exp(-s) learnable_softmin layers.py:716-720 */
};
/* mask activation: sigmoid, or the raw logits the "new_sigmoid" quirk leaves (layers.py:381-385) */
enum { MMA_ACT_SIGMOID = 0, MMA_ACT_RAW = 1 };

/* dropout of the mask (F.dropout(mask0, p), training=True always: layers.py:219).
 *   mode NONE: p == 0.  mode HASH: counter-based keep bits, DESIGN.md "dropout RNG"; every `drop_thr` of this header is a
 *   16-bit threshold (ABI 34): P(drop) = drop_thr / 65536, 0..65535, survivors scale by 65536 / (65536 - drop_thr) - any p to
 *   2^-17.  mode EXPLICIT: keep[(k*E + e)*H + h] in {0,1} supplied by the caller (parity tests), scale by the same formula. */
enum { MMA_DROP_NONE = 0, MMA_DROP_HASH = 1, MMA_DROP_EXPLICIT = 2 };

int mma_abi_version(void);
const char* mma_last_error(void);
/* SHA-256 (16 hex digits) over the kernel sources THIS library was built from (csrc/Makefile -> build_stamp.h; ABI 29): the identity a
 * recorded measurement carries (tools/build_stamp.py), taken from the .so that ran, not from re-hashing the tree. */
const char* mma_build_stamp(void);

/* ---- K1: fused masked message + K-aggregator segmented reduce, node classification form ----------
 * Replaces, for all K selected aggregators at once, the per-node Python loop of
 * layers.py:205-226 (gather / tile / cat / mm / sigmoid / dropout / mul / sum / combine):
 *     z_k(i,j) = P[i, k*H:(k+1)*H] + Q[j, k*H:(k+1)*H]        ( = [x_i || x_j] @ W_k, split as
 *                P = x @ W_k[:H], Q = x @ W_k[H:], computed once per layer by a dense GEMM )
 *     s_k[i]   = sum_{j in N(i)} drop(act_k(z_k(i,j))) * x_j
 *     m[k,i,:] = combine_k(x_i, s_k[i], d_i)
 * Work is described by `items`: int32 quadruples {node, edge_begin, edge_end, slot}. slot < 0: the
 * item covers the node's whole CSR segment and writes m directly. slot >= 0: the item is one chunk
 * of a hub node; it writes partial sums to `partial[slot]` and `hubs` {node, slot_begin, slot_end, 0}
 * lists the nodes whose chunks are then summed in slot order (deterministic) by a second launch.
 * When T and sel are non-NULL (training) the kernel also saves, per node, what backward needs:
 *     T[i, k*H+h]   = sum_j drop * act_k'(z) * x_j[h]          (so grad_P = gs * T needs no edge pass)
 *     sel[i, k*H+h] = 0: x_i selected  1: s selected  2: tie (0.5/0.5, torch.max/min backward)  3: NaN
 * and/or (ABI 27) the packed code row the shared-gradient backward gathers per edge:
 *     crow[i] = [ 1/max(d_i,1), 0, 0, 0 | for every max/min/softmax/softmin mask, in mask order: ceil(H/4) words of bytes
 *                 = 2 * dm/ds of the element (0: x_i selected, 1: tie, 2: s selected, 255: NaN gradient) ]
 *     (round 2 had K2a write a [g | 1/d | codes] row per target: 0.8 GB per step at C4 that K1 now leaves in place of `sel`).
 */
int mma_nc_fused_fwd(
    const float* x, int64_t ldx,                 /* (n_src,H): rows [0,N) are the targets, the rest source-only (halo) */
    const float* P, int64_t ldp,                 /* (N,K*H)     P[i, k*H+h] = (x_i @ W_k[:H])[h] */
    const float* Q, int64_t ldq,                 /* (n_src,K*H) Q[j, k*H+h] = (x_j @ W_k[H:])[h] */
    const int32_t* rowptr,                       /* (N+1) CSR by target: d_i = rowptr[i+1]-rowptr[i] */
    const int32_t* col,                          /* (E) source node of each edge, target-major */
    const int32_t* items, int64_t n_items,       /* (n_items,4) */
    int64_t n_wave_items,                        /* items [0,n_wave_items) run one per wavefront (long segments), the rest one per
                                                    H/4-lane group, 64/(H/4) per wavefront (short segments: more items in flight);
                                                    pass n_items to run all per wavefront.  Speed only, results are identical. */
    const int32_t* hubs, int64_t n_hubs,         /* (n_hubs,4), may be NULL when n_hubs == 0 */
    float* partial, int64_t n_slots,             /* (n_slots, 2, K, H) scratch, NULL when n_slots == 0 */
    float* m,                                    /* (K,N,H) out: m[k] = learnable_<k>(x); may be NULL if m_sum is given */
    float* m_sum, int64_t ldms,                  /* (N,H) out: sum_k m[k] (all MMA.forward needs, since
                                                    sum_k A (m_k W) = A ((sum_k m_k) W)); may be NULL */
    float* T, uint8_t* sel, int64_t ldt,         /* (N,K*H) out; T NULL: nothing is saved; with T give sel, crow, or both */
    float* crow, int64_t ldc,                    /* (N,ldc) out or NULL; ldc >= mma_nc_crow_floats(H, K, kinds), a multiple of 4 */
    int64_t N, int64_t E, int32_t H, int32_t K,
    const uint8_t* kind_host, const uint8_t* act_host,   /* K codes each, HOST memory */
    int32_t drop_mode, uint32_t drop_thr, uint64_t seed,
    const uint64_t* seed_dev,                    /* optional DEVICE seed (overrides `seed`): a hipGraph replay then draws fresh
                                                    dropout bits without re-capture, the host only rewrites 8 bytes */
    int64_t drop_edge_base,                      /* HASH key = edge position + base */
    const uint8_t* keep,                         /* EXPLICIT: (K,E,H), else NULL */
    int32_t* sync,                               /* optional DEVICE counter, zero before the first call and left zero by every call:
                                                    with it a small plan (H % 4 == 0, H <= 256, K in {1,2,3,4,8}, no EXPLICIT mask,
                                                    few hubs) runs as ONE launch - wave items, grouped items and the hub sums, the
                                                    last workgroup to finish doing the latter.  Same results.  One counter per
                                                    stream of calls (calls sharing it must not overlap). */
    void* stream);

/* ---- K2a: node-level backward of the combine (element-wise) -------------------------------------
 * From g = dL/dm (K,N,H): gs = dL/ds (N,K*H), gP = gs * T (N,K*H) = dL/dP, gxs = sum_k dL/dx_i
 * through the combine (N,H).  Mirrors autograd of layers.py:221,326-329,452,562 (ties split 0.5/0.5).
 * gs may be NULL (shared-gradient form: K2b rebuilds it on the fly).  The selection state is read from `sel`, or - when crow is
 * given - from the packed code rows K1 wrote.  (With the epilogue of mma_nc_fused_bwd fused, this call is not needed at all.) */
int mma_nc_bwd_node(
    const float* g, int64_t g_kstride, int64_t ldgr, /* g[k*g_kstride + i*ldgr + h]; g_kstride = 0: one (N,H) gradient for all k */
    const uint8_t* sel, const float* T, int64_t ldt, const int32_t* rowptr,
    float* gs, int64_t ldgs,                     /* (N,K*H) out, or NULL */
    const float* crow, int64_t ldc,              /* the code rows of mma_nc_fused_fwd instead of sel, or NULL */
    float* gP, int64_t ldgp, float* gxs, int64_t ldgx,
    float* row_max,                              /* optional (N,) in/out, zeroed by the caller: row_max[i] = max(row_max[i], max |gP[i,:]|)
                                                    by atomicMax on the bit pattern - with K2b's contribution the row scale of the
                                                    three-product dL/dx GEMM (mma_gemm_f16x2_n128) */
    int64_t N, int32_t H, int32_t K, const uint8_t* kind_host, void* stream);
int64_t mma_nc_crow_floats(int32_t H, int32_t K, const uint8_t* kind_host);   /* row length of crow in floats (-1: bad arguments) */

/* ---- K2b: edge-level backward, source-major (no atomics, deterministic) ---------------------------
 * Walks the TRANSPOSED CSR (edges grouped by source j): t_col[e'] = target i, t_eid[e'] = position of
 * that edge in the forward CSR (keys the dropout bits).  Per source j:
 *     gQ[j, k*H+h] = x_j[h] * sum_i gs[i,k,h] * drop * act_k'(z_k(i,j))
 *     gx[j, h]     = gxs[j,h] + sum_i sum_k gs[i,k,h] * drop * act_k(z_k(i,j))
 * items/hubs/partial as in mma_nc_fused_fwd but over the transposed segments; partial is
 * (n_slots, K+1, H).  N here is the number of SOURCE rows (n_src); gs, g, crow, T, gP and P have one row per target.
 * Three forms:
 *   gs given                      : gathers the materialised dL/ds rows (any upstream gradient; K2a made them);
 *   gs NULL (shared gradient)     : all masks share ONE upstream gradient g (n_targets,H) (MMA.forward: the gradient of sum_k m_k);
 *                                   gs_k[i] is rebuilt per edge from g[i] and crow[i]: (1 + K_sel/4) H floats + 16 B gathered per
 *                                   edge instead of K H.  gxs comes from mma_nc_bwd_node;
 *   gs NULL and T given (ABI 27)  : the same, and the node-level backward (K2a) runs in the per-source epilogue: source j <
 *                                   n_targets is also target j, so gP[j] = g[j] dm/ds T[j] and the direct term sum_k g[j] dm/dx_i
 *                                   are formed where gx[j] is stored - mma_nc_bwd_node, its launch and the gxs round trip go away;
 *                                   row_max then receives max(|gP[j,:]|, |gQ[j,:]|).  Sources >= n_targets (halo rows) have no
 *                                   target role. */
int mma_nc_fused_bwd(
    const float* x, int64_t ldx, const float* P, int64_t ldp, const float* Q, int64_t ldq,
    const float* gs, int64_t ldg,                /* (n_targets,K*H) from K2a, or NULL for the shared-gradient forms: */
    const float* g, int64_t ldgg,                /*   the upstream gradient (n_targets,H), */
    const float* crow, int64_t ldc,              /*   the code rows K1 wrote, */
    const uint8_t* kind_host,                    /*   and the K kinds (HOST) */
    const float* gxs, int64_t ldgx,              /* (N,H) from K2a; unused (may be NULL) when T is given */
    const float* T, int64_t ldt,                 /* NULL, or K1's saved T (n_targets,K*H): fuse the node-level backward */
    float* gP, int64_t ldgp, int64_t n_targets,  /*   -> gP (n_targets,K*H) out; 0 <= n_targets <= N */
    const int32_t* t_col, const int32_t* t_eid,
    const int32_t* items, int64_t n_items, int64_t n_wave_items, const int32_t* hubs, int64_t n_hubs,
    float* partial, int64_t n_slots,
    float* gQ, int64_t ldgq, float* gx, int64_t ldgxo,
    float* row_max,                              /* optional (n_src,), as in mma_nc_bwd_node: row_max[j] = max(row_max[j], max |gQ[j,:]|) */
    int64_t N, int64_t E, int32_t H, int32_t K,
    const uint8_t* act_host,
    int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, int64_t drop_edge_base, const uint8_t* keep,
    int32_t* sync,                               /* optional, as in mma_nc_fused_fwd (shared-gradient form only): ONE launch */
    void* stream);

/* ---- K5: CSR SpMM over a K-times column-stacked adjacency ----------------------------------------
 * out[i,:] = bias + sum_{k<K} sum_{e in row i} val[e] * B[k*rows_per_block + col[e], :]
 * = torch.spmm(cat((adj,)*K, 1), support) + bias            (layers.py:861-865; K=1: layers.py:41)
 * val may be NULL (all ones); bias may be NULL. */
int mma_csr_spmm(
    const int32_t* rowptr, const int32_t* col, const float* val,
    const float* B, int64_t ldb, int64_t rows_per_block, int32_t K,
    const float* bias, float* out, int64_t ldo,
    int64_t n_rows, int32_t C, void* stream);
/* the same, and row_max[i] = max(row_max[i], max_c |out[i,c]|) (ABI 34; row_max (n_rows,) zeroed or pre-filled by the caller; the bits of
 * a non-negative float merged by atomicMax): the dV segment sum of graph regression's backward leaves the row scales of the GEMMs behind it */
int mma_csr_spmm_rm(
    const int32_t* rowptr, const int32_t* col, const float* val,
    const float* B, int64_t ldb, int64_t rows_per_block, int32_t K,
    const float* bias, float* out, int64_t ldo,
    int64_t n_rows, int32_t C, float* row_max, void* stream);

/* K5, item-driven form for K = 1 (the form MMA.forward and GraphConvolution use): rows cut into work items
 * {row, ebeg, eend, slot} (longest first) with hub partials, exactly as in mma_nc_fused_fwd; partial is (n_slots, C). */
int mma_csr_spmm_items(
    const int32_t* col, const float* val, const float* B, int64_t ldb, const float* bias, float* out, int64_t ldo,
    const int32_t* items, int64_t n_items,
    int64_t n_wave_items,                        /* as in mma_nc_fused_fwd: the head of the (longest-first) list runs one item per
                                                    wavefront, the short tail one per C/4-lane group; speed only */
    const int32_t* hubs, int64_t n_hubs, float* partial, int64_t n_slots,
    int32_t C, void* stream);

/* ---- GEMM-pre/post: fp32-accurate tall-skinny GEMM on the bf16 matrix cores ("bf16x3") -------------------
 * C (M,N) = A (M,K) @ B (K,N) for M >> N,K: the x @ W_k products of layers.py:215-216 hoisted out of the node loop
 * (P = x Wtop, Q = x Wbot) and their dL/dx.  gfx950 has no TF32/xf32 and its fp32-input MFMA runs at the vector
 * rate, so A and B are split exactly into three bf16 pieces each and the six significant piece products are
 * accumulated in fp32 by v_mfma_f32_32x32x16_bf16: accuracy of an fp32 GEMM at 6/16 of the fp32 matrix-core time.
 * Bt3: (3,N,K) bf16 = the three pieces of B^T (k contiguous), made by mma_split_bf16x3 from a row-major (N,K) fp32
 * matrix.  Requires N % 32 == 0, K % 128 == 0, (K == 128 or N <= 128) and row pitches < 2^24 floats.  accumulate = 1
 * adds the product to C with ONE fp32 addition per element (bit-identical to C + (A B); on the N == 128 long-K path the
 * addition is done by the L2 atomic unit - every element has exactly one writer, so the result is deterministic). */
int mma_split_bf16x3(const float* in, int64_t n, void* out_3n_bf16, void* stream);
int mma_gemm_bf16x3(const float* A, int64_t lda, const void* Bt3, float* C, int64_t ldc,
                    int64_t M, int32_t N, int32_t K, int32_t accumulate /* 0: C = A B, 1: C += A B */, void* stream);
/* out[i] = max_j |A[i,j]| of a (M, cols) fp32 matrix (a NaN or inf in the row is returned as such): the row maxima the
 * three-product kernels below scale by, for callers whose producers do not leave them (one pass over A, 32 lanes per row). */
int mma_row_absmax(const float* A, int64_t lda, int64_t M, int32_t cols, float* out, void* stream);
/* The B operand of the three-product kernels below from w (K,N) fp32 with element strides (stride_k, stride_n) - a transposed view
 * is fine: Bt2 = (2, N, K) fp16, piece 0 = hi and piece 1 = lo * 2^11 of w^T scaled per column by the power of two that puts the
 * column maximum into [2^14, 2^15); col_unscale (N,) = the reciprocal scales.  One launch.  plain_lo != 0 (ABI 30): piece 1 = lo itself,
 * not pre-scaled - the operand of the one-accumulator kernels (mma_gemm_f16x2_nlp). */
int mma_split_f16x2(const float* w, int64_t stride_k, int64_t stride_n, int32_t K, int32_t N, void* bt2, float* col_unscale,
                    int32_t plain_lo, void* stream);
/* Three-product form for K = 128 (the forward [P|Q] = x [Wtop|Wbot] and every other tall product whose reduction fits one
 * 128-wide chunk): fp16 x 2 pieces, a = s_row (a_hi + 2^-11 a_lo), b = s_col (b_hi + 2^-11 b_lo), a b ~= hi hi + 2^-11 (hi lo +
 * lo hi) - half the MFMAs of the six-product bf16 form at the accuracy of an fp32 GEMM (measured 1.1e-7 sum|a||b|).  The
 * power-of-two row scales of A are formed in the kernel; the CALLER prepares B: Bt2 = (2, N, 128) fp16, piece 0 = hi and
 * piece 1 = lo * 2^11 of B^T scaled per column by a power of two that puts the column maximum into [2^14, 2^15), and
 * col_unscale (N,) fp32 = the reciprocal of that scale.  N % 128 == 0, N <= 4096; any M (ragged tails handled inside). */
int mma_gemm_f16x2(const float* A, int64_t lda, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                   float* a_row_max,             /* optional (M,) out: max |a| of every row (the kernel forms it for its row scales
                                                    anyway) - the x_row_max of the weight-gradient product mma_gemm_f16x2_tn */
                   int64_t M, int32_t N, void* stream);
/* ABI 35: the same product for K in {64, 96, 128} (Bt2 (2, N, K), lda >= K): the zero-padded tall Linears of graph regression (50 + 1 ->
 * 64 columns, 75 + 1 -> 96) stop loading, splitting and multiplying pad columns up to 128; the narrow B slabs leave room for THREE
 * resident column groups per workgroup (N / 128 a multiple of 3: A is read once for 384 output columns). */
int mma_gemm_f16x2_k(const float* A, int64_t lda, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                     float* a_row_max, int64_t M, int32_t N, int32_t K, void* stream);
/* The three-product column-group form for K = 256 (the forward product of hidden width 256: C5): as mma_gemm_f16x2 with Bt2 = (2, N, 256),
 * but the row scales come from the caller's row_max (M,) >= max |a| of every row (mma_row_absmax, or a producer's bound; 0 marks an
 * all-zero row): 256 floats per row do not fit the registers beside their own fp16 pieces while an in-kernel maximum forms. */
int mma_gemm_f16x2_k256(const float* A, int64_t lda, const float* row_max, const void* Bt2, const float* col_unscale, float* C,
                        int64_t ldc, int64_t M, int32_t N, void* stream);
/* ABI 35: the same product with A PACKED once (the 32 column groups of hidden width 256 each loaded the rows in the MFMA's fragment shape
 * - 8x the line requests of a coalesced load - and split them again): mma_pack_f16x2_k256 writes, per 32-row unit, the 16 k-steps x 2 fp16
 * pieces in fragment order (Ap: mma_pack_f16x2_k256_bytes(M) bytes, 16-byte aligned), the rows' scale exponents sce (M,) and, optionally,
 * their maxima row_max (M,); mma_gemm_f16x2_k256p multiplies from that.  Bit-equal to mma_gemm_f16x2_k256 given the exact row maxima. */
int64_t mma_pack_f16x2_k256_bytes(int64_t M);
int mma_pack_f16x2_k256(const float* A, int64_t lda, int64_t M, void* Ap, int32_t* sce, float* row_max, void* stream);
int mma_gemm_f16x2_k256p(const void* Ap, const int32_t* sce, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                         int64_t M, int32_t N, void* stream);
/* The same three-product form for N = 128 and a long reduction (dL/dx += [gP|gQ] [Wtop|Wbot]^T, K % 64 == 0): the row scales
 * cannot be formed in the kernel (a row is consumed in 64-wide chunks), so the caller passes row_max (M,) >= the maximum
 * |a| of every row (the backward kernels produce it: mma_nc_bwd_node / mma_nc_fused_bwd); 0 marks an all-zero row.
 * Bt2 = (2, 128, K) fp16 and col_unscale (128,) as for mma_gemm_f16x2.  accumulate != 0: C += A B (a plain read - add - store of the tile;
 * MMA_DX_ACC=atomic in the environment: one float atomic per element, round 3's form - same bits). */
int mma_gemm_f16x2_n128(const float* A, int64_t lda, const float* row_max, const void* Bt2, const float* col_unscale,
                        float* C, int64_t ldc, int64_t M, int32_t K, int32_t accumulate, void* stream);
/* ABI 30: the same product, N = 128 or 256 (hidden width 256: all of dL/dx in ONE pass over [gP|gQ]), K % 128 == 0, K >= 256, with
 * PLAIN lo pieces in both operands (a = s (a_hi + a_lo), Bt2 from mma_split_f16x2 with plain_lo = 1) accumulated into one fp32 tile -
 * the TN kernel's numerics: 22 bits for every element within 2^16 of its row / column maximum, an absolute 2^-25 of the scaled maximum
 * below - which frees the registers for a second chunk of A in flight (the two-accumulator form waits out one memory round trip per
 * 64-deep chunk).  accumulate != 0: C += A B by a plain read - add - store (every element has exactly one writer). */
int mma_gemm_f16x2_nlp(const float* A, int64_t lda, const float* row_max, const void* Bt2, const float* col_unscale,
                       float* C, int64_t ldc, int64_t M, int32_t N, int32_t K, int32_t accumulate, void* stream);
/* ABI 31: the forward product C (M, N) = A (M, K) B for K = 128 or 256 and N a multiple of 256 with N / 256 dividing 32, on the
 * W-STATIONARY kernel: a wave keeps its 32 columns of B in registers for the whole launch, the eight waves of a workgroup load each
 * 64-row (K = 256: 32-row) slice of A together, row-major (whole lines), form the row maxima and power-of-two row scales and split
 * once, and share the fp16 pieces through LDS.  Same products in the same order as mma_gemm_f16x2: bit-identical results.  Bt2 = (2, N, K)
 * and col_unscale (N,) from mma_split_f16x2 (plain_lo = 0).  a_row_max (M,) or NULL: receives max |A[i,:]| (K = 256 callers no longer
 * need the mma_row_absmax pass).  MMA_FWD_WS=1 in the environment makes mma_gemm_f16x2 take this kernel when the shape fits (default: the
 * column-group kernel, which is faster as measured in round 4: 1.06 vs 1.37 ms at M = 2^20, N = 1024).
 * A MEASUREMENT kernel (ABI 34): compiled only with -DMMA_EXPERIMENTAL_FWD (make EXTRA=...); a product library validates the arguments,
 * then returns an error whose text says so, and does not read MMA_FWD_WS. */
int mma_gemm_f16x2_ws(const float* A, int64_t lda, const void* Bt2, const float* col_unscale, float* C, int64_t ldc,
                      float* a_row_max, int64_t M, int32_t N, int32_t K, void* stream);
/* TN form for the weight gradients (autograd's x^T g of layers.py:215-216's torch.mm): C (KA,NC) = X^T G with X (M,KA),
 * G (M,NC) fp32 row-major, C contiguous.  Any 1 <= KA <= 128 and NC >= 1 (ragged tiles are clamped on load and guarded on
 * store; KA % 32 == 0 with NC % 128 == 0 runs without the guards).  Both operands are split to bf16x3 on the
 * fly; the reduction over M runs in fixed row ranges whose partial tiles (ws) are summed in a fixed order.
 * ws: mma_gemm_bf16x3_tn_workspace_floats(M, KA, NC) floats (0 => may be NULL). */
int64_t mma_gemm_bf16x3_tn_workspace_floats(int64_t M, int32_t KA, int32_t NC);
int mma_gemm_bf16x3_tn(const float* X, int64_t ldx, const float* G, int64_t ldg, float* C, float* ws, int64_t ws_floats,
                       int64_t M, int32_t KA, int32_t NC, void* stream);
/* B independent TN products in one launch: C (B,KA,NC), C[b] = X_b^T G_b with X_b = X + b*xb and G_b = G + b*gb - column blocks of
 * wider rows (the per-tower weight gradients of MMAConv's post-NN: mma_tower_post_bwd's gys and K3's aggregates).  Six-product form,
 * same shape rules as mma_gemm_bf16x3_tn; ws: mma_gemm_bf16x3_tn_batched_workspace_floats(M, KA, NC, B) floats. */
int64_t mma_gemm_bf16x3_tn_batched_workspace_floats(int64_t M, int32_t KA, int32_t NC, int32_t B);
int mma_gemm_bf16x3_tn_batched(const float* X, int64_t ldx, int64_t xb, const float* G, int64_t ldg, int64_t gb, float* C, float* ws,
                               int64_t ws_floats, int64_t M, int32_t KA, int32_t NC, int32_t B, void* stream);
/* The TN product in the three-product fp16 x 2 form (half the MFMAs).  The reduction runs over the rows, so the power-of-two scales
 * are per ROW of both operands, balanced between them (x_i 2^a and g_i 2^-a leave x_i g_i unchanged) so that ONE un-scaling serves
 * the whole product; they are derived on the device from the row maxima: x_row_max / g_row_max (M,) >= max |.| of every row of X /
 * G (0 marks an all-zero row), either may be NULL (then one extra pass over that operand forms them; the backward kernels
 * produce G's: mma_nc_bwd_node / mma_nc_fused_bwd).  |error| <= 2^-22 sum|x||g| + M 2^-39 max_i(max|x_i| max|g_i|).  If some row's
 * products lie more than 2^40 below the largest row's, or a row maximum is inf / NaN / subnormal, the six-product kernel
 * (mma_gemm_bf16x3_tn) does the call instead - decided on the device, no synchronisation.  Shapes as mma_gemm_bf16x3_tn, M < 2^30,
 * and KA up to 256 (round 4: eight waves share one staged tile of G, which is then read once for hidden width 256 - same bits as
 * one call per 128-column block of X).
 * ws: mma_gemm_f16x2_tn_workspace_floats(M, KA, NC) floats, 16-byte aligned (never NULL). */
int64_t mma_gemm_f16x2_tn_workspace_floats(int64_t M, int32_t KA, int32_t NC);
int mma_gemm_f16x2_tn(const float* X, int64_t ldx, const float* G, int64_t ldg, const float* x_row_max, const float* g_row_max,
                      float* C, float* ws, int64_t ws_floats, int64_t M, int32_t KA, int32_t NC, void* stream);

/* ---- K7: halo pack / unpack for the 1-D node-sharded multi-GPU path --------------------------------
 * pack:   dst[r,:] = src[idx[r],:]            (send buffer for the all-to-all of halo rows)
 * unpack: dst[idx[r],:] += src[r,:]           (reverse exchange in backward; idx rows of one call are
 *                                              unique, so no atomics are needed) */
int mma_pack_rows(const float* src, int64_t lds, const int32_t* idx, int64_t n_idx,
                  float* dst, int64_t ldd, int32_t width, void* stream);
int mma_unpack_add_rows(const float* src, int64_t lds, const int32_t* idx, int64_t n_idx,
                        float* dst, int64_t ldd, int32_t width, void* stream);
/* the whole reverse exchange in one launch: dst[rows[t],:] += sum_{q in [segptr[t], segptr[t+1])} src[pos[q],:] (fixed order;
 * rows unique; pos indexes the concatenated receive buffer of all peers) - a row read by several peers has one writer */
int mma_unpack_add_rows_csr(const float* src, int64_t lds, const int32_t* rows, const int32_t* segptr, const int32_t* pos,
                            int64_t n_rows, float* dst, int64_t ldd, int32_t width, void* stream);

/* ---- K8: column sums of a tall row-major matrix (bias gradients) ----------------------------------------
 * out[c] = sum_r g[r*ldg + c], r < R, c < C, in a fixed order (two passes over ~2*sqrt(R)-row blocks, no atomics).
 * Replaces the `grad_output.sum(0)` that autograd runs for the bias of every Linear on the path (mma_conv.py:82,99-105,
 * mask_aggr.py:50, layers.py:48,865): torch's column reduction takes 2-7 ms on (2e5..4e5) x 375 because 375 % 4 != 0.
 * ws: mma_col_sum_workspace_floats(R, C) floats (0 => may be NULL). */
int64_t mma_col_sum_workspace_floats(int64_t R, int32_t C);
int mma_col_sum(const float* g, int64_t ldg, int64_t R, int32_t C, float* out, float* ws, int64_t ws_floats,
                void* stream);

/* ---- K9: backward of MMAConv's per-tower post-NN Linear on the aggregates (mma_conv.py:132-134) ------------------------
 * a (N,T,C) contiguous, W (T,O,C) contiguous, gy (N,T,O) contiguous, O <= 16, C % 4 == 0:
 *   ga[n,t,c] = sum_o gy[n,t,o] W[t,o,c]   and   part[b,t,o,c] = sum over the nodes of block b of gy[n,t,o] a[n,t,c],
 * b < n_blocks = mma_tower_linear_bwd_blocks(N); the weight gradient is mma_col_sum over the n_blocks rows of `part`.
 * Replaces autograd's two batched GEMMs with a 15-wide dimension for ZINC (0.79 + 0.94 ms -> one pass over `a`). */
int64_t mma_tower_linear_bwd_blocks(int64_t N);
int mma_tower_linear_bwd(const float* gy, const float* a, const float* W, float* ga, float* part, int64_t n_blocks,
                         int64_t N, int32_t T, int32_t O, int32_t C, void* stream);

/* ---- K13 / K14: MMAConv's per-tower post-NN on the UNSCALED aggregates, degree scalers applied as row factors (ABI 27) -------------
 * mma_conv.py:181-196 builds out (N,T,S*K*F) = cat_q(agg * prod_{q' <= q} scaler_q'(deg)) and :132-134 applies post_nns[t] to cat[x, out].
 * The scalers are per-target row factors, so with pre_q = the running product (reference order) of the S scalers at deg = clamp(d_n, 1):
 *   fwd: y[n, t*O + o]      = sum_q pre_q sum_kf agg[n, t*KF + kf] Wa[t][kf][q*16 + o]        (agg (N, lda): K3 with the identity scaler)
 *   bwd: gagg[n, t*KF + kf] = sum_q pre_q sum_o gy[n, t*O + o] Wb[t][q*16 + o][kf]            (what K4 takes as its upstream gradient)
 *        gys[n, (t*S + q)*16 + o] = pre_q gy[n, t*O + o]   (optional: the left operand of gW[t] = gys_t^T agg_t, a TN product)
 * on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain).  The caller lays the post-NN weight columns
 * Wo[t][o][(q*K + k)*F + f] out twice, zero-padded: Wa (T, KFp, S*16) and Wb (T, S*16, KFp + 16) with KFp = mma_tower_post_kfp(KF);
 * KF = K*F <= 512, O <= 16, S <= 5; pre: the table of mma_tower_post_pre (from the CSR-by-target row pointers = degrees).  Neither `out` nor its
 * gradient is ever materialised. */
int64_t mma_tower_post_kfp(int32_t KF);
/* ABI 32: both padded layouts from the contiguous weight columns Wo (T, O, S*KF) in one launch: Wa (T, KFp, S*16), Wb (T, S*16, KFp + 16),
 * zero padding included (the caller need not clear them). */
int mma_tower_post_weights(const float* Wo, int32_t T, int32_t O, int32_t S, int32_t KF, float* Wa, float* Wb, void* stream);
/* ABI 32: mma_tower_post_gw's partial tiles (n_chunks <= 1024 rows of T*S*16*kfp16 floats, kfp16 = KF rounded up to 16) summed in
 * mma_col_sum's order and written in the weight layout gWo (T, O, S*KF) - one launch instead of the reduction + a permuting copy. */
/* ABI 32: the two padded layouts of a Linear weight W (O, K) contiguous for mma_skinny_linear_fwd / _bwd_dx in one launch:
 * Wa (KFp, S*16), Wb (S*16, KFp + 16), S = ceil(O / 16), KFp = mma_tower_post_kfp(K); zero padding included. */
int mma_skinny_linear_weights(const float* W, int32_t O, int32_t K, float* Wa, float* Wb, void* stream);
int mma_tower_post_gw_reduce(const float* part, int64_t n_chunks, int32_t T, int32_t S, int32_t O, int32_t KF, float* gWo, void* stream);
/* 1 when K13 / K14 (and their PLAIN form K16, with S = ceil(O / 16)) take this shape: 1 <= KF <= 512, 1 <= S <= 5 and the tower's staged
 * weights plus the four wave tiles fit the 160 KB of LDS in BOTH layouts (forward KFp*S*16 floats, backward S*16*(KFp + 16)); 0 otherwise
 * - the caller then keeps the unfactored / library path (ABI 29: the host-side gates ask the library instead of restating its limits). */
int mma_tower_post_fits(int32_t KF, int32_t S);
/* pre (N, 8): pre[n][q] = prod_{q' <= q} scaler_q'(clamp(rowptr[n+1] - rowptr[n], 1)), q < S - the table the three kernels below read */
int mma_tower_post_pre(const int32_t* rowptr, float* pre, int64_t N, int32_t S, const uint8_t* scaler_host, float avg_log, float avg_lin,
                       void* stream);
int mma_tower_post_fwd(const float* agg, int64_t lda, const float* pre, const float* Wa, float* y, int64_t ldy,
                       int64_t N, int32_t T, int32_t KF, int32_t S, int32_t O, const uint8_t* scaler_host, float avg_log, float avg_lin,
                       void* stream);
int mma_tower_post_bwd(const float* gy, int64_t ldg, const float* pre, const float* Wb, float* gagg, int64_t lda,
                       float* gys, int64_t ldgs,
                       int64_t N, int32_t T, int32_t KF, int32_t S, int32_t O,
                       const uint8_t* scaler_host, float avg_log, float avg_lin, void* stream);
/* K15: the weight gradient of the same product, part (n_chunks, T, S*16, KFp16) with KFp16 = KF rounded up to 16 and n_chunks =
 * mma_tower_post_gw_chunks(N, T): part[c][t][q*16 + o][kf] = sum over the nodes n of chunk c of pre_q(deg_n) gy[n, t*O + o] agg[n, t*KF + kf]
 * (exact fp32 on the matrix cores, the node is the reduction index); summing the chunks in order (mma_col_sum over the n_chunks rows)
 * gives gW[t][q*16 + o][kf] = the gradient of Wb.  Rows o >= O and columns kf >= KF are zero. */
int64_t mma_tower_post_gw_chunks(int64_t N, int32_t T);
int mma_tower_post_gw(const float* gy, int64_t ldg, const float* agg, int64_t lda,
                      const float* pre,            /* (N, 8) from mma_tower_post_pre */
                      float* part, int64_t n_chunks,
                      int64_t N, int32_t T, int32_t KF, int32_t S, int32_t O, const uint8_t* scaler_host, float avg_log, float avg_lin,
                      void* stream);

/* ---- K16: a plain skinny Linear on the K13 / K14 kernels (exact fp32 on the matrix cores): the 75 -> 75 Linear layers around MMAConv's
 * fused kernels (x-part of the post-NN and `lin`, mma_conv.py:99-105,132-136; F.linear in the reference) --------------------------------
 *   fwd:    y (N,O)  = x (N,K) W^T + bias            Wa (KFp, S*16): Wa[k][o] = W[o][k], zero-padded, S = ceil(O/16), KFp = mma_tower_post_kfp(K)
 *   bwd_dx: gx (N,K) = gy (N,O) W                    Wb (S*16, KFp + 16): Wb[o][k] = W[o][k], zero-padded
 * O <= 80, K <= 512, any row pitches (16-byte aligned rows take the float4 path).  bias may be NULL. */
int mma_skinny_linear_fwd(const float* x, int64_t ldx, const float* Wa, const float* bias,
                          const float* addend, int64_t ldadd,      /* ABI 34, may be NULL: y = (x W^T + bias) + addend (N, ldadd >= O) - the two
                                                                      halves of MMAConv's post-NN (mma_conv.py:132-134) meet in the epilogue */
                          float* y, int64_t ldy, int64_t N, int32_t K, int32_t O, void* stream);
int mma_skinny_linear_bwd_dx(const float* gy, int64_t ldg, const float* Wb, float* gx, int64_t ldx,
                             int64_t N, int32_t K, int32_t O, void* stream);
/* ABI 35: gw (O,K) = gy^T x and gb (O) = column sums of gy (may be NULL) from ONE pass over gy (N,O) and x (N,K) - K15's kernel in its plain
 * form, the bias gradient as the product with a ones column behind x; part: workspace of mma_skinny_linear_gw_part(N, K, O) floats (one
 * partial tile per workgroup, summed in a fixed order: deterministic).  Replaces the TN GEMM + column sum of autograd's F.linear backward. */
int64_t mma_skinny_linear_gw_part(int64_t N, int32_t K, int32_t O);
int mma_skinny_linear_gw(const float* gy, int64_t ldg, const float* x, int64_t ldx, float* part, int64_t n_part,
                         float* gw, float* gb, int64_t N, int32_t K, int32_t O, void* stream);

/* ---- K10: fused log_softmax + nll_loss of the training step (models.py:68 F.log_softmax(x, dim=1) + train.py:77
 * F.nll_loss(output[idx_train], labels[idx_train])) ------------------------------------------------------------------------
 * fwd: logp (N,C) = log_softmax(x) for every row (the model's return value); loss (may be NULL) = -mean_i logp[idx[i], labels[idx[i]]]
 *      (labels is indexed by NODE, like the reference's labels[idx_train]; int64 like torch's LongTensors).
 * bwd: gx (N,C) = gloss * (exp(logp) - onehot(labels)) / n_idx on the rows idx (unique), 0 elsewhere; gloss: DEVICE scalar. */
int mma_logsoftmax_nll_fwd(const float* x, int64_t ldx, const int64_t* idx, const int64_t* labels, int64_t n_idx,
                           float* logp, int64_t ldo, float* loss, int64_t N, int32_t C, void* stream);
int mma_logsoftmax_nll_bwd(const float* logp, int64_t ldo, const int64_t* idx, const int64_t* labels, int64_t n_idx,
                           const float* gloss, float* gx, int64_t ldg, int64_t N, int32_t C, void* stream);

/* ---- K17: BatchNorm1d (training mode) + ReLU over the first *n_valid rows of a padded batch (mma.py:121 F.relu(batch_norm(conv(...)))
 * inside the graphed Net step, where batches are padded to a static shape) ---------------------------------------------------------
 * fwd: per column, mean / biased variance over rows [0, *n_valid) (n_valid: DEVICE int64 scalar, clamped to [1, N]); y = relu?(gamma
 *      (x - mean) rstd + beta) for ALL N rows; mean_out / rstd_out (C,) saved for backward; running_mean / running_var (both or neither)
 *      updated with `momentum` (the unbiased variance, like torch) and *n_tracked incremented (may be NULL).  gamma / beta may be NULL.
 * bwd: g = gy [y > 0] (if relu); gx = gamma rstd (g - sum g / n - xhat sum(g xhat) / n), ggamma = sum g xhat, gbeta = sum g (may be NULL);
 *      the sums run over all N rows (rows past n_valid belong to the dummy graph the loss never reads: their g is 0). */
int mma_masked_bn_relu_fwd(const float* x, int64_t ldx, const int64_t* n_valid, const float* gamma, const float* beta, float* y, int64_t ldy,
                           float* mean_out, float* rstd_out, float* running_mean, float* running_var, int64_t* n_tracked, float momentum,
                           float eps, int64_t N, int32_t C, int32_t relu, void* stream);
int mma_masked_bn_relu_bwd(const float* gy, int64_t ldg, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* mean,
                           const float* rstd, const float* gamma, const int64_t* n_valid, float* gx, int64_t ldgx, float* ggamma, float* gbeta,
                           int64_t N, int32_t C, int32_t relu, void* stream);

/* ---- K18: block pack / unpack (the weight plumbing of one MMAConv call, mma_conv.py:96-118 Parameters -> the padded matrices the
 * fused kernels take, and their gradients back) ------------------------------------------------------------------------------------
 * table: DEVICE (n_blocks, 10) int64, one 2-D block per row:
 *     [ a, lda, rows, cols, b_index, b_off, ldb, b_rows, b_cols, flags ]
 *   a: the A side - an absolute device address of fp32 data when a_base is NULL or flags & 1, else a float offset from a_base;
 *   flags & 2: unpack ADDS onto A (a gradient buffer that outlives the call: the reference's unregistered mask Linears, G2, are
 *   never zeroed by optimizer.zero_grad(), so their .grad accumulates over the whole training run);
 *   b_index in [0,8): which of b0..b7 holds the B side, at float offset b_off with row pitch ldb.
 * unpack == 0:  B[r*ldb + c] = (r < rows && c < cols) ? A[r*lda + c] : 0   for r < b_rows, c < b_cols   (zero padding written here)
 * unpack != 0:  A[r*lda + c] (+)= B[r*ldb + c]                              for r < rows,   c < cols
 * One workgroup per block, no overlap checks: blocks of one call must not write the same element.  The table is trusted (addresses
 * are the caller's): a wrong entry is an out-of-bounds access, as with any pointer argument. */
/* ABI 33: out (M_out, width) = [x | col | 0 ...] from x (M, fin) with row pitches ldx / ldo, fin < width, width % 4 == 0, out 16-byte
 * aligned; col (M,) or NULL = a column of ones; rows M .. M_out - 1 are zero.  The operands of the zero-padded tall Linear layers (graph
 * regression: the pre-NN on [x_i | x_j | e], mma_conv.py:81-84,170-176, as ONE GEMM per operand with the bias riding on a ones column):
 * A = [x | 1 | 0], and [W | b | 0] - the weight operand of the forward (as its transposed view) and of dL/dx (its leading columns) - one
 * launch each; was torch's pad (a zero fill of the whole buffer + a strided copy) and a strided fill per operand. */
int mma_pad_rows(const float* x, int64_t ldx, int64_t M, int32_t fin, const float* col /* (M,) or NULL */,
                 const int32_t* row_index /* ABI 34: (M,) or NULL - out row r takes x row row_index[r] (the edge rows of graph regression in
                                             target-sorted position order: the permuting copy in front of the pad is gone) */,
                 float* out, int64_t ldo, int32_t width, int64_t M_out, void* stream);
int mma_pack_blocks(const int64_t* table, int64_t n_blocks, float* a_base, float* b0, float* b1, float* b2, float* b3, float* b4,
                    float* b5, float* b6, float* b7, int32_t unpack, void* stream);

/* ---- K19: the edge encoder folded into the pre-NN's edge block (mma_conv.py:141-146: enc(e) We^T = e (We Wenc)^T + We benc) ----------
 * fwd: wz (TF,ED) = We Wenc, bz (TF) = We benc.   We (TF,F) row pitch ldw; Wenc (F,ED) and the outputs contiguous; benc / bz both or
 *      neither.  bwd: gWe (TF,F) = gwz Wenc^T + gbz (x) benc, gWenc (F,ED) = We^T gwz, gbenc (F) = We^T gbz (gbz / gbenc both or neither).
 * Plain fp32 FMA chains in a fixed order; F, ED <= 512.  One launch each (seven torch launches at ZINC's 380 x 75 x 50). */
int mma_edge_fold_fwd(const float* We, int64_t ldw, const float* Wenc, const float* benc, float* wz, float* bz, int64_t TF, int32_t F,
                      int32_t ED, void* stream);
int mma_edge_fold_bwd(const float* We, int64_t ldw, const float* Wenc, const float* benc, const float* gwz, const float* gbz, float* gWe,
                      float* gWenc, float* gbenc, int64_t TF, int32_t F, int32_t ED, void* stream);

/* ---- dropout seeds of a captured (hipGraph) step -------------------------------------------------------------------------------
 * seeds: DEVICE (2n,) uint64 - [0,n) the seeds the fused kernels read through their `seed_dev` argument (one per launch group of 8
 * masks), [n,2n) the splitmix64 states behind them (initialised by the caller from its own generator).  Each call advances every
 * state by the golden-ratio increment and writes the finalised value: a fresh seed per replay from ONE captured launch (the
 * reference redraws its F.dropout masks from torch's generator on every forward, layers.py:223; a generator op inside a captured
 * graph costs two extra fills ahead of every replay). */
int mma_seed_advance(uint64_t* seeds, int32_t n, void* stream);

/* ---- K12: the graph-regression loss of the training step (graph_regression/mma.py:156 (out.squeeze() - data.y).abs().mean()) ----
 * fwd: loss (DEVICE scalar) = mean_i |pred[i] - target[i]| over n >= 1 contiguous values, fixed summation order.
 * bwd: gpred[i] = gloss * sign(pred[i] - target[i]) / n with sign(0) = 0 (torch's abs backward); gloss: DEVICE scalar. */
int mma_l1_loss_fwd(const float* pred, const float* target, int64_t n, float* loss, void* stream);
int mma_l1_loss_bwd(const float* pred, const float* target, int64_t n, const float* gloss, float* gpred, void* stream);

/* ---- K11: Adam over ALL parameter tensors in one launch (train.py:69 optim.Adam(model.parameters(), lr, weight_decay)) ---------
 * torch.optim.Adam semantics (no amsgrad; weight decay added to the gradient; bias-corrected).  table (DEVICE memory,
 * mma_adam_table_bytes(n_tensors, total_chunks) bytes): n_tensors records {float* p; const float* g; float* m; float* v;
 * int64 n; int64 chunk0} followed by total_chunks int32 tensor ids, one per workgroup chunk of mma_adam_chunks(1) = 4096
 * elements (tensor t owns chunks [chunk0, chunk0 + mma_adam_chunks(n))), followed by one int32 ZERO (the ticket counter of the
 * launch: the last workgroup to finish moves `step`; left at zero).  step: DEVICE float, the number of steps taken so far;
 * incremented by the call (so a captured hipGraph replays correctly).
 * mma_adam_step_grads: the gradient pointers come with the CALL - grad_ptrs_host = n_tensors little-endian 8-byte device addresses in
 * HOST memory, n_tensors <= mma_adam_max_grads_by_value() - and the table's g fields are ignored: the caller may hand over fresh
 * gradient tensors every step (optimizer.zero_grad(set_to_none=True): no zero-fill, no accumulating add per parameter). */
int64_t mma_adam_table_bytes(int64_t n_tensors, int64_t total_chunks);
int64_t mma_adam_chunks(int64_t n_elements);
int mma_adam_step(const void* table, int64_t n_tensors, int64_t total_chunks, float* step, double lr, double beta1, double beta2,
                  float eps, float weight_decay, void* stream);   /* lr, betas: double, like torch's Python floats (1 - 0.999f != 1e-3) */
int64_t mma_adam_max_grads_by_value(void);
int mma_adam_step_grads(const void* table, int64_t n_tensors, int64_t total_chunks, float* step, double lr, double beta1, double beta2,
                        float eps, float weight_decay, const uint8_t* grad_ptrs_host, void* stream);

/* ---- K6: CSR by key, built on the device (graph-regression batches change every call) ----------------
 * Stable radix sort (rocPRIM) of edge positions by key[e] (int64 node ids as PyG's edge_index holds them):
 *   rowptr (N+1), perm (E) = original edge positions grouped by key, ascending inside a group (so min/max ties
 *   resolve to the lowest edge position, like torch_scatter), other_sorted[p] = other[perm[p]] (may be NULL).
 * Replaces the implicit grouping inside torch_scatter.scatter / PyG propagate (mma_conv.py:130,166). */
/*   long_nodes (may be NULL): mma_gr_long_nodes_len(E) int32 = [count, ids of the nodes whose group holds more than
 *   MMA_GR_LONG_SEGMENT entries ..., error flag] (ids in no particular order): K3/K4 hand exactly these segments to their
 *   wave-per-node pass, everything shorter runs in the block kernels.  The LAST word (ABI 34) is an error flag: zeroed here, set by
 *   K3 / K4 when they meet a count beyond the list's capacity (1) or an id that is not a node (2) - which they skip; read it
 *   (outside any graph capture) to learn that a list was clobbered. */
#define MMA_GR_LONG_SEGMENT 64
int64_t mma_csr_workspace_bytes(int64_t E, int64_t N);
int64_t mma_gr_long_nodes_len(int64_t E);
int mma_build_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N,
                  int32_t* rowptr, int32_t* perm, int32_t* other_sorted, int32_t* long_nodes,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* aggregator / scaler codes of the graph-regression path (mma_conv.py:164-172, 181-194) */
enum { MMA_GR_SUM = 0, MMA_GR_MEAN = 1, MMA_GR_MIN = 2, MMA_GR_MAX = 3, MMA_GR_VAR = 4, MMA_GR_STD = 5 };
enum { MMA_SC_IDENTITY = 0, MMA_SC_AMPLIFICATION = 1, MMA_SC_ATTENUATION = 2, MMA_SC_LINEAR = 3, MMA_SC_INVERSE_LINEAR = 4 };

/* ---- K3: fused message + K-aggregator scatter-reduce + degree scalers, graph-regression form --------
 * One pass over each target's edge segment replaces MMAConv.message + MMAConv.aggregate
 * (mma_conv.py:138-196: cat, per-tower Linear, dropout, K x torch_scatter.scatter, degree, compounding scalers, cat):
 *   h_e = drop(U[i] + V[j] + Z[r])      with i = target, j = src[p], e = perm[p]   (fused-message mode), or
 *   h_e = inputs[e]                                                                (given-messages mode = aggregate())
 *   out[n, t, s*K*F + k*F + f] = aggr_k over {h_e[t*F+f] : target(e) = n}, times the running product of scalers 0..s.
 * U,V: (N,T*F) with U = x @ W_i^T + b, V = x @ W_j^T; Z: (E,T*F) = enc(edge_attr) @ W_e^T or NULL, its row r = the
 * target-sorted POSITION p when by_pos != 0 (the caller permuted edge_attr by `perm` before its GEMM, so Z streams
 * contiguously), the original edge id e otherwise.  The dropout bits are keyed by e in both cases.
 * Saved for backward when non-NULL (row pitch ldsave >= T*F for all of them):
 *   amin8/amax8 (N,ldsave) BYTES: offset of the extremal edge inside its target's segment (ties -> lowest position, the
 *     torch_scatter CPU rule; 0xFF: none) for segments of < 256 edges; longer segments store int32 offsets in row
 *     ceil(rowptr[n]/256) of amin_side/amax_side (mma_gr_arg_side_rows(E), ldsave) - both of a pair or neither;
 *   mean/var (N,ldsave) floats for var/std.
 * When every operand's pitch is T*F rounded up to a multiple of 4 (zero padding columns) and 16-byte aligned, a lane
 * moves one dwordx4 per row; otherwise one dword.  Row pitches beyond that (lduv, ldz, ldg, ldgu a multiple of 128 floats:
 * the zero-padded buffers of the tall GEMMs around the kernels) are taken as they are.
 * z_index (categorical edge features, e.g. ZINC's 4 bond types: mma.py:88,103 embeds them, so Z has only 4 distinct rows):
 * Z is then the (n_types, ldz) table enc(emb) @ W_e^T and the edge at row index r = position p (by_pos != 0) or edge id e
 * (otherwise) reads Z[z_index[r]] - 1 byte per edge instead of T*F floats; dL/dZ_table = onehot(z_index)^T gmsg is the
 * caller's (K4 still returns gmsg per edge).
 * E == 0 is a legal graph: every target is empty and gets 0 for sum/mean/min/max/var but sqrt(0 + 1e-5) for std, times the
 * scalers of the clamped degree 1; in the given-messages form `inputs` may then be NULL (an (0,T*F) tensor has no address). */
int64_t mma_gr_arg_side_rows(int64_t E);
int mma_gr_fused_fwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos,
    const uint8_t* z_index,                      /* NULL, or categorical edge features: Z is a TABLE (<= 256 rows) and edge r takes row z_index[r] */
    const float* inputs, int64_t ldi,
    float* out, uint8_t* amin8, uint8_t* amax8, int32_t* amin_side, int32_t* amax_side, float* mean, float* var, int64_t ldsave,
    const int32_t* long_nodes,                   /* from mma_build_csr, or NULL (the second pass then scans all N row pointers) */
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream);

/* ---- K4: backward of K3 w.r.t. every edge message: gmsg (E,T*F), row = position p (by_pos != 0) or original edge id ------
 * min/max route the gradient to the saved arg edge only (torch_scatter), mean divides by the count, var/std use the
 * saved mean/var; in fused-message mode the dropout factor of the edge is applied, so gmsg = dL/d(U[i]+V[j]+Z[r]).
 * gU (may be NULL): (N, ldgu >= T*F) receives dL/dU[i] = the sum of gmsg over target i's segment (a zero row for an empty
 * target) from the same pass - the kernel walks exactly those segments, so the separate segment sum is not needed.
 * gmsg_row_max (E,) / gu_row_max (N,) (ABI 34; both or neither, gu_row_max needs gU; the CALLER ZEROES them): every gmsg row r and
 * every gU row n merge max |row| into its entry (the bits of a non-negative float, by atomicMax - mma_csr_spmm_rm merges the dV rows
 * into the same (N,) array): the row scales of the three-product GEMMs that take these gradients (mma_gemm_f16x2_tn / _nlp). */
int mma_gr_fused_bwd(
    const int32_t* rowptr, const int32_t* src, const int32_t* perm,
    const float* U, const float* V, int64_t lduv, const float* Z, int64_t ldz, int32_t by_pos, const uint8_t* z_index,
    const float* inputs, int64_t ldi,
    const float* gout, const uint8_t* amin8, const uint8_t* amax8, const int32_t* amin_side, const int32_t* amax_side,
    const float* mean, const float* var, int64_t ldsave, const int32_t* long_nodes, float* gmsg, int64_t ldg, float* gU, int64_t ldgu,
    float* gmsg_row_max, float* gu_row_max,
    int64_t N, int64_t E, int32_t T, int32_t F, const uint8_t* aggr_host, int32_t K, const uint8_t* scaler_host, int32_t S,
    float avg_log, float avg_lin, int32_t drop_mode, uint32_t drop_thr, uint64_t seed, const uint64_t* seed_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMA_AMD_H */
