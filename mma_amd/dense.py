"""Dense pre/post transforms of the hot path: plain library GEMMs (rocBLAS / hipBLASLt fp32 MFMA through torch.mm),
with one MI355X-specific twist in backward.

The weight gradients are (in, N) @ (N, out) products whose reduction dimension is the node count (1 M at C4) and whose
output is tiny (128 x 512): a single GEMM call leaves most of the 256 CUs idle (measured 49-75 TFLOP/s, and 3.4 TFLOP/s
for the (128,N)@(N,16) tail).  Splitting the reduction into ~N/8192 batches (strided-batched GEMM + a sum over the
batch) fills the chip: 140 TFLOP/s, 89 % of the fp32 MFMA peak (0.97 ms instead of 2.7 ms per mask-weight half)."""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


def _span(name, nbytes=0, flops=0, mfma=None):
    from . import functional as Fn      # late: functional imports this module
    return Fn._span(name, nbytes, flops, mfma)

_ROWS_PER_BATCH = 8192
USE_BF16X3 = True     # fp32-accurate GEMM on the bf16 matrix cores (csrc/gemm_x3.hip) where the shape allows
_MIN_ROWS_X3 = 4096
_MIN_COLS_TN = int(__import__("os").environ.get("MMA_MIN_COLS_TN", "1"))
_MIN_ROWS_TN = int(__import__("os").environ.get("MMA_MIN_ROWS_TN", "1024"))    # the TN kernel from here on (Cora's 2 708 rows: the library's
                                                                              # 128 x 256 x 2708 TN product takes 22-25 us, a fifth of the layer replay)


def _x3_ok(a, w):
    """The kernel takes K == 128 with any N, or any K % 128 == 0 with N <= 128 (accumulator tiles live in registers); a
    product with both K > 128 and N > 128 (hidden width 256: C5) is run as N/128 column blocks of the second form."""
    K, N = w.shape
    return (USE_BF16X3 and a.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32 and a.shape[0] >= _MIN_ROWS_X3
            and N % 32 == 0 and K % 128 == 0 and (K == 128 or N <= 128 or N % 128 == 0))


USE_F16X2 = True      # K == 128 products on the three-product fp16 x 2 kernel (csrc/gemm_x3.hip) instead of the six-product bf16 x 3 one


F16X2_K = (64, 96, 128)      # reduction widths of the whole-row three-product kernel (mma_gemm_f16x2_k)


def gemm_f16x2(a, w, out=None, row_max_out=None):
    """a (M,K) @ w (K,N), K in F16X2_K, N % 128 == 0: the three-product form (fp16 hi/lo pieces, power-of-two row and column scales).
    row_max_out (M,): the kernel also leaves max |a[i,:]| there (it forms them for its row scales anyway)."""
    M, K = a.shape
    N = w.shape[1]
    bt2, cu = _split_f16x2(w)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    with _span("gemm_x3_k128", nbytes=4 * M * (K + N), flops=2 * M * K * N, mfma="f16x3"):        # A in, C out (B is 0.5 MB)
        call("mma_gemm_f16x2_k", ptr(a), a.stride(0), ptr(bt2), ptr(cu), ptr(out), out.stride(0), ptr(row_max_out), M, N, K, stream_ptr())
    return out


def _split_f16x2(w, plain_lo=False):
    """w (K,N) fp32, any strides -> ((2,N,K) fp16 pieces of B^T scaled per column by a power of two that puts the column maximum
    into [2^14, 2^15), (N,) fp32 reciprocal scales): one launch (mma_split_f16x2).  plain_lo: the lo piece as it is, not times 2^11
    (the one-accumulator kernels)."""
    K, N = w.shape
    bt2 = torch.empty((2, N, K), device=w.device, dtype=torch.float16)
    cu = torch.empty((N,), device=w.device, dtype=torch.float32)
    call("mma_split_f16x2", ptr(w), w.stride(0), w.stride(1), K, N, ptr(bt2), ptr(cu), 1 if plain_lo else 0, stream_ptr())
    return bt2, cu


PACK_K256 = __import__("os").environ.get("MMA_PACK_K256", "1") != "0"        # 0: round 4's K = 256 forward (fp32 rows split per column group)
USE_F16X2_N128 = __import__("os").environ.get("MMA_F16X2_DX", "1") != "0"
USE_F16X2_K256 = __import__("os").environ.get("MMA_F16X2_K256", "1") != "0"


def ws_ok(N):
    """mma_gemm_f16x2_ws takes N: whole 256-column groups, a divisor of the 32 workgroup slots of an XCD.  Opt-in (MMA_FWD_WS=1) AND only in
    a library built with -DMMA_EXPERIMENTAL_FWD [r5]: measured slower than the column-group kernels (gemm_x3.hip)."""
    return N % 256 == 0 and 32 % (N // 256) == 0 and __import__("os").environ.get("MMA_FWD_WS", "0") == "1"


def f16x2_n128_ok(M, K, N):
    return (USE_F16X2 and USE_F16X2_N128 and USE_BF16X3 and N % 128 == 0 and N <= 512 and K % 64 == 0 and K > 128 and M >= (1 << 16))


def row_absmax(a):
    """max |a[i,:]| per row in one pass (mma_row_absmax); torch's a.abs().amax(1) is two kernels and a full-size temporary."""
    M, C = a.shape
    out = torch.empty((M,), device=a.device, dtype=torch.float32)
    if a.stride(1) != 1:
        a = a.contiguous()
    call("mma_row_absmax", ptr(a), a.stride(0) if M > 1 else C, M, C, ptr(out), stream_ptr())
    return out


USE_NLP = __import__("os").environ.get("MMA_DX_NLP", "1") != "0"      # round 4: the pipelined one-accumulator form (0: round 3's kernel)


def gemm_f16x2_n128(a, row_max, w, out, accumulate=False):
    """out (M,N) (+)= a (M,K) @ w (K,N), N a multiple of 128, on the three-product kernels; row_max (M,) >= max |a[i,:]| (0 for an
    all-zero row).  N in {128, 256} with K % 128 == 0 takes the pipelined one-accumulator kernel (mma_gemm_f16x2_nlp: ONE pass over
    `a`, also for N = 256 - hidden width 256, C5); other shapes one launch of the two-accumulator kernel per 128-column block."""
    M, K = a.shape
    N = w.shape[1]
    if USE_NLP and N in (128, 256) and K % 128 == 0 and K >= 256 and out.stride(1) == 1:
        bt2, cu = _split_f16x2(w, plain_lo=True)
        with _span("gemm_x3_acc" if accumulate else "gemm_x3_persist", nbytes=4 * M * (K + (2 if accumulate else 1) * N), flops=2 * M * K * N, mfma="f16x3"):
            call("mma_gemm_f16x2_nlp", ptr(a), a.stride(0), ptr(row_max), ptr(bt2), ptr(cu), ptr(out), out.stride(0), M, N, K,
                 1 if accumulate else 0, stream_ptr())
        return out
    bt2, cu = _split_f16x2(w)                                            # (2, N, K), (N,)
    # A in once (algorithmic: the column blocks of a wider output re-read it), C out - and in, when accumulating
    with _span("gemm_x3_acc" if accumulate else "gemm_x3_persist", nbytes=4 * M * (K + (2 if accumulate else 1) * N), flops=2 * M * K * N, mfma="f16x3"):
        # (walking A in row slabs that stay in the Infinity Cache between the column blocks was measured at C5's forward shape -
        # 32 blocks over a 1 GB A: 11.3 -> 11.2 ms, i.e. the kernel, not the re-reads of A, is what the product costs)
        for b in range(N // 128):
            blk = bt2[:, 128 * b:128 * b + 128].contiguous() if N > 128 else bt2
            call("mma_gemm_f16x2_n128", ptr(a), a.stride(0), ptr(row_max), ptr(blk), ptr(cu[128 * b:]), ptr(out[:, 128 * b:]), out.stride(0), M, K,
                 1 if accumulate else 0, stream_ptr())
    return out


def rows_mm_add_scaled_(acc, a, w, row_max):
    """acc += a @ w like rows_mm_add_, on the three-product kernel when the rows' maxima are known (row_max (M,) >= max |a[i,:]|,
    as K2a / K2b leave them) and the shape is the dL/dx one; else the six-product / library path."""
    if (row_max is not None and a.shape[0] > 0 and f16x2_n128_ok(a.shape[0], a.shape[1], w.shape[1]) and acc.stride(1) == 1
            and a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0):
        return gemm_f16x2_n128(a, row_max, w, acc, accumulate=True)
    return rows_mm_add_(acc, a, w)


def gemm_bf16x3(a, w, out=None, accumulate=False, row_max_box=None):
    """a (M,K) @ w (K,N) with fp32 accuracy on the bf16 MFMA path (three-piece split of both operands).  `a` may be a
    row-strided view (a column block of a wider buffer); accumulate=True adds the product to `out`.  row_max_box: a list that
    receives the (M,) row maxima of |a| when the path taken forms them anyway (the three-product forms)."""
    if a.stride(1) != 1 or a.stride(0) % 4 or a.data_ptr() % 16:
        a = a.contiguous()
    M, K = a.shape
    N = w.shape[1]
    if (USE_F16X2 and K in F16X2_K and not accumulate and N % 128 == 0 and N <= 4096 and M >= _MIN_ROWS_X3
            and (out is None or (out.stride(1) == 1 and out.dtype == torch.float32))):
        rm = torch.empty((M,), device=a.device, dtype=torch.float32) if row_max_box is not None else None
        if rm is not None:
            row_max_box.append(rm)
        return gemm_f16x2(a, w, out, rm)
    if (USE_F16X2 and USE_F16X2_N128 and not accumulate and K > 128 and N > 128 and K % 64 == 0 and N % 128 == 0 and M >= (1 << 16)
            and (out is None or (out.stride(1) == 1 and out.dtype == torch.float32))):
        # hidden width 256 (C5): K = 256 does not fit the whole-row form of mma_gemm_f16x2; the chunked three-product kernel
        # takes the row maxima from one cheap pass over `a` (M x K floats read against M x N written)
        if out is None:
            out = torch.empty((M, N), device=a.device, dtype=torch.float32)
        if K == 256 and ws_ok(N):
            # W-stationary kernel (round 4): forms the row maxima itself (and leaves them for the weight-gradient product) - no separate
            # pass over `a`
            rm = torch.empty((M,), device=a.device, dtype=torch.float32)
            if row_max_box is not None:
                row_max_box.append(rm)
            bt2, cu = _split_f16x2(w)
            with _span("gemm_x3_persist", nbytes=4 * M * (K + N), flops=2 * M * K * N, mfma="f16x3"):
                call("mma_gemm_f16x2_ws", ptr(a), a.stride(0), ptr(bt2), ptr(cu), ptr(out), out.stride(0), ptr(rm), M, N, K, stream_ptr())
            return out
        if K == 256 and N <= 4096 and USE_F16X2_K256 and PACK_K256:
            # [r5] A packed once into fp16 fragment order (one pass: row maxima, scale exponents, both pieces) - the 32 column groups of
            # hidden width 256 stop re-loading the rows in the MFMA's fragment shape and re-splitting them (C5 shard: 7.7 -> see DESIGN.md)
            rm = torch.empty((M,), device=a.device, dtype=torch.float32)
            if row_max_box is not None:
                row_max_box.append(rm)
            ap = torch.empty((int(_lib.query("mma_pack_f16x2_k256_bytes", M)),), device=a.device, dtype=torch.uint8)
            sce = torch.empty((M,), device=a.device, dtype=torch.int32)
            bt2, cu = _split_f16x2(w)
            with _span("pack_f16x2", nbytes=8 * M * K, flops=0):
                call("mma_pack_f16x2_k256", ptr(a), a.stride(0), M, ptr(ap), ptr(sce), ptr(rm), stream_ptr())
            with _span("gemm_x3_persist", nbytes=4 * M * (K + N), flops=2 * M * K * N, mfma="f16x3"):
                call("mma_gemm_f16x2_k256p", ptr(ap), ptr(sce), ptr(bt2), ptr(cu), ptr(out), out.stride(0), M, N, stream_ptr())
            return out
        rm = row_absmax(a)
        if row_max_box is not None:
            row_max_box.append(rm)
        if K == 256 and N <= 4096 and USE_F16X2_K256:
            # column-group form with the whole 256-deep B slab resident: A is read once per column group through L2 instead of once per
            # 128-column launch from HBM (C5 forward, N = 4096: 32 launches of the chunked kernel)
            bt2, cu = _split_f16x2(w)
            with _span("gemm_x3_persist", nbytes=4 * M * (K + N), flops=2 * M * K * N, mfma="f16x3"):
                call("mma_gemm_f16x2_k256", ptr(a), a.stride(0), ptr(rm), ptr(bt2), ptr(cu), ptr(out), out.stride(0), M, N, stream_ptr())
            return out
        return gemm_f16x2_n128(a, rm, w, out)
    wt = w.t().contiguous()                                  # (N,K): B^T, k contiguous
    bt3 = torch.empty((3, N, K), device=a.device, dtype=torch.bfloat16)
    call("mma_split_bf16x3", ptr(wt), N * K, ptr(bt3), stream_ptr())
    assert out is not None or not accumulate
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    acc = 1 if accumulate else 0
    with _span("gemm_x3_acc" if accumulate else ("gemm_x3_k128" if K == 128 else "gemm_x3_persist"),
               nbytes=4 * M * (K + (2 if accumulate else 1) * N), flops=2 * M * K * N, mfma="bf16x6"):
        if K == 128 or N <= 128:
            call("mma_gemm_bf16x3", ptr(a), a.stride(0), ptr(bt3), ptr(out), out.stride(0), M, N, K, acc, stream_ptr())
        else:                                                # K > 128 and N > 128: one launch per 128-column block of the output
            blocks = bt3.view(3, N // 128, 128, K).permute(1, 0, 2, 3).contiguous()    # (N/128, 3, 128, K): a block's own three pieces
            for b in range(N // 128):
                call("mma_gemm_bf16x3", ptr(a), a.stride(0), ptr(blocks[b]), ptr(out[:, 128 * b:128 * b + 128]), out.stride(0), M, 128,
                     K, acc, stream_ptr())
    return out


def mm_into(a, w, out, row_max_box=None):
    """out[...] = a @ w (no autograd): the forward GEMMs of the sharded layer write row blocks of one buffer.  row_max_box: see
    gemm_bf16x3 (stays empty when the path taken does not form the row maxima of a)."""
    if a.shape[0] == 0:
        return out
    if _x3_ok(a, w):
        return gemm_bf16x3(a, w, out, row_max_box=row_max_box)
    return torch.mm(a, w, out=out)


def rows_mm(a, w):
    """a @ w for tall a, no autograd (bf16x3 kernel or row-batched library GEMM)."""
    return _rows_mm(a, w)


def rows_mm_add_(acc, a, w):
    """acc += a @ w in place (no autograd): the bf16x3 kernel folds the addition into its epilogue."""
    if a.shape[0] == 0:
        return acc
    if _x3_ok(a, w) and acc.stride(1) == 1:
        return gemm_bf16x3(a, w, acc, accumulate=True)
    if a.shape[0] // _ROWS_PER_BATCH < 4:                  # small: the library GEMM adds in its epilogue (beta = 1) - one launch, not two
        with _span("lib_mm", nbytes=4 * a.shape[0] * (a.shape[1] + 2 * w.shape[1]), flops=2 * a.shape[0] * a.shape[1] * w.shape[1], mfma="f32"):
            return acc.addmm_(a, w)
    return acc.add_(_rows_mm(a, w))


def _rows_mm(a, w):
    """a (N,in) @ w (in,out) for tall a, issued as a strided-batched GEMM over row blocks (w broadcast): rocBLAS then
    picks a kernel that runs 15-25 % faster than the single tall-skinny GEMM (C4: 94 -> 114 TFLOP/s forward,
    105 -> 135 TFLOP/s for g @ W^T)."""
    if _x3_ok(a, w):
        return gemm_bf16x3(a, w)
    N = a.shape[0]
    B = N // _ROWS_PER_BATCH
    with _span("lib_mm", nbytes=4 * N * (a.shape[1] + w.shape[1]), flops=2 * N * a.shape[1] * w.shape[1], mfma="f32"):      # rocBLAS fp32
        if B < 4:
            return torch.mm(a, w)
        a = a.contiguous()
        n = B * _ROWS_PER_BATCH
        out = torch.empty((N, w.shape[1]), device=a.device, dtype=a.dtype)
        torch.bmm(a[:n].view(B, _ROWS_PER_BATCH, -1), w.unsqueeze(0).expand(B, -1, -1), out=out[:n].view(B, _ROWS_PER_BATCH, -1))
        if n < N:
            torch.mm(a[n:], w, out=out[n:])
        return out


class _MM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _rows_mm(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _rows_mm(g, w.t())
        if ctx.needs_input_grad[1]:
            gw = xt_g(x, g)
        return gx, gw


def _x3_tn_ok(x, g):
    """The kernel takes up to 128 x columns (any count: ragged tiles are guarded); wider x (hidden width 256: C5) runs as
    128-column blocks of x.  Any g width (the odd-width Linear layers of graph regression: 75 x 76; the 3- / 7-class output
    weights of node classification, where the library's TN product takes 22 us on Cora and 80 us on PubMed)."""
    KA = x.shape[1]
    ok = (USE_BF16X3 and x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32 and x.shape[0] >= _MIN_ROWS_TN
          and (8 <= KA <= 128 or KA % 128 == 0) and g.shape[1] >= _MIN_COLS_TN and x.stride(1) == 1 and g.stride(1) == 1)
    if not ok:
        return False
    # the kernel addresses one row range through a 32-bit buffer window: (rows per split) x (row pitch) must stay < 2 GB
    M, NC, kb = x.shape[0], g.shape[1], min(KA, 128)
    splits = max(1, int(_lib.lib().mma_gemm_bf16x3_tn_workspace_floats(M, kb, NC)) // (kb * NC))
    rows = -(-(-(-M // splits)) // 32) * 32
    return (rows + 32) * max(x.stride(0), g.stride(0)) * 4 < 2 ** 31


def gemm_bf16x3_tn(x, g):
    """x^T @ g (in,out) for tall fp32 x (N,in), g (N,out) on the bf16x3 TN kernel (row-strided operands are fine)."""
    N, KA = x.shape
    NC = g.shape[1]
    out = torch.empty((KA, NC), device=x.device, dtype=torch.float32)
    kb = KA if KA <= 128 else 128
    n_ws = int(_lib.lib().mma_gemm_bf16x3_tn_workspace_floats(N, kb, NC))
    ws = torch.empty((n_ws,), device=x.device, dtype=torch.float32) if n_ws else None
    with _span("gemm_x3_tn", nbytes=4 * N * (KA + NC), flops=2 * N * KA * NC, mfma="bf16x6"):
        for j in range(0, KA, kb):                           # one launch per 128-column block of x (= row block of the result)
            call("mma_gemm_bf16x3_tn", ptr(x[:, j:j + kb]), x.stride(0), ptr(g), g.stride(0), ptr(out[j:j + kb]), ptr(ws), n_ws, N, kb,
                 NC, stream_ptr())
    return out


USE_F16X2_TN = __import__("os").environ.get("MMA_F16X2_TN", "1") != "0"
_MIN_ROWS_F16X2_TN = 1 << 16


TN_KA256 = __import__("os").environ.get("MMA_TN_KA256", "1") != "0"       # 0: 128-column blocks of x, one launch each (round 3)


def gemm_f16x2_tn(x, g, x_row_max=None, g_row_max=None):
    """x^T @ g like gemm_bf16x3_tn on the three-product fp16 x 2 kernel: row scales balanced between the operands, derived on
    the device from the row maxima ((N,) upper bounds of max |x[i,:]| / max |g[i,:]|; None = one extra pass over that operand);
    rows too far apart in size send the call to the six-product kernel (decided on the device)."""
    N, KA = x.shape
    NC = g.shape[1]
    out = torch.empty((KA, NC), device=x.device, dtype=torch.float32)
    # [r4] x up to 256 columns wide in ONE launch (hidden width 256: G is read once, not once per 128-column block of x)
    kb = KA if KA <= 128 else (256 if TN_KA256 and KA % 256 == 0 else 128)
    n_ws = int(_lib.lib().mma_gemm_f16x2_tn_workspace_floats(N, kb, NC))
    ws = torch.empty((n_ws,), device=x.device, dtype=torch.float32)
    with _span("gemm_x3_tn", nbytes=4 * N * (KA + NC), flops=2 * N * KA * NC, mfma="f16x3"):
        for j in range(0, KA, kb):          # wider x (C5): 128-column blocks; the maxima of the whole row bound every block's
            call("mma_gemm_f16x2_tn", ptr(x[:, j:j + kb]), x.stride(0), ptr(g), g.stride(0),
                 ptr(x_row_max), ptr(g_row_max), ptr(out[j:j + kb]), ptr(ws), n_ws, N, kb,
                 NC, stream_ptr())
    return out


def xt_g(x, g, x_row_max=None, g_row_max=None):
    """x^T @ g for tall x (N,in), g (N,out): the TN kernels where the shape allows (three products when the caller brings g's row
    maxima along - a pass over the wide operand would cost what the form saves -, else six), else split-N batched GEMM + sum."""
    if _x3_tn_ok(x, g):
        if USE_F16X2 and USE_F16X2_TN and g_row_max is not None and x.shape[0] >= _MIN_ROWS_F16X2_TN:
            return gemm_f16x2_tn(x, g, x_row_max, g_row_max)
        return gemm_bf16x3_tn(x, g)
    N = x.shape[0]
    B = N // _ROWS_PER_BATCH
    if B < 4:
        return torch.mm(x.t(), g)
    n = B * _ROWS_PER_BATCH
    x = x if x.stride(1) == 1 else x.contiguous()            # row-strided operands (column blocks of a wider buffer) are
    g = g if g.stride(1) == 1 else g.contiguous()            # fine: the row blocks stay strided-batched GEMM operands
    out = torch.bmm(x[:n].unflatten(0, (B, _ROWS_PER_BATCH)).transpose(1, 2), g[:n].unflatten(0, (B, _ROWS_PER_BATCH))).sum(0)
    if n < N:
        out = out + torch.mm(x[n:].t(), g[n:])
    return out


def xt_g_batched(x, ka, g, nc, B):
    """(B, ka, nc): out[b] = x[:, b*ka:(b+1)*ka]^T @ g[:, b*nc:(b+1)*nc] - B independent TN products over the same rows in ONE launch
    (mma_gemm_bf16x3_tn_batched) where the shape allows, else one xt_g per block."""
    M = x.shape[0]
    ok = (USE_BF16X3 and x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32 and M >= _MIN_ROWS_X3 and 8 <= ka <= 128
          and nc >= 32 and x.stride(1) == 1 and g.stride(1) == 1 and max(x.stride(0), g.stride(0)) < (1 << 24))
    if ok:
        n_ws = int(_lib.lib().mma_gemm_bf16x3_tn_batched_workspace_floats(M, ka, nc, B))
        splits = max(1, n_ws // (B * ka * nc))
        rows = -(-(-(-M // splits)) // 32) * 32
        ok = (rows + 32) * max(x.stride(0), g.stride(0)) * 4 < 2 ** 31          # one split's rows through a 32-bit buffer window
    if not ok:
        return torch.stack([xt_g(x[:, b * ka:(b + 1) * ka], g[:, b * nc:(b + 1) * nc]) for b in range(B)])
    out = torch.empty((B, ka, nc), device=x.device, dtype=torch.float32)
    ws = torch.empty((n_ws,), device=x.device, dtype=torch.float32) if n_ws else None
    with _span("gemm_x3_tn", nbytes=4 * M * B * (ka + nc), flops=2 * M * B * ka * nc, mfma="bf16x6"):
        call("mma_gemm_bf16x3_tn_batched", ptr(x), x.stride(0), ka, ptr(g), g.stride(0), nc, ptr(out), ptr(ws), n_ws, M, ka, nc, B,
             stream_ptr())
    return out


def mm(x, w):
    """x @ w with the split-reduction weight gradient."""
    return _MM.apply(x, w)


def col_sum(g):
    """g.sum(0) of a tall (R,C) fp32 matrix on the K8 kernel (fixed summation order).  torch's own column reduction
    falls off a cliff when C % 4 != 0 (3.2 ms for 204552 x 375 on MI355X; this kernel: HBM rate)."""
    _lib.require_gpu(g)
    assert g.dim() == 2 and g.dtype == torch.float32
    if g.stride(1) != 1:
        g = g.contiguous()
    R, C = g.shape
    out = torch.empty((C,), device=g.device, dtype=torch.float32)
    n_ws = int(_lib.lib().mma_col_sum_workspace_floats(R, C))
    ws = torch.empty((n_ws,), device=g.device, dtype=torch.float32) if n_ws else None
    call("mma_col_sum", ptr(g), g.stride(0) if R > 1 else C, R, C, ptr(out), ptr(ws), n_ws, stream_ptr())
    return out


USE_SKINNY = __import__("os").environ.get("MMA_SKINNY_LINEAR", "1") != "0"
_SKINNY_MIN_ROWS = 4096


def _skinny_ok(x2, weight):
    """K16: tall fp32 rows through a narrow Linear (out <= 80, in <= 512): the 75 -> 75 layers of graph regression.  rocBLAS runs
    them at ~0.11 ms per GEMM on 2e5 rows (61 MB in, 61 MB out); the fp32 matrix-core kernels stream them."""
    O, K = weight.shape
    return (USE_SKINNY and x2.is_cuda and x2.dtype == torch.float32 and weight.dtype == torch.float32 and x2.dim() == 2
            and x2.shape[0] >= _SKINNY_MIN_ROWS and O <= 80 and K <= 512 and x2.stride(1) == 1
            and tower_post_fits(K, -(-O // 16)))     # the kernels' own limits (weights + wave tiles inside 160 KB of LDS, both layouts)


def tower_post_fits(KF, S):
    """mma_tower_post_fits: do K13 / K14 (K16 with S = ceil(O/16)) take a (KF, S) product?  The library answers - the gates here do
    not restate its limits (a shape inside a Python gate but outside the kernel's used to raise MMALibraryError instead of taking the
    library GEMM)."""
    return bool(_lib.query("mma_tower_post_fits", int(KF), int(S)))


def _skinny_weights(weight):
    """(Wa (KFp, S*16), Wb (S*16, KFp+16)) zero-padded copies of W (O, K) in the layouts the kernels stage into LDS."""
    O, K = weight.shape
    S = -(-O // 16)
    kfp = int(_lib.lib().mma_tower_post_kfp(K))
    Wb = torch.empty((S * 16, kfp + 16), device=weight.device, dtype=torch.float32)
    Wa = torch.empty((kfp, S * 16), device=weight.device, dtype=torch.float32)
    call("mma_skinny_linear_weights", ptr(weight.contiguous()), O, K, ptr(Wa), ptr(Wb), stream_ptr())      # one launch (a fill + two copies before)
    return Wa, Wb


SKINNY_GW = __import__("os").environ.get("MMA_SKINNY_GW", "1") != "0"      # 0: round 4's TN GEMM + column sum (A/B)


def skinny_gw(g2, x2, want_bias=True):
    """(gw (O, K), gb (O) or None) = (g2^T x2, column sums of g2) for tall fp32 rows, O <= 80, K <= 512: mma_skinny_linear_gw."""
    rows, O = g2.shape
    K = x2.shape[1]
    n_part = int(_lib.query("mma_skinny_linear_gw_part", rows, K, O))
    part = torch.empty(n_part, device=g2.device, dtype=torch.float32)
    gw = torch.empty((O, K), device=g2.device, dtype=torch.float32)
    gb = torch.empty(O, device=g2.device, dtype=torch.float32) if want_bias else None
    with _span("skinny_linear_gw", nbytes=4 * rows * (K + O), flops=2 * rows * (-(-O // 16) * 16) * (-(-(K + 1) // 16) * 16), mfma="f32"):
        call("mma_skinny_linear_gw", ptr(g2), g2.stride(0), ptr(x2), x2.stride(0), ptr(part), n_part, ptr(gw), ptr(gb), rows, K, O, stream_ptr())
    return gw, gb


class _Linear(torch.autograd.Function):
    """y = x W^T + b over the last dimension (torch_geometric Linear / F.linear: mma_conv.py:82,99-105, mask_aggr.py:50).
    Forward is the library GEMM - or, for tall rows through a narrow layer, the K16 fp32 matrix-core kernel; backward replaces
    autograd's weak spots for tall inputs: the weight gradient is the split-reduction GEMM (xt_g), the bias gradient the K8 column
    sum, dL/dx the K16 kernel again."""

    @staticmethod
    def forward(ctx, x, weight, bias, addend=None):
        """addend (rows, out), optional: y = linear(x) + addend - on the K16 path the addition is the kernel's epilogue (one pass less over
        two (rows, out) tensors and one launch less: MMAConv's post-NN, y_aggregates + x W_x^T + b)."""
        ctx.has_bias = bias is not None
        x2 = x.reshape(-1, x.shape[-1])
        ctx.skinny = _skinny_ok(x2, weight)
        if ctx.skinny:
            O, K = weight.shape
            Wa, Wb = _skinny_weights(weight)
            y = torch.empty((x2.shape[0], O), device=x.device, dtype=torch.float32)
            add2 = None
            if addend is not None:
                add2 = addend.reshape(-1, O)
                add2 = add2 if add2.stride(1) == 1 else add2.contiguous()
            x3 = add2 is None and __import__("os").environ.get("MMA_SKINNY_X3", "1") != "0" and __import__("os").environ.get("MMA_POST_EXACT") != "1"
            with _span("skinny_linear_fwd", nbytes=4 * x2.shape[0] * (K + O * (2 if add2 is not None else 1)), flops=2 * x2.shape[0] * Wa.shape[0] * Wa.shape[1],
                       mfma="bf16x6" if x3 else "f32"):
                call("mma_skinny_linear_fwd", ptr(x2), x2.stride(0), ptr(Wa), ptr(bias.contiguous() if bias is not None else None), ptr(add2),
                     add2.stride(0) if add2 is not None else 0, ptr(y), O, x2.shape[0], K, O, stream_ptr())
            ctx.save_for_backward(x, weight, Wb)
            return y.view(x.shape[:-1] + (O,))
        ctx.save_for_backward(x, weight)
        y = torch.nn.functional.linear(x, weight, bias)
        return y if addend is None else y + addend.reshape(y.shape)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors[:2]
        g2 = g.reshape(-1, g.shape[-1])
        gx = gw = gb = None
        g_add = g if (len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3]) else None      # d(y)/d(addend) = identity
        if ctx.needs_input_grad[0]:
            if ctx.skinny:
                O, K = weight.shape
                g2 = g2 if g2.stride(1) == 1 else g2.contiguous()
                gx2 = torch.empty((g2.shape[0], K), device=g.device, dtype=torch.float32)
                Wb_ = ctx.saved_tensors[2]
                with _span("skinny_linear_bwd", nbytes=4 * g2.shape[0] * (K + O), flops=2 * g2.shape[0] * Wb_.shape[0] * (Wb_.shape[1] - 16), mfma="f32"):
                    call("mma_skinny_linear_bwd_dx", ptr(g2), g2.stride(0), ptr(ctx.saved_tensors[2]), ptr(gx2), K, g2.shape[0], K, O, stream_ptr())
                gx = gx2.view(x.shape)
            else:
                gx = torch.mm(g2, weight).view(x.shape)
        x2 = x.reshape(-1, x.shape[-1])
        if ctx.skinny and SKINNY_GW and ctx.needs_input_grad[1] and x2.stride(1) == 1:
            # K15 in its plain form: the weight and the bias gradient from ONE pass over g and x (the bias gradient as the product with a
            # ones column behind x) - the TN library GEMM + the column sum read g twice and took ~0.09 ms per 75 -> 75 layer at C2L
            gw, gb = skinny_gw(g2 if g2.stride(1) == 1 else g2.contiguous(), x2, ctx.has_bias and ctx.needs_input_grad[2])
            return gx, gw, gb, g_add
        if ctx.has_bias and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and x2.shape[0] >= 4 * _ROWS_PER_BATCH and not ctx.skinny:
            # tall input: the bias gradient rides on the weight-gradient GEMM as the row of a ones column appended to x, so the
            # (rows, out) gradient is read once instead of twice (C2L: 2 x 0.13 ms of column sums over 0.62 / 0.65 GB)
            gw1 = xt_g(g2, torch.cat([x2, x2.new_ones((x2.shape[0], 1))], 1))               # (out, in + 1)
            return gx, gw1[:, :-1].contiguous(), gw1[:, -1].contiguous(), g_add
        if ctx.needs_input_grad[1]:
            gw = xt_g(g2, x2)                                           # (out, in)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = col_sum(g2)
        return gx, gw, gb, g_add


def linear(x, weight, bias=None, addend=None):
    """F.linear (+ addend) with the tall-input backward (GPU tensors only)."""
    return _Linear.apply(x, weight, bias, addend)


# ---- tall Linear layers with odd widths on the bf16x3 kernels (graph regression: 75 -> 760 on 2e5 rows, 50 -> 380 on 4e5) ------
# rocBLAS runs these at 30-60 TFLOP/s (K or N in {50, 75, 76, 380, 760}: nothing is a multiple of anything); zero-padded to
# K = 128 and N % 128 == 0 they are ordinary inputs of the bf16x3 kernels.  The ones column that carries the bias sits in the
# K padding, so forward folds the bias in and the TN weight-gradient GEMM returns the bias gradient as one more row.
X3_LINEAR = True
FUSED_PAD = __import__("os").environ.get("MMA_PAD_ONES", "1") != "0"        # 0: torch's pad + a strided fill (round 3)
X3_LINEAR_MIN_ROWS = 32768
X3_ROW_MAX = __import__("os").environ.get("MMA_X3_ROW_MAX", "1") != "0"     # 0: round 4's six-product backward of the tall Linears (A/B)
X3_NARROW_K = __import__("os").environ.get("MMA_X3_NARROW_K", "1") != "0"   # 0: every padded A operand 128 columns wide (A/B)
_PADDED = {}          # data_ptr -> weakref to a (rows, pitch) fp32 buffer whose columns beyond the payload are ZERO


def _round_up(v, m):
    return -(-v // m) * m


def padded_empty(rows, cols, device, multiple=128, row_max=None):
    """The (rows, cols) leading-columns view of a new (rows, round_up(cols, multiple)) fp32 buffer whose pad columns are zero
    and which is REGISTERED: a consumer (linear_x3's backward) can take the whole buffer as a GEMM operand without a copy.
    Producers that fill such a view must leave the pad columns alone.  row_max (rows,), optional: the producer will leave
    max |row| there (K4 / the dV segment sum, round 5) - it travels with the buffer, and the consumer's GEMMs take the three-product form."""
    import weakref
    pitch = _round_up(cols, multiple)
    buf = torch.empty((rows, pitch), device=device, dtype=torch.float32)
    buf._mma_row_max = row_max
    if pitch > cols:
        buf[:, cols:].zero_()
    for k in [k for k, r in _PADDED.items() if r() is None]:
        del _PADDED[k]
    _PADDED[buf.data_ptr()] = weakref.ref(buf)
    return buf[:, :cols]


def _padded_parent(v, pitch):
    ref = _PADDED.get(v.data_ptr())
    buf = ref() if ref is not None else None
    if (buf is not None and v.dim() == 2 and tuple(buf.shape) == (v.shape[0], pitch) and v.stride(1) == 1 and v.stride(0) == pitch
            and v.storage_offset() == buf.storage_offset() and v.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()):
        return buf
    return None


class _LinearX3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, row_index=None, inv_index=None):
        """row_index (N,) int32 / inv_index (N,) int64, a permutation and its inverse (optional): y = linear(x[row_index]) - the gather rides
        on the pad launch (graph regression's edge rows in target-sorted position order), dL/dx is gathered back with inv_index."""
        N, fin = x.shape
        ctx.inv_index = inv_index
        fout = weight.shape[0]
        OP = _round_up(fout, 128)
        fused = FUSED_PAD and x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and weight.dtype == torch.float32
        # [r5] the A operand is padded to 64 / 96 columns where [x | 1] fits (edge features 50 + 1, node features 75 + 1): the forward and
        # the weight-gradient product stop reading, splitting and multiplying pad columns up to 128
        KP = next(k for k in F16X2_K if fin + 1 <= k) if (X3_NARROW_K and USE_F16X2 and N >= _MIN_ROWS_X3) else 128
        if row_index is not None and not fused:
            x = x.index_select(0, row_index.long())
        if fused:
            # [r4] both padded operands in ONE launch each (mma_pad_rows): A = [x | 1 | 0] (N, 128), and wpad = [W | b | 0] (OP, 128), whose
            # transposed VIEW is the forward's B and whose leading columns are dL/dx's B - no second padded copy of W in backward
            xp = torch.empty((N, KP), device=x.device, dtype=torch.float32)
            call("mma_pad_rows", ptr(x), x.stride(0), N, fin, None, ptr(row_index), ptr(xp), KP, KP, N, stream_ptr())
            w2 = weight if weight.stride(1) == 1 else weight.contiguous()
            wpad = torch.empty((OP, 128), device=x.device, dtype=torch.float32)
            if bias is not None:
                call("mma_pad_rows", ptr(w2), w2.stride(0), fout, fin, ptr(bias.contiguous()), None, ptr(wpad), 128, 128, OP, stream_ptr())
            else:
                call("mma_pad_rows", ptr(w2), w2.stride(0), fout, fin + 0, ptr(torch.zeros((fout,), device=x.device)), None, ptr(wpad), 128, 128, OP, stream_ptr())
            wt = wpad.t()[:KP]                                       # (KP, OP) view: rows beyond fin + 1 are zero anyway
        else:
            xp = torch.nn.functional.pad(x, (0, KP - fin))          # (N, KP): [x | 1 | 0 ...]
            xp[:, fin] = 1.0
            wt = weight.new_zeros((KP, OP))                          # [W^T ; b ; 0 ...], pad columns zero
            wt[:fin, :fout] = weight.t()
            if bias is not None:
                wt[fin, :fout] = bias
            wpad = None
        box = []
        y = gemm_bf16x3(xp, wt, row_max_box=box)                     # (N, OP); columns beyond fout are exact zeros
        ctx.x_rm = box[0] if box else None                           # max |[x | 1]| per row, formed by the three-product forward for its own scales
        if wpad is not None:
            ctx.save_for_backward(xp, weight, wpad)
        else:
            ctx.save_for_backward(xp, weight)
        ctx.dims = (fin, fout, OP, bias is not None)
        return y[:, :fout]

    @staticmethod
    def backward(ctx, g):
        xp, weight = ctx.saved_tensors[:2]
        wpad = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        fin, fout, OP, has_bias = ctx.dims
        gp = _padded_parent(g, OP)                                   # the producer's own zero-padded buffer: no copy
        if gp is None:
            gp = torch.nn.functional.pad(g, (0, OP - fout))
        gx = gw = gb = None
        # [r5] the producer of g (K4 + the dV segment sum) left max |g row| with its padded buffer: both products take the THREE-product
        # fp16 x 2 kernels (round 4: six bf16 products, because nobody knew the row maxima - 15 % of the C2L step)
        g_rm = getattr(gp, "_mma_row_max", None) if (USE_F16X2 and USE_F16X2_TN and X3_ROW_MAX) else None
        three = g_rm is not None and xp.shape[0] >= _MIN_ROWS_F16X2_TN
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            KA = _round_up(fin + 1, 32)
            gwb = gemm_f16x2_tn(xp[:, :KA], gp, ctx.x_rm, g_rm) if three else gemm_bf16x3_tn(xp[:, :KA], gp)   # (KA, OP) = [x | 1]^T g
            gw = gwb[:fin, :fout].t().contiguous()
            gb = gwb[fin, :fout].contiguous() if has_bias else None
        if ctx.needs_input_grad[0] and three and wpad is not None and f16x2_n128_ok(gp.shape[0], OP, 128) and OP % 128 == 0 and OP >= 256:
            # dL/dx = g [W | b | 0]: one pass over g on the one-accumulator kernel; column `fin` (the bias column) is dropped by the slice
            gx = gemm_f16x2_n128(gp, g_rm, wpad, torch.empty((gp.shape[0], 128), device=gp.device, dtype=torch.float32))[:, :fin]
        elif ctx.needs_input_grad[0]:
            NP = _round_up(fin, 32)
            if wpad is not None:
                wp = wpad[:, :NP]           # column fin (the bias) lands in a pad column of gx that the slice below drops
            else:
                wp = weight.new_zeros((OP, NP))
                wp[:fout, :fin] = weight
            gx = gemm_bf16x3(gp, wp)[:, :fin]
        if gx is not None and ctx.inv_index is not None:
            gx = gx.index_select(0, ctx.inv_index)                   # back to the caller's row order (deterministic: a gather, no index_add)
        return gx, gw, gb, None, None


def linear_x3_ok(x, weight):
    return (X3_LINEAR and USE_BF16X3 and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= X3_LINEAR_MIN_ROWS
            and x.shape[1] + 1 <= 128 and weight.shape[0] > 128)      # narrower outputs: measured no better than the library (A is split per 4 tiles only)


def linear_tall(x, weight, bias=None):
    """F.linear for a tall 2-D x: zero-padded onto the bf16x3 kernels where that pays (see above), else `linear`."""
    if linear_x3_ok(x, weight):
        return _LinearX3.apply(x, weight, bias, None, None)
    return _Linear.apply(x, weight, bias, None)


def linear_tall_rows(x, row_index, inv_index, weight, bias=None):
    """linear_tall(x[row_index]) for a PERMUTATION row_index (int32) with inverse inv_index (int64): on the zero-padded path the gather
    is part of the pad launch; None: the caller permutes first."""
    if linear_x3_ok(x, weight) and FUSED_PAD and x.stride(1) == 1:
        return _LinearX3.apply(x, weight, bias, row_index, inv_index)
    return None


class _BiasAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, b):
        return y + b

    @staticmethod
    def backward(ctx, g):
        return g, (col_sum(g.reshape(-1, g.shape[-1])) if ctx.needs_input_grad[1] else None)


def bias_add(y, b):
    """y (..., C) + b (C,) whose bias gradient is the K8 column sum."""
    return _BiasAdd.apply(y, b)


class _TowerLinear(torch.autograd.Function):
    """y[n,t,:] = a[n,t,:] @ W[t]^T for a (N,T,C), W (T,O,C): the per-tower post-NN Linear of MMAConv (mma_conv.py:132-134)
    on the aggregates, without slicing towers apart.  Backward writes the gradient of `a` tower by tower straight into
    its (N,T,C) layout (autograd's batched GEMM returns it tower-major and costs a 1.9 GB re-layout copy at C2L) and takes
    the weight gradients as split-reduction GEMMs over the row-strided tower slices."""

    @staticmethod
    def forward(ctx, a, W):
        ctx.save_for_backward(a, W)
        return torch.bmm(a.transpose(0, 1), W.transpose(1, 2)).transpose(0, 1)

    @staticmethod
    def backward(ctx, g):
        a, W = ctx.saved_tensors
        T, O, C = W.shape
        N = a.shape[0]
        if (a.is_cuda and O <= 16 and C % 4 == 0 and N > 0 and a.is_contiguous() and ctx.needs_input_grad[0]
                and ctx.needs_input_grad[1]):
            # K9: both gradients in one pass over `a` (the 15-wide batched GEMMs below run at ~15 TFLOP/s)
            g = g.contiguous()
            Wc = W.contiguous()
            ga = torch.empty_like(a)
            nb = int(_lib.lib().mma_tower_linear_bwd_blocks(N))
            part = torch.empty((nb, T * O * C), device=a.device, dtype=torch.float32)
            call("mma_tower_linear_bwd", ptr(g), ptr(a), ptr(Wc), ptr(ga), ptr(part), nb, N, T, O, C, stream_ptr())
            return ga, col_sum(part).view(T, O, C)
        ga = gW = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(a, memory_format=torch.contiguous_format)
            torch.bmm(g.transpose(0, 1), W, out=ga.transpose(0, 1))          # strided-batched: ldc = T*C, batch stride C
        if ctx.needs_input_grad[1]:
            gW = torch.stack([xt_g(g[:, t], a[:, t]) for t in range(T)])
        return ga, gW


def tower_linear(a, W):
    return _TowerLinear.apply(a, W)
