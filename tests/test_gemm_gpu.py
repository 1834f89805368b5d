"""The bf16x3 GEMM (fp32 accuracy on the bf16 matrix cores) against an fp64 product: its error must be at the level of a
plain fp32 GEMM's (which is what the reference computes with), far inside the 1e-5 parity bar."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,K,N", [(5000, 128, 512), (4099, 512, 128), (9001, 256, 64), (4096, 128, 32), (70000, 384, 96),
                                   (5003, 256, 384), (4100, 1024, 256),       # these two: column-blocked (K > 128 and N > 128)
                                   (40037, 128, 1024), (70001, 128, 512), (140001, 128, 256), (263000, 128, 128),    # column-group kernel + tail
                                   (66001, 256, 128), (131075, 1024, 128)])                                         # N == 128 long-K kernel + tail
@pytest.mark.parametrize("f16x2", [True, False], ids=["f16x2", "bf16x3"])
def test_gemm_bf16x3_accuracy(M, K, N, f16x2, monkeypatch):
    """f16x2: the K == 128, N % 128 == 0 products take the three-product fp16 kernel (mma_gemm_f16x2); bf16x3: everything on the
    six-product kernels."""
    from mma_amd import dense
    monkeypatch.setattr(dense, "USE_F16X2", f16x2)
    if f16x2 and not (K == 128 and N % 128 == 0):
        pytest.skip("not a shape of the three-product kernel")
    rng = np.random.default_rng(M + K + N)
    a = torch.from_numpy((rng.standard_normal((M, K)) * np.exp(rng.uniform(-6, 6, (M, 1)))).astype(np.float32)).to(DEV)
    w = torch.from_numpy(((rng.random((K, N)) * 2 - 1) / np.sqrt(K)).astype(np.float32)).to(DEV)
    ref = a.double() @ w.double()
    got = dense.gemm_bf16x3(a, w)
    f32 = a @ w
    scale = (a.double().abs() @ w.double().abs())            # per-element sum of |terms|
    e_got = ((got.double() - ref).abs() / scale).max().item()
    e_f32 = ((f32.double() - ref).abs() / scale).max().item()
    assert e_got < 5e-7 and e_got < 1.5 * e_f32 + 1e-7, (e_got, e_f32)   # fp32 eps 6e-8; the fp32 library GEMM sits at 1-3e-7
    # exactness of the split on values that fit one piece, and determinism
    assert torch.equal(dense.gemm_bf16x3(a, w), got)
    ones = torch.ones(M, K, device=DEV)
    wi = torch.zeros(K, N, device=DEV); wi[:N, :] = torch.eye(N, device=DEV) if K >= N else 0
    if K >= N:
        assert torch.equal(dense.gemm_bf16x3(ones, wi), torch.ones(M, N, device=DEV))


def test_mm_autograd_uses_x3_and_matches():
    from mma_amd import dense
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((20000, 128)).astype(np.float32)).to(DEV).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((128, 512)) * 0.1).astype(np.float32)).to(DEV).requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((20000, 512)).astype(np.float32)).to(DEV)
    y = dense.mm(x, w)
    gx, gw = torch.autograd.grad((y * cot).sum(), [x, w])
    yr = x.double() @ w.double()
    gxr = cot.double() @ w.double().t()
    gwr = x.double().t() @ cot.double()
    for a, b, name in ((y, yr, "y"), (gx, gxr, "gx"), (gw, gwr, "gw")):
        err = (a.double() - b).abs().max().item() / b.abs().max().item()
        assert err < 1e-5, (name, err)


@pytest.mark.parametrize("R,C,pad", [(0, 7, 0), (1, 1, 0), (3, 75, 0), (1000, 375, 0), (1025, 64, 0), (204552, 375, 0),
                                     (427376, 75, 0), (50000, 15, 5), (300001, 130, 2), (257, 3, 0), (1344, 760, 0), (8192, 75, 1), (8193, 75, 0),
                                     (1411, 2049, 0), (2708, 7, 0)])
def test_col_sum(R, C, pad):
    """K8 column sums: fixed-order (bitwise repeatable), fp32-accurate against an fp64 sum, strided rows, ragged sizes."""
    from mma_amd import dense
    rng = np.random.default_rng(R + C)
    full = torch.from_numpy(rng.standard_normal((R, C + pad)).astype(np.float32)).to(DEV)
    g = full[:, :C] if pad else full
    got = dense.col_sum(g)
    ref = g.double().sum(0)
    tol = 1e-6 * g.double().abs().sum(0) + 1e-30
    assert got.shape == (C,) and bool(((got.double() - ref).abs() <= tol).all())
    assert torch.equal(dense.col_sum(g), got)


def test_linear_backward_matches_torch():
    """dense.linear / bias_add: same forward as F.linear, gradients through the split-reduction dW and the K8 column sum."""
    from mma_amd import dense
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((70001, 75)).astype(np.float32)).to(DEV).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((375, 75)) * 0.1).astype(np.float32)).to(DEV).requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal(375).astype(np.float32)).to(DEV).requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((70001, 375)).astype(np.float32)).to(DEV)
    y = dense.bias_add(dense.linear(x, w, b), b)
    got = torch.autograd.grad((y * cot).sum(), [x, w, b])
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = torch.nn.functional.linear(xd, wd, bd) + bd
    ref = torch.autograd.grad((yd * cot.double()).sum(), [xd, wd, bd])
    assert torch.allclose(y.double(), yd, rtol=1e-5, atol=1e-5)
    for g_, r_ in zip(got, ref):
        assert (g_.double() - r_).abs().max().item() <= 2e-6 * r_.abs().max().item() + 1e-6 * 70001 ** 0.5


@pytest.mark.parametrize("M,KA,NC,padx,padg", [(5000, 128, 1024, 0, 0), (4097, 64, 96, 0, 32), (70001, 128, 512, 128, 0),
                                               (33, 32, 32, 0, 0), (262144 + 17, 96, 160, 0, 0), (6000, 256, 512, 0, 0),
                                               (204552, 75, 76, 0, 0), (5001, 9, 33, 3, 1), (7000, 128, 129, 0, 0), (4100, 1, 200, 0, 0),
                                               (2708, 64, 7, 0, 0), (19717, 16, 3, 0, 0), (2708, 64, 256, 0, 0), (1500, 50, 25, 0, 0),
                                               (3000, 25, 1, 0, 3)])
def test_gemm_bf16x3_tn_accuracy(M, KA, NC, padx, padg):
    """x^T g on the TN kernel: error at the level of an fp32 GEMM's against fp64, repeatable, row-strided operands."""
    from mma_amd import dense
    rng = np.random.default_rng(M + KA + NC)
    xf = torch.from_numpy((rng.standard_normal((M, KA + padx)) * np.exp(rng.uniform(-3, 3, (M, 1)))).astype(np.float32)).to(DEV)
    gf = torch.from_numpy(rng.standard_normal((M, NC + padg)).astype(np.float32)).to(DEV)
    x, g = xf[:, padx:], gf[:, :NC]
    got = dense.gemm_bf16x3_tn(x, g)
    ref = x.double().t() @ g.double()
    f32 = x.t() @ g
    scale = x.double().abs().t() @ g.double().abs()
    e_got = ((got.double() - ref).abs() / scale).max().item()
    e_f32 = ((f32.double() - ref).abs() / scale).max().item()
    assert got.shape == (KA, NC) and e_got < 5e-7 and e_got < 1.5 * e_f32 + 1e-7, (e_got, e_f32)
    assert torch.equal(dense.gemm_bf16x3_tn(x, g), got)


@pytest.mark.parametrize("M,KA,NC,padx,padg", [(70001, 128, 1024, 0, 0), (4097, 64, 96, 0, 32), (70001, 128, 512, 128, 0),
                                               (33, 32, 32, 0, 0), (262144 + 17, 96, 160, 0, 0), (66000, 256, 512, 0, 0),
                                               (70001, 75, 76, 0, 0), (5001, 9, 33, 3, 1), (300007, 384, 160, 0, 4)])
@pytest.mark.parametrize("maxima", ["given", "loose", "own"])
def test_gemm_f16x2_tn_accuracy(M, KA, NC, padx, padg, maxima):
    """x^T g on the THREE-product TN kernel (fp16 x 2 pieces, row scales balanced between the operands): error at the level of an
    fp32 GEMM's against fp64 with the row sizes of both operands spread over e^+-3 (a spread of ~2^17 in the row products),
    ReLU zeros in x, all-zero rows, exact / 8x-loose / kernel-made row maxima, ragged M, row-strided operands; repeatable."""
    from mma_amd import dense
    rng = np.random.default_rng(M + KA + NC)
    xf = np.maximum(rng.standard_normal((M, KA + padx)), 0) * np.exp(rng.uniform(-3, 3, (M, 1)))
    gf = rng.standard_normal((M, NC + padg)) * 1e-3 * np.exp(rng.uniform(-3, 3, (M, 1)))
    xf[::11] = 0
    gf[::13] = 0
    xf, gf = torch.from_numpy(xf.astype(np.float32)).to(DEV), torch.from_numpy(gf.astype(np.float32)).to(DEV)
    x, g = xf[:, padx:], gf[:, :NC]
    xm, gm = x.abs().amax(1), g.abs().amax(1)
    if maxima == "loose":
        xm, gm = xm * 8.0, gm * 3.0
    elif maxima == "own":
        xm = gm = None
    got = dense.gemm_f16x2_tn(x, g, xm, gm)
    six = dense.gemm_bf16x3_tn(x, g)
    ref = x.double().t() @ g.double()
    f32 = x.t() @ g
    scale = x.double().abs().t() @ g.double().abs()
    e_got = ((got.double() - ref).abs() / scale).max().item()
    e_six = ((six.double() - ref).abs() / scale).max().item()
    e_f32 = ((f32.double() - ref).abs() / scale).max().item()
    assert got.shape == (KA, NC) and e_got < 5e-7 and e_got < 1.5 * e_f32 + 1e-7, (e_got, e_six, e_f32)
    assert not torch.equal(got, six)                          # the three-product kernel did the work, not the fall-back
    assert torch.equal(dense.gemm_f16x2_tn(x, g, xm, gm), got)


@pytest.mark.parametrize("M,KA,NC,padx", [(70001, 256, 2048, 0), (66000, 128, 2048, 4), (65537, 200, 2100, 0), (70001, 64, 384, 0), (9000, 96, 256, 0)])
def test_gemm_f16x2_tn_packed_x_is_bit_equal(M, KA, NC, padx, monkeypatch):
    """[r5] X packed once per call in fragment order (tn_pack_x_kernel; default from 16 column blocks of 128 on, MMA_TN_PACKX forces it, read
    per call): the same pieces and the same products as the kernel that loads and splits the fp32 rows per column block - the same bits,
    for whole and ragged tiles, a row-strided X, splits that end inside the last chunk, and the device-side fall-back (a spread of 2^50)."""
    from mma_amd import dense
    rng = np.random.default_rng(M + KA + NC)
    xf = torch.from_numpy((rng.standard_normal((M, KA + padx)) * np.exp(rng.uniform(-4, 4, (M, 1)))).astype(np.float32)).to(DEV)
    g = torch.from_numpy((rng.standard_normal((M, NC)) * np.exp(rng.uniform(-4, 4, (M, 1)))).astype(np.float32)).to(DEV)
    x = xf[:, padx:]
    xm, gm = x.abs().amax(1), g.abs().amax(1)
    res = {}
    for v in ("0", "1"):
        monkeypatch.setenv("MMA_TN_PACKX", v)
        res[v] = dense.gemm_f16x2_tn(x, g, xm, gm)
    assert torch.equal(res["0"], res["1"])
    ref = x.double().t() @ g.double()
    scale = x.double().abs().t() @ g.double().abs()
    assert ((res["1"].double() - ref).abs() / scale).max().item() < 5e-7
    monkeypatch.delenv("MMA_TN_PACKX")
    assert torch.equal(dense.gemm_f16x2_tn(x, g, xm, gm), res["0"])          # the default, whichever form the shape takes
    x2 = x.clone(); x2[5] *= 2.0 ** 50                                           # fall-back decided on the device: packed or not, six products
    xm2 = x2.abs().amax(1)
    monkeypatch.setenv("MMA_TN_PACKX", "1")
    assert torch.equal(dense.gemm_f16x2_tn(x2, g, xm2, gm), dense.gemm_bf16x3_tn(x2, g))


@pytest.mark.parametrize("KA", [128, 256])
@pytest.mark.parametrize("case", ["spread", "inf", "nan", "subnormal", "zero"])
def test_gemm_f16x2_tn_falls_back_on_the_device(case, KA):
    """Rows whose products lie more than 2^40 apart, or an inf / NaN / subnormal row maximum: the scale kernels flag the call and
    the six-product kernel does it - the result is bit-for-bit gemm_bf16x3_tn's.  All-zero operands give exact zeros.  KA = 256: the
    128-column blocks of x (hidden width 256, C5)."""
    from mma_amd import dense
    rng = np.random.default_rng(7)
    M, NC = 70001, 256
    x = rng.standard_normal((M, KA)).astype(np.float32)
    g = rng.standard_normal((M, NC)).astype(np.float32)
    if case == "spread":
        x *= np.exp(rng.uniform(-25, 25, (M, 1))).astype(np.float32)
    elif case == "inf":
        x[1234, 5::128] = np.inf                  # in every 128-column block of x: the blocks decide on their own
    elif case == "nan":
        g[4321, 7] = np.nan
    elif case == "subnormal":
        x[77] = 1e-41
        x[77, 3] = 3e-39
    elif case == "zero":
        x[:] = 0
    x, g = torch.from_numpy(x).to(DEV), torch.from_numpy(g).to(DEV)
    got = dense.gemm_f16x2_tn(x, g)
    if case == "zero":
        assert torch.equal(got, torch.zeros_like(got))
        return
    six = dense.gemm_bf16x3_tn(x, g)
    assert torch.equal(got.isnan(), six.isnan()) and torch.equal(got.nan_to_num(0.0, 1.0, -1.0), six.nan_to_num(0.0, 1.0, -1.0))


@pytest.mark.parametrize("K,N,transposed", [(128, 1024, False), (1024, 128, True), (256, 4096, False), (64, 1, False), (1, 7, True)])
def test_split_f16x2_one_launch_matches_the_torch_expression(K, N, transposed):
    """mma_split_f16x2 (B operand of the three-product kernels: column scales, hi / lo pieces, k contiguous) against the torch
    expression it replaced, bit for bit, for plain and transposed-view weights, zero columns and columns near the ends of the range."""
    from mma_amd import dense
    rng = np.random.default_rng(K + N)
    w = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-40, 40, (1, N)))).astype(np.float32)
    w[:, N // 2] = 0
    w = torch.from_numpy(w).to(DEV)
    if transposed:
        w = w.t().contiguous().t()                                       # same values, (K,N) view of an (N,K) buffer
    bt2, cu = dense._split_f16x2(w)
    amax = w.abs().amax(0).clamp_min(1e-30)
    s = torch.exp2(14.0 - torch.floor(torch.log2(amax.double()))).float()
    x = (w * s).t().contiguous()
    hi = x.half()
    assert torch.equal(cu, 1.0 / s) and torch.equal(bt2[0], hi) and torch.equal(bt2[1], ((x - hi.float()) * 2048.0).half())
    bt2p, cup = dense._split_f16x2(w, plain_lo=True)                     # the one-accumulator kernels' operand: lo as it is
    assert torch.equal(cup, cu) and torch.equal(bt2p[0], hi) and torch.equal(bt2p[1], (x - hi.float()).half())


@pytest.mark.parametrize("M,C,pad", [(1, 1, 0), (7, 33, 3), (70001, 256, 0), (5000, 75, 5), (1025, 4096, 0)])
def test_row_absmax(M, C, pad):
    from mma_amd import dense
    rng = np.random.default_rng(M + C)
    a = torch.from_numpy(rng.standard_normal((M, C + pad)).astype(np.float32)).to(DEV)[:, :C]
    assert torch.equal(dense.row_absmax(a), a.abs().amax(1))
    if M > 3:
        a = a.clone(); a[2, C // 2] = float("nan"); a[3, 0] = float("inf")
        r = dense.row_absmax(a)
        assert bool(r[2].isnan()) and bool(r[3].isinf())


@pytest.mark.parametrize("M,N", [(66001, 4096), (70000, 384), (131075, 1024), (65536, 256)])
def test_gemm_f16x2_k256(M, N, monkeypatch):
    """K = 256 column-group three-product kernel (C5 forward shape) against fp64: rows spread over e^+-6, zero rows, ragged M; the
    chunked N = 128 kernel on the same operands gives the same error level."""
    from mma_amd import dense
    rng = np.random.default_rng(M + N)
    a = (rng.standard_normal((M, 256)) * np.exp(rng.uniform(-6, 6, (M, 1)))).astype(np.float32)
    a[::17] = 0
    a = torch.from_numpy(a).to(DEV)
    w = torch.from_numpy(((rng.random((256, N)) * 2 - 1) / 16).astype(np.float32)).to(DEV)
    got = dense.gemm_bf16x3(a, w)
    f32 = a @ w
    idx = torch.from_numpy(rng.choice(M, 4096, replace=False)).to(DEV)          # fp64 truth on a row sample
    ref = a[idx].double() @ w.double()
    scale = a[idx].double().abs() @ w.double().abs() + 1e-300
    e_got = ((got[idx].double() - ref).abs() / scale).max().item()
    e_f32 = ((f32[idx].double() - ref).abs() / scale).max().item()
    assert e_got < 5e-7 and e_got < 1.5 * e_f32 + 1e-7, (e_got, e_f32)
    assert torch.equal(got[::17], torch.zeros_like(got[::17]))
    assert torch.equal(dense.gemm_bf16x3(a, w), got)
    box = []
    dense.gemm_bf16x3(a, w, row_max_box=box)
    assert len(box) == 1 and torch.equal(box[0], a.abs().amax(1))
    # [r5] the default path packs A once (mma_pack_f16x2_k256 + mma_gemm_f16x2_k256p); round 4's kernel splits the fp32 rows per column
    # group: the same pieces, the same products in the same order - the same bits; a row-strided A as well
    monkeypatch.setattr(dense, "PACK_K256", False)
    assert torch.equal(dense.gemm_bf16x3(a, w), got)
    monkeypatch.setattr(dense, "PACK_K256", True)
    buf = torch.zeros((M, 256 + 8), device=DEV)
    buf[:, :256] = a
    assert torch.equal(dense.gemm_bf16x3(buf[:, :256], w), got)


def test_forward_gemm_exports_the_row_maxima():
    """mma_gemm_f16x2 leaves max |a[i,:]| in a_row_max (ragged M), and the layer-level autograd product through xt_g matches."""
    from mma_amd import dense
    rng = np.random.default_rng(3)
    a = torch.from_numpy((rng.standard_normal((70001, 128)) * np.exp(rng.uniform(-6, 6, (70001, 1)))).astype(np.float32)).to(DEV)
    a[17] = 0
    w = torch.from_numpy((rng.standard_normal((128, 256)) * 0.1).astype(np.float32)).to(DEV)
    box = []
    out = torch.empty((70001, 256), device=DEV)
    dense.mm_into(a, w, out, row_max_box=box)
    assert len(box) == 1 and torch.equal(box[0], a.abs().amax(1))
    g = torch.from_numpy(rng.standard_normal((70001, 256)).astype(np.float32)).to(DEV)
    got = dense.xt_g(a, g, box[0], g.abs().amax(1))
    assert torch.equal(got, dense.gemm_f16x2_tn(a, g, box[0], g.abs().amax(1)))


@pytest.mark.parametrize("N,T,O,C", [(1, 1, 1, 4), (7, 2, 3, 12), (1000, 5, 15, 456), (70001, 3, 16, 260), (513, 1, 15, 456)])
def test_tower_linear_backward(N, T, O, C):
    """K9: both gradients of y[n,t,:] = a[n,t,:] W[t]^T in one pass over `a`, against an fp64 einsum; repeatable."""
    from mma_amd import dense
    rng = np.random.default_rng(N + T + O + C)
    a = torch.from_numpy(rng.standard_normal((N, T, C)).astype(np.float32)).to(DEV).requires_grad_(True)
    W = torch.from_numpy((rng.standard_normal((T, O, C)) * 0.1).astype(np.float32)).to(DEV).requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((N, T, O)).astype(np.float32)).to(DEV)
    y = dense.tower_linear(a, W)
    ga, gW = torch.autograd.grad((y * cot).sum(), [a, W])
    yd = torch.einsum('ntc,toc->nto', a.double(), W.double())
    assert torch.allclose(y.double(), yd, rtol=1e-5, atol=1e-5)
    ga_ref = torch.einsum('nto,toc->ntc', cot.double(), W.double())
    gW_ref = torch.einsum('nto,ntc->toc', cot.double(), a.double())
    assert (ga.double() - ga_ref).abs().max().item() <= 1e-6 * (cot.double().abs().unsqueeze(-1) * W.double().abs().unsqueeze(0)).sum(2).max().item() + 1e-30
    scale = torch.einsum('nto,ntc->toc', cot.double().abs(), a.double().abs())
    assert bool(((gW.double() - gW_ref).abs() <= 2e-6 * scale + 1e-30).all())
    ga2, gW2 = torch.autograd.grad((dense.tower_linear(a, W) * cot).sum(), [a, W])
    assert torch.equal(ga, ga2) and torch.equal(gW, gW2)


@pytest.mark.parametrize("M,K,N", [(5000, 128, 512), (4099, 512, 128), (9001, 1024, 128), (70001, 512, 128)])
def test_gemm_bf16x3_accumulate(M, K, N, monkeypatch):
    """accumulate=True: C += A B in the kernel epilogue (used for dL/dx = direct part + g [Wtop|Wbot]^T)."""
    from mma_amd import dense
    monkeypatch.setattr(dense, "USE_F16X2", False)          # `plain` below must be the six-product kernel's product as well
    rng = np.random.default_rng(M + K + N + 1)
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(DEV)
    w = torch.from_numpy(((rng.random((K, N)) * 2 - 1) / np.sqrt(K)).astype(np.float32)).to(DEV)
    c0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV)
    plain = dense.gemm_bf16x3(a, w)
    acc = c0.clone()
    got = dense.rows_mm_add_(acc, a, w)
    assert got.data_ptr() == acc.data_ptr()
    ref = c0.double() + a.double() @ w.double()
    assert torch.equal(acc, c0 + plain)                       # one fp32 addition on top of the same product (M >= 65536: by an L2 atomic)
    again = c0.clone(); dense.rows_mm_add_(again, a, w)
    assert torch.equal(again, acc)
    assert (acc.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("N,fin,fout,bias,padded_grad", [(70001, 75, 760, True, True), (66000, 50, 380, True, False), (40000, 75, 380, False, True),
                                                         (33000, 127, 129, True, False)])
def test_linear_tall_on_the_bf16x3_kernels(N, fin, fout, bias, padded_grad):
    """dense.linear_tall: odd-width tall Linear layers zero-padded onto the bf16x3 kernels (bias in the ones column of the K padding,
    bias gradient = one more row of the TN product), against fp64; the upstream gradient arrives either as a view of a registered
    zero-padded buffer (what K4 hands over: no copy) or as a plain tensor (padded by a copy)."""
    from mma_amd import dense
    rng = np.random.default_rng(N + fin + fout)
    x = torch.from_numpy(rng.standard_normal((N, fin)).astype(np.float32)).to(DEV).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((fout, fin)) * 0.1).astype(np.float32)).to(DEV).requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal(fout).astype(np.float32)).to(DEV).requires_grad_(True) if bias else None
    assert dense.linear_x3_ok(x, w)
    y = dense.linear_tall(x, w, b)
    assert y.shape == (N, fout) and y.stride(0) == -(-fout // 128) * 128
    cot_np = rng.standard_normal((N, fout)).astype(np.float32)
    if padded_grad:
        cot = dense.padded_empty(N, fout, DEV)
        cot.copy_(torch.from_numpy(cot_np))
        assert dense._padded_parent(cot, y.stride(0)) is not None
    else:
        cot = torch.from_numpy(cot_np).to(DEV)
    ins = [x, w] + ([b] if bias else [])
    got = torch.autograd.grad(y, ins, grad_outputs=cot)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    yd = torch.nn.functional.linear(xd, wd, bd)
    ref = torch.autograd.grad(yd, [xd, wd] + ([bd] if bias else []), grad_outputs=torch.from_numpy(cot_np).double().to(DEV))
    scale = xd.detach().abs() @ wd.detach().abs().t() + (bd.detach().abs() if bias else 0)
    assert ((y.double() - yd).abs() / scale).max().item() < 5e-7
    for g_, r_ in zip(got, ref):
        assert g_.shape == r_.shape
        assert (g_.double() - r_).abs().max().item() <= 2e-6 * r_.abs().max().item() + 1e-6 * N ** 0.5
    dense.X3_LINEAR = False
    try:
        assert not dense.linear_x3_ok(x, w)
    finally:
        dense.X3_LINEAR = True


@pytest.mark.parametrize("K", [64, 96])
@pytest.mark.parametrize("M,N", [(4099, 128), (9001, 256), (40037, 384), (70001, 768), (5000, 640), (33001, 1152)])
def test_gemm_f16x2_narrow_reductions(M, N, K, monkeypatch):
    """[r5] mma_gemm_f16x2_k, K in {64, 96}: the column-group kernel with 4 / 6 k-steps and one, two or THREE resident 128-column groups
    (N / 128 = 1, 2, 3, 6, 5, 9): the accuracy of the K = 128 form, the same bits as that form on the operand padded with zero columns
    (zero products add exactly), the row maxima, a ragged last block, a row-strided A, and the exported G2 switch."""
    from mma_amd import dense
    rng = np.random.default_rng(M + N + K)
    a = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-6, 6, (M, 1)))).astype(np.float32)
    a[3] = 0.0
    a[5] = 0.0; a[5, K - 1] = 1.0e-30
    buf = torch.zeros((M, 128 + 12), device=DEV)
    buf[:, :K] = torch.from_numpy(a).to(DEV)
    av = buf[:, :K]                                              # row-strided view (pitch 140 floats)
    w = torch.from_numpy(((rng.random((K, N)) * 2 - 1) * np.exp(rng.uniform(-3, 3, (1, N)))).astype(np.float32)).to(DEV)
    rm = torch.empty(M, device=DEV)
    got = dense.gemm_f16x2(av, w, row_max_out=rm)
    ref = av.double() @ w.double()
    scale = av.double().abs() @ w.double().abs()
    err = (got.double() - ref).abs() / scale.clamp_min(1e-300)
    assert torch.isfinite(got).all() and (got[3] == 0).all()
    assert err[scale > 0].max().item() < 5e-7, err[scale > 0].max().item()
    assert torch.equal(rm, av.abs().amax(1))
    assert torch.equal(got, dense.gemm_f16x2(av, w))
    wide = torch.zeros((128, N), device=DEV); wide[:K] = w
    assert torch.equal(got, dense.gemm_f16x2(buf[:, :128], wide)), "zero pad columns change no bit"
    assert torch.equal(got, dense.gemm_f16x2(av.contiguous(), w))
    monkeypatch.setenv("MMA_F16X2_G2", "1")                      # read per call: one resident group per workgroup
    assert torch.equal(got, dense.gemm_f16x2(av, w))


def test_gemm_f16x2_narrow_rejects_other_widths():
    from mma_amd import dense
    from mma_amd._lib import MMALibraryError
    a, w = torch.randn(5000, 80, device=DEV), torch.randn(80, 128, device=DEV)
    with pytest.raises(MMALibraryError, match="K=80"):
        dense.gemm_f16x2(a, w)


def test_gemm_f16x2_edge_rows():
    """Three-product kernel: all-zero rows, rows of one huge / one tiny value (the power-of-two row scale must keep both exact),
    a ragged last block, a row-strided A, and a wide dynamic range INSIDE rows."""
    from mma_amd import dense
    rng = np.random.default_rng(5)
    M, K, N = 9000 + 77, 128, 256
    a = rng.standard_normal((M, K)).astype(np.float32)
    a[3] = 0.0
    a[4] = 0.0; a[4, 17] = 1.0e36              # x w up to 2e37: finite only if the two un-scalings are applied as one
    a[5] = 0.0; a[5, 99] = 1.0e-30
    a[6] *= np.exp(rng.uniform(-20, 20, K)).astype(np.float32)
    buf = torch.zeros((M, K + 64), device=DEV)
    buf[:, :K] = torch.from_numpy(a).to(DEV)
    av = buf[:, :K]                                              # row-strided view
    w = torch.from_numpy(((rng.random((K, N)) * 2 - 1) * np.exp(rng.uniform(-3, 3, (1, N)))).astype(np.float32)).to(DEV)
    got = dense.gemm_f16x2(av, w)
    ref = av.double() @ w.double()
    scale = av.double().abs() @ w.double().abs()
    err = (got.double() - ref).abs() / scale.clamp_min(1e-300)
    assert torch.isfinite(got).all()
    assert (got[3] == 0).all()
    assert err[scale > 0].max().item() < 5e-7, err[scale > 0].max().item()
    assert torch.equal(got, dense.gemm_f16x2(av, w))


@pytest.mark.parametrize("M,K,N,acc", [(70001, 1024, 256, True), (66000, 4096, 256, False), (65536 + 255, 256, 128, True), (300, 512, 256, True)])
def test_gemm_f16x2_nlp_one_accumulator_form(M, K, N, acc, monkeypatch):
    """Round 4: the pipelined dL/dx kernel (mma_gemm_f16x2_nlp) - PLAIN lo pieces in both operands, one fp32 accumulator tile, two raw
    chunks of A in flight, N = 128 or 256 in ONE pass - against float64 under the bound of the two-accumulator form (5e-7 sum|a||b| and
    the one fp32 addition onto C), with rows of very different sizes, a wide range INSIDE rows (elements 2^-20 of their row maximum keep
    their relative precision in the bound's terms), all-zero rows, loose row bounds, ragged M (also fewer rows than one 256-row unit),
    bit-repeatable; and against the two-accumulator kernel it replaces."""
    from mma_amd import dense
    rng = np.random.default_rng(M + K + N)
    a = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-12, 12, (M, 1)))).astype(np.float32)
    a[7] *= np.exp(rng.uniform(-14, 0, K)).astype(np.float32)            # a wide range inside one row
    a[5] = 0.0
    a[M - 1] = 0.0
    at = torch.from_numpy(a).to(DEV)
    w = torch.from_numpy(((rng.random((K, N)) * 2 - 1) / np.sqrt(K) * np.exp(rng.uniform(-2, 2, (1, N)))).astype(np.float32)).to(DEV)
    c0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV) if acc else torch.zeros((M, N), device=DEV)
    rm = at.abs().amax(1)
    rm[::3] *= 8.0                                               # a bound, not the exact maximum
    assert dense.USE_NLP
    out = c0.clone()
    dense.gemm_f16x2_n128(at, rm, w, out, accumulate=acc)
    prod = at.double() @ w.double()
    scale = at.double().abs() @ w.double().abs()
    got_prod = out.double() - (c0.double() if acc else 0)
    nz = scale > 0
    tol = 5e-7 * scale + (1.2e-7 * c0.double().abs() if acc else 0)      # + the one fp32 addition onto c0
    assert ((got_prod - prod).abs()[nz] <= tol[nz]).all(), ((got_prod - prod).abs() / scale.clamp_min(1e-300))[nz].max().item()
    assert torch.equal(out[5], c0[5]) and torch.equal(out[M - 1], c0[M - 1])
    again = c0.clone()
    dense.gemm_f16x2_n128(at, rm, w, again, accumulate=acc)
    assert torch.equal(again, out)
    if M >= (1 << 16):                                           # the two-accumulator kernel it replaces (its own launcher wants tall inputs)
        monkeypatch.setattr(dense, "USE_NLP", False)
        old = c0.clone()
        dense.gemm_f16x2_n128(at, rm, w, old, accumulate=acc)
        assert ((old.double() - out.double()).abs()[nz] <= 2 * tol[nz]).all()


@pytest.mark.parametrize("M,K,acc", [(70001, 1024, True), (66000, 512, False), (65536 + 255, 192, True)])
def test_gemm_f16x2_n128(M, K, acc):
    """Three-product N == 128 kernel (the dL/dx shape): caller-supplied row maxima (exact, or a loose bound 8x above), rows of
    very different magnitudes, all-zero rows with row_max 0, ragged M, C += by one atomic per element (bit-repeatable)."""
    from mma_amd import dense
    rng = np.random.default_rng(M + K)
    a = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-12, 12, (M, 1)))).astype(np.float32)
    a[5] = 0.0
    a[M - 1] = 0.0
    at = torch.from_numpy(a).to(DEV)
    w = torch.from_numpy(((rng.random((K, 128)) * 2 - 1) / np.sqrt(K)).astype(np.float32)).to(DEV)
    c0 = torch.from_numpy(rng.standard_normal((M, 128)).astype(np.float32)).to(DEV) if acc else torch.zeros((M, 128), device=DEV)
    rm = at.abs().amax(1)
    rm[::3] *= 8.0                                               # a bound, not the exact maximum
    out = c0.clone()
    dense.gemm_f16x2_n128(at, rm, w, out, accumulate=acc)
    prod = at.double() @ w.double()
    scale = at.double().abs() @ w.double().abs()
    got_prod = out.double() - (c0.double() if acc else 0)
    nz = scale > 0
    tol = 5e-7 * scale + (1.2e-7 * c0.double().abs() if acc else 0)      # + the one fp32 addition onto c0
    assert ((got_prod - prod).abs()[nz] <= tol[nz]).all(), ((got_prod - prod).abs() / scale.clamp_min(1e-300))[nz].max().item()
    assert torch.equal(out[5], c0[5]) and torch.equal(out[M - 1], c0[M - 1])
    again = c0.clone()
    dense.gemm_f16x2_n128(at, rm, w, again, accumulate=acc)
    assert torch.equal(again, out)


def _need_experimental_fwd():
    """[r5] The W-stationary / loader-wave forward kernels are measurement forms (slower than the column-group kernels): the product
    library is built without them (`make EXTRA=-DMMA_EXPERIMENTAL_FWD` builds them) and mma_gemm_f16x2_ws then says so."""
    from mma_amd import dense
    from mma_amd._lib import call, ptr, stream_ptr, MMALibraryError
    a = torch.zeros((64, 128), device=DEV)
    w = torch.zeros((128, 256), device=DEV)
    bt2, cu = dense._split_f16x2(w)
    out = torch.empty((64, 256), device=DEV)
    try:
        call("mma_gemm_f16x2_ws", ptr(a), 128, ptr(bt2), ptr(cu), ptr(out), 256, None, 64, 256, 128, stream_ptr())
    except MMALibraryError as e:
        if "MMA_EXPERIMENTAL_FWD" in str(e):
            pytest.skip("libmma_amd.so is built without -DMMA_EXPERIMENTAL_FWD (the measurement forms of the forward GEMM)")
        raise


@pytest.mark.parametrize("M,K,N", [(1, 128, 256), (63, 128, 256), (64, 128, 512), (4099, 128, 1024), (70001, 128, 2048), (140001, 128, 256),
                                   (1, 256, 256), (31, 256, 512), (4100, 256, 1024), (66001, 256, 4096), (131075, 256, 256)])
def test_gemm_f16x2_ws_is_bit_equal_to_the_column_group_kernels(M, K, N, monkeypatch):
    """The W-stationary forward kernel (round 4: W in registers, x loaded row-major by the whole workgroup and shared through LDS) forms
    the same three products in the same order as the column-group kernels: identical bits, identical row maxima, ragged M, zero rows, rows
    spread over e^+-6, a row pitch wider than K, an output that is a column block of a wider buffer; and the fp64 error bar on a sample."""
    from mma_amd import dense
    from mma_amd._lib import call, ptr, stream_ptr
    _need_experimental_fwd()
    rng = np.random.default_rng(M + K + N)
    a_np = (rng.standard_normal((M, K + 8)) * np.exp(rng.uniform(-6, 6, (M, 1)))).astype(np.float32)
    a_np[::17] = 0
    a = torch.from_numpy(a_np).to(DEV)[:, :K]                                     # lda = K + 8
    w = torch.from_numpy(((rng.random((K, N)) * 2 - 1) / 16).astype(np.float32)).to(DEV)
    bt2, cu = dense._split_f16x2(w)
    wide = torch.full((M, N + 64), 7.0, device=DEV)
    out, rm = wide[:, 32:32 + N], torch.full((M,), -1.0, device=DEV)
    call("mma_gemm_f16x2_ws", ptr(a), a.stride(0), ptr(bt2), ptr(cu), ptr(out), out.stride(0), ptr(rm), M, N, K, stream_ptr())
    assert torch.equal(rm, a.abs().amax(1))
    assert torch.equal(wide[:, :32], torch.full_like(wide[:, :32], 7.0)) and torch.equal(wide[:, 32 + N:], torch.full_like(wide[:, 32 + N:], 7.0))
    old = torch.empty((M, N), device=DEV)
    if K == 128:
        monkeypatch.delenv("MMA_FWD_WS", raising=False)                            # the default: the column-group kernel
        call("mma_gemm_f16x2", ptr(a), a.stride(0), ptr(bt2), ptr(cu), ptr(old), old.stride(0), None, M, N, stream_ptr())
        # the loader-wave kernel (MMA_FWD_WS=2): eight multiplier waves + two loader waves, same bits and row maxima
        monkeypatch.setenv("MMA_FWD_WS", "2")
        lw, rm2 = torch.full((M, N), 3.0, device=DEV), torch.full((M,), -1.0, device=DEV)
        call("mma_gemm_f16x2", ptr(a), a.stride(0), ptr(bt2), ptr(cu), ptr(lw), lw.stride(0), ptr(rm2), M, N, stream_ptr())
        monkeypatch.delenv("MMA_FWD_WS")
        assert torch.equal(lw, old) and torch.equal(rm2, a.abs().amax(1))
    else:
        call("mma_gemm_f16x2_k256", ptr(a), a.stride(0), ptr(rm), ptr(bt2), ptr(cu), ptr(old), old.stride(0), M, N, stream_ptr())
    assert torch.equal(out, old)
    idx = torch.from_numpy(rng.choice(M, min(M, 2048), replace=False)).to(DEV)
    ref = a[idx].double() @ w.double()
    scale = a[idx].double().abs() @ w.double().abs() + 1e-300
    assert ((out[idx].double() - ref).abs() / scale).max().item() < 5e-7
    # NULL a_row_max; the layer-level call takes this kernel by itself
    out2 = torch.empty((M, N), device=DEV)
    call("mma_gemm_f16x2_ws", ptr(a), a.stride(0), ptr(bt2), ptr(cu), ptr(out2), out2.stride(0), None, M, N, K, stream_ptr())
    assert torch.equal(out2, old)
    if M >= dense._MIN_ROWS_X3 and (K == 128 or M >= (1 << 16)):        # the layer-level call, with the switch on and off
        for sw in ("1", "0"):
            monkeypatch.setenv("MMA_FWD_WS", sw)
            box = []
            assert torch.equal(dense.gemm_bf16x3(a, w, row_max_box=box), old) and torch.equal(box[0], rm)
        monkeypatch.delenv("MMA_FWD_WS")


def test_gemm_f16x2_ws_rejects_shapes_it_does_not_take():
    from mma_amd import dense
    from mma_amd._lib import call, ptr, stream_ptr, MMALibraryError as MMAError
    _need_experimental_fwd()
    a = torch.zeros((64, 128), device=DEV)
    for N, K in ((768, 128), (128, 128), (256, 64), (256, 512)):
        w = torch.zeros((2, N, K), device=DEV, dtype=torch.float16)
        with pytest.raises(MMAError):
            call("mma_gemm_f16x2_ws", ptr(a), 128, ptr(w), ptr(torch.ones(N, device=DEV)), ptr(torch.zeros((64, N), device=DEV)), N, None, 64, N, K,
                 stream_ptr())


@pytest.mark.parametrize("M,NC", [(70001, 512), (131072, 4096), (66000, 96)])
def test_gemm_f16x2_tn_256_columns_of_x_in_one_launch(M, NC):
    """Round 4: x up to 256 columns wide (hidden width 256) is one launch of the eight-wave TN kernel - the staged G tile is shared, G
    is read once - with the bits of one launch per 128-column block of x; the on-device fall-back (a row 2^50 below the rest) too."""
    from mma_amd import dense
    rng = np.random.default_rng(M + NC)
    x = torch.from_numpy(rng.standard_normal((M, 256)).astype(np.float32)).to(DEV)
    g = torch.from_numpy((rng.standard_normal((M, NC)) * np.exp(rng.uniform(-3, 3, (M, 1)))).astype(np.float32)).to(DEV)
    for bad in (False, True):
        if bad:
            x = x.clone(); x[5] *= 2.0 ** -60                      # sends the call to the six-product kernel (decided on the device)
        xm, gm = x.abs().amax(1), g.abs().amax(1)
        dense.TN_KA256 = False
        try:
            ref = dense.gemm_f16x2_tn(x, g, xm, gm)
        finally:
            dense.TN_KA256 = True
        got = dense.gemm_f16x2_tn(x, g, xm, gm)
        assert torch.equal(got, ref)
        truth = x.double().t() @ g.double()
        scale = x.double().abs().t() @ g.double().abs() + 1e-300
        assert ((got.double() - truth).abs() / scale).max().item() < 5e-7
