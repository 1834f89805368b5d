"""K13 / K14 / K15 (MMAConv's post-NN on the unscaled aggregates, degree scalers as row factors) and K16 (the plain skinny Linear on the
same kernels) against plain torch in float64: the reference computes  out = cat_q(agg * prod_{q'<=q} scaler_q'(deg))  (mma_conv.py:181-196)
and  post_nns[t](cat[x, out])  (:132-134); here  y[n,t,o] = sum_q pre_q(deg_n) sum_kf agg[n,t,kf] Wo[t][o][q*KF+kf]  on the fp32 matrix
cores.  Shapes: ragged node counts (not a multiple of 64 / 256), KF not a multiple of the 32-column tile, O < 16, S = 1..5, empty targets."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SC = ["identity", "amplification", "attenuation", "linear", "inverse_linear"]


def _pre(deg, scalers, avg_log, avg_lin):
    d = deg.clamp(min=1).double()
    lg = torch.log(d + 1)
    f = {"identity": torch.ones_like(d), "amplification": lg / avg_log, "attenuation": avg_log / lg, "linear": d / avg_lin, "inverse_linear": avg_lin / d}
    run, out = torch.ones_like(d), []
    for s in scalers:
        run = run * f[s]
        out.append(run)
    return torch.stack(out, 1)                                    # (N, S)


@pytest.mark.parametrize("N,T,KF,O,scalers", [(1000, 5, 152, 15, SC[:1] + SC[1:2] + SC[3:4]), (77, 1, 4, 1, SC[:1]), (4097, 3, 36, 16, SC),
                                              (300, 2, 100, 7, SC[1:3]), (64, 4, 32, 16, SC[4:5] + SC[:1])])
def test_tower_post_forward_backward_against_float64(N, T, KF, O, scalers):
    from mma_amd import functional as Fn
    g = torch.Generator().manual_seed(N + KF)
    S = len(scalers)
    deg = torch.randint(0, 7, (N,), generator=g)
    deg[::17] = 0                                                 # empty targets: clamp(1)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(deg, 0)]).to(torch.int32).to(DEV)
    agg = torch.randn(N, T, KF, generator=g)
    Wo = torch.randn(T, O, S * KF, generator=g) / np.sqrt(S * KF)
    cot = torch.randn(N, T * O, generator=g)
    avg_log, avg_lin = 1.3, 2.2
    # float64 reference
    a64, w64 = agg.double().requires_grad_(True), Wo.double().requires_grad_(True)
    pre = _pre(deg, scalers, avg_log, avg_lin)                    # (N,S)
    out = torch.cat([a64 * pre[:, q].view(N, 1, 1) for q in range(S)], -1)       # (N,T,S*KF), q-major: mma_conv.py:195-196
    y64 = torch.einsum("ntc,toc->nto", out, w64).reshape(N, T * O)
    ga64, gw64 = torch.autograd.grad((y64 * cot.double()).sum(), [a64, w64])
    # HIP
    ad, wd = agg.to(DEV).requires_grad_(True), Wo.to(DEV).requires_grad_(True)
    y = Fn.tower_post(ad, wd, rowptr, scalers, avg_log, avg_lin)
    ga, gw = torch.autograd.grad((y * cot.to(DEV)).sum(), [ad, wd])
    mag = (out.abs().unsqueeze(2) * w64.detach().abs().unsqueeze(0)).sum(-1).reshape(N, T * O)      # sum |a||w| per output
    assert ((y.cpu().double() - y64.detach()).abs() <= 3e-7 * mag + 1e-9).all(), "y"
    for got, ref, what in ((ga, ga64, "gagg"), (gw, gw64, "gWo")):
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item() + 1e-7, (what, err, ref.abs().max().item())


@pytest.mark.parametrize("N,K,O,bias", [(5000, 75, 75, True), (4097, 50, 80, False), (4200, 128, 16, True), (4500, 3, 1, True), (6000, 129, 33, True)])
def test_skinny_linear_against_float64(N, K, O, bias):
    from mma_amd import dense
    g = torch.Generator().manual_seed(N + K + O)
    x = torch.randn(N, K, generator=g)
    W = torch.randn(O, K, generator=g) / np.sqrt(K)
    b = torch.randn(O, generator=g) if bias else None
    cot = torch.randn(N, O, generator=g)
    x64, w64 = x.double().requires_grad_(True), W.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if bias else None
    y64 = torch.nn.functional.linear(x64, w64, b64)
    gr = torch.autograd.grad((y64 * cot.double()).sum(), [x64, w64] + ([b64] if bias else []))
    xd, wd = x.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if bias else None
    assert dense._skinny_ok(xd, wd), "this shape must take the K16 path"
    y = dense.linear(xd, wd, bd)
    gg = torch.autograd.grad((y * cot.to(DEV)).sum(), [xd, wd] + ([bd] if bias else []))
    mag = x.double().abs() @ W.double().abs().t() + (b.double().abs() if bias else 0)
    assert ((y.cpu().double() - y64.detach()).abs() <= 3e-7 * mag + 1e-9).all()
    for got, ref, what in zip(gg, gr, ("gx", "gW", "gb")):
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item() + 1e-7, (what, err)
    # a row-strided input (column block of a wider buffer) takes the dword path
    wide = torch.randn(N, K + 5, generator=g).to(DEV)
    y2 = dense.linear(wide[:, 2:2 + K], wd.detach(), bd.detach() if bias else None)
    ref2 = torch.nn.functional.linear(wide[:, 2:2 + K].double(), wd.detach().double(), bd.detach().double() if bias else None)
    assert (y2.double() - ref2).abs().max().item() <= 1e-5 * ref2.abs().max().item()


def test_batched_tn_product_against_float64():
    from mma_amd import dense
    g = torch.Generator().manual_seed(1)
    M, ka, nc, B = 70001, 48, 152, 5
    x = torch.randn(M, B * ka, generator=g).to(DEV)
    y = torch.randn(M, B * nc, generator=g).to(DEV)
    out = dense.xt_g_batched(x, ka, y, nc, B)
    for b in range(B):
        xb, yb = x[:, b * ka:(b + 1) * ka].double(), y[:, b * nc:(b + 1) * nc].double()
        ref, mag = xb.t() @ yb, xb.abs().t() @ yb.abs()
        assert ((out[b].double() - ref).abs() <= 5e-7 * mag).all(), b          # the six-product kernel's bound (tests/test_gemm_gpu.py)
    one = torch.stack([dense.xt_g(x[:, b * ka:(b + 1) * ka], y[:, b * nc:(b + 1) * nc]) for b in range(B)])
    assert (out - one).abs().max().item() <= 5e-7 * mag.max().item()
