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


@pytest.mark.parametrize("x3", ["1", "0"], ids=["bf16x3_forward", "fp32_forward"])
@pytest.mark.parametrize("N,K,O,bias", [(5000, 75, 75, True), (4097, 50, 80, False), (4200, 128, 16, True), (4500, 3, 1, True), (6000, 129, 33, True)])
def test_skinny_linear_against_float64(N, K, O, bias, x3, monkeypatch):
    """[r5] the forward runs on three bf16 pieces per operand (the K13 kernel's PLAIN form) by default, MMA_SKINNY_X3=0 (read per call) is
    the exact-fp32 MFMA kernel: both inside 3e-7 sum|x||w| of float64."""
    from mma_amd import dense
    monkeypatch.setenv("MMA_SKINNY_X3", x3)
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


@pytest.mark.parametrize("N,K,O,bias", [(5000, 75, 75, True), (199999, 75, 75, True), (4097, 80, 80, True), (4100, 16, 5, False),
                                        (70000, 512, 64, True), (4099, 1, 1, True), (4, 7, 3, True)])
def test_skinny_weight_and_bias_gradient_in_one_pass(N, K, O, bias):
    """[r5] mma_skinny_linear_gw (K15's kernel in its plain form): gw = g^T x and gb = column sums of g from one pass, the ones column at
    k = K (K = 80 / 16 / 512: a tile of its own).  Inside 1e-6 sum|g||x| of float64 (fp32 matrix cores, fp32 partial sums), the same
    bits run after run, row-strided operands included."""
    from mma_amd import dense
    gen = torch.Generator().manual_seed(N + 3 * K + O)
    gbuf = torch.randn(N, O + 3, generator=gen).to(DEV)
    xbuf = torch.randn(N, K + 6, generator=gen).to(DEV)
    g2, x2 = gbuf[:, 1:1 + O], xbuf[:, 4:4 + K]                 # row-strided views (pitches O + 3 / K + 6: the dword loads)
    gw, gb = dense.skinny_gw(g2, x2, bias)
    gw_b, gb_b = dense.skinny_gw(g2, x2, bias)
    assert torch.equal(gw, gw_b) and (not bias or torch.equal(gb, gb_b)), "fixed summation order: run-to-run bit equality"
    g64, x64 = g2.double(), x2.double()
    ref = g64.t() @ x64
    mag = g64.abs().t() @ x64.abs()
    assert ((gw.double() - ref).abs() <= 1e-6 * mag + 1e-9).all().item(), (gw.double() - ref).abs().max().item()
    if bias:
        assert ((gb.double() - g64.sum(0)).abs() <= 1e-6 * g64.abs().sum(0) + 1e-9).all().item()
    else:
        assert gb is None
    # contiguous operands take the same kernel: same values up to the loads' alignment (no arithmetic differs) -> bit-equal
    gw_c, _ = dense.skinny_gw(g2.contiguous(), x2.contiguous(), bias)
    assert torch.equal(gw, gw_c)


def test_skinny_gradient_switch_matches_the_library_path(monkeypatch):
    """MMA_SKINNY_GW=0 (module switch) is round 4's TN GEMM + column sum: the two backward forms of a 75 -> 75 layer agree to fp32 noise."""
    from mma_amd import dense
    gen = torch.Generator().manual_seed(11)
    x, W, b = torch.randn(9000, 75, generator=gen).to(DEV), (torch.randn(75, 75, generator=gen) / 8).to(DEV), torch.randn(75, generator=gen).to(DEV)
    cot = torch.randn(9000, 75, generator=gen).to(DEV)
    res = []
    for on in (True, False):
        monkeypatch.setattr(dense, "SKINNY_GW", on)
        xd, wd, bd = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        res.append(torch.autograd.grad((dense.linear(xd, wd, bd) * cot).sum(), [xd, wd, bd]))
    for a, c in zip(*res):
        assert (a - c).abs().max().item() <= 2e-5 * c.abs().max().item()
    assert torch.equal(res[0][0], res[1][0]), "dL/dx does not depend on the switch"


def test_shapes_inside_the_old_gates_but_outside_the_kernels_take_the_library_path():
    """Linear(512 -> 80) on >= 4096 rows passed `_skinny_ok` (O <= 80, K <= 512) but its 160 KB of staged weights do
    not fit the LDS beside the wave tiles - MMA_REQUIRE raised where round 2 ran the library GEMM.  The gate now asks
    mma_tower_post_fits; the layer works and matches float64."""
    from mma_amd import dense
    g = torch.Generator().manual_seed(5)
    N, K, O = 4200, 512, 80
    x, W, b = torch.randn(N, K, generator=g), torch.randn(O, K, generator=g) / np.sqrt(K), torch.randn(O, generator=g)
    xd, wd, bd = x.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    assert not dense._skinny_ok(xd, wd)
    y = dense.linear(xd, wd, bd)
    gx, gw, gb = torch.autograd.grad(y.square().sum(), [xd, wd, bd])
    x64, w64, b64 = x.double().requires_grad_(True), W.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = torch.nn.functional.linear(x64, w64, b64)
    r = torch.autograd.grad(y64.square().sum(), [x64, w64, b64])
    assert (y.cpu().double() - y64.detach()).abs().max().item() <= 2e-5 * y64.abs().max().item()
    for got, ref in zip((gx, gw, gb), r):
        assert (got.cpu().double() - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()


def test_mmaconv_with_wide_aggregate_blocks_takes_the_unfactored_post_nn():
    """4 aggregators at F_in = 150 (K*Fw = 608 > 512) with F_out <= 16 and S <= 5: inside the `factored` gate of round 3, outside K13's
    limits.  The layer must run (unfactored post-NN) and agree with the factored form's float64 restatement = the oracle."""
    import mma_amd
    from mma_amd import functional as Fn
    from oracle import gr_oracle as G
    from tools.synth import molecule_batch
    rng = np.random.default_rng(3)
    torch.manual_seed(0)
    T, F = 1, 150
    conv = mma_amd.MMAConv(F, 12, ["sum", "mean", "min", "max"], ["identity", "amplification"], torch.tensor([0, 10, 30, 50, 10]),
                           edge_dim=None, towers=T).to(DEV)
    assert conv.F_out == 12 and not __import__("mma_amd").dense.tower_post_fits(4 * conv.fused_width(), 2)
    ei, N = molecule_batch(rng, 40)
    x = rng.standard_normal((N, F)).astype(np.float32)
    conv.drop_override = Fn.DropoutSpec(0.0)
    xg = torch.from_numpy(x).to(DEV).requires_grad_(True)
    got = conv(xg, torch.from_numpy(ei).to(DEV))
    gx, = torch.autograd.grad(got.square().sum(), [xg])
    from test_gr_gpu import conv_params, to64
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    want = G.conv_forward(x64, torch.from_numpy(ei), None, to64(conv_params(conv)), conv.aggregators, conv.scalers, conv.avg_deg, T, False, None, 0.0)
    g64, = torch.autograd.grad(want.square().sum(), [x64])
    assert (got.cpu().double() - want.detach()).abs().max().item() <= 2e-5 * want.abs().max().item()
    assert (gx.cpu().double() - g64).abs().max().item() <= 5e-5 * g64.abs().max().item()


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


@pytest.mark.parametrize("T,O,S,KF", [(5, 15, 3, 152), (1, 1, 1, 4), (3, 16, 5, 100), (2, 7, 2, 512)])
def test_tower_post_weight_layouts_in_one_launch(T, O, S, KF):
    """mma_tower_post_weights (round 4): both zero-padded layouts of the post-NN weight columns from one launch equal the three torch
    operations they replace (fill, strided copy, transposing copy), padding included (the buffers start as garbage)."""
    from mma_amd import _lib
    from mma_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(T + O + S + KF)
    Wo = torch.randn(T, O, S * KF, generator=g).to(DEV)
    KFp = int(_lib.lib().mma_tower_post_kfp(KF))
    Wb_ref = torch.zeros((T, S * 16, KFp + 16), device=DEV)
    Wb_ref.view(T, S, 16, KFp + 16)[:, :, :O, :KF] = Wo.view(T, O, S, KF).permute(0, 2, 1, 3)
    Wa_ref = Wb_ref[:, :, :KFp].transpose(1, 2).contiguous()
    Wb = torch.full((T, S * 16, KFp + 16), float("nan"), device=DEV)
    Wa = torch.full((T, KFp, S * 16), float("nan"), device=DEV)
    call("mma_tower_post_weights", ptr(Wo), T, O, S, KF, ptr(Wa), ptr(Wb), stream_ptr())
    assert torch.equal(Wb, Wb_ref) and torch.equal(Wa, Wa_ref)


@pytest.mark.parametrize("R,T,S,O,KF", [(4, 5, 3, 15, 152), (1, 1, 1, 1, 4), (300, 2, 5, 16, 100), (1024, 4, 2, 7, 36), (256, 1, 1, 3, 8)])
def test_tower_post_gw_reduce_equals_col_sum_and_permute(R, T, S, O, KF):
    """mma_tower_post_gw_reduce (round 4): the partial tiles summed in mma_col_sum's order, written in the weight layout - the bits of
    the K8 launch + the permuting copy it replaces."""
    from mma_amd import dense
    from mma_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(R + T + KF)
    kfp16 = -(-KF // 16) * 16
    part = torch.randn(R, T * S * 16 * kfp16, generator=g).to(DEV)
    ref = dense.col_sum(part).view(T, S, 16, kfp16)[:, :, :O, :KF].permute(0, 2, 1, 3).reshape(T, O, S * KF)
    got = torch.full((T, O, S * KF), float("nan"), device=DEV)
    call("mma_tower_post_gw_reduce", ptr(part), R, T, S, O, KF, ptr(got), stream_ptr())
    assert torch.equal(got, ref)


@pytest.mark.parametrize("O,K", [(75, 75), (1, 1), (80, 512), (16, 50), (33, 130)])
def test_skinny_linear_weight_layouts_in_one_launch(O, K):
    """mma_skinny_linear_weights (round 4) equals the fill + copy + transposing copy it replaces."""
    from mma_amd import _lib
    from mma_amd._lib import call, ptr, stream_ptr
    W = torch.randn(O, K, generator=torch.Generator().manual_seed(O * K)).to(DEV)
    S, kfp = -(-O // 16), int(_lib.lib().mma_tower_post_kfp(K))
    Wb_ref = torch.zeros((S * 16, kfp + 16), device=DEV)
    Wb_ref[:O, :K] = W
    Wb = torch.full_like(Wb_ref, float("nan")); Wa = torch.full((kfp, S * 16), float("nan"), device=DEV)
    call("mma_skinny_linear_weights", ptr(W), O, K, ptr(Wa), ptr(Wb), stream_ptr())
    assert torch.equal(Wb, Wb_ref) and torch.equal(Wa, Wb_ref[:, :kfp].t().contiguous())


def _post_case(N, T, KF, O, scalers, seed):
    g = torch.Generator().manual_seed(seed)
    S = len(scalers)
    deg = torch.randint(0, 7, (N,), generator=g)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(deg, 0)]).to(torch.int32).to(DEV)
    agg = torch.randn(N, T, KF, generator=g)
    Wo = torch.randn(T, O, S * KF, generator=g) / np.sqrt(S * KF)
    cot = torch.randn(N, T * O, generator=g)
    return deg, rowptr, agg, Wo, cot


def _post_run(agg, Wo, rowptr, scalers, cot):
    from mma_amd import functional as Fn
    ad, wd = agg.to(DEV).requires_grad_(True), Wo.to(DEV).requires_grad_(True)
    y = Fn.tower_post(ad, wd, rowptr, scalers, 1.3, 2.2)
    ga, gw = torch.autograd.grad((y * cot.to(DEV)).sum(), [ad, wd])
    torch.cuda.synchronize()
    return y.detach().cpu(), ga.cpu(), gw.cpu()


@pytest.mark.parametrize("N,T,KF,O,scalers", [(3000, 5, 152, 15, SC[:1] + SC[1:2] + SC[3:4]), (333, 3, 36, 16, SC), (130, 2, 100, 7, SC[1:3])])
def test_k13_on_bf16_pieces_and_on_fp32_agree_with_float64(N, T, KF, O, scalers, monkeypatch):
    """[r4] K13 runs on three bf16 pieces per operand (six piece products, the five small ones summed apart); MMA_POST_EXACT=1 (read per
    call) runs the fp32-MFMA kernel.  Both inside the same bound against float64, and the two forms within 2 ulp-of-the-sum of each other."""
    deg, rowptr, agg, Wo, cot = _post_case(N, T, KF, O, scalers, N + KF)
    S = len(scalers)
    pre = _pre(deg, scalers, 1.3, 2.2)
    out = torch.cat([agg.double() * pre[:, q].view(N, 1, 1) for q in range(S)], -1)
    y64 = torch.einsum("ntc,toc->nto", out, Wo.double()).reshape(N, T * O)
    mag = (out.abs().unsqueeze(2) * Wo.double().abs().unsqueeze(0)).sum(-1).reshape(N, T * O)
    ys = {}
    for exact in ("0", "1"):
        monkeypatch.setenv("MMA_POST_EXACT", exact)
        ys[exact] = _post_run(agg, Wo, rowptr, scalers, cot)[0].double()
        assert ((ys[exact] - y64).abs() <= 3e-7 * mag + 1e-9).all(), exact
    assert not torch.equal(ys["0"], ys["1"]), "the switch selected the same kernel twice"
    assert ((ys["0"] - ys["1"]).abs() <= 3e-7 * mag + 1e-9).all()


def test_tower_post_results_do_not_depend_on_the_workgroup_order(monkeypatch):
    """[r4] post_block(): the tower is the fastest-varying workgroup coordinate (whole agg rows are read / written by the workgroups that run
    together); MMA_POST_TOWER_MAJOR=1 is round 3's order.  The arithmetic of a (tower, node block) does not know which workgroup id it has:
    K13, K14 and K15 bit-equal in both orders, on the bf16-piece and the fp32 forward."""
    N, T, KF, O, scalers = 2500, 5, 152, 15, SC[:1] + SC[1:2] + SC[3:4]
    deg, rowptr, agg, Wo, cot = _post_case(N, T, KF, O, scalers, 7)
    for exact in ("0", "1"):
        monkeypatch.setenv("MMA_POST_EXACT", exact)
        monkeypatch.setenv("MMA_POST_TOWER_MAJOR", "0")
        a = _post_run(agg, Wo, rowptr, scalers, cot)
        monkeypatch.setenv("MMA_POST_TOWER_MAJOR", "1")
        b = _post_run(agg, Wo, rowptr, scalers, cot)
        for x, y, what in zip(a, b, ("y", "gagg", "gWo")):
            assert torch.equal(x, y), (what, exact)


@pytest.mark.parametrize("N,fin,pitch", [(5000, 75, 75), (4099, 50, 64), (33000, 127, 127), (1, 3, 3), (2000, 76, 380)])
def test_pad_rows_equals_pad_and_fill(N, fin, pitch):
    """[r4] mma_pad_rows: [x | 1 | 0 ...] (and [W | b | 0 ...] with zero rows below) of the zero-padded tall Linear in one launch = torch's
    pad + a strided fill, bit for bit (row-strided x included: a column block of a wider buffer)."""
    from mma_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(N + fin)
    wide = torch.randn(N, pitch, generator=g).to(DEV)
    x = wide[:, :fin]
    want = torch.nn.functional.pad(x, (0, 128 - fin))
    want[:, fin] = 1.0
    got = torch.full((N, 128), float("nan"), device=DEV)
    call("mma_pad_rows", ptr(x), x.stride(0), N, fin, None, None, ptr(got), 128, 128, N, stream_ptr())
    assert torch.equal(got, want)
    # [r5] with a row index: out row r = x row idx[r] (the permutation of graph regression's edge rows rides on the pad)
    idx = torch.randperm(N, generator=g).to(DEV)
    got_p = torch.full((N, 128), float("nan"), device=DEV)
    call("mma_pad_rows", ptr(x), x.stride(0), N, fin, None, ptr(idx.to(torch.int32)), ptr(got_p), 128, 128, N, stream_ptr())
    assert torch.equal(got_p, want[idx])
    # a given column and zero rows below (the weight operand)
    col = torch.randn(N, generator=g).to(DEV)
    rows_out = -(-N // 128) * 128 + 128
    want2 = torch.zeros((rows_out, 128), device=DEV)
    want2[:N, :fin] = x
    want2[:N, fin] = col
    got2 = torch.full((rows_out, 128), float("nan"), device=DEV)
    call("mma_pad_rows", ptr(x), x.stride(0), N, fin, ptr(col), None, ptr(got2), 128, 128, rows_out, stream_ptr())
    assert torch.equal(got2, want2)


def test_linear_tall_with_and_without_the_fused_pad(monkeypatch):
    from mma_amd import dense
    g = torch.Generator().manual_seed(5)
    x = torch.randn(40000, 75, generator=g).to(DEV)
    W = (torch.randn(760, 75, generator=g) / 9).to(DEV)
    b = torch.randn(760, generator=g).to(DEV)
    assert dense.linear_x3_ok(x, W)
    for bias in (True, False):
        outs = []
        for flag in (True, False):
            monkeypatch.setattr(dense, "FUSED_PAD", flag)
            xr, wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
            br = b.clone().requires_grad_(True) if bias else None
            y = dense.linear_tall(xr, wr, br)
            gr = torch.autograd.grad((y * y).sum(), [xr, wr] + ([br] if bias else []))
            outs.append((y.detach(),) + gr)
        for a, c in zip(*outs):
            assert torch.equal(a, c), bias
