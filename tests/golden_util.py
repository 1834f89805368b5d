"""Load tests/golden/*.npz fixtures (made by tests/golden/gen_golden.py from the reference) and
regenerate their seeded inputs."""
import os

import numpy as np
import torch

from golden.inputs import make_inputs, keep_mask, sha

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["toy6_h8", "ringhub40_h20", "cora_h64", "cora_h16", "pubmed_h16", "citeseer_h128"]


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        z = self.z
        self.name = name
        self.N, self.H, self.C, self.seed = int(z["N"]), int(z["H"]), int(z["C"]), int(z["seed"])
        self.rowptr = z["rowptr"].astype(np.int64)
        self.col = z["col"].astype(np.int64)
        self.E = len(self.col)
        self.rows = z["rows"]
        x, masks, weight, bias, cot = make_inputs(self.seed, self.N, self.H, self.C)
        # the inputs are regenerated from seeds: make sure they are the ones the reference saw
        assert sha(x) == str(z["sha_x"]) and sha(weight) == str(z["sha_w"])
        assert sha(masks["sum"]) == str(z["sha_mask_sum"]) and sha(cot) == str(z["sha_cot"])
        self.x, self.masks, self.weight, self.bias, self.cot = x, masks, weight, bias, cot
        self.keys = [str(k) for k in z["keys"]]
        self.add_all = [self.col[self.rowptr[i]:self.rowptr[i + 1]] for i in range(self.N)]

    def keep(self, agg, p):
        return keep_mask(self.seed, agg, self.E, self.H, p) if p > 0 else None

    def set_keys(self):
        return [k for k in self.keys if k.startswith("set/")]

    def single_keys(self):
        return [k for k in self.keys if k.startswith("single/")]

    @staticmethod
    def parse(key):
        kind, act, ptag, aggs = key.split("/")
        return kind, act, int(ptag[1:]) / 100.0, aggs.split(",")

    def torch_inputs(self, device="cpu"):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        return t(self.x), {k: t(v) for k, v in self.masks.items()}, t(self.weight), t(self.bias), t(self.cot)


# ---- parity bars -----------------------------------------------------------------------------------------
# STRICT bar (SURVEY Appendix A / north_star): |got - want| <= 1e-5 + 1e-5*|want| element-wise.  Every comparison
# counts and REPORTS the elements outside it (REPORT list below, dumped to gpurun_out/parity_strict_report_*.jsonl by the
# conftest session hook), whatever bar it asserts.
# Long SIGNED sums (layer output = spmm of mm, gradients) cannot meet 1e-5 ABSOLUTE under any fp32 re-association once an
# element cancels to ~0 out of terms of size ~1e3 ("Parity definition") - the reference's own fp32 result is
# that far from the exact value of its own formula.  Their absolute slack therefore FOLLOWS THE DATA:
#   * `truth` given = the same quantity evaluated by the CPU oracle in float64 (the exact value of the reference's
#     formula on these inputs).  |want - truth| is the fp32 evaluation noise of the reference/oracle itself, element by
#     element; rows share one summation structure, so a row's noise level is its max.  Bar:
#         |got - want| <= 1e-5 + 1e-5*|want| + NOISE_C * rowmax|want - truth|        (NOISE_C = 6; round 3: 8, round 2: 16)
#     i.e. the HIP path may differ from the reference by a few times what the reference differs from exact arithmetic
#     (measured on MI355X, round 2: the largest multiple any golden / small-graph comparison needs is 2.95; the fp32 CPU
#     oracle itself needs 0.69 against the goldens.  Round 2 also compared the mask-weight gradients of the WHOLE C5-shape graph
#     with a torch-CPU oracle and saw 4.7-6.1 on single elements of the mean-kind masks, varying with the box because that
#     oracle's own summation order does: that comparison no longer exists - since round 3 the full-size tests compare sampled
#     target rows and the gradients of the induced sub-problem, whose needs are 0.72-0.97 (profiles/r04_parity_strict_report.md),
#     so nothing asserted today sits near 6.  Round 3: 16 -> 8, i.e. 1.3x the worst element seen (round-2 VERDICT:
#     16 was 2.6x slack); an element that then fails on some box is to be REPORTED, not absorbed by a wider constant.
#     Round 4: 8 -> 6 (round-3 VERDICT item 4): the largest need is the 3-epoch training trajectory's, 4.3 - explained in
#     tests/test_train_golden.py::test_trajectory_noise_is_single_step_rounding_amplified_by_adam; the session tail prints the
#     run's largest multiple.)
#   * no truth available:   atol = max(1e-5, SIGNED_SUM_ATOL * max|want|), SIGNED_SUM_ATOL = 1e-6
#     (round 1 used 1e-5 * max|want| everywhere: 10x looser than this fallback, ~1000x looser than the truth-based bar).
SIGNED_SUM_ATOL = 1e-6
NOISE_C = 6.0
REPORT = []          # one dict per comparison: what, n, strict_outside, max_err, max_ref, bar, slack_max


def _rows_of(a, rows, n_want):
    a = np.asarray(a)
    return a[np.asarray(rows)] if (rows is not None and a.shape[0] != n_want) else a


def check_close(got, want_rows, rows, stats, rtol=1e-5, atol=1e-5, what="", signed_sum=False, truth=None):
    """Assert parity of `got` with the stored/oracle rows and record the strict-bar failure count (see above)."""
    got = got.detach().cpu()
    g = got[torch.as_tensor(rows)].numpy() if rows is not None else got.numpy()
    want_rows = np.asarray(want_rows)
    assert g.shape == want_rows.shape, "%s: shape %s vs %s" % (what, g.shape, want_rows.shape)
    finite = np.isfinite(want_rows)
    assert np.array_equal(np.isnan(g), np.isnan(want_rows)) and np.array_equal(np.isinf(g), np.isinf(want_rows)) and \
        np.array_equal(g[np.isinf(want_rows)], want_rows[np.isinf(want_rows)]), what + ": NaN/inf pattern differs"
    err = np.where(finite, np.abs(np.where(finite, g, 0).astype(np.float64) - np.where(finite, want_rows, 0).astype(np.float64)), 0.0)
    aw = np.where(finite, np.abs(want_rows), 0.0).astype(np.float64)
    strict_tol = 1e-5 + 1e-5 * aw
    n_strict = int((err > strict_tol).sum())
    bar, slack, need_c = "strict", 0.0, None
    tol = atol + rtol * aw
    if signed_sum:
        if truth is not None:
            t = _rows_of(truth, rows, want_rows.shape[0]).astype(np.float64)
            assert t.shape == want_rows.shape, "%s: truth shape %s vs %s" % (what, t.shape, want_rows.shape)
            noise = np.where(finite & np.isfinite(t), np.abs(np.where(finite, want_rows, 0).astype(np.float64) - np.where(np.isfinite(t), t, 0)), 0.0)
            row_noise = noise.reshape(noise.shape[0], -1).max(1).reshape((-1,) + (1,) * (noise.ndim - 1)) if noise.ndim > 1 \
                else np.full(noise.shape, noise.max() if noise.size else 0.0)
            tol = atol + rtol * aw + NOISE_C * row_noise
            bar, slack = "1e-5 + 1e-5|ref| + %g*rowmax|ref - fp64 oracle|" % NOISE_C, float((NOISE_C * row_noise).max()) if noise.size else 0.0
            over = np.maximum(err - strict_tol, 0.0) / np.maximum(np.broadcast_to(row_noise, err.shape), 1e-30)
            need_c = float(over[np.broadcast_to(row_noise, err.shape) > 0].max()) if (np.broadcast_to(row_noise, err.shape) > 0).any() else 0.0
        else:
            a = max(atol, SIGNED_SUM_ATOL * float(aw.max()) if aw.size else atol)
            tol = a + rtol * aw
            bar, slack = "max(1e-5, 1e-6*max|ref|)", a
    REPORT.append({"what": what, "n": int(err.size), "strict_outside": n_strict,
                   "max_err": float(err.max()) if err.size else 0.0, "max_ref": float(aw.max()) if aw.size else 0.0,
                   "bar": bar, "abs_slack_max": slack, "noise_multiple_needed": need_c})
    bad = err > tol
    if bad.any():
        idx = np.argwhere(bad)[:5]
        detail = "; ".join("%s got %.6g want %.6g tol %.3g" % (tuple(i), g[tuple(i)], want_rows[tuple(i)], np.broadcast_to(tol, err.shape)[tuple(i)])
                           for i in idx)
        raise AssertionError("%s: %d/%d outside the [%s] bar (max err %.3g), %d outside the strict bar: %s" % (
            what, int(bad.sum()), err.size, bar, err.max(), n_strict, detail))
    if stats is not None:
        gd = got.double()
        assert abs(gd.sum().item() - stats[0]) <= 1e-5 * stats[1] + 1e-5, what + ": checksum(sum)"
        assert abs(gd.abs().sum().item() - stats[1]) <= 1e-5 * stats[1] + 1e-5, what + ": checksum(abs)"


# ---- float64 evaluations of the oracle (the "truth" of the bars above) ------------------------------------------
_TRUTH = {}


def nc_truth(gold, key):
    """MMA.forward + gradients of one golden set key by the CPU oracle in FLOAT64 (oracle/nc_oracle.py, pinned to the
    reference by test_oracle_golden.py): {"out","gx","gweight","gbias","gmask/<a>","m/<a>"} -> float64 arrays (all rows)."""
    ck = (gold.name, key)
    if ck not in _TRUTH:
        from oracle import nc_oracle as O
        _, act, p, aggs = gold.parse(key)
        x, masks, weight, bias, cot = [t.double() if torch.is_tensor(t) else {k: v.double() for k, v in t.items()}
                                       for t in gold.torch_inputs()]
        x.requires_grad_(True); weight.requires_grad_(True); bias.requires_grad_(True)
        Ws = {a: masks[a].clone().requires_grad_(True) for a in aggs}
        keeps = {a: gold.keep(a, p) for a in aggs} if p > 0 else None
        out, ms = O.mma_forward(aggs, x, Ws, weight, bias, gold.rowptr, gold.col, gold.z["adj_row"], gold.z["adj_col"],
                                gold.z["adj_val"], act, p, keeps, return_m=True)
        grads = torch.autograd.grad((out * cot).sum(), [x, weight, bias] + [Ws[a] for a in aggs])
        t = {"out": out.detach().numpy(), "gx": grads[0].numpy(), "gweight": grads[1].numpy(), "gbias": grads[2].numpy()}
        for a, m, g in zip(aggs, ms, grads[3:]):
            t["m/" + a] = m.detach().numpy()
            t["gmask/" + a] = g.numpy()
        _TRUTH[ck] = t
    return _TRUTH[ck]
