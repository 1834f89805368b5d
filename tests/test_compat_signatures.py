"""The drop-in story of INTEGRATION.md 1: prepend mma_amd/compat to sys.path and the reference's scripts import OUR
modules flat (`from layers import MMA`, `from mma_conv import MMAConv`, ...).  This test does exactly that and compares the
public call surface - class names, method names, parameter names, order and defaults - with
tests/golden/reference_signatures.json, which tests/golden/gen_signatures.py recorded from the reference's source text
(layers.py:57-61,853; mma_conv.py:47-51,121-122,159-160; mask_aggr.py:13-23; scalers.py:10-64; models.py:10-16; utils.py)."""
import importlib
import inspect
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "mma_amd", "compat")
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))

# Deliberate, documented deviations from the reference surface:
EXTRA_OK = {("layers", "MMA", "__init__"): {"chunk", "strict_reference", "scalers", "compound_scalers", "avg_d"},      # trailing keyword extensions
            ("mma_conv", "MMAConv", "aggregate"): {"_graph"},                                   # private plan hand-over
            ("utils", "load_data"): {"data_dir"}}                                                # the reference hard-codes "data/"
HELPERS_NOT_NEEDED = {("utils", "sample_mask"), ("utils", "normalize"), ("utils", "sparse_mx_to_torch_sparse_tensor")}


@pytest.fixture(scope="module")
def flat():
    saved_path, saved_mods = list(sys.path), {k: sys.modules.get(k) for k in FIX}
    sys.path.insert(0, COMPAT)
    for k in FIX:
        sys.modules.pop(k, None)
    mods = {k: importlib.import_module(k) for k in FIX}
    for k, m in mods.items():
        assert os.path.dirname(os.path.abspath(m.__file__)) == COMPAT, "%s resolved to %s, not to the shim" % (k, m.__file__)
    yield mods
    sys.path[:] = saved_path
    for k, m in saved_mods.items():
        if m is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = m


def _params(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        name = {p.VAR_POSITIONAL: "*", p.VAR_KEYWORD: "**"}.get(p.kind, "") + p.name
        out.append((name, None if p.default is p.empty else p.default))
    return out


def _same_default(ours, ref_src):
    if ref_src is None:
        return ours is None
    try:
        return ours == eval(ref_src, {})          # literals only: None, True, 1, 'x'
    except Exception:
        return True                               # non-literal default expression: name match is what is checked


def _check(key, ours, ref):
    names_ref = [r["name"] for r in ref]
    names_ours = [n for n, _ in ours]
    extra = EXTRA_OK.get(key, set())
    core = [n for n in names_ours if n not in extra]
    assert core == names_ref, "%s: parameters %s != reference %s" % (key, core, names_ref)
    d = dict(ours)
    for r in ref:
        if not r["name"].startswith("*"):
            assert _same_default(d[r["name"]], r["default"]), "%s: default of %s is %r, reference has %s" % (
                key, r["name"], d[r["name"]], r["default"])


def test_flat_imports_resolve_to_the_shims_and_match_the_reference_surface(flat):
    checked = 0
    for mod, spec in FIX.items():
        m = flat[mod]
        for fname, ref in spec["functions"].items():
            if (mod, fname) in HELPERS_NOT_NEEDED and not hasattr(m, fname):
                continue
            assert hasattr(m, fname), "%s.%s missing" % (mod, fname)
            _check((mod, fname), _params(getattr(m, fname)), ref)
            checked += 1
        for cname, cspec in spec["classes"].items():
            cls = getattr(m, cname)
            for meth, ref in cspec["methods"].items():
                if meth == "__repr__":
                    continue
                assert hasattr(cls, meth), "%s.%s.%s missing" % (mod, cname, meth)
                _check((mod, cname, meth), _params(getattr(cls, meth)), ref)
                checked += 1
    assert checked >= 40


def test_extension_parameters_are_optional(flat):
    for key, extra in EXTRA_OK.items():
        obj = flat[key[0]]
        for part in key[1:]:
            obj = getattr(obj, part)
        sig = inspect.signature(obj)
        for n in extra:
            if n in sig.parameters:
                assert sig.parameters[n].default is not inspect.Parameter.empty, (key, n)


def test_public_scalers_evaluate_like_the_reference_formulas(flat):
    """scalers.py:22-62 on real neighbour lists (not the degenerate call of layers.py:856): log(d+1)/avg and its inverse."""
    import numpy as np
    import torch
    sc = flat["scalers"]
    add_all = [np.arange(d) for d in (1, 3, 7, 2)]
    x = torch.arange(8, dtype=torch.float32).view(4, 2) + 1
    deg = torch.tensor([1, 3, 7, 2])
    lg = torch.log(deg + 1)
    avg = lg.mean()
    assert torch.equal(sc.scale_identity(x, add_all, 1), x)
    assert torch.allclose(sc.scale_amplification(x, add_all, 1), x * (lg / avg).unsqueeze(-1), rtol=1e-6)
    assert torch.allclose(sc.scale_attenuation(x, add_all, 1), x * (avg / lg).unsqueeze(-1), rtol=1e-6)
    x2 = torch.cat([x, x], 0)                    # K = 2 stacked (scalers.py:33-40 tiles the factor K times)
    assert torch.allclose(sc.scale_amplification(x2, add_all, 2), x2 * torch.cat([lg / avg] * 2).unsqueeze(-1), rtol=1e-6)
    assert set(sc.SCALERS) == {"identity", "amplification", "attenuation"}
    assert torch.allclose(sc.avg_d_log(deg), avg) and torch.isfinite(sc.avg_d_exp(deg.float()))
