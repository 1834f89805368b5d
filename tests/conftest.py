import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) where no GPU is visible, e.g. a plain `pytest tests/` here.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Dump the strict-bar (1e-5 + 1e-5|ref|) failure count of every parity comparison of this session."""
    try:
        import json
        import golden_util
        if not golden_util.REPORT:
            return
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        import torch
        tag = "gpu" if torch.cuda.is_available() else "cpu"
        with open(os.path.join(out_dir, "parity_strict_report_%s.jsonl" % tag), "w") as f:
            for r in golden_util.REPORT:
                f.write(json.dumps(r) + "\n")
        n = len(golden_util.REPORT)
        bad = [r for r in golden_util.REPORT if r["strict_outside"]]
        print("\n[parity] %d comparisons, %d with elements outside the strict 1e-5 bar (%d elements of %d)" % (
            n, len(bad), sum(r["strict_outside"] for r in bad), sum(r["n"] for r in golden_util.REPORT)))
        worst = max((r for r in golden_util.REPORT if r.get("noise_multiple_needed") is not None), key=lambda r: r["noise_multiple_needed"],
                    default=None)
        if worst is not None:       # round-3 VERDICT item 4: the figure belongs in the test tail; the bar (golden_util.NOISE_C) fails above it
            print("[parity] largest multiple of the reference's own fp32 noise any comparison needed: %.2f (%s); asserted bar: %g" % (
                worst["noise_multiple_needed"], worst["what"], golden_util.NOISE_C))
    except Exception as e:      # reporting must never turn a green run red
        print("[parity] report skipped:", e)
