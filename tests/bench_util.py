"""Reading bench.py's output the way the driver does: stdout holds exactly ONE line - the compact metric record, strict JSON, < 4096
bytes; the verbose record ({"detail": ...}) is on stderr and in $MMA_BENCH_DETAIL (round-4 VERDICT item 1)."""
import json


def _no_constants(name):
    raise ValueError("non-finite constant %s in the bench line" % name)


def strict_loads(s):
    return json.loads(s, parse_constant=_no_constants)


def parse_bench(r, need_roofline=True, need_cpu=False):
    """r: CompletedProcess(capture_output=True, text=True).  Returns (compact line, detail record)."""
    out = [l for l in r.stdout.splitlines() if l.strip()]
    jl = [l for l in out if l.startswith("{")]
    assert len(jl) == 1, r.stdout[-2000:]                       # exactly ONE JSON line on stdout
    last = out[-1]
    assert last == jl[0], "the metric line must be the LAST stdout line"
    assert len(last.encode()) < 4096, len(last.encode())
    d = strict_loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    if need_roofline:
        ro = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in ro, k
        assert 0 < ro["frac"] == ro["frac"] and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    if need_cpu:
        c = d["cpu_baseline"]
        assert c["value"] > 0 and c["kind"] in ("port", "reference") and c["cores"] >= 1 and len(c["sample"]) <= 200
    det = [l for l in r.stderr.splitlines() if l.startswith('{"detail"')]
    assert len(det) == 1, r.stderr[-2000:]
    return d, strict_loads(det[0])["detail"]
