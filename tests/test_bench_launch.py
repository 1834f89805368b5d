"""bench.py's launch paths, without a GPU: `python bench.py --gpus N` with no launcher on the command line must start N
ranks itself (before touching the GPU), and the documented launcher form must keep working."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly ONE JSON line (rank 0's), relayed by the parent
    return json.loads(lines[0])


def test_self_launch_spawns_the_ranks():
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--rendezvous-only"])
    assert d == {"rendezvous": "ok", "world": 2, "rank_sum": 1.0}


def test_launcher_form_still_works():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "3", "--rendezvous-only"])
    assert d == {"rendezvous": "ok", "world": 3, "rank_sum": 3.0}


def test_self_launch_propagates_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rendezvous-only", "--workload", "bogus"], capture_output=True,
                       text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0


def test_metric_line_is_compact_and_strict_json(capsys, tmp_path, monkeypatch):
    """Round-4 VERDICT item 1: whatever the verbose record holds (the 25 KB of round 4, NaN / Infinity from a failed secondary config),
    stdout gets ONE strict-JSON line < 4096 bytes with the contract's keys; the rest goes to stderr and $MMA_BENCH_DETAIL."""
    import bench
    from bench_util import strict_loads
    monkeypatch.setenv("MMA_BENCH_DETAIL", str(tmp_path / "d" / "detail.json"))
    kernels = {"kernel_%d" % i: {"launches": 10, "avg_ms": 0.1 * i, "algorithmic_bytes": 1e9, "achieved_GBs": float("nan"), "frac": 0.5} for i in range(40)}
    extra = {"C%d" % i: {"config": "x" * 300, "ms_per_step_eager": 1.0 + i, "ms_per_step_hipgraph": float("inf"), "kernels": kernels} for i in range(12)}
    extra["bad"] = {"error": "RuntimeError: " + "y" * 500}
    roof = {"bound": "hbm", "kernel": "nc_fused_bwd", "achieved": 7000.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.875, "traffic": 3.86e10,
            "traffic_source": "z" * 400, "algorithmic_bytes": 41346253264, "frac_rocprof": None, "duration": "w" * 300}
    cpu = {"value": 115780.2, "unit": "edges/s", "cores": 16, "kind": "port", "form": "vectorised", "sample": "s" * 500,
           "loop": [{"config": "C1", "value": 7000.0, "unit": "edges/s", "cores": 16, "kind": "port", "form": "faithful per-node loop", "sample": "q" * 300}] * 2}
    line = {"metric": "aggregated edges/sec (fwd+bwd) MultiMaskConv", "value": 7.5e8, "unit": "edges/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 14.45, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4: " + "c" * 250, "nodes": 1048576, "edges": 10900000, "hidden": 128, "K": 4, "parallelism": "single GPU"},
            "roofline": bench._compact_roofline(roof), "cpu_baseline": bench._compact_cpu(cpu),
            "kernels_ms": {k: v["avg_ms"] for k, v in kernels.items()}, "extra_summary": bench._extra_summary(extra)}
    bench.emit(line, {"roofline": roof, "kernels": kernels, "cpu_baseline": cpu, "extra": extra})
    cap = capsys.readouterr()
    out = [l for l in cap.out.splitlines() if l.strip()]
    assert len(out) == 1 and len(out[0].encode()) < 4096
    d = strict_loads(out[0])
    assert d["roofline"]["frac"] == 0.875 and d["cpu_baseline"]["value"] > 0 and len(d["cpu_baseline"]["sample"]) <= 200
    assert len(d["cpu_baseline"]["loop"]) == 2 and d["value"] == 7.5e8
    det = strict_loads([l for l in cap.err.splitlines() if l.startswith('{"detail"')][0])["detail"]
    assert det["extra"]["C3"]["ms_per_step_hipgraph"] is None and len(det["kernels"]) == 40            # inf -> null, nothing dropped
    assert strict_loads(open(tmp_path / "d" / "detail.json").read())["detail"]["metric"] == d["metric"]
