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
