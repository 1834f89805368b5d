"""N>1 path: world_size-2/3 gloo runs.  CPU: plan + all-to-all plumbing + sharding algebra (oracle as compute).
GPU: the real ShardedMMA (HIP kernels) on two ranks sharing cuda:0, against the single-GPU layer."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(mode, world, *extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "sharded_worker.py"), mode, *extra]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and ("SHARDED_%s_OK" % mode.upper()) in r.stdout, r.stdout[-6000:] + r.stderr[-1500:]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_sharded_cpu_gloo(world):
    _launch("cpu", world)


@pytest.mark.parametrize("world", [2, 3])
def test_bucketed_gradient_allreduce_gloo(world):
    """the one exchange of the data-parallel graph-regression replicas (and of the sharded NC layer's parameters)."""
    _launch("grads", world)


@pytest.mark.gpu
def test_gr_replicas_bench_rehearsal_two_ranks_one_device():
    """`bench.py --workload c2l --gpus 2` (self-launched, gloo, both ranks on cuda:0): replicas with their own molecule batches,
    gradients averaged per step; ONE JSON line, weak scaling, edges of both ranks counted."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--workload", "c2l", "--gpus", "2", "--backend", "gloo", "--molecules", "300",
                        "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    from bench_util import parse_bench
    d, _ = parse_bench(r)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "replicas x2" in d["config"]["parallelism"]
    assert 2 * 300 * 30 < d["config"]["edges"] < 2 * 300 * 60          # both ranks' molecules (about 43 directed edges each)


@pytest.mark.gpu
def test_sharded_bench_line_explains_itself_two_ranks_one_device():
    """Round-3 VERDICT item 3: the N > 1 line of `bench.py` carries one record per rank (device identity, own / halo rows, edges, bytes
    sent and received per step, the rank's own ms per step, exchange time from events on a side stream in a separate instrumented pass, the compute stream's halo
    waits) in the DETAIL record (stderr), rank-max AND rank-mean step time, and the result of `--verify` (the sharded layer against the unsharded one on a 2^14-node
    R-MAT, run before anything is timed - the default for N > 1).  Rehearsed here with gloo, both ranks on cuda:0."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--scale", "15", "--edges", "200000", "--steps", "3",
                        "--warmup", "1", "--cpu-sample", "0", "--no-extra"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    from bench_util import parse_bench
    d, det = parse_bench(r)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["verify"] == {"ok": True}
    ranks = det["ranks"]
    assert [q["rank"] for q in ranks] == [0, 1]
    need = {"device", "own_rows", "halo_rows", "rows_sent", "local_edges", "bytes_sent", "bytes_received", "ms_per_step", "exchange_ms_instrumented",
            "halo_wait_ms", "exchanges_per_step", "nc_fused_fwd_ms", "nc_fused_bwd_ms", "gemm_ms"}
    for q in ranks:
        assert need <= set(q), need - set(q)
        assert q["device"]["name"] and q["own_rows"] > 0 and q["ms_per_step"] > 0 and q["exchange_ms_instrumented"] > 0 and q["exchanges_per_step"] == 4
    assert sum(q["own_rows"] for q in ranks) == d["config"]["nodes"] and sum(q["local_edges"] for q in ranks) == d["config"]["edges"]
    # two ranks: what one sends the other receives, row for row - forward x rows (H floats) and tail rows (C floats), and both back
    H, C = d["config"]["hidden"], d["config"]["nclass"]
    assert ranks[0]["bytes_sent"] == ranks[1]["bytes_received"] == (ranks[0]["rows_sent"] + ranks[0]["halo_rows"]) * (H + C) * 4
    assert ranks[0]["halo_rows"] == ranks[1]["rows_sent"] and ranks[1]["halo_rows"] == ranks[0]["rows_sent"]
    assert d["ms_per_step_rank_max"] >= d["ms_per_step_rank_mean"] > 0 and d["ms_per_step"] >= 0.99 * d["ms_per_step_rank_max"]
    assert d["distinct_devices"] == 1 and "NOT a multi-GPU measurement" in d["multi_gpu_note"]          # the rehearsal says what it is
    v = det["verify"]
    assert v["ok"] and v["checks"]["out"]["rows_outside"] == 0 and v["checks"]["dL/dx"]["ok"] and all(c["ok"] for c in v["checks"].values())


@pytest.mark.gpu
def test_sharded_on_the_rccl_backend_single_rank():
    """RCCL refuses two ranks on one device, so the multi-rank tests above run over gloo; this one takes the REAL backend
    (`nccl` = RCCL: plan exchange and all_to_all_single on device tensors, async work handles) at world size 1 and checks the
    sharded layer against the plain one on the same graphs."""
    _launch("nccl", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_on_rccl_one_gpu_per_rank(world):
    """the overlapped exchanges (async all_to_all_single handles on RCCL's stream beside the compute stream, send / receive
    buffer lifetimes, uneven and zero counts) with world > 1 on the REAL backend - hub, directed (a rank without halo), sweep
    (messy / tiny / empty graphs, true-degree scalers, K=8 / S=5, tall shards) and, at world 3, an empty shard.  Needs one GPU per
    rank: skipped on the single-GPU test box (RCCL refuses two ranks on one device), runs wherever the driver has a multi-GPU node."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs (this box has %d)" % (world, torch.cuda.device_count()))
    _launch("nccl_multi", world)


@pytest.mark.gpu
def test_sharded_gpu_two_ranks_one_device():
    _launch("gpu", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,world", [("directed", 2), ("empty", 3), ("sweep", 2), ("sweep", 3)])
def test_sharded_gpu_rank_without_halo_still_joins_the_exchange(variant, world):
    """a rank with no halo rows of its own (directed graph) or an empty shard must still take part in the
    reverse all-to-all and receive its peers' dL/dx contributions."""
    _launch("gpu", world, variant)


def test_partition_bounds_balance():
    import numpy as np
    from mma_amd.sharded import partition_bounds
    rng = np.random.default_rng(0)
    deg = rng.poisson(6, 10000); deg[17] = 5000
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    for world in (1, 2, 4, 8):
        b = partition_bounds(rowptr, world)
        assert b[0] == 0 and b[-1] == 10000 and len(b) == world + 1 and (np.diff(b) >= 0).all()
        e = rowptr[b[1:]] - rowptr[b[:-1]]
        assert e.max() <= rowptr[-1] / world + deg.max()


def test_halo_volume_of_the_partition_is_bounded_and_relabel_does_not_pay():
    """tools/halo_report.py's finding at a CPU-sized scale (R-MAT 2^16 nodes): under the generator's locality-free order the
    edge-balanced cut keeps rows AND edges balanced and the halo within known factors of the own rows; a degree order
    concentrates rows on one rank (what DESIGN.md 5 records as the reason not to relabel this graph)."""
    import numpy as np
    from mma_amd.sharded import halo_report, partition_bounds
    from tools.synth import rmat_graph
    from tools.halo_report import relabel
    rowptr, col = rmat_graph(16, 320_000, seed=42)
    N, E = len(rowptr) - 1, len(col)
    bound = {2: 0.85, 4: 1.7, 8: 2.6}                 # halo rows / own rows (C4 at scale 20 measures 0.73 / 1.48 / 2.29)
    for w in (2, 4, 8):
        h = np.array(halo_report(rowptr, col, w))
        assert h[:, 0].max() <= 1.1 * N / w and h[:, 2].max() <= 1.02 * E / w + np.diff(rowptr).max()
        assert (h[:, 1] / h[:, 0]).max() <= bound[w], (w, (h[:, 1] / h[:, 0]).max())
    rp, c = relabel(rowptr, col, np.argsort(-np.diff(rowptr), kind="stable"))
    h_rand, h_deg = np.array(halo_report(rowptr, col, 8)), np.array(halo_report(rp, c, 8))
    assert (h_deg[:, 0] + h_deg[:, 1]).max() > 1.5 * (h_rand[:, 0] + h_rand[:, 1]).max()      # rows per rank: worse, not better
    # the row-weighted cut is monotone and covers all nodes
    b = partition_bounds(rp, 8, row_cost=E / N)
    assert b[0] == 0 and b[-1] == N and (np.diff(b) >= 0).all()
