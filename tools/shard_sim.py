"""Single-GPU rehearsal of ONE rank of an N-rank run: the halo exchange is replaced by a local stand-in (rows filled with
random data, no communication), everything else is the real sharded layer WITH ITS REAL HALO ROWS.  Gives the per-rank compute
time and peak memory - a prediction of the compute side, not a multi-GPU measurement.

    python tools/shard_sim.py                      # C4 (BASELINE configs[3]) at 1 / 2 / 4 / 8 ranks, ranks 0, mid, last
    python tools/shard_sim.py --c5 [--ranks 0,4]   # C5 (configs[4]: scale 23, 128 M directed edges, feat 256, K=8, S=5 true-degree
                                                   # scalers): rank r of 8 - the whole graph does not fit one GPU, one rank does"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mma_amd
from mma_amd import sharded as S, functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--c5", action="store_true")
ap.add_argument("--ranks", default=None, help="comma-separated ranks to rehearse (default: 0, mid, last)")
ap.add_argument("--worlds", default=None, help="comma-separated world sizes (default: 1,2,4,8; --c5: 8)")
ap.add_argument("--steps", type=int, default=None)
args = ap.parse_args()
dev = torch.device('cuda:0')
t_gen = time.perf_counter()
if args.c5:
    P5 = bench.C5_PRESET
    rowptr, col = bench.rmat_graph(P5["scale"], P5["edges"], seed=42)
    H, C, names = P5["hidden"], P5["nclass"], P5["aggregators"].split(",")
    EXTRA = dict(strict_reference=False, scalers=["identity", "amplification", "attenuation", "linear", "inverse_linear"], compound_scalers=True)
else:
    rowptr, col = bench.rmat_graph(20, 5_000_000, seed=42)
    H, C, names = 128, 16, ["sum", "mean", "max", "min"]
    EXTRA = {}
N, E = len(rowptr) - 1, int(rowptr[-1])
print("graph: %d nodes / %d directed edges, H=%d, K=%d (%s), generated in %.1f s" % (N, E, H, len(names), ",".join(names), time.perf_counter() - t_gen), flush=True)
if EXTRA:
    d_ = np.maximum(np.diff(rowptr), 1).astype(np.float32)
    EXTRA["avg_d"] = {"log": float(np.log(d_ + 1).mean()), "lin": float(d_.mean())}

class FakeHandle:
    def __init__(self, r): self.r = r
    def wait(self): return self.r
def fake_start(send, send_counts, recv_counts, group=None, out=None):
    n = int(sum(recv_counts))
    r = out if out is not None else torch.empty((n,) + tuple(send.shape[1:]), device=send.device)
    r.normal_()
    return FakeHandle(r)
def fake_rows(send, send_counts, recv_counts, group=None):
    return torch.randn((int(sum(recv_counts)),) + tuple(send.shape[1:]), device=send.device)
S.all_to_all_rows_start = fake_start; S.all_to_all_rows = fake_rows

def make_plan(rank, world):
    bounds = S.partition_bounds(rowptr, world)
    p = object.__new__(S.HaloPlan)
    p.rank, p.world, p.group = rank, world, None
    p.lo, p.hi = int(bounds[rank]), int(bounds[rank + 1]); p.n_own = p.hi - p.lo
    e0, e1 = int(rowptr[p.lo]), int(rowptr[p.hi])
    cg = np.asarray(col[e0:e1], dtype=np.int64)
    own = (cg >= p.lo) & (cg < p.hi)
    p.halo_ids = np.unique(cg[~own]); p.n_halo = len(p.halo_ids); p.n_src = p.n_own + p.n_halo
    owner = np.searchsorted(bounds, p.halo_ids, side="right") - 1
    p.recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
    cl = np.where(own, cg - p.lo, 0); cl[~own] = p.n_own + np.searchsorted(p.halo_ids, cg[~own])
    p.rowptr, p.col = np.asarray(rowptr[p.lo:p.hi + 1], dtype=np.int64) - e0, cl
    # rows of this rank the others need
    sc, give = [], []
    for q in range(world):
        if q == rank: sc.append(0); continue
        lo, hi = int(bounds[q]), int(bounds[q + 1])
        cq = np.asarray(col[int(rowptr[lo]):int(rowptr[hi])], dtype=np.int64)
        need = np.unique(cq[(cq >= p.lo) & (cq < p.hi)])
        sc.append(len(need)); give.append(need)
    p.send_counts = np.array(sc, dtype=np.int64)
    p.send_idx = (np.concatenate(give) - p.lo).astype(np.int64) if give else np.zeros(0, np.int64)
    p.send_offsets = np.concatenate([[0], np.cumsum(p.send_counts)]).astype(np.int64)
    p.n_total = N
    p.build_unpack()
    return p, e0

def run(rank, world, steps=5):
    t0 = time.perf_counter()
    plan, e0 = make_plan(rank, world)
    t_plan = time.perf_counter() - t0
    g = torch.Generator().manual_seed(42); b = 1.0 / np.sqrt(H)
    P = lambda *s: torch.nn.Parameter(((torch.rand(*s, generator=g) * 2 - 1) * b).to(dev))
    masks = {n: P(2 * H, H) for n in names}
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    sh = S.ShardedMMA(plan, dev, H, C, names, masks, P(H, C), P(C), 0.5, edge_base=e0, **EXTRA)
    t_graph = time.perf_counter() - t0
    x = sh.feature_buffer()                        # heads the (S,H) source table: no per-call copy of the own rows
    with torch.no_grad():
        x.copy_(torch.relu(torch.randn(plan.n_own, H, device=dev)))
    cot = torch.randn(plan.n_own, C, device=dev)
    def step():
        x.grad = None
        for prm in sh.owned: prm.grad = None
        sh(x).backward(cot)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    Fn.TIMER = t = bench.KernelTimer(); t.enabled = True
    for _ in range(3): step()
    sp = t.summary(); Fn.TIMER = None
    ks = {k: round(v[1] / 3, 3) for k, v in sp.items()}
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    # the same shard plan built on the device (what ShardedMMA.build does through HaloPlan(plan_device=...)), timed beside the host one
    rp_d, cl_d = torch.from_numpy(plan.rowptr).to(dev), torch.from_numpy(plan.col).to(dev)
    mma_amd.NCGraph.from_device_csr(rp_d, cl_d, n_src=plan.n_src, edge_base=e0, H=H)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mma_amd.NCGraph.from_device_csr(rp_d, cl_d, n_src=plan.n_src, edge_base=e0, H=H)
    torch.cuda.synchronize(); t_graph_dev = time.perf_counter() - t0
    del rp_d, cl_d
    print(f"world {world} rank {rank}: own {plan.n_own} halo {plan.n_halo} send {int(plan.send_counts.sum())} edges {sh.local_edges}  "
          f"{dt:.2f} ms/step  peak {peak:.1f} GiB  plan {t_plan:.1f} s + graph {t_graph:.1f} s (host numpy; the graph plan on the device: {t_graph_dev:.3f} s)  {ks}", flush=True)
    del sh, x, cot
    torch.cuda.empty_cache()
    return dt

if __name__ == "__main__":
    worlds = [int(w) for w in args.worlds.split(",")] if args.worlds else ([8] if args.c5 else [1, 2, 4, 8])
    for world in worlds:
        ranks = [int(r) for r in args.ranks.split(",")] if args.ranks else sorted(set([0, world // 2, world - 1]))
        ts = [run(r, world, args.steps or (2 if args.c5 else 5)) for r in ranks if r < world]
        print(f"== world {world}: max rank time {max(ts):.2f} ms", flush=True)
