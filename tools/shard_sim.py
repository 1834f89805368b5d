"""Single-GPU rehearsal of ONE rank of an N-rank run: the halo exchange is replaced by a local stand-in (rows filled with
random data, no communication), everything else is the real sharded layer.  Gives the per-rank compute time at 2/4/8 ranks."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mma_amd
from mma_amd import sharded as S, functional as Fn

dev = torch.device('cuda:0')
rowptr, col = bench.rmat_graph(20, 5_000_000, seed=42)
N, E = len(rowptr) - 1, int(rowptr[-1])
H, C, names = 128, 16, ["sum", "mean", "max", "min"]

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
    plan, e0 = make_plan(rank, world)
    g = torch.Generator().manual_seed(42); b = 1.0 / np.sqrt(H)
    P = lambda *s: torch.nn.Parameter(((torch.rand(*s, generator=g) * 2 - 1) * b).to(dev))
    masks = {n: P(2 * H, H) for n in names}
    sh = S.ShardedMMA(plan, dev, H, C, names, masks, P(H, C), P(C), 0.5, edge_base=e0)
    x = torch.relu(torch.randn(plan.n_own, H, device=dev)).requires_grad_(True)
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
    print(f"world {world} rank {rank}: own {plan.n_own} halo {plan.n_halo} send {int(plan.send_counts.sum())} edges {sh.local_edges}  {dt:.2f} ms/step  {ks}", flush=True)
    return dt

if __name__ == "__main__":
    for world in (1, 2, 4, 8):
        ts = [run(r, world) for r in sorted(set([0, world // 2, world - 1]))]
        print(f"== world {world}: max rank time {max(ts):.2f} ms", flush=True)
