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
ap.add_argument("--link-gbs", type=float, default=153.0, help="xGMI model: GB/s per direction of ONE point-to-point link (MI355X: 7 links per GPU, "
                "~153 GB/s each; the 8 GPUs of a node are fully connected, so an all-to-all drives every link at once)")
ap.add_argument("--a2a-latency-us", type=float, default=20.0, help="xGMI model: fixed cost of one all-to-all-v (launch + synchronisation of "
                "8 peers), an ASSUMPTION - nothing was measured on a multi-GPU node")
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
    return S.HaloPlan.from_full_graph(rowptr, col, rank, world)

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
    pred = dt
    if world > 1:
        pred = exchange_model(plan, sh, ks, dt, rank, world)
    del sh, x, cot
    torch.cuda.empty_cache()
    return dt, pred


def exchange_model(plan, sh, ks, dt, rank, world):
    """The four all-to-all-v of one step under a STATED xGMI model (round-3 VERDICT item 3) - a prediction, nothing here was measured on
    more than one GPU.  An all-to-all-v between fully connected GPUs moves every (sender, receiver) block over its own link, all links
    at once: t = latency + max over peers of max(bytes to that peer, bytes from that peer) / link rate.  Each exchange is set against
    the compute the layer queues behind it (mma_amd/sharded.py): only the part that does not fit is added to the rehearsed step."""
    bw, lat = args.link_gbs * 1e9, args.a2a_latency_us * 1e-3
    sc, rc = plan.send_counts.astype(np.float64), plan.recv_counts.astype(np.float64)
    own_rows, halo_rows = plan.n_own, plan.n_halo
    e_own = int(sh.sg_own.col.numel())
    f_edge = e_own / max(sh.local_edges, 1)                                   # share of the edges whose source is an own row
    f_rows = 2.0 * own_rows / max(2.0 * own_rows + halo_rows, 1.0)              # share of the forward GEMM columns-x-rows done on own rows
    gemm_fwd = ks.get("gemm_x3_k128", 0.0) + ks.get("gemm_x3_persist", 0.0)
    ex = [  # name, bytes out per peer, bytes in per peer, compute queued behind it (ms)
        ("forward x rows (H floats)", sc * H * 4, rc * H * 4, gemm_fwd * f_rows),
        ("forward tail rows (C floats)", sc * C * 4, rc * C * 4, ks.get("csr_spmm_fwd", 0.0) * f_edge),
        ("backward tail gradients", rc * C * 4, sc * C * 4, ks.get("csr_spmm_bwd", 0.0) * f_edge),
        ("backward dL/dx of the halo rows", rc * H * 4, sc * H * 4,
         ks.get("nc_fused_bwd", 0.0) * f_edge + ks.get("gemm_x3_acc", 0.0) * own_rows / max(own_rows + halo_rows, 1) + ks.get("gemm_x3_tn", 0.0)),
    ]
    total = dt
    parts = []
    for name, out_b, in_b, cover in ex:
        t = lat + max(float(out_b.max()), float(in_b.max())) / bw * 1e3
        exposed = max(0.0, t - cover)
        total += exposed
        parts.append("%s: %.1f MB out / %.1f MB in, largest peer block %.2f MB -> %.3f ms, behind %.3f ms of compute -> %.3f ms exposed" % (
            name, out_b.sum() / 1e6, in_b.sum() / 1e6, max(float(out_b.max()), float(in_b.max())) / 1e6, t, cover, exposed))
    print("   xGMI model (%.0f GB/s per link and direction, %.0f us per all-to-all-v, all %d links at once): %s" % (
        args.link_gbs, args.a2a_latency_us, world - 1, " | ".join(parts)), flush=True)
    print("   => modelled step of rank %d WITH exchange: %.2f ms (rehearsed compute %.2f ms + %.3f ms exposed exchange)" % (
        rank, total, dt, total - dt), flush=True)
    return total

if __name__ == "__main__":
    worlds = [int(w) for w in args.worlds.split(",")] if args.worlds else ([8] if args.c5 else [1, 2, 4, 8])
    for world in worlds:
        ranks = [int(r) for r in args.ranks.split(",")] if args.ranks else sorted(set([0, world // 2, world - 1]))
        ts = [run(r, world, args.steps or (2 if args.c5 else 5)) for r in ranks if r < world]
        tmax, pmax = max(t for t, _ in ts), max(q for _, q in ts)
        print(f"== world {world}: max rank time {tmax:.2f} ms compute only; {pmax:.2f} ms with the modelled exchange "
              f"=> {E / pmax * 1e3 / 1e6:.0f} M edges/s aggregate (a PREDICTION: one GPU, exchange stubbed + modelled)", flush=True)
