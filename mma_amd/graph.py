"""Static graph plan for the NC path: CSR by target (from the reference's `add_all` neighbour lists,
utils.py:97-100), its transpose by source, and the work-item lists the kernels walk.

The reference captures `add_all` at construction time (layers.py:75), so all of this is built once per
graph on the host with numpy and then lives in HBM as int32."""
import numpy as np
import torch

DEFAULT_CHUNK = None   # edges per work item (longer segments - hubs - are split and summed in a second pass); None: by graph size
SPMM_CHUNK = 512       # the same for the SpMM plans (64-byte rows: as tuned in round 1)
SPMM_GROUP_BELOW = 64  # SpMM rows shorter than this run one per C/4-lane group (a group walks its row alone, four gathers in flight)


SMALL_GRAPH_EDGES = 1_000_000


def auto_plan(E, H=None):
    """(chunk, group_below, t_group_below) for a graph (or shard) of E edges; H: the feature width, when the caller knows it.  chunk: a hub chunk is walked by ONE wavefront, ~0.4 us
    per edge, so it must stay a small part of the kernel's time - which scales with E; group_below / t_group_below: items shorter
    than this run one per H/4-lane group (several per wavefront) in the forward (by-target) / backward (by-source) lists.
    Swept on the C4 R-MAT graph and on its 2 / 4 / 8-way shards after the nt policy went in (ms per step, same box each):
      E = 10.9 M (whole graph)  (512,16,16) 15.12   (1024,32,64) 14.80   (1024,48,64) 14.72   (1536,32,64) 14.83
      E = 5.4 M  (1 of 2 ranks) (512,16,16) 9.55    (512,48,64) 9.45     (1024,48,64) 9.64    (256,32,32) 9.50
      E = 2.7 M  (1 of 4)       (512,16,16) 5.80    (256,16,16) 5.71     (256,32,32) 5.60     (1024,48,64) 6.47
      E = 1.4 M  (1 of 8)       (512,16,16) 3.82    (256,16,16) 3.76     (256,32,32) 3.63     (1024,48,64) 4.74"""
    if E >= 8_000_000:
        return 1024, 48, 64
    if E >= 4_000_000:
        return 512, 48, 64
    if E < SMALL_GRAPH_EDGES and H is not None:
        # Small graphs (Cora 10.6 k edges, Pubmed 88.6 k: BASELINE configs[0], [2]) move a few MB per kernel: the kernel's time is the
        # LONGEST serial chain of dependent gathers any wavefront walks, not bytes.  A wavefront gathers EPG = 64 / (H/4 lanes per row)
        # rows per step, a grouped item (one H/4-lane group) ONE row per step: the 256-edge chunks above leave Cora's 168-edge hub as one
        # wavefront's 42-step walk (K1 60 us for 11.6 MB), and Pubmed's 31-edge items as 31-step walks of a 4-lane group (K2b 110 us).
        # So: chunks of 16 wavefront steps, grouped items below 4 edges.  Swept on the Cora / Pubmed structures (round 3, HIP-event spans
        # of the K1 / K2b calls in us, eager; (steps, group)):  Cora H=64: old plan 58.9 / 70.5, (4,4) 40.3 / 40.5, (8,4) 44.0 / 45.1,
        # (16,4) 44.1 / 47.8, (8,16) 47.6 / 51.9;  Pubmed H=16: old 68.7 / 76.4, (4,4) 62.1 / 43.4, (8,4) 67.8 / 43.3, (16,4) 51.0 / 37.7,
        # (16,16) 64.6 / 54.4.  Layer replay as one hipGraph: Cora 0.249 -> 0.197 ms, Pubmed 0.371 -> 0.319 ms.
        lpr = 1
        while lpr < min(-(-H // 4), 64):
            lpr *= 2
        epg = 64 // lpr
        return max(16, SMALL_STEPS * epg), SMALL_GROUP_BELOW, SMALL_GROUP_BELOW
    return 256, 32, 32


SMALL_STEPS = int(__import__("os").environ.get("MMA_SMALL_STEPS", "16"))
SMALL_GROUP_BELOW = int(__import__("os").environ.get("MMA_SMALL_GROUP", "4"))



def make_items(rowptr, chunk):
    """Split every CSR segment into work items of at most `chunk` edges.

    Returns items (n,4) int32 {node, ebeg, eend, slot}, hubs (h,4) int32 {node, slot_beg, slot_end, 0},
    n_slots.  slot = -1: the item is the node's whole segment (also for degree 0)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    N = len(rowptr) - 1
    deg = np.diff(rowptr)
    nch = np.maximum(1, -(-deg // chunk))
    total = int(nch.sum())
    node = np.repeat(np.arange(N, dtype=np.int64), nch)
    first = np.cumsum(nch) - nch
    ci = np.arange(total, dtype=np.int64) - first[node]
    ebeg = rowptr[node] + ci * chunk
    eend = np.minimum(ebeg + chunk, rowptr[node + 1])
    is_hub = nch[node] > 1
    slot = np.where(is_hub, np.cumsum(is_hub) - 1, -1)
    items = np.stack([node, ebeg, eend, slot], 1).astype(np.int32)
    # Longest first.  Waves take items round-robin (item = wave + r * n_waves), so with the list sorted by length
    # every wave receives one item of each size band and all waves finish together; in natural order the few
    # waves that happen to draw several hub chunks set the kernel time (measured on the C4 R-MAT graph:
    # forward 7.5 -> 5.7 ms).  Stable, so equal-length items keep ascending node order.
    items = items[np.argsort(-(eend - ebeg), kind="stable")]
    hub_nodes = np.nonzero(nch > 1)[0]
    n_slots = int(is_hub.sum())
    if len(hub_nodes):
        sb = np.cumsum(nch[hub_nodes]) - nch[hub_nodes]
        hubs = np.stack([hub_nodes, sb, sb + nch[hub_nodes], np.zeros_like(sb)], 1).astype(np.int32)
    else:
        hubs = np.zeros((0, 4), dtype=np.int32)
    return items, hubs, n_slots


def transpose_csr(rowptr, col, n_src):
    """Edges grouped by source: t_rowptr (n_src+1), t_col (target of each), t_eid (position in the forward CSR).
    Stable, so within a source the targets ascend."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    N = len(rowptr) - 1
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr))
    perm = np.argsort(col, kind="stable")
    t_rowptr = np.zeros(n_src + 1, dtype=np.int64)
    np.cumsum(np.bincount(col, minlength=n_src), out=t_rowptr[1:])
    return t_rowptr, dst[perm], perm


class NCGraph:
    """Device-resident plan.  n_src >= N allows extra source rows (halo rows in the sharded path)."""

    def __init__(self, rowptr, col, device, n_src=None, chunk=DEFAULT_CHUNK, edge_base=0, group_below=None, t_group_below=None, H=None):
        rowptr = np.asarray(rowptr, dtype=np.int64)
        col = np.asarray(col, dtype=np.int64)
        self.N = len(rowptr) - 1
        self.E = int(rowptr[-1])
        self.n_src = self.N if n_src is None else int(n_src)
        assert len(col) == self.E and (self.E == 0 or (col.min() >= 0 and col.max() < self.n_src)), "bad CSR"
        assert self.n_src < 2 ** 31 and self.E < 2 ** 31
        auto = auto_plan(self.E, H)
        chunk = auto[0] if chunk is None else chunk
        group_below = auto[1] if group_below is None else group_below
        t_group_below = auto[2] if t_group_below is None else t_group_below
        self.chunk = int(chunk)
        self.edge_base = int(edge_base)   # global position of this shard's first edge (keys the dropout hash)
        items, hubs, self.n_slots = make_items(rowptr, self.chunk)
        t_rowptr, t_col, t_eid = transpose_csr(rowptr, col, self.n_src)
        t_items, t_hubs, self.t_n_slots = make_items(t_rowptr, self.chunk)
        # the lists are sorted longest first: the head runs one item per wavefront, the short tail grouped
        self.n_wave_items = int(((items[:, 2] - items[:, 1]) >= group_below).sum())
        self.t_n_wave_items = int(((t_items[:, 2] - t_items[:, 1]) >= t_group_below).sum())
        dev = torch.device(device)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        self.device = dev
        self.rowptr, self.col = i32(rowptr), i32(col)
        self.items, self.hubs = i32(items), i32(hubs)
        self.t_rowptr, self.t_col, self.t_eid = i32(t_rowptr), i32(t_col), i32(t_eid)
        self.t_items, self.t_hubs = i32(t_items), i32(t_hubs)
        # sharded path: sources >= N are halo rows.  Their backward items go first (their gradient has to travel), the
        # own sources' items run while it is on the wire.  Each part keeps the longest-first order.
        self.t_parts = None
        if self.n_src > self.N:
            parts = []
            for sel_halo in (True, False):
                mi = (t_items[:, 0] >= self.N) == sel_halo
                mh = (t_hubs[:, 0] >= self.N) == sel_halo
                it = t_items[mi]
                parts.append((i32(it), int(((it[:, 2] - it[:, 1]) >= t_group_below).sum()), i32(t_hubs[mh])))
            self.t_parts = parts
        deg = np.diff(rowptr)
        self.max_degree = int(deg.max()) if self.N else 0
        self.inv_deg = torch.from_numpy((1.0 / np.maximum(deg, 1)).astype(np.float32)).to(dev)   # mean-kind backward

    @classmethod
    def from_add_all(cls, add_all, device, chunk=DEFAULT_CHUNK, H=None):
        rowptr = np.zeros(len(add_all) + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum([len(a) for a in add_all])
        col = (np.concatenate([np.asarray(a, dtype=np.int64) for a in add_all])
               if len(add_all) and rowptr[-1] > 0 else np.zeros(0, np.int64))
        return cls(rowptr, col, device, chunk=chunk, H=H)


class SpmmGraph:
    """CSR (and its transpose) of the adjacency the reference hands to torch.spmm (layers.py:861-862)."""

    def __init__(self, row, col, val, n_rows, n_cols, device):
        row = np.asarray(row, dtype=np.int64); col = np.asarray(col, dtype=np.int64)
        val = None if val is None else np.asarray(val, dtype=np.float32)
        dev = torch.device(device)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        order = np.lexsort((col, row))
        rp = np.zeros(n_rows + 1, dtype=np.int64); np.cumsum(np.bincount(row, minlength=n_rows), out=rp[1:])
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.rowptr, self.col = i32(rp), i32(col[order])
        self.val = None if val is None or np.all(val == 1.0) else f32(val[order])
        t_order = np.lexsort((row, col))
        trp = np.zeros(n_cols + 1, dtype=np.int64); np.cumsum(np.bincount(col, minlength=n_cols), out=trp[1:])
        self.t_rowptr, self.t_col = i32(trp), i32(row[t_order])
        self.t_val = None if self.val is None else f32(val[t_order])
        it, hb, self.n_slots = make_items(rp, SPMM_CHUNK)
        self.items, self.hubs = i32(it), i32(hb)
        self.n_wave_items = int(((it[:, 2] - it[:, 1]) >= SPMM_GROUP_BELOW).sum())     # longest first: head per wave, tail grouped
        it, hb, self.t_n_slots = make_items(trp, SPMM_CHUNK)
        self.t_items, self.t_hubs = i32(it), i32(hb)
        self.t_n_wave_items = int(((it[:, 2] - it[:, 1]) >= SPMM_GROUP_BELOW).sum())

    @classmethod
    def from_torch_sparse(cls, adj, device=None):
        """`device`: where the plan lives (the layer passes its input's device: a CPU sparse adj next to GPU features must
        not hand host pointers to the kernels)."""
        a = adj.coalesce()
        idx = a.indices().cpu().numpy()
        return cls(idx[0], idx[1], a.values().cpu().numpy(), a.shape[0], a.shape[1], adj.device if device is None else device)
