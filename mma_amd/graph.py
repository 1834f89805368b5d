"""Static graph plan for the NC path: CSR by target (from the reference's `add_all` neighbour lists,
utils.py:97-100), its transpose by source, and the work-item lists the kernels walk.

The reference captures `add_all` at construction time (layers.py:75), so all of this is built once per
graph on the host with numpy and then lives in HBM as int32."""
import numpy as np
import torch

DEVICE_PLAN = __import__("os").environ.get("MMA_DEVICE_PLAN", "1") != "0"     # 0: the host numpy builders (round 2)
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
        # So: chunks of 16 wavefront steps, grouped items below 4 edges (8 since the one-launch form: see the end of this comment).  Swept on the Cora / Pubmed structures (round 3, HIP-event spans
        # of the K1 / K2b calls in us, eager; (steps, group)):  Cora H=64: old plan 58.9 / 70.5, (4,4) 40.3 / 40.5, (8,4) 44.0 / 45.1,
        # (16,4) 44.1 / 47.8, (8,16) 47.6 / 51.9;  Pubmed H=16: old 68.7 / 76.4, (4,4) 62.1 / 43.4, (8,4) 67.8 / 43.3, (16,4) 51.0 / 37.7,
        # (16,16) 64.6 / 54.4.  Layer replay as one hipGraph: Cora 0.249 -> 0.197 ms, Pubmed 0.371 -> 0.319 ms.
        lpr = 1
        while lpr < min(-(-H // 4), 64):
            lpr *= 2
        epg = 64 // lpr
        # One-launch form (late round 3; one call inside a replayed hipGraph, us, K1 / K2b; (forward, backward) thresholds): Cora (4,8) 12.0 /
        # 17.4, (8,8) 12.2 / 17.8, (16,8) 13.3 / 17.4, (8,16) 11.9 / 19.8; Pubmed (4,8) 28.2 / 27.3, (8,8) 24.4 / 27.2, (16,8) 21.9 / 27.1,
        # (32,8) 32.5 / 27.2, (8,32) 24.4 / 48.1: forward items shorter than one wavefront step (EPG rows) are better off grouped.
        return max(16, SMALL_STEPS * epg), max(SMALL_GROUP_BELOW, epg), SMALL_T_GROUP_BELOW
    return 256, 32, 32


SMALL_STEPS = int(__import__("os").environ.get("MMA_SMALL_STEPS", "16"))
SMALL_GROUP_BELOW = int(__import__("os").environ.get("MMA_SMALL_GROUP", "8"))
SMALL_T_GROUP_BELOW = int(__import__("os").environ.get("MMA_SMALL_T_GROUP", str(SMALL_GROUP_BELOW)))
ONE_LAUNCH = __import__("os").environ.get("MMA_ONE_LAUNCH", "1") != "0"



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

    def sync(self, which):
        """Device counter for the one-launch form of K1 (which = 0) / K2b (1) on small graphs (mma_amd.h `sync`), else None."""
        if not ONE_LAUNCH or self.E >= SMALL_GRAPH_EDGES or self.n_src != self.N:
            return None
        if getattr(self, "_sync", None) is None:
            self._sync = torch.zeros(2, dtype=torch.int32, device=self.device)
        return self._sync[which:]

    @classmethod
    def from_add_all(cls, add_all, device, chunk=DEFAULT_CHUNK, H=None):
        rowptr = np.zeros(len(add_all) + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum([len(a) for a in add_all])
        col = (np.concatenate([np.asarray(a, dtype=np.int64) for a in add_all])
               if len(add_all) and rowptr[-1] > 0 else np.zeros(0, np.int64))
        if DEVICE_PLAN and torch.device(device).type == "cuda":       # the neighbour lists go up once; the plan is built in HBM
            assert len(col) == 0 or (col.min() >= 0 and col.max() < len(add_all)), "bad neighbour lists"
            return cls.from_device_csr(torch.from_numpy(rowptr).to(device), torch.from_numpy(col).to(device), chunk=chunk, H=H)
        return cls(rowptr, col, device, chunk=chunk, H=H)


# ---- device-side plan builders (graph ingest without host preprocessing) -----------------------------------------------
# The same plans as above, bit for bit, from a CSR that already lives in HBM: the two groupings (edges by source; work items by
# length) are stable radix sorts of K6 (mma_build_csr, rocPRIM), everything else is scans and gathers.  The host only reads back
# four counts (items, hub slots, the per-wavefront / grouped split points).
def _i64(t):
    return t.to(torch.int64)


def device_make_items(rowptr, chunk):
    """make_items() on the device: (items (n,4) int32, hubs (h,4) int32, n_slots, lens (n,) int64 sorted like items)."""
    from .functional import DeviceCSR
    dev = rowptr.device
    rp = _i64(rowptr)
    N = rp.numel() - 1
    deg = rp[1:] - rp[:-1]
    nch = torch.clamp((deg + (chunk - 1)) // chunk, min=1)
    csum = torch.cumsum(nch, 0)
    total = int(csum[-1]) if N else 0
    node = torch.repeat_interleave(torch.arange(N, device=dev), nch, output_size=total)
    first = csum - nch
    ci = torch.arange(total, device=dev) - first[node]
    ebeg = rp[node] + ci * chunk
    eend = torch.minimum(ebeg + chunk, rp[node + 1])
    is_hub = nch[node] > 1
    hub_rank = torch.cumsum(is_hub.to(torch.int64), 0)
    slot = torch.where(is_hub, hub_rank - 1, torch.full_like(hub_rank, -1))
    items = torch.stack([node, ebeg, eend, slot], 1)
    lens = eend - ebeg
    n_slots = int(hub_rank[-1]) if total else 0
    if total:
        # longest first, stable (equal lengths keep ascending node / chunk order): radix sort of (max_len - len) by K6
        mx = int(lens.max())
        order = DeviceCSR((mx - lens).contiguous(), None, mx + 1).perm[:total].long()
        items, lens = items[order], lens[order]
    hub_nodes = torch.nonzero(nch > 1).flatten()
    if hub_nodes.numel():
        hn = nch[hub_nodes]
        sb = torch.cumsum(hn, 0) - hn
        hubs = torch.stack([hub_nodes, sb, sb + hn, torch.zeros_like(sb)], 1)
    else:
        hubs = torch.zeros((0, 4), dtype=torch.int64, device=dev)
    return items.to(torch.int32).contiguous(), hubs.to(torch.int32).contiguous(), n_slots, lens


def device_transpose_csr(rowptr, col, n_src):
    """transpose_csr() on the device: edges grouped by source with a stable radix sort (K6): (t_rowptr, t_col, t_eid) int32."""
    from .functional import DeviceCSR
    rp = _i64(rowptr)
    N, E = rp.numel() - 1, int(col.numel())
    if E == 0:
        z = torch.zeros((0,), dtype=torch.int32, device=rowptr.device)
        return torch.zeros((n_src + 1,), dtype=torch.int32, device=rowptr.device), z, z
    dst = torch.repeat_interleave(torch.arange(N, device=rowptr.device), rp[1:] - rp[:-1], output_size=E)
    t = DeviceCSR(_i64(col).contiguous(), dst, n_src)
    return t.rowptr, t.other[:E].contiguous(), t.perm[:E].contiguous()


def _ncgraph_from_device_csr(cls, rowptr, col, n_src=None, chunk=DEFAULT_CHUNK, edge_base=0, group_below=None, t_group_below=None, H=None):
    """NCGraph from a CSR by target that already lives on the GPU (rowptr (N+1), col (E), any integer dtype): the plan of
    NCGraph.__init__ bit for bit, built on the device."""
    assert rowptr.is_cuda and col.is_cuda
    g = object.__new__(cls)
    dev = rowptr.device
    g.N = rowptr.numel() - 1
    g.E = int(col.numel())
    g.n_src = g.N if n_src is None else int(n_src)
    assert g.n_src < 2 ** 31 and g.E < 2 ** 31
    if g.E:
        torch._assert_async(((col >= 0) & (col < g.n_src)).all())
    auto = auto_plan(g.E, H)
    chunk = auto[0] if chunk is None else chunk
    group_below = auto[1] if group_below is None else group_below
    t_group_below = auto[2] if t_group_below is None else t_group_below
    g.chunk, g.edge_base, g.device = int(chunk), int(edge_base), dev
    g.rowptr, g.col = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
    g.items, g.hubs, g.n_slots, lens = device_make_items(g.rowptr, g.chunk)
    g.t_rowptr, g.t_col, g.t_eid = device_transpose_csr(g.rowptr, g.col, g.n_src)
    g.t_items, g.t_hubs, g.t_n_slots, t_lens = device_make_items(g.t_rowptr, g.chunk)
    g.n_wave_items = int((lens >= group_below).sum())
    g.t_n_wave_items = int((t_lens >= t_group_below).sum())
    g.t_parts = None
    if g.n_src > g.N:
        parts = []
        for sel_halo in (True, False):
            mi = (g.t_items[:, 0] >= g.N) == sel_halo
            mh = (g.t_hubs[:, 0] >= g.N) == sel_halo
            parts.append((g.t_items[mi].contiguous(), int(((t_lens >= t_group_below) & mi).sum()), g.t_hubs[mh].contiguous()))
        g.t_parts = parts
    deg = (g.rowptr[1:] - g.rowptr[:-1])
    g.max_degree = int(deg.max()) if g.N else 0
    g.inv_deg = (1.0 / torch.clamp(deg, min=1).to(torch.float64)).to(torch.float32)          # as the numpy plan: fp64 reciprocal, rounded once
    return g


NCGraph.from_device_csr = classmethod(_ncgraph_from_device_csr)


class SpmmGraph:
    """CSR (and its transpose) of the adjacency the reference hands to torch.spmm (layers.py:861-862)."""

    @classmethod
    def from_device_csr(cls, rowptr, col, n_cols=None):
        """The plan of an all-ones adjacency that is already a CSR by row on the GPU with ascending columns inside a row (the raw
        0/1 adjacency of utils.py:71,114: MMA.forward's `adj` is the same CSR as its `add_all`): built on the device."""
        g = object.__new__(cls)
        g.n_rows = rowptr.numel() - 1
        g.n_cols = g.n_rows if n_cols is None else int(n_cols)
        g.rowptr, g.col = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
        g.val = g.t_val = None
        g.t_rowptr, g.t_col, _ = device_transpose_csr(g.rowptr, g.col, g.n_cols)
        g.items, g.hubs, g.n_slots, lens = device_make_items(g.rowptr, SPMM_CHUNK)
        g.n_wave_items = int((lens >= SPMM_GROUP_BELOW).sum())
        g.t_items, g.t_hubs, g.t_n_slots, t_lens = device_make_items(g.t_rowptr, SPMM_CHUNK)
        g.t_n_wave_items = int((t_lens >= SPMM_GROUP_BELOW).sum())
        return g

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

    def scaled_by_source(self, f):
        """A shallow copy of the plan with values A_ij * f_j (f: (n_cols,) / (n_cols,1) per source column, or one element for all):
        the scaler stage of MMA.forward - a row factor on the support matrix (layers.py:856-860) - folded into the SpMM's edge values,
        forward and transposed, instead of an element-wise launch each way."""
        import copy
        g = copy.copy(self)
        f = f.detach().reshape(-1).to(torch.float32)
        E = self.col.numel()
        if f.numel() == 1:
            v = tv = f.expand(E)
        else:
            assert f.numel() == self.n_cols
            v = f.index_select(0, self.col.long())
            counts = (self.t_rowptr[1:] - self.t_rowptr[:-1]).long()
            tv = torch.repeat_interleave(f, counts, output_size=E)
        g.val = (v if self.val is None else v * self.val).contiguous()
        g.t_val = (tv if self.t_val is None else tv * self.t_val).contiguous()
        return g

    @classmethod
    def from_torch_sparse(cls, adj, device=None):
        """`device`: where the plan lives (the layer passes its input's device: a CPU sparse adj next to GPU features must
        not hand host pointers to the kernels)."""
        a = adj.coalesce()
        dev = torch.device(adj.device if device is None else device)
        if DEVICE_PLAN and dev.type == "cuda" and a.values().numel() and bool((a.values() == 1).all()):
            # the raw 0/1 adjacency of utils.py:71,114: coalesced COO = sorted by (row, col) -> its CSR is one scan away, on the device
            idx = a.indices().to(dev)
            rowptr = torch.zeros(a.shape[0] + 1, dtype=torch.int64, device=dev)
            rowptr[1:] = torch.cumsum(torch.bincount(idx[0], minlength=a.shape[0]), 0)
            return cls.from_device_csr(rowptr, idx[1], n_cols=a.shape[1])
        idx = a.indices().cpu().numpy()
        return cls(idx[0], idx[1], a.values().cpu().numpy(), a.shape[0], a.shape[1], dev)
