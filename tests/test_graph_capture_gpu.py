"""hipGraph capture (torch.cuda.graph) of the drop-in layers: launch-bound sizes (Cora / ZINC batches) replay the whole
forward+backward as one graph.  The dropout seed lives in a device buffer re-drawn inside the graph, so every replay
gets fresh keep bits without re-capture; results equal the eager path run with the same seed, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nc_layer(p):
    import bench
    import mma_amd
    rng = np.random.default_rng(0)
    N, H, C, names = 2708, 64, 7, ["mean", "mean2"]
    deg = rng.poisson(3.9, N) + 1
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    col = np.concatenate([np.sort(rng.choice(N, d, replace=False)) for d in deg])
    graph = mma_amd.NCGraph(rowptr, col, DEV)
    layer = bench.make_layer(mma_amd, graph, H, C, names, p, DEV)
    adj = mma_amd.graph.SpmmGraph(np.repeat(np.arange(N), deg), col, None, N, N, DEV)
    x = torch.relu(torch.randn(N, H, device=DEV)).requires_grad_(True)
    cot = torch.randn(N, C, device=DEV)
    return layer, adj, x, cot


@pytest.mark.parametrize("p", [0.0, 0.5])
def test_mma_layer_graph_replay_matches_eager(p):
    from mma_amd import functional as Fn
    layer, adj, x, cot = _nc_layer(p)
    layer.graph_capturable = True

    def step():
        x.grad = None
        out = layer(x, adj)
        out.backward(cot)
        return out

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_static = step()
    grad_static = x.grad
    outs, seeds = [], []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        outs.append((out_static.clone(), grad_static.clone()))
        seeds.append(int(layer._seed_buf.item()) if p > 0 else 0)
    if p > 0:
        assert len(set(seeds)) == 3 and not torch.equal(outs[0][0], outs[1][0])    # fresh dropout bits per replay
    # eager run with the seed of the last replay == the replayed result, bit for bit
    layer.graph_capturable = False
    layer.drop_override = Fn.DropoutSpec(p, seed=seeds[-1] & 0xFFFFFFFFFFFFFFFF)
    x.grad = None
    ref = layer(x, adj)
    ref.backward(cot)
    assert torch.equal(ref, outs[-1][0]) and torch.equal(x.grad, outs[-1][1])


def test_mmaconv_graph_replay():
    import mma_amd
    from test_gr_gpu import molecule_batch
    rng = np.random.default_rng(0)
    ei, N = molecule_batch(rng, 64)
    E = ei.shape[1]
    hist = np.bincount(np.bincount(ei[1], minlength=N), minlength=5)
    conv = mma_amd.MMAConv(75, 75, ["min", "max"], ["identity", "amplification", "linear"], torch.tensor(hist), edge_dim=50,
                           towers=5).to(DEV)
    conv.graph_capturable = True
    x = torch.randn(N, 75, device=DEV, requires_grad=True)
    ea = torch.randn(E, 50, device=DEV)
    eig = torch.from_numpy(ei).to(DEV)
    cot = torch.randn(N, 75, device=DEV)

    def step():
        x.grad = None
        out = conv(x, eig, ea)
        out.backward(cot)
        return out

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_static = step()
    g.replay(); torch.cuda.synchronize(); a = out_static.clone()
    g.replay(); torch.cuda.synchronize(); b = out_static.clone()
    assert torch.isfinite(a).all() and not torch.equal(a, b)        # the always-on dropout re-draws per replay
    # timing: eager vs replay (printed with -s; the point is the order of magnitude)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 20 * 1e3
    print("MMAConv ZINC-like batch 64: eager %.3f ms, hipGraph replay %.3f ms per layer fwd+bwd" % (eager, rep))
    assert rep < eager


def _net_batches(n_batches, n_graphs, seed=5):
    from test_gr_gpu import molecule_batch
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_batches):
        ei, N, sizes = molecule_batch(rng, n_graphs, return_sizes=True)
        out.append(dict(x=torch.from_numpy(rng.integers(0, 21, (N, 1))).to(DEV), ei=torch.from_numpy(ei).to(DEV),
                        ea=torch.from_numpy(rng.integers(0, 4, ei.shape[1])).to(DEV),
                        batch=torch.from_numpy(np.repeat(np.arange(n_graphs), sizes)).to(DEV),
                        y=torch.from_numpy(rng.standard_normal(n_graphs).astype(np.float32)).to(DEV), N=N, E=ei.shape[1]))
    return out


def _make_net(seed, hist, p, aggregators=("min", "max")):
    import mma_amd
    from mma_amd.net import Net
    torch.manual_seed(seed)
    net = Net(list(aggregators), ["identity", "amplification", "linear"], hist).to(DEV)
    for conv in net.convs:
        conv.dropout = p
    return net, mma_amd.FusedAdam([q for q in net.parameters() if q.requires_grad], lr=1e-3)


def test_graphed_net_step_equals_the_eager_padded_step_bit_for_bit():
    """Round-2 VERDICT item 2b: the whole graph-regression TRAINING step (mma.py:150-160: Net forward incl. the per-batch CSR build,
    L1 loss, backward, Adam) as ONE hipGraph over padded static-shape buffers.  At p = 0 the replay must equal the same padded step run
    eagerly BIT FOR BIT - loss and every parameter, over several batches of different sizes - and the padded step must equal the
    plain unpadded step of the reference's loop within fp32 tolerance (BatchNorm sums in another order; dummy rows are masked out)."""
    import mma_amd
    batches = _net_batches(5, 16)
    hist = torch.bincount(torch.bincount(batches[0]["ei"][1].cpu(), minlength=batches[0]["N"]), minlength=5)
    n_pad = max(b["N"] for b in batches) + 9
    e_pad = max(b["E"] for b in batches) + 14
    net_g, opt_g = _make_net(3, hist, 0.0)
    net_e, opt_e = _make_net(3, hist, 0.0)
    sg = mma_amd.GraphedNetStep(net_g, opt_g, 16, n_pad, e_pad, DEV, warmup=2)
    se = mma_amd.GraphedNetStep(net_e, opt_e, 16, n_pad, e_pad, DEV)
    b0 = batches[0]
    lg = sg(b0["x"], b0["ei"], b0["ea"], b0["batch"], b0["y"])         # 2 warm-up steps, UNDONE, + the replay: ONE optimizer
    se.load(b0["x"], b0["ei"], b0["ea"], b0["batch"], b0["y"])         # step on the first batch, as mma.py:150-160 takes one per batch
    le = se.step_eager()
    assert torch.equal(lg, le)
    assert all(int(m.num_batches_tracked) == 1 for m in net_g.batch_norms)          # BatchNorm saw the first batch once
    assert float(opt_g.state[next(iter(net_g.parameters()))]["step"]) == 1.0         # ... and Adam stepped once
    for qg, qe in zip(net_g.convs[0].unregistered_parameters(), net_e.convs[0].unregistered_parameters()):
        assert (qg.grad is None) == (qe.grad is None) and (qg.grad is None or torch.equal(qg.grad, qe.grad))   # G2: one backward's worth
    for b in batches[1:] + batches[:2]:
        lg = sg(b["x"], b["ei"], b["ea"], b["batch"], b["y"]).clone()
        se.load(b["x"], b["ei"], b["ea"], b["batch"], b["y"])
        le = se.step_eager()
        assert torch.equal(lg, le), (lg.item(), le.item())
    for (n1, p1), (n2, p2) in zip(net_g.named_parameters(), net_e.named_parameters()):
        assert torch.equal(p1, p2), n1
    for m1, m2 in zip(net_g.batch_norms, net_e.batch_norms):
        assert torch.equal(m1.running_mean, m2.running_mean) and torch.equal(m1.running_var, m2.running_var)
    # the padded step against the reference's own (unpadded) step on the same first batch, from the same initial parameters.
    # Gradients, not parameters after Adam: the first Adam step is g / (|g| + 1e-8), ill-conditioned wherever |g| is near eps.
    # (a) [sum, mean]: no routing decision anywhere, so the padded step must reproduce the unpadded one entry by entry to 2e-5 of a
    #     gradient's largest entry (round-3's bar; round-4 ADVICE: the bar below alone cannot see a 1e-3-level regression of the padded step).
    # (b) [min, max], mma.py's setting: the two forwards differ in their last bits (BatchNorm's reduction order) and a min / max aggregator
    #     hands its gradient to ONE edge: a near-tie can go to another edge in the other run, and the rows of a gradient that edge feeds
    #     then differ at 1e-3 of the largest entry (measured with the fp32 kernels of round 3 as well: seed 4, batches 3 and 4 of this
    #     generator: 5e-4 and 5e-3; with K13 on bf16 pieces seed 3, batch 0: 1e-3 in two rows of node_emb, and - the edge sits in the last
    #     layer - 1e-4 in everything the back-propagation reaches from there).  A discontinuity of the reference's own formula, so this
    #     leg keeps the loose bar (norm 5e-3, entries 1e-2 of the largest) and is a sanity check only; the loss, which no routing
    #     decision enters, agrees to 1e-5 in both legs.
    _padded_step_vs_unpadded(batches[0], hist, n_pad, e_pad, ("sum", "mean"), 2e-5, 2e-5)
    _padded_step_vs_unpadded(batches[0], hist, n_pad, e_pad, ("min", "max"), 1e-2, 5e-3)


def _padded_step_vs_unpadded(b, hist, n_pad, e_pad, aggregators, elem_bar, norm_bar):
    import mma_amd
    net_u, opt_u = _make_net(3, hist, 0.0, aggregators)
    net_u.train()
    opt_u.zero_grad(set_to_none=False)
    out = net_u(b["x"], b["ei"], b["ea"], b["batch"])
    loss_u = mma_amd.fused_l1_loss(out.squeeze(-1), b["y"])
    net_p, opt_p = _make_net(3, hist, 0.0, aggregators)
    sp = mma_amd.GraphedNetStep(net_p, opt_p, 16, n_pad, e_pad, DEV)
    sp.load(b["x"], b["ei"], b["ea"], b["batch"], b["y"])
    loss_p = sp.forward_backward()
    assert abs(loss_u.item() - loss_p.item()) <= 1e-5 * max(1.0, abs(loss_u.item())), (aggregators, loss_u.item(), loss_p.item())
    loss_u.backward()
    worst = 0.0
    for (n1, p1), (_, p2) in zip(net_u.named_parameters(), net_p.named_parameters()):
        g1 = p1.grad if p1.grad is not None else torch.zeros_like(p1)
        g2 = p2.grad if p2.grad is not None else torch.zeros_like(p2)
        d = (g1 - g2).abs()                                # (a bias in front of BatchNorm has an exactly-zero gradient: both sides hold rounding noise)
        worst = max(worst, d.max().item() / max(g1.abs().max().item(), 1e-6))
        assert d.max().item() <= max(elem_bar * g1.abs().max().item(), 1e-6), (aggregators, n1, d.max().item(), g1.abs().max().item())
        assert (g1 - g2).norm().item() <= norm_bar * g1.norm().item() + 1e-6, (aggregators, n1, (g1 - g2).norm().item(), g1.norm().item())
    print("padded vs unpadded step, aggregators %s: worst |dg| / max|g| over the parameters = %.2e (bar %g)" % (",".join(aggregators), worst, elem_bar))


def test_graphed_net_step_refuses_a_batch_that_does_not_fit():
    import mma_amd
    batches = _net_batches(1, 8)
    b = batches[0]
    hist = torch.bincount(torch.bincount(b["ei"][1].cpu(), minlength=b["N"]), minlength=5)
    net, opt = _make_net(0, hist, 0.5)
    st = mma_amd.GraphedNetStep(net, opt, 8, b["N"], b["E"], DEV)       # no room for the dummy node
    with pytest.raises(ValueError, match="does not fit"):
        st.load(b["x"], b["ei"], b["ea"], b["batch"], b["y"])


@pytest.mark.parametrize("N,nv,C", [(333, 301, 75), (5, 3, 2), (64, 64, 4), (1345, 1286, 75), (5000, 4999, 130), (513, 2, 7)])
def test_fused_masked_bn_relu_equals_the_torch_formulation(N, nv, C):
    """K17 against net.masked_batch_norm + relu (torch ops) and, on the valid rows, against torch's own BatchNorm1d on the unpadded
    rows: outputs, input / weight / bias gradients, running statistics; fewer rows than row lanes, more than one 8-row batch per lane,
    column counts that are no multiple of the 4 columns a workgroup owns."""
    from mma_amd import net as NN
    g = torch.Generator().manual_seed(3)
    x0 = (torch.randn(N, C, generator=g) * 2 + 0.5).to(DEV)
    cot = torch.randn(N, C, generator=g).to(DEV)
    cot[nv:] = 0                                      # rows of the dummy graph: the loss never reads them
    n_valid = torch.tensor(nv, device=DEV)
    res = []
    for fused in (True, False):
        NN.FUSED_BN = fused
        bn = torch.nn.BatchNorm1d(C).to(DEV)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.5, 0.5, C))
        x = x0.clone().requires_grad_(True)
        y = NN.masked_bn_relu(x, bn, n_valid)
        y.backward(cot)
        res.append((y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    NN.FUSED_BN = True
    assert torch.equal(res[0][1][nv:], torch.zeros(N - nv, C, device=DEV))        # padded rows: no gradient (their g is 0)
    for a, b, what in zip(res[0][:6], res[1][:6], ("y", "gx", "gweight", "gbias", "running_mean", "running_var")):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-5), (what, (a - b).abs().max().item())
    assert res[0][6] == res[1][6] == 1
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.5, 0.5, C))
    xr = x0[:nv].clone().requires_grad_(True)
    yr = torch.relu(bn(xr)); yr.backward(cot[:nv])
    assert torch.allclose(res[0][0][:nv], yr.detach(), rtol=2e-5, atol=2e-5) and torch.allclose(res[0][1][:nv], xr.grad, rtol=2e-5, atol=2e-5)
    assert torch.allclose(res[0][4], bn.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(res[0][5], bn.running_var, rtol=1e-5, atol=1e-6)


def test_global_add_pool_takes_an_unsorted_batch_vector():
    """PyG's global_add_pool (mma.py:124) accepts any batch vector; the segment kernel's contiguous-range form is only for
    callers that guarantee a sorted one.  Unsorted (and sorted) vectors must give index_add_'s sums, in a fixed order, with gradients."""
    from mma_amd.net import global_add_pool
    g = torch.Generator().manual_seed(0)
    N, C, G = 1000, 75, 17
    x = torch.randn(N, C, generator=g).to(DEV).requires_grad_(True)
    for batch in (torch.randint(0, G, (N,), generator=g), torch.sort(torch.randint(0, G, (N,), generator=g)).values,
                  torch.full((N,), 3, dtype=torch.int64)):            # unsorted, sorted, one graph (others empty)
        b = batch.to(DEV)
        ref = torch.zeros(G, C, dtype=torch.float64, device=DEV).index_add_(0, b, x.detach().double())
        out = global_add_pool(x, b, G)
        assert (out.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        assert torch.equal(out, global_add_pool(x, b, G))              # fixed order: bit-identical from call to call
        cot = torch.randn(G, C, generator=g).to(DEV)
        gx, = torch.autograd.grad((out * cot).sum(), [x])
        assert torch.equal(gx, cot[b])
    srt = torch.sort(torch.randint(0, G, (N,), generator=g)).values.to(DEV)
    assert torch.equal(global_add_pool(x, srt, G, assume_sorted=True), global_add_pool(x, srt, G))


def test_csr_build_inside_a_graph_rezeroes_the_long_list_and_a_bad_count_is_not_believed():
    """Round 4: bench.py's C2net graph faulted on its fifth replay - gr_fwd_list_kernel read a count of 0x03030303 from the long-segment
    list (bytes that had lived at that address before the capture's pool took it over; the count was zeroed by a 4-byte memset NODE).
    Now (a) the count is zeroed by the build's first kernel, replay after replay, and (b) the list kernels clamp the count to the list's
    capacity and skip ids that are not nodes: a poisoned list can cost time, never an access outside the graph's arrays."""
    import mma_amd
    from mma_amd import functional as Fn
    g = torch.Generator().manual_seed(5)
    E, N = 2816, 1344
    key = torch.randint(0, N, (E,), generator=g).to(DEV)
    other = torch.randint(0, N, (E,), generator=g).to(DEV)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        Fn.DeviceCSR(key, other, N, long_list=True)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        csr = Fn.DeviceCSR(key, other, N, long_list=True)
    for _ in range(3):
        csr.long_nodes.fill_(0x03030303)
        graph.replay()
        torch.cuda.synchronize()
        assert int(csr.long_nodes[0]) == 0 and int(csr.rowptr[-1]) == E
    # the layer kernels with a list whose count word and ids are garbage: same result as with the true (empty) list
    T, F = 2, 8
    ei = torch.stack([other, key])
    x = torch.randn(N, T * F, generator=g).to(DEV)
    conv_graph = Fn.gr_graph(ei, N)
    UV = torch.randn(N, 2 * T * F, generator=g).to(DEV).requires_grad_(True)

    def run():
        UV.grad = None
        out = Fn.gr_fused_conv(UV, None, conv_graph, T, F, ["min", "max", "mean"], ["identity", "amplification"], 1.0, 2.0, Fn.DropoutSpec(0.0))
        out.sum().backward()
        return out.detach().clone(), UV.grad.clone()
    ref = run()
    conv_graph.by_target.check()                               # a healthy list: the error flag behind it is clear
    ln = conv_graph.by_target.long_nodes
    assert int(ln[0]) == 0 and int(ln[-1]) == 0
    ln[:-1].fill_(0x7f7f7f7f)                                 # count and ids: garbage far beyond the list and beyond N
    got = run()
    ln[0] = 0
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    # [r5] ... and the skip is not silent (round-4 ADVICE): the list kernels raised the flag behind the list, check() reports it
    assert int(ln[-1]) in (1, 2)                               # the count beyond the capacity (1), then the garbage ids it did walk (2)
    with pytest.raises(mma_amd._lib.MMALibraryError, match="clobbered"):
        conv_graph.by_target.check()
    ln[-1] = 0
    ln[0] = 3; ln[1:4] = N + 7                                # a believable count, ids that are not nodes
    got = run()
    assert torch.equal(ref[0], got[0]) and int(ln[-1]) == 2
    ln[0] = 0; ln[-1] = 0


def test_bench_c2net_flow_eager_phase_then_graphed_step_replays_without_a_fault():
    """The exact process history in which bench.py's default run faulted in round 4 (docs/DESIGN_rounds_1-4.md section 0, item 0): an eager training phase
    of the Net on rotating batches, THEN a second Net captured as one hipGraph and replayed past the fifth replay.  A fault would take the
    test process down; the replayed losses must also be finite and the long-segment list of the captured CSR empty after every replay."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mma_amd import functional as Fn
    r = bench.gr_model_config("C2net (test)", 64, torch.device(DEV), reps=8, graphed=True)
    torch.cuda.synchronize()
    assert np.isfinite(r["loss_after_hipgraph"]) and all(np.isfinite(v) for v in r["loss_first_steps_hipgraph"])
    assert r["ms_per_step_hipgraph"] < r["ms_per_step_eager"]
    for g in Fn._GR_GRAPHS.values():
        assert int(g.by_target.long_nodes[0]) == 0
