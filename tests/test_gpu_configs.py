"""BASELINE configs[2] and configs[3] at workload size against the CPU oracle (run on the GPU box: -m gpu).

  cfg3  PPI GraphSAGE, 24 graphs per batch (scripts/ppi_mle/run.py:71-77, dgl.batch at :12-14): one
        block-diagonal graph, N = 56,944, E = 818,716; mean aggregation at D = 50 (PPI's input width:
        layer 1) and D = 256 (hidden: layers 2, 3); stag/zoo/graph_sage.py:70-75,107.
  cfg4  ogbg-molhiv GIN, 4096 molecules per batch (scripts/molhiv_mle/run.py:112-123): N ~ 106 k,
        E ~ 218 k, degree 1-4; sum aggregation at D = 9 (atom features) and D = 128, then the per-graph
        mean readout (stag/layers.py:168-178); stag/zoo/gin.py:4-11 = DGL GINConv:
        rst = Linear((1 + eps) * x_dst + sum_in w (.) x_src).
Also: AmortizedDistribution.condition on device tensors (its split-GEMM + kernel-gather branch) against the
fixtures the reference's own condition() produced (stag/distributions.py:221-233).

Bars: every aggregated feature |a - b| <= 1e-5 (1 + |b|), flat, against the oracle drawing the device's
normals (util.hw_normals) and, on rows of up to 256 in-edges, against the oracle's own libm normals;
layer outputs behind a dense transform and gradients: the same bar (measured: <= 2.4e-6).
"""
import numpy as np
import pytest
import torch

from util import TOL, assert_close, hw_normals, oracle_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ppi(dev):
    import stag_amd
    from stag_amd import synthetic
    src, dst, sizes = synthetic.ppi_like()
    n = int(sizes.sum())
    assert (n, len(src), len(sizes)) == (synthetic.PPI_NODES, synthetic.PPI_EDGES, 24)
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n,
                       batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)
    return g, n, sizes


@pytest.fixture(scope="module")
def molecules(dev):
    import stag_amd
    from stag_amd import synthetic
    src, dst, sizes = synthetic.molecules_like(4096)
    n = int(sizes.sum())
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n,
                       batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)
    return g, n, sizes


def _normal(g, D, seed, offset, **kw):
    import stag_amd
    from stag_amd import _lib
    return stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, 1.0, 0.5, seed=seed, offset=offset, **kw)


@pytest.mark.parametrize("D", [50, 256])
def test_cfg3_ppi_sage_mean_aggregation(dev, oracle, ppi, D):
    from stag_amd import ops
    g, n, sizes = ppi
    og = oracle_graph(oracle, g)
    x = torch.randn(n, D, generator=torch.Generator().manual_seed(D))
    xd = x.to(dev)
    spec = oracle.make_spec("normal", 1.0, 0.5, seed=31, offset=2, Dn=D, n_edges=g.number_of_edges())
    got = ops.aggregate(g, xd, _normal(g, D, 31, 2), reduce="mean")
    with hw_normals(oracle, dev):
        ref = oracle.agg_fwd(og, x.numpy(), spec, reduce=oracle.REDUCE_MEAN)
    assert_close(got, ref, what=f"cfg3 SAGE mean D={D}, every row")
    ref_libm = oracle.agg_fwd(og, x.numpy(), spec, reduce=oracle.REDUCE_MEAN)
    assert_close(got, ref_libm, what=f"cfg3 SAGE mean D={D} vs libm normals")      # mean: sums / deg, well inside
    # block-diagonal batch: a graph's rows depend on that graph's rows only
    offs = np.concatenate([[0], np.cumsum(sizes)])
    x2 = xd.clone()
    x2[offs[3]:offs[4]] = 0.0                              # wipe graph 3
    got2 = ops.aggregate(g, x2, _normal(g, D, 31, 2), reduce="mean")
    keep = torch.ones(n, dtype=torch.bool, device=dev)
    keep[offs[3]:offs[4]] = False
    assert torch.equal(got2[keep], got[keep]) and float(got2[~keep].abs().max()) == 0.0
    # Bernoulli + in-norm, the other distribution of the scripts (scripts/ppi_mle/run.py --distribution)
    import stag_amd
    from stag_amd import _lib
    b = stag_amd.EdgeNoise(g, D, _lib.NOISE_BERNOULLI, 0.7, None, seed=5, offset=1, in_norm=True)
    refb = oracle.agg_fwd(og, x.numpy(), oracle.make_spec("bernoulli", 0.7, in_norm=True, seed=5, offset=1, Dn=D,
                                                           n_edges=g.number_of_edges()), reduce=oracle.REDUCE_MEAN)
    assert_close(ops.aggregate(g, xd, b, reduce="mean"), refb, what=f"cfg3 Bernoulli + in-norm D={D}")


@pytest.mark.parametrize("D", [256, 128])
def test_cfg3_whole_graphs_per_xcd_at_full_size(dev, oracle, D, monkeypatch):
    """BASELINE configs[2] at full size with the XCD-aware order built from the range table (whole graphs per XCD, the one
    family of stripes for the launch that does not draw, two blocks of rows in flight at D = 256): bit-identical to the
    plan-order launch — no draw, Normal, Bernoulli + in-norm; forward, and the dx pass on the transposed view — and the
    forward against the oracle; every unit of the plan appears exactly once in the order and no graph straddles two stripes."""
    import importlib
    import stag_amd
    from stag_amd import _lib, ops, synthetic
    G = importlib.import_module("stag_amd.graph")
    src, dst, sizes = synthetic.ppi_like()
    n = int(sizes.sum())
    mk = lambda: stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n,
                                batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)
    monkeypatch.setattr(G, "XCD_ORDER", "1")
    ga = mk()
    for view in (ga.csr, ga.csr_t):
        view.xcd_graphs = True                      # (D = 128 too: by default only rows of 1 KB and up)
        view.plan(64, need=True)
    monkeypatch.setattr(G, "XCD_ORDER", "0")
    gb = mk()
    assert ga.csr.plan(64)["xcd_on"] and not gb.csr.plan(64, need=True).get("xcd_on")
    order, (sh, sl), tag = ga.csr.xcd_order(ga.csr.plan(64), D)
    assert tag == 1000 + D + (512 if D > 128 else 0) and (sh == 0) == (D > 128)
    rec = order.cpu().numpy()[_lib.XCD_HEADER:].reshape(-1, 4)
    plan = ga.csr.plan(64)
    units = plan["units"].cpu().numpy()[:plan["n_units"]]
    real = rec[rec[:, 0] >= 0]
    assert len(real) == len(units) and sorted(map(tuple, real)) == sorted(map(tuple, units))
    cuts, keys, fine = ga.csr.xcd_ranges(D)[:3]
    edge_cuts = ga.csr.indptr.cpu().numpy().astype(np.int64)[np.concatenate([[0], np.cumsum(sizes)])]
    stripe_of = lambda pos: keys[np.clip(np.searchsorted(cuts, pos, side="right") - 1, 0, len(keys) - 1)] // fine
    for gi in range(len(sizes)):
        pos = np.arange(edge_cuts[gi], edge_cuts[gi + 1], 97)
        assert len(set(stripe_of(pos).tolist())) == 1, "a graph never straddles two XCD stripes"
    x = torch.randn(n, D, generator=torch.Generator().manual_seed(D))
    xd, gout = x.to(dev), torch.randn(n, D, device=dev)
    og = oracle_graph(oracle, ga)
    cases = (("none", lambda g: None, oracle.make_spec("none")),
             ("normal", lambda g: _normal(g, D, 31, 2), oracle.make_spec("normal", 1.0, 0.5, seed=31, offset=2, Dn=D, n_edges=len(src))),
             ("bernoulli+norm", lambda g: stag_amd.EdgeNoise(g, D, _lib.NOISE_BERNOULLI, 0.7, None, seed=5, offset=1, in_norm=True),
              oracle.make_spec("bernoulli", 0.7, in_norm=True, seed=5, offset=1, Dn=D, n_edges=len(src))))
    for name, noise, spec in cases:
        outs = []
        for g in (ga, gb):
            xr = xd.clone().requires_grad_(True)
            y = ops.aggregate(g, xr, noise(g), reduce="mean")
            y.backward(gout)
            outs.append((y.detach(), xr.grad))
        assert torch.equal(outs[0][0], outs[1][0]), f"cfg3 D={D} {name}: whole graphs per XCD changed a bit of the forward"
        assert torch.equal(outs[0][1], outs[1][1]), f"cfg3 D={D} {name}: ... of d/dx"
        with hw_normals(oracle, dev):
            ref = oracle.agg_fwd(og, x.numpy(), spec, reduce=oracle.REDUCE_MEAN)
        assert_close(outs[0][0], ref, what=f"cfg3 D={D} {name}, whole graphs per XCD, vs oracle")


def test_cfg3_sage_layer_stack_against_oracle_formula(dev, oracle, ppi):
    """StagLayer(GraphSAGE) 50 -> 256 -> 256 on the batch: every layer's output against
    fc_self(h) + fc_neigh(mean_in(w (.) h)) + bias computed from the oracle's aggregation in fp64
    (stag/zoo/graph_sage.py:70-75, 107-111)."""
    import stag_amd
    g, n, _ = ppi
    og = oracle_graph(oracle, g)
    torch.manual_seed(3)
    dims = [50, 256, 256]
    layers = [stag_amd.layers.StagLayer(stag_amd.zoo.GraphSAGE(a, b, aggregator_type="mean"),
                                        q_a=torch.distributions.Normal(1.0, 0.5)).to(dev)
              for a, b in zip(dims[:-1], dims[1:])]
    h = torch.randn(n, 50, generator=torch.Generator().manual_seed(8)).to(dev)
    stag_amd.manual_seed(4242)
    with torch.no_grad(), hw_normals(oracle, dev):
        for layer in layers:
            out = layer(g, h)
            nz = layer._edge_weight_handle
            D = h.shape[1]
            spec = oracle.make_spec("normal", 1.0, 0.5, seed=nz.seed, offset=nz.offset, Dn=D, n_edges=g.number_of_edges())
            agg = oracle.agg_fwd(og, h.cpu().numpy(), spec, reduce=oracle.REDUCE_MEAN).astype(np.float64)
            b = layer.base_layer
            ref = (h.cpu().numpy().astype(np.float64) @ b.fc_self.weight.detach().cpu().numpy().astype(np.float64).T
                   + agg @ b.fc_neigh.weight.detach().cpu().numpy().astype(np.float64).T
                   + b.bias.detach().cpu().numpy().astype(np.float64))
            assert_close(out, ref, what=f"cfg3 SAGE layer {D}->{out.shape[1]}")
            h = torch.relu(out)


@pytest.mark.parametrize("D", [9, 128])
def test_cfg4_molhiv_gin_sum_and_readout(dev, oracle, molecules, D):
    import stag_amd
    from stag_amd import ops
    g, n, sizes = molecules
    assert len(sizes) == 4096 and int(g.in_degrees().max()) <= 64       # tiny rows: no segments at all
    og = oracle_graph(oracle, g)
    x = torch.randn(n, D, generator=torch.Generator().manual_seed(D))
    xd = x.to(dev)
    spec = oracle.make_spec("normal", 1.0, 0.5, seed=77, offset=9, Dn=D, n_edges=g.number_of_edges())
    got = ops.aggregate(g, xd, _normal(g, D, 77, 9))
    assert_close(got, oracle.agg_fwd(og, x.numpy(), spec), what=f"cfg4 GIN sum D={D} vs libm normals")
    with hw_normals(oracle, dev):
        assert_close(got, oracle.agg_fwd(og, x.numpy(), spec), what=f"cfg4 GIN sum D={D}")
    # the per-graph mean readout of the batch (stag/layers.py:168-178 = dgl.mean_nodes)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    read = stag_amd.layers.MeanNodes()(g, got)
    assert read.shape == (4096, D)
    assert_close(read, oracle.segment_reduce(got.cpu().numpy(), offs, oracle.REDUCE_MEAN), what=f"cfg4 MeanNodes D={D}")
    tot = stag_amd.layers.SumNodes()(g, got)
    assert_close(tot, oracle.segment_reduce(got.cpu().numpy(), offs, oracle.REDUCE_SUM), what=f"cfg4 SumNodes D={D}")


@pytest.mark.parametrize("agg,eps,learn", [("sum", 0.0, False), ("sum", 0.3, True), ("mean", 0.1, False)])
def test_gin_module_against_oracle_formula(dev, oracle, molecules, agg, eps, learn):
    """zoo.GIN (stag/zoo/gin.py:4-11; DGL GINConv with apply_func = Linear):
    rst = W ((1 + eps) x + sum|mean_in w (.) x_src) + b, w explicit or drawn; forward and d/dx."""
    import stag_amd
    g, n, _ = molecules
    og = oracle_graph(oracle, g)
    D, out_f = 9, 32
    torch.manual_seed(11)
    gin = stag_amd.zoo.GIN(D, out_f, aggregator_type=agg, init_eps=eps, learn_eps=learn).to(dev)
    assert ("eps" in dict(gin.named_parameters())) == learn
    x = torch.randn(n, D, generator=torch.Generator().manual_seed(2))
    W = gin.apply_func.weight.detach().cpu().numpy().astype(np.float64)
    bias = gin.apply_func.bias.detach().cpu().numpy().astype(np.float64)
    red = oracle.REDUCE_MEAN if agg == "mean" else oracle.REDUCE_SUM
    # explicit edge weight (the boundary's call shape: base_layer.forward(graph, feat, edge_weight=w))
    w = torch.rand(g.number_of_edges(), D, generator=torch.Generator().manual_seed(3)) + 0.5
    neigh = oracle.agg_fwd(og, x.numpy(), oracle.make_spec("explicit", w.numpy()), reduce=red).astype(np.float64)
    ref = ((1.0 + eps) * x.numpy().astype(np.float64) + neigh) @ W.T + bias
    xd = x.to(dev).requires_grad_(True)
    out = gin(g, xd, edge_weight=w.to(dev))
    assert_close(out, ref, what=f"GIN {agg} explicit weight")
    # d/dx of sum(out * G): (1 + eps) G W + A^T (G W) with the same weights, from the oracle's transposed pass
    G = torch.randn(n, out_f, generator=torch.Generator().manual_seed(4))
    out.backward(G.to(dev))
    GW = G.numpy().astype(np.float64) @ W
    ogt = oracle_graph(oracle, g, transposed=True)
    gscale = (1.0 / np.maximum(g.in_degrees().cpu().numpy(), 1)).astype(np.float32) if agg == "mean" else None
    back = oracle.agg_fwd(ogt, GW.astype(np.float32), oracle.make_spec("explicit", w.numpy()), src_scale=gscale)
    assert_close(xd.grad, (1.0 + eps) * GW + back, what=f"GIN {agg} d/dx")
    # drawn weights through StagLayer
    layer = stag_amd.layers.StagLayer(gin, q_a=torch.distributions.Normal(1.0, 0.5)).to(dev)
    stag_amd.manual_seed(6)
    with torch.no_grad(), hw_normals(oracle, dev):
        out = layer(g, x.to(dev))
        nz = layer._edge_weight_handle
        spec = oracle.make_spec("normal", 1.0, 0.5, seed=nz.seed, offset=nz.offset, Dn=D, n_edges=g.number_of_edges())
        neigh = oracle.agg_fwd(og, x.numpy(), spec, reduce=red).astype(np.float64)
    assert_close(out, ((1.0 + eps) * x.numpy().astype(np.float64) + neigh) @ W.T + bias,
                 what=f"StagLayer(GIN {agg})")


@pytest.mark.parametrize("tag,of", [("re", 1), ("rec", 16)])
def test_amortized_condition_on_device_golden(dev, golden, tag, of):
    """AmortizedDistribution.condition's device branch (one GEMM over the N node rows per half of the
    embedding weight, two kernel-backed gathers, the parameter heads; stag_amd/distributions.py) against the
    per-edge loc / log_scale the reference's own condition() produced (stag/distributions.py:221-233), and
    the layer output with the reference's sampled weights injected."""
    import stag_amd
    from stag_amd.distributions import AmortizedDistribution
    q = AmortizedDistribution(16, of).to(dev)
    sd = {k[len(f"amort_{tag}_sd_"):]: torch.from_numpy(golden[k]) for k in golden.files
          if k.startswith(f"amort_{tag}_sd_")}
    q.load_state_dict(sd)
    g = stag_amd.Graph(torch.from_numpy(golden["hub40_src"]), torch.from_numpy(golden["hub40_dst"]), 40, device=dev)
    x = torch.from_numpy(golden[f"amort_{tag}_x"]).to(dev).requires_grad_(True)
    q.condition(g, x)
    assert q.new_parameters["loc"].is_cuda and q.new_parameters["loc"].shape == (g.number_of_edges(), of)
    assert_close(q.new_parameters["loc"], golden[f"amort_{tag}_loc"], what=f"{tag} loc")
    assert_close(q.new_parameters["log_scale"], golden[f"amort_{tag}_log_scale"], what=f"{tag} log_scale")
    # gradients reach the MLP and the features through the kernel-backed gathers: compare with the plain
    # torch form of the same expression (the reference's dataflow: cat -> Linear, stag/distributions.py:225-231)
    (q.new_parameters["loc"].sum() + (q.new_parameters["log_scale"] ** 2).sum()).backward()
    got = {k: p.grad.clone() for k, p in q.named_parameters()}
    gx = x.grad.clone()
    q.zero_grad()
    x2 = x.detach().clone().requires_grad_(True)
    src, dst = g.edges()
    h = q.embedding_mlp(torch.cat([x2[src], x2[dst]], dim=-1))
    loc, ls = q.parameters_mlp["loc"](h), q.parameters_mlp["log_scale"](h)
    (loc.sum() + (ls ** 2).sum()).backward()
    assert_close(gx, x2.grad.cpu().numpy(), what=f"{tag} d/dx")
    for k, p in q.named_parameters():
        ref = p.grad.cpu().numpy()
        s = max(1.0, float(np.abs(ref).max()))
        assert_close(got[k] / s, ref / s, what=f"{tag} d/d{k}")
    # the reference's layer output with ITS sampled weights injected through the explicit-weight kernel
    from stag_amd import ops
    w = torch.from_numpy(golden[f"amort_{tag}_w"]).to(dev)
    out = ops.aggregate(g, x.detach(), w.expand(g.number_of_edges(), 16).contiguous())
    assert_close(out, golden[f"amort_{tag}_out"], what=f"{tag} layer output (SumBase)")


def test_cfg1_cora_two_layer_gcn_model(dev, oracle):
    """BASELINE configs[0] — Cora-sized 2-layer GCN, hidden 16 (scripts/citation_mle/gcn/run.py:44-66: remove +
    add self loops, StagLayer(GCN(1433, 16, relu)), StagLayer(GCN(16, 7))) — on the HIP path: every layer's output
    against `((sum_in w (.) x outdeg^-1/2) W) indeg^-1/2 + b` (stag/zoo/gcn.py:67-111) from the oracle's aggregation,
    the StagModel loss and its backward."""
    import stag_amd
    rng = np.random.default_rng(0)
    n, e0 = 2708, 10556
    src = rng.integers(0, n, e0)
    dst = rng.integers(0, n, e0)
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    g.ndata["feat"] = (torch.rand(n, 1433, generator=torch.Generator().manual_seed(1)) < 0.012).float().to(dev)   # bag of words
    g = stag_amd.add_self_loop(stag_amd.remove_self_loop(g))
    assert g.number_of_edges() == int((src != dst).sum()) + n and g.ndata["feat"].shape == (n, 1433)
    x = g.ndata["feat"]
    torch.manual_seed(5)
    l1 = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(1433, 16, activation=torch.relu), q_a=torch.distributions.Normal(1.0, 0.5)).to(dev)
    l2 = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(16, 7), q_a=torch.distributions.Normal(1.0, 0.5)).to(dev)
    og = oracle_graph(oracle, g)
    ind = np.maximum(g.in_degrees().cpu().numpy(), 1).astype(np.float64)
    outd = np.maximum(g.out_degrees().cpu().numpy(), 1).astype(np.float64)
    stag_amd.manual_seed(77)
    h = x
    with torch.no_grad(), hw_normals(oracle, dev):
        for layer, act in ((l1, True), (l2, False)):
            out = layer(g, h)
            nz = layer._edge_weight_handle
            D = h.shape[1]
            spec = oracle.make_spec("normal", 1.0, 0.5, seed=nz.seed, offset=nz.offset, Dn=D, n_edges=g.number_of_edges())
            agg = oracle.agg_fwd(og, h.cpu().numpy(), spec, src_scale=(outd ** -0.5).astype(np.float32)).astype(np.float64)
            W = layer.base_layer.weight.detach().cpu().numpy().astype(np.float64)
            ref = (agg @ W) * (ind ** -0.5)[:, None] + layer.base_layer.bias.detach().cpu().numpy().astype(np.float64)
            if act:
                ref = np.maximum(ref, 0.0)
            assert_close(out, ref, what=f"cfg1 GCN layer {D}->{out.shape[1]}")
            h = out
    model = stag_amd.models.StagModel([l1, l2, stag_amd.layers.FeatOnlyLayer(torch.nn.Softmax(dim=-1))])
    y = torch.randint(0, 7, (n,), generator=torch.Generator().manual_seed(2)).to(dev)
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    mask[:140] = True
    loss = model.loss(g, x, y, mask=mask, n_samples=2)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in l1.base_layer.parameters())
    pred = model(g, x, n_samples=4, return_parameters=True)
    assert pred.shape == (n, 7) and torch.allclose(pred.sum(-1), torch.ones(n, device=dev), atol=1e-5)


def test_amortized_vi_layer_kl_and_gradients_golden(dev, golden, oracle):
    """The reference's amortised vi=True layer (fixture amort_kl: AmortizedDistribution(16, 1), a learned Normal
    prior, loss = <gout, out> + kl_divergence(); stag/layers.py:93-145) through the HIP path: condition() on the
    device (ops.node_project + ops.edge_mlp), the fused KL (ops.normal_kl_mean), and every gradient of the loss
    with the reference's own noise draw injected (eps = (w - loc) / scale from the fixture)."""
    import stag_amd
    from stag_amd import ops
    from stag_amd.distributions import AmortizedDistribution
    g = stag_amd.Graph(torch.from_numpy(golden["hub40_src"]), torch.from_numpy(golden["hub40_dst"]), 40, device=dev)
    q = AmortizedDistribution(16, 1)
    layer = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(16, 16), q_a=q, p_a=torch.distributions.Normal(0.8, 0.6), vi=True)
    sd = {k[len("amort_kl_sd_"):]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith("amort_kl_sd_")}
    missing = layer.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("base_layer.") for k in missing.missing_keys)
    layer = layer.to(dev)
    x = torch.from_numpy(golden["amort_kl_x"]).to(dev).requires_grad_(True)
    q = layer.q_a
    q.condition(g, x)
    loc, ls = q.new_parameters["loc"], q.new_parameters["log_scale"]
    assert "EdgeMlp" in type(loc.grad_fn).__name__
    assert_close(loc, golden["amort_kl_loc"], what="loc vs reference")
    assert_close(ls, golden["amort_kl_log_scale"], what="log_scale vs reference")
    kl = layer.kl_divergence()
    assert "NormalKlMean" in type(kl.grad_fn).__name__
    ref_kl = float(golden["amort_kl_value"][0])
    assert abs(float(kl.detach()) - ref_kl) <= 1e-5 * abs(ref_kl)
    assert abs(oracle.normal_kl_mean(loc.detach().cpu().numpy(), ls.detach().cpu().numpy(), 0.8, 0.6) - ref_kl) <= 1e-5 * abs(ref_kl)
    # the reference's draw: w = loc + exp(log_scale) * eps, [E, 16]
    w_ref = torch.from_numpy(golden["amort_kl_w"]).to(dev)
    eps = (w_ref - torch.from_numpy(golden["amort_kl_loc"]).to(dev)) / torch.from_numpy(golden["amort_kl_log_scale"]).to(dev).exp()
    w = loc + ls.exp() * eps
    out = ops.aggregate(g, x, w)                       # SumBase: sum_e w_e * x[src_e]
    assert_close(out, golden["amort_kl_out"], what="layer output")
    gout = torch.from_numpy(golden["amort_kl_gout"]).to(dev)
    ((out * gout).sum() + kl).backward()
    checks = [("x", x.grad)] + [(k, p.grad) for k, p in layer.named_parameters() if not k.startswith("base_layer.")]
    for k, got in checks:
        ref = golden[f"amort_kl_grad_{k}"]
        sc = max(1.0, float(np.abs(ref).max()))
        assert_close(got / sc, ref / sc, tol=2e-5, what=f"d loss / d {k}")


@pytest.mark.gpu
def test_batch_concatenates_the_parts_csr_on_the_device(dev, oracle):
    """graph.batch on the device: the union's CSR views come from the parts' views (no sort per batch) and equal, array
    for array, the views built from the union's COO; a GraphSAGE-mean aggregation with noise on the batch equals the
    oracle's on the union."""
    import stag_amd
    from stag_amd import _lib, ops, synthetic
    from util import assert_close, oracle_graph
    s, d, sizes = synthetic.ppi_like(n_graphs=5, n_nodes=2000, n_edges=30000, seed=4)
    off = np.concatenate([[0], np.cumsum(sizes)])
    gid = np.searchsorted(off, s, side="right") - 1
    parts = [stag_amd.Graph(torch.from_numpy(s[gid == i] - off[i]).to(dev), torch.from_numpy(d[gid == i] - off[i]).to(dev),
                            int(sizes[i]), device=dev) for i in range(len(sizes))]
    b = stag_amd.batch(parts)
    assert b._csr is not None and b._csr.indptr.is_cuda
    ref = stag_amd.Graph(*b.edges(), b.number_of_nodes(), device=dev)
    for v in ("csr", "csr_t"):
        for f in ("indptr", "indices", "eid", "nidx"):
            x, y = getattr(getattr(b, v), f), getattr(getattr(ref, v), f)
            assert (x is None and y is None) or torch.equal(x, y), (v, f)
    n, D = b.number_of_nodes(), 50
    x = torch.randn(n, D, device=dev)
    noise = stag_amd.EdgeNoise(b, D, _lib.NOISE_NORMAL, 1.0, 0.3, seed=5, offset=2)
    got = ops.aggregate(b, x, noise, reduce="mean")
    spec = oracle.make_spec("normal", 1.0, 0.3, seed=5, offset=2, Dn=D, n_edges=b.number_of_edges())
    want = oracle.agg_fwd(oracle_graph(oracle, ref), x.cpu().numpy(), spec, reduce=oracle.REDUCE_MEAN)
    assert_close(got, want, what="aggregation on a batch whose CSR was concatenated")

