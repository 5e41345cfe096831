#!/usr/bin/env python
"""Generate tests/golden/*.npz by EXECUTING the reference's own Python source.

Run in the build container only (needs /root/reference; the GPU box never sees
it):  python tests/golden/make_golden.py

How: the reference (yuanqing-wang/stag) is pure Python over `dgl`, which is not
installed here and cannot be fetched.  `stag/distributions.py` needs only torch.
For `stag/layers.py`, `stag/models.py` and `stag/zoo/*.py` this script installs a
small stand-in `dgl` module (written here from DGL's documented semantics:
u_mul_e / copy_e / u_add_v messages, sum / mean reducers, edge_softmax, and the
constructor attributes of GraphConv / SAGEConv / GATConv that the reference's
`forward` overrides read) and then imports the reference from /root/reference.
Everything ABOVE the DGL primitives in the fixtures is therefore computed by the
reference's own code: noise sampling, relu, `_in_norm`, the Dn rule, GCN 'both'
normalisation, SAGE mean + fc_self/fc_neigh, GAT noisy-logit softmax, the
StagModel loss.  The DGL primitives themselves are "parity unpinned" (SURVEY.md
§8c): the reference's tests hold no values for them.

The fixtures are data only: inputs, the torch RNG seed, and the tensors the
reference produced.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("STAG_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# stand-in dgl (test infrastructure; DGL semantics restated, not reference code)
# --------------------------------------------------------------------------- #
class _Msg:
    def __init__(self, kind, *fields):
        self.kind, self.fields = kind, fields


class _Red:
    def __init__(self, kind, msg, out):
        self.kind, self.msg, self.out = kind, msg, out


class _EdgeBatch:
    def __init__(self, g):
        self.src = {k: v[g._src] for k, v in g.srcdata.items()}
        self.dst = {k: v[g._dst] for k, v in g.dstdata.items()}
        self.data = g.edata


class StandInGraph:
    is_block = False

    def __init__(self, src, dst, num_nodes, batch_num_nodes=None):
        self._src = torch.as_tensor(src, dtype=torch.int64)
        self._dst = torch.as_tensor(dst, dtype=torch.int64)
        self._n = int(num_nodes)
        self.ndata, self.edata = {}, {}
        self._batch_num_nodes = batch_num_nodes

    srcdata = property(lambda self: self.ndata)
    dstdata = property(lambda self: self.ndata)

    def local_var(self):
        g = StandInGraph(self._src, self._dst, self._n, self._batch_num_nodes)
        g.ndata, g.edata = dict(self.ndata), dict(self.edata)
        return g

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata.clear(); self.ndata.update(nd)
            self.edata.clear(); self.edata.update(ed)

    def number_of_edges(self): return int(self._src.shape[0])
    def number_of_nodes(self): return self._n
    def number_of_dst_nodes(self): return self._n
    num_dst_nodes = number_of_dst_nodes
    def in_degrees(self): return torch.bincount(self._dst, minlength=self._n)
    def out_degrees(self): return torch.bincount(self._src, minlength=self._n)
    def to(self, device): return self

    def _message(self, m):
        if m.kind in ("copy_u",):
            return self.srcdata[m.fields[0]][self._src], m.fields[1]
        if m.kind == "copy_e":
            return self.edata[m.fields[0]], m.fields[1]
        if m.kind == "u_mul_e":
            return self.srcdata[m.fields[0]][self._src] * self.edata[m.fields[1]], m.fields[2]
        if m.kind == "u_add_v":
            return self.srcdata[m.fields[0]][self._src] + self.dstdata[m.fields[1]][self._dst], m.fields[2]
        raise NotImplementedError(m.kind)

    def update_all(self, msg, red):
        m, _ = self._message(msg)
        out = torch.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype)
        if red.kind in ("sum", "mean"):
            out.index_add_(0, self._dst, m)
            if red.kind == "mean":
                deg = self.in_degrees().clamp(min=1).to(m.dtype)
                out = out / deg.reshape((-1,) + (1,) * (m.dim() - 1))
        elif red.kind == "max":
            idx = self._dst.reshape((-1,) + (1,) * (m.dim() - 1)).expand_as(m)
            out = out.scatter_reduce(0, idx, m, reduce="amax", include_self=False)
        else:
            raise NotImplementedError(red.kind)
        self.ndata[red.out] = out

    def apply_edges(self, func):
        if isinstance(func, _Msg):
            m, name = self._message(func)
            self.edata[name] = m
        else:
            self.edata.update(func(_EdgeBatch(self)))


def _edge_softmax(graph, e):
    n = graph.number_of_nodes()
    idx = graph._dst.reshape((-1,) + (1,) * (e.dim() - 1)).expand_as(e)
    mx = torch.full((n,) + tuple(e.shape[1:]), -float("inf"), dtype=e.dtype)
    mx = mx.scatter_reduce(0, idx, e, reduce="amax", include_self=True)
    ex = torch.exp(e - mx[graph._dst])
    den = torch.zeros_like(mx).index_add_(0, graph._dst, ex)
    return ex / den[graph._dst]


def _install_standin():
    nn = torch.nn
    dgl = types.ModuleType("dgl")
    fn = types.ModuleType("dgl.function")
    fn.copy_edge = fn.copy_e = lambda e, out: _Msg("copy_e", e, out)
    fn.copy_src = fn.copy_u = lambda u, out: _Msg("copy_u", u, out)
    fn.u_mul_e = lambda u, e, out: _Msg("u_mul_e", u, e, out)
    fn.u_add_v = lambda u, v, out: _Msg("u_add_v", u, v, out)
    fn.sum = lambda msg, out: _Red("sum", msg, out)
    fn.mean = lambda msg, out: _Red("mean", msg, out)
    fn.max = lambda msg, out: _Red("max", msg, out)

    base = types.ModuleType("dgl.base")

    class DGLError(Exception):
        pass
    base.DGLError = DGLError

    utils = types.ModuleType("dgl.utils")
    utils.expand_as_pair = lambda feat, g=None: feat if isinstance(feat, tuple) else (feat, feat)
    utils.check_eq_shape = lambda feat: None

    dnn = types.ModuleType("dgl.nn")
    dnn.edge_softmax = _edge_softmax

    class GraphConv(nn.Module):
        def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True,
                     activation=None, allow_zero_in_degree=False):
            super().__init__()
            self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
            self._allow_zero_in_degree = allow_zero_in_degree
            self.weight = nn.Parameter(torch.empty(in_feats, out_feats)) if weight else None
            self.bias = nn.Parameter(torch.zeros(out_feats)) if bias else None
            if self.weight is not None:
                nn.init.xavier_uniform_(self.weight)
            self._activation = activation

    class SAGEConv(nn.Module):
        def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, bias=True,
                     norm=None, activation=None):
            super().__init__()
            self._in_src_feats = self._in_dst_feats = in_feats
            self._out_feats, self._aggre_type = out_feats, aggregator_type
            self.norm, self.activation = norm, activation
            self.feat_drop = nn.Dropout(feat_drop)
            if aggregator_type == "pool":
                self.fc_pool = nn.Linear(in_feats, in_feats)
            self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
            self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
            self.bias = nn.Parameter(torch.zeros(out_feats)) if bias else None

        def _compatibility_check(self):
            pass

    class GATConv(nn.Module):
        def __init__(self, in_feats, out_feats, num_heads, feat_drop=0.0, attn_drop=0.0,
                     negative_slope=0.2, residual=False, activation=None,
                     allow_zero_in_degree=False, bias=True):
            super().__init__()
            self._num_heads, self._out_feats = num_heads, out_feats
            self._in_src_feats = self._in_dst_feats = in_feats
            self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
            self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
            self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
            self.feat_drop, self.attn_drop = nn.Dropout(feat_drop), nn.Dropout(attn_drop)
            self.leaky_relu = nn.LeakyReLU(negative_slope)
            self.bias = nn.Parameter(torch.zeros(num_heads * out_feats)) if bias else None
            self.res_fc = (nn.Linear(in_feats, num_heads * out_feats, bias=False)
                           if residual else None)
            self.activation = activation
            self.reset_parameters()

    class GINConv(nn.Module):
        def __init__(self, apply_func=None, aggregator_type="sum", init_eps=0, learn_eps=False):
            super().__init__()
            self.apply_func = apply_func

    dnn.GraphConv, dnn.SAGEConv, dnn.GATConv, dnn.GINConv = GraphConv, SAGEConv, GATConv, GINConv
    dgl.function, dgl.nn, dgl.base, dgl.utils = fn, dnn, base, utils
    for name, mod in (("dgl", dgl), ("dgl.function", fn), ("dgl.nn", dnn),
                      ("dgl.base", base), ("dgl.utils", utils)):
        sys.modules[name] = mod
    return dgl


# --------------------------------------------------------------------------- #
def _graphs():
    """name -> (src, dst, N).  Edge cases the domain has: the reference test's
    rand_graph(3, 9) shape, a zero-in-degree node, duplicate edges, self loops, a
    hub row, an isolated node."""
    rng = np.random.default_rng(20261003)
    gs = {}
    gs["rand3x9"] = (rng.integers(0, 3, 9), rng.integers(0, 3, 9), 3)
    src = np.concatenate([rng.integers(0, 40, 150), np.arange(40), np.full(30, 7), [1, 1, 1]])
    dst = np.concatenate([rng.integers(1, 39, 150), np.full(40, 5), rng.integers(1, 39, 30), [2, 2, 2]])
    gs["hub40"] = (src, dst, 40)   # node 0 and 39 have zero in-degree, node 5 is a hub, (1->2) x3
    src = rng.integers(0, 12, 30); dst = rng.integers(0, 12, 30)
    gs["selfloop12"] = (np.concatenate([src, np.arange(12)]), np.concatenate([dst, np.arange(12)]), 12)
    return gs


def main():
    _install_standin()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import stag  # noqa: F401  (the reference)
    from stag.distributions import ParametrizedDistribution, AmortizedDistribution
    from stag.layers import StagLayer, FeatOnlyLayer
    from stag.models import StagModel
    import stag.zoo as zoo
    import dgl.function as fn

    class SumBase(torch.nn.Module):
        """minimal base layer: exactly the aggregation lines of stag/zoo/gcn.py:61-63,94-96"""
        def forward(self, graph, feat, edge_weight=None):
            with graph.local_scope():
                graph.edata["_edge_weight"] = edge_weight
                graph.srcdata["h"] = feat
                graph.update_all(fn.u_mul_e("h", "_edge_weight", "m"), fn.sum("m", "h"))
                return graph.dstdata["h"]

    def quiet(f, *a, **k):
        with contextlib.redirect_stdout(io.StringIO()):
            return f(*a, **k)

    fx = {}

    # ---- (a) distributions: parameter names / values / shapes ------------------
    d = ParametrizedDistribution(torch.distributions.Normal(1.0, 0.5), vi=True)
    fx["pd_vi_names"] = np.array(sorted(n for n, _ in d.named_parameters()))
    fx["pd_vi_loc"] = d.loc.detach().numpy()
    fx["pd_vi_log_scale"] = d.log_scale.detach().numpy()
    d = ParametrizedDistribution(torch.distributions.Normal(1.0, 0.5))
    fx["pd_buf_names"] = np.array(sorted(n for n, _ in d.named_buffers()))
    half = 0.3 * np.sqrt(3.0)
    d = ParametrizedDistribution(torch.distributions.Uniform(1.0 - half, 1.0 + half, validate_args=False))
    fx["pd_uniform_names"] = np.array(sorted(n for n, _ in d.named_buffers()))
    fx["pd_uniform_low_high"] = np.array([d.low.item(), d.high.item()], np.float32)
    prob = 0.5 * (1.0 + np.sqrt(1 - 4.0 * 0.3 ** 2))
    d = ParametrizedDistribution(torch.distributions.Bernoulli(probs=prob))
    fx["pd_bernoulli_names"] = np.array(sorted(n for n, _ in d.named_buffers()))
    fx["pd_bernoulli_probs"] = np.array([d.probs.item()], np.float32)
    torch.manual_seed(11)
    d = ParametrizedDistribution(torch.distributions.Normal(torch.zeros(10, 8), torch.ones(10, 8)))
    fx["pd_expand_shape"] = np.array(d.expand(torch.Size([12, 11, 10, 8])).rsample().shape)

    gs = _graphs()
    for gname, (src, dst, n) in gs.items():
        fx[f"{gname}_src"], fx[f"{gname}_dst"] = np.asarray(src, np.int64), np.asarray(dst, np.int64)
        fx[f"{gname}_n"] = np.array(n)

    def G(name):
        src, dst, n = gs[name]
        return StandInGraph(src, dst, n)

    # ---- (b) StagLayer.forward over a bare sum aggregator ----------------------
    # modes: r1 (scalar), rc (per-channel), relu, Uniform, Bernoulli + norm
    cases = []
    for gname in gs:
        for D in (16, 5):
            cases += [
                (f"layer_{gname}_D{D}_r1", gname, D, dict(q_a=torch.distributions.Normal(1.0, 0.5))),
                (f"layer_{gname}_D{D}_rc", gname, D, dict(q_a=torch.distributions.Normal(
                    torch.linspace(0.5, 1.5, D), torch.linspace(0.1, 1.0, D)))),
                (f"layer_{gname}_D{D}_relu", gname, D, dict(q_a=torch.distributions.Normal(0.2, 1.0), relu=True)),
                (f"layer_{gname}_D{D}_unif", gname, D, dict(q_a=torch.distributions.Uniform(
                    1.0 - half, 1.0 + half, validate_args=False))),
                (f"layer_{gname}_D{D}_bern", gname, D, dict(q_a=torch.distributions.Bernoulli(probs=0.7), norm=True)),
            ]
    names = []
    for i, (cname, gname, D, kw) in enumerate(cases):
        torch.manual_seed(1000 + i)
        g = G(gname)
        x = torch.randn(g.number_of_nodes(), D)
        layer = StagLayer(SumBase(), **kw)
        torch.manual_seed(2000 + i)
        out = quiet(layer, g, x)
        fx[cname + "_x"] = x.numpy()
        fx[cname + "_w"] = layer._edge_weight_sample.numpy()
        fx[cname + "_out"] = out.detach().numpy()
        names.append(cname)
    fx["layer_cases"] = np.array(names)

    # ---- (b') amortised (re / rec): condition() outputs given fixed weights -----
    for tag, of in (("re", 1), ("rec", 16)):
        torch.manual_seed(31 + of)
        g = G("hub40")
        x = torch.randn(40, 16)
        q = AmortizedDistribution(16, of, init_like=torch.distributions.Normal(1.0, 0.3))
        layer = StagLayer(SumBase(), q_a=q)
        torch.manual_seed(77)
        out = quiet(layer, g, x)
        for k, v in q.state_dict().items():
            fx[f"amort_{tag}_sd_{k}"] = v.numpy()
        fx[f"amort_{tag}_x"] = x.numpy()
        fx[f"amort_{tag}_loc"] = q.new_parameters["loc"].detach().numpy()
        fx[f"amort_{tag}_log_scale"] = q.new_parameters["log_scale"].detach().numpy()
        fx[f"amort_{tag}_w"] = layer._edge_weight_sample.numpy()
        fx[f"amort_{tag}_out"] = out.detach().numpy()

    # ---- (b'') amortised + vi=True: the KL term against a learned Normal prior and every gradient of
    #      loss = <gout, out> + kl  (stag/layers.py:132-145; scripts/arxiv_rec/gcn/run.py:85, 148-155)
    torch.manual_seed(53)
    g = G("hub40")
    x = torch.randn(40, 16, requires_grad=True)
    q = AmortizedDistribution(16, 1, init_like=torch.distributions.Normal(1.0, 0.3))
    with torch.no_grad():           # the default heads are near-constant: make every weight matter
        for prm in q.parameters():
            prm.copy_(torch.randn_like(prm) * 0.5)
    layer = StagLayer(SumBase(), q_a=q, p_a=torch.distributions.Normal(0.8, 0.6), vi=True)
    for k, v in layer.state_dict().items():
        fx[f"amort_kl_sd_{k}"] = v.numpy().copy()
    torch.manual_seed(78)
    out = quiet(layer, g, x)
    kl = layer.kl_divergence()
    gout = torch.randn(40, 16)
    ((out * gout).sum() + kl).backward()
    fx["amort_kl_x"], fx["amort_kl_gout"] = x.detach().numpy(), gout.numpy()
    fx["amort_kl_w"] = layer._edge_weight_sample.detach().numpy()
    fx["amort_kl_out"] = out.detach().numpy()
    fx["amort_kl_value"] = np.array([kl.item()], np.float32)
    fx["amort_kl_loc"] = q.new_parameters["loc"].detach().numpy()
    fx["amort_kl_log_scale"] = q.new_parameters["log_scale"].detach().numpy()
    fx["amort_kl_grad_x"] = x.grad.numpy()
    for k, prm in layer.named_parameters():
        fx[f"amort_kl_grad_{k}"] = prm.grad.numpy().copy()

    # ---- (c) zoo layers with an explicit edge_weight ---------------------------
    torch.manual_seed(5)
    g = G("hub40")
    x = torch.randn(40, 16)
    w = torch.rand(g.number_of_edges(), 16) + 0.5
    gcn = zoo.GCN(16, 8)
    with torch.no_grad():
        gcn.bias.copy_(torch.linspace(-0.1, 0.1, 8))
    fx["zoo_x"], fx["zoo_w"] = x.numpy(), w.numpy()
    fx["gcn_weight"], fx["gcn_bias"] = gcn.weight.detach().numpy(), gcn.bias.detach().numpy()
    fx["gcn_out"] = gcn(g, x, edge_weight=w).detach().numpy()
    fx["gcn_out_noweight"] = gcn(g, x).detach().numpy()

    sage = zoo.GraphSAGE(16, 8, activation=torch.relu)
    with torch.no_grad():
        sage.bias.copy_(torch.linspace(-0.2, 0.2, 8))
    for k, v in sage.state_dict().items():
        fx[f"sage_sd_{k}"] = v.numpy()
    fx["sage_out"] = sage(g, x, edge_weight=w).detach().numpy()

    for last in (False, True):
        gat = zoo.GAT(16, 4, num_heads=3, last=last)
        with torch.no_grad():
            gat.bias.copy_(torch.linspace(-0.1, 0.1, 12))
        wh = torch.rand(g.number_of_edges(), 3) + 0.5
        tag = "gat_last" if last else "gat"
        for k, v in gat.state_dict().items():
            fx[f"{tag}_sd_{k}"] = v.numpy()
        fx[f"{tag}_w"] = wh.numpy()
        out, attn = gat(g, x, get_attention=True, edge_weight=wh)
        fx[f"{tag}_out"], fx[f"{tag}_attn"] = out.detach().numpy(), attn.detach().numpy()

    # ---- (c') GatedGCN (stag/zoo/gated_gcn.py:6-61): with edge weights it aggregates h itself,
    #      without them B(h); batch norm in training mode, residual on
    torch.manual_seed(17)
    gated = zoo.GatedGCN(16, 16, dropout=0.0, batch_norm=True, residual=True)
    for k, v in gated.state_dict().items():
        fx[f"gated_sd_{k}"] = v.numpy().copy()
    gated.train()
    fx["gated_out"] = quiet(gated, g, x, edge_weight=w).detach().numpy()
    gated2 = zoo.GatedGCN(16, 16, dropout=0.0, batch_norm=True, residual=True)
    gated2.load_state_dict({k[len("gated_sd_"):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("gated_sd_")})
    gated2.train()
    fx["gated_out_noweight"] = quiet(gated2, g, x).detach().numpy()
    gated3 = zoo.GatedGCN(16, 8, batch_norm=False, residual=True)      # residual silently off: 16 != 8
    for k, v in gated3.state_dict().items():
        fx[f"gated3_sd_{k}"] = v.numpy().copy()
    fx["gated3_out_noweight"] = quiet(gated3, g, x).detach().numpy()

    # ---- (d) StagModel.loss_terms on a 2-layer GCN stack ------------------------
    torch.manual_seed(9)
    l1 = StagLayer(zoo.GCN(16, 8, activation=torch.relu), q_a=torch.distributions.Normal(1.0, 0.4), vi=True)
    l2 = StagLayer(zoo.GCN(8, 4, activation=lambda t: torch.softmax(t, dim=-1)),
                   q_a=torch.distributions.Normal(1.0, 0.4), vi=True)
    model = StagModel(layers=torch.nn.ModuleList([l1, l2]), kl_scaling=0.5)
    y = torch.randint(0, 4, (40,))
    torch.manual_seed(123)
    nll, reg = quiet(model.loss_terms, g, x, y, n_samples=1)
    fx["model_y"] = y.numpy()
    fx["model_w1"] = l1._edge_weight_sample.detach().numpy()
    fx["model_w2"] = l2._edge_weight_sample.detach().numpy()
    for i, l in enumerate((l1, l2)):
        fx[f"model_l{i}_weight"] = l.base_layer.weight.detach().numpy()
        fx[f"model_l{i}_bias"] = l.base_layer.bias.detach().numpy()
    fx["model_nll_reg"] = np.array([nll.item(), reg.item()], np.float32)
    (nll + reg).backward()
    fx["model_l0_weight_grad"] = l1.base_layer.weight.grad.numpy()
    fx["model_l0_qa_loc_grad"] = l1.q_a.loc.grad.numpy()
    fx["model_l0_qa_log_scale_grad"] = l1.q_a.log_scale.grad.numpy()

    # ---- (e) autograd gradients of the zoo layers with explicit weights (round 3) ---------------
    #      loss = <gout, layer(g, x, edge_weight=w)>: d x, d w and every parameter gradient, by the reference's own
    #      forward under torch autograd (stag/zoo/gcn.py:58-116, graph_sage.py:44-119, gat.py:74-149).  Placed after
    #      every earlier draw so the older fixtures keep their values; the layers are rebuilt from the stored
    #      state dicts.
    torch.manual_seed(31)
    g = G("hub40")
    x0, w0 = torch.from_numpy(fx["zoo_x"]), torch.from_numpy(fx["zoo_w"])

    def grads_of(tag, layer, w_init, out_shape):
        x = x0.clone().requires_grad_(True)
        w = w_init.clone().requires_grad_(True)
        gout = torch.randn(*out_shape)
        out = layer(g, x, edge_weight=w)
        assert tuple(out.shape) == tuple(out_shape), (tag, out.shape)
        (out * gout).sum().backward()
        fx[f"{tag}_gout"] = gout.numpy()
        fx[f"{tag}_grad_x"], fx[f"{tag}_grad_w"] = x.grad.numpy().copy(), w.grad.numpy().copy()
        for k, prm in layer.named_parameters():
            fx[f"{tag}_grad_{k}"] = prm.grad.numpy().copy()

    gcn2 = zoo.GCN(16, 8)
    gcn2.load_state_dict({"weight": torch.from_numpy(fx["gcn_weight"]), "bias": torch.from_numpy(fx["gcn_bias"])})
    grads_of("gcn", gcn2, w0, (40, 8))
    sage2 = zoo.GraphSAGE(16, 8, activation=torch.relu)
    sage2.load_state_dict({k[len("sage_sd_"):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sage_sd_")})
    grads_of("sage", sage2, w0, (40, 8))
    for last in (False, True):
        tag = "gat_last" if last else "gat"
        gat2 = zoo.GAT(16, 4, num_heads=3, last=last)
        gat2.load_state_dict({k[len(tag + "_sd_"):]: torch.from_numpy(v) for k, v in fx.items()
                              if k.startswith(tag + "_sd_")})
        grads_of(tag, gat2, torch.from_numpy(fx[f"{tag}_w"]), (40, 4) if last else (40, 12))

    path = os.path.join(OUT, "stag_reference.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path}: {len(fx)} arrays, {os.path.getsize(path)} bytes; torch {torch.__version__}")


if __name__ == "__main__":
    main()
