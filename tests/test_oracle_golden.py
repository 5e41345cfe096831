"""The oracle against what pins it (CPU only):
  - Philox4x32-10 against the Random123 known-answer vectors;
  - the aggregation / relu / in-norm / GCN / SAGE / GAT restatement against fixtures
    produced by executing the reference's own source (tests/golden/make_golden.py);
  - the noise transforms against scipy's exact distributions.
"""
import numpy as np
import pytest
from scipy import stats

from util import assert_close


def _g(oracle, golden, name):
    src, dst, n = golden[f"{name}_src"], golden[f"{name}_dst"], int(golden[f"{name}_n"])
    indptr, indices, eid, ind, outd = oracle.csr_build(src, dst, n, n)
    return oracle.CsrGraph(indptr, indices, eid, n_src=n), src, dst, n, ind, outd


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert [int(v) for v in oracle.philox4x32_10(ctr, key)] == want


def test_philox_counter_layout(oracle):
    # ctr = {lo32(gpos), chunk | hi32(gpos) << 20, lo32(offset), hi32(offset)}, key = seed halves
    seed, offset, gpos, chunk = 0x1234567890ABCDEF, 0x0FEDCBA987654321, (5 << 32) | 77, 9
    got = oracle.philox_raw(seed, offset, gpos, 1, chunk + 1)[0, chunk]
    want = oracle.philox4x32_10([77, chunk | (5 << 20), offset & 0xFFFFFFFF, offset >> 32],
                                [seed & 0xFFFFFFFF, seed >> 32])
    assert np.array_equal(got, want)


def test_csr_build_is_stable(oracle):
    rng = np.random.default_rng(0)
    src, dst = rng.integers(0, 50, 2000), rng.integers(0, 50, 2000)
    indptr, indices, eid, ind, outd = oracle.csr_build(src, dst, 50, 50)
    order = np.argsort(dst, kind="stable")
    assert np.array_equal(eid, order) and np.array_equal(indices, src[order])
    assert np.array_equal(np.diff(indptr), np.bincount(dst, minlength=50))
    assert np.array_equal(ind, np.bincount(dst, minlength=50))
    assert np.array_equal(outd, np.bincount(src, minlength=50))
    # empty graph
    indptr, indices, eid, _, _ = oracle.csr_build(np.zeros(0, int), np.zeros(0, int), 4, 4)
    assert np.array_equal(indptr, np.zeros(5)) and len(indices) == 0


def test_layer_cases_explicit_weights(oracle, golden):
    """out = sum_in w (.) x with the reference's sampled w (stag/layers.py:96-113 over the
    aggregation lines of stag/zoo/gcn.py:61-63,94-96)."""
    for name in golden["layer_cases"]:
        name = str(name)
        gname = name.split("_")[1]
        g, *_ = _g(oracle, golden, gname)
        x, w, out = golden[name + "_x"], golden[name + "_w"], golden[name + "_out"]
        assert_close(oracle.agg_fwd(g, x, oracle.make_spec("explicit", w)), out, what=name)


def test_in_norm_matches_reference(oracle, golden):
    """Bernoulli + norm=True: the fixture holds w AFTER _in_norm (stag/layers.py:8-36); the raw
    draws are its support.  The oracle's in-norm of the raw draws must reproduce it."""
    seen = 0
    for name in golden["layer_cases"]:
        name = str(name)
        if not name.endswith("_bern"):
            continue
        g, src, dst, n, ind, _ = _g(oracle, golden, name.split("_")[1])
        w_final = golden[name + "_w"]
        w_raw = (w_final != 0).astype(np.float32)
        got = oracle.noise_materialize(g, oracle.make_spec("explicit", w_raw, in_norm=True), w_raw.shape[1])
        assert_close(got, w_final, what=name + " in-norm")
        # kept weights of every destination sum to its in-degree (where any survived)
        s = np.zeros((n, w_final.shape[1]))
        np.add.at(s, dst, got)
        alive = np.zeros_like(s, dtype=bool)
        np.logical_or.at(alive, dst, w_raw != 0)
        assert np.allclose(s[alive], np.broadcast_to(ind[:, None], s.shape)[alive], rtol=1e-5)
        # fused form: out = s * sum(w_raw x) equals the reference's layer output
        x, out = golden[name + "_x"], golden[name + "_out"]
        assert_close(oracle.agg_fwd(g, x, oracle.make_spec("explicit", w_raw, in_norm=True)), out,
                     what=name + " fused in-norm")
        seen += 1
    assert seen >= 6


def test_relu_cases(oracle, golden):
    for name in golden["layer_cases"]:
        name = str(name)
        if name.endswith("_relu"):
            w = golden[name + "_w"]
            assert w.min() >= 0.0 and (w == 0).any()
            g, *_ = _g(oracle, golden, name.split("_")[1])
            got = oracle.noise_materialize(g, oracle.make_spec("explicit", w, relu=True), w.shape[1])
            assert np.array_equal(got, w)   # relu is idempotent on the reference's output


def test_gcn_golden(oracle, golden):
    """GCN norm='both' (stag/zoo/gcn.py:67-75, 94-108) = fused src/dst degree scales."""
    g, src, dst, n, ind, outd = _g(oracle, golden, "hub40")
    x, w = golden["zoo_x"], golden["zoo_w"]
    ss = np.maximum(outd, 1).astype(np.float32) ** -0.5
    ds = np.maximum(ind, 1).astype(np.float32) ** -0.5
    agg = oracle.agg_fwd(g, x, oracle.make_spec("explicit", w), src_scale=ss, dst_scale=ds)
    assert_close(agg @ golden["gcn_weight"] + golden["gcn_bias"], golden["gcn_out"], what="gcn")
    agg = oracle.agg_fwd(g, x, oracle.make_spec("none"), src_scale=ss, dst_scale=ds)
    assert_close(agg @ golden["gcn_weight"] + golden["gcn_bias"], golden["gcn_out_noweight"], what="gcn no weight")


def test_sage_golden(oracle, golden):
    """GraphSAGE mean path (stag/zoo/graph_sage.py:70-75, 107-118)."""
    g, *_ = _g(oracle, golden, "hub40")
    x, w = golden["zoo_x"], golden["zoo_w"]
    neigh = oracle.agg_fwd(g, x, oracle.make_spec("explicit", w), reduce=oracle.REDUCE_MEAN)
    rst = x @ golden["sage_sd_fc_self.weight"].T + neigh @ golden["sage_sd_fc_neigh.weight"].T + golden["sage_sd_bias"]
    assert_close(np.maximum(rst, 0), golden["sage_out"], what="sage")


def test_gated_gcn_golden(oracle, golden):
    """GatedGCN (stag/zoo/gated_gcn.py:25-55): with edge weights the layer aggregates h itself
    (`u_mul_e('h', w)`), without them B(h); then A(h) + sum, batch norm with batch statistics,
    relu, residual."""
    g, *_ = _g(oracle, golden, "hub40")
    x, w = golden["zoo_x"].astype(np.float64), golden["zoo_w"]

    def finish(h, sd, bn=True, residual=True):
        if bn:
            mu, var = h.mean(0), h.var(0)            # training mode: biased batch statistics
            h = (h - mu) / np.sqrt(var + 1e-5) * golden[sd + "bn_node_h.weight"] + golden[sd + "bn_node_h.bias"]
        h = np.maximum(h, 0)
        return x + h if residual else h
    A = lambda sd: x @ golden[sd + "A.weight"].T.astype(np.float64) + golden[sd + "A.bias"]
    B = lambda sd: x @ golden[sd + "B.weight"].T.astype(np.float64) + golden[sd + "B.bias"]
    agg = oracle.agg_fwd(g, x.astype(np.float32), oracle.make_spec("explicit", w))
    assert_close(finish(A("gated_sd_") + agg, "gated_sd_"), golden["gated_out"], what="gated, edge weights")
    agg = oracle.agg_fwd(g, B("gated_sd_").astype(np.float32), oracle.make_spec("none"))
    assert_close(finish(A("gated_sd_") + agg, "gated_sd_"), golden["gated_out_noweight"], what="gated, no weights")
    agg = oracle.agg_fwd(g, B("gated3_sd_").astype(np.float32), oracle.make_spec("none"))
    assert_close(finish(A("gated3_sd_") + agg, "gated3_sd_", bn=False, residual=False), golden["gated3_out_noweight"],
                 what="gated 16->8: no batch norm, residual silently off")


@pytest.mark.parametrize("tag", ["gat", "gat_last"])
def test_gat_golden(oracle, golden, tag):
    """GAT with per-head weights scaling the logits before the softmax (stag/zoo/gat.py:93-141)."""
    g, *_ = _g(oracle, golden, "hub40")
    x, wh = golden["zoo_x"], golden[f"{tag}_w"]
    H, F = 3, 4
    ft = (x @ golden[f"{tag}_sd_fc.weight"].T).reshape(-1, H, F)
    el = (ft * golden[f"{tag}_sd_attn_l"]).sum(-1)
    er = (ft * golden[f"{tag}_sd_attn_r"]).sum(-1)
    out, attn = oracle.gat_fwd(g, el, er, ft, 0.2, oracle.make_spec("explicit", wh), want_attn=True)
    out = out + golden[f"{tag}_sd_bias"].reshape(1, H, F)
    out = out.mean(-2) if tag == "gat_last" else out.reshape(out.shape[0], -1)
    assert_close(out, golden[f"{tag}_out"], what=tag)
    assert_close(attn, golden[f"{tag}_attn"].reshape(-1, H), what=tag + " attention")


def test_amortized_golden(oracle, golden):
    """AmortizedDistribution.condition (stag/distributions.py:221-233) then the layer."""
    g, src, dst, *_ = _g(oracle, golden, "hub40")
    for tag in ("re", "rec"):
        x = golden[f"amort_{tag}_x"]
        sd = {k[len(f"amort_{tag}_sd_"):]: golden[k] for k in golden.files if k.startswith(f"amort_{tag}_sd_")}
        h = np.concatenate([x[src], x[dst]], -1) @ sd["embedding_mlp.0.weight"].T + sd["embedding_mlp.0.bias"]
        h = h / (1.0 + np.exp(-h))   # SiLU
        loc = h @ sd["parameters_mlp.loc.weight"].T + sd["parameters_mlp.loc.bias"]
        log_scale = h @ sd["parameters_mlp.log_scale.weight"].T + sd["parameters_mlp.log_scale.bias"]
        assert_close(loc, golden[f"amort_{tag}_loc"], what="amortised loc")
        assert_close(log_scale, golden[f"amort_{tag}_log_scale"], what="amortised log_scale")
        assert_close(oracle.agg_fwd(g, x, oracle.make_spec("explicit", golden[f"amort_{tag}_w"])),
                     golden[f"amort_{tag}_out"], what=f"amortised {tag} out")


def test_noise_distributions(oracle):
    """Fused draws against the exact distributions (KS) — the counterpart of
    `q_a.expand([E, Dn]).sample()` (stag/layers.py:117-127)."""
    E, D = 40000, 8
    g = oracle.CsrGraph(np.arange(0, E + 1, 50, dtype=np.int32), np.zeros(E, np.int32), n_src=1)
    w = oracle.noise_materialize(g, oracle.make_spec("normal", 1.0, 0.5, seed=3, Dn=D, n_edges=E), D).ravel()
    assert stats.kstest(w, "norm", args=(1.0, 0.5)).pvalue > 1e-3
    assert abs(w.mean() - 1.0) < 4e-3 and abs(w.std() - 0.5) < 4e-3
    w = oracle.noise_materialize(g, oracle.make_spec("uniform", 0.25, 1.75, seed=4, Dn=D, n_edges=E), D).ravel()
    assert stats.kstest(w, "uniform", args=(0.25, 1.5)).pvalue > 1e-3
    assert w.min() >= 0.25 and w.max() < 1.75
    w = oracle.noise_materialize(g, oracle.make_spec("bernoulli", 0.9, seed=5, Dn=D, n_edges=E), D).ravel()
    assert set(np.unique(w)) == {0.0, 1.0} and abs(w.mean() - 0.9) < 3e-3
    # independence across channels / edges / offsets: correlations vanish
    a = oracle.noise_materialize(g, oracle.make_spec("normal", 0.0, 1.0, seed=3, offset=0, Dn=D, n_edges=E), D)
    b = oracle.noise_materialize(g, oracle.make_spec("normal", 0.0, 1.0, seed=3, offset=1, Dn=D, n_edges=E), D)
    assert abs(np.corrcoef(a[:, 0], a[:, 1])[0, 1]) < 0.02      # Box-Muller pair
    assert abs(np.corrcoef(a[:, 0], a[:, 2])[0, 1]) < 0.02
    assert abs(np.corrcoef(a[:-1, 0], a[1:, 0])[0, 1]) < 0.02   # neighbouring edges
    assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 0.01  # next offset
    # per-channel and per-edge parameters follow the broadcasting of q_a.expand([E, Dn])
    loc = np.linspace(-1, 1, D).astype(np.float32)
    w = oracle.noise_materialize(g, oracle.make_spec("normal", loc, np.full(D, 0.1, np.float32), seed=6, Dn=D, n_edges=E), D)
    assert np.allclose(w.mean(0), loc, atol=5e-3)
    pe = np.linspace(0, 5, E).astype(np.float32)[:, None]
    w = oracle.noise_materialize(g, oracle.make_spec("normal", pe, np.full((E, 1), 1e-3, np.float32), seed=7, Dn=D, n_edges=E), D)
    assert np.allclose(w, np.broadcast_to(pe, (E, D)), atol=1e-2)


def test_transposed_oracle_is_the_gradient(oracle):
    """d(out)/d(x) applied to g == aggregation over the source-major CSR with nidx."""
    rng = np.random.default_rng(1)
    n, E, D = 30, 200, 6
    src, dst = rng.integers(0, n, E), rng.integers(0, n, E)
    indptr, indices, eid, *_ = oracle.csr_build(src, dst, n, n)
    g = oracle.CsrGraph(indptr, indices, eid, n_src=n)
    gt = g.transpose()
    spec = oracle.make_spec("normal", 1.0, 0.5, seed=9, offset=2, Dn=D, n_edges=E)
    w = oracle.noise_materialize(g, spec, D)
    gout = rng.standard_normal((n, D)).astype(np.float32)
    ref = np.zeros((n, D))
    np.add.at(ref, src, w.astype(np.float64) * gout[dst])
    assert_close(oracle.agg_fwd(gt, gout, spec), ref, what="transposed")


def test_segment_reduce_oracle(oracle):
    x = np.arange(24, dtype=np.float32).reshape(8, 3)
    offs = np.array([0, 3, 3, 8], np.int32)
    assert np.allclose(oracle.segment_reduce(x, offs), [x[:3].sum(0), np.zeros(3), x[3:].sum(0)])
    assert np.allclose(oracle.segment_reduce(x, offs, oracle.REDUCE_MEAN), [x[:3].mean(0), np.zeros(3), x[3:].mean(0)])


# ---- gradients: the oracle's backward twins against the reference's autograd (round 3) -----------------------
# tests/golden/make_golden.py (e): loss = <gout, layer(g, x, edge_weight=w)> through the reference's own forward;
# the fixtures hold d x, d w and every parameter gradient.
def _relerr(got, ref):
    """gradients are compared relative to their own scale (a weight gradient sums 40 rows)"""
    ref = np.asarray(ref, np.float64)
    sc = max(1.0, float(np.abs(ref).max()))
    return np.asarray(got, np.float64) / sc, ref / sc


def test_gcn_gradients_golden(oracle, golden):
    """d/dx = the transposed aggregation, d/dw = agg_bwd_w, around GCN's dense transform (stag/zoo/gcn.py:67-108)."""
    g, src, dst, n, ind, outd = _g(oracle, golden, "hub40")
    x, w, gout = golden["zoo_x"], golden["zoo_w"], golden["gcn_gout"].astype(np.float64)
    W = golden["gcn_weight"].astype(np.float64)
    ss = np.maximum(outd, 1).astype(np.float32) ** -0.5
    ds = np.maximum(ind, 1).astype(np.float32) ** -0.5
    spec = oracle.make_spec("explicit", w)
    agg = oracle.agg_fwd(g, x, spec, src_scale=ss, dst_scale=ds).astype(np.float64)
    assert_close(*_relerr(agg.T @ gout, golden["gcn_grad_weight"]), what="gcn d weight")
    assert_close(*_relerr(gout.sum(0), golden["gcn_grad_bias"]), what="gcn d bias")
    g_agg = (gout @ W.T).astype(np.float32)                     # d L / d agg
    dx = oracle.agg_fwd(g.transpose(), g_agg, spec, src_scale=ds, dst_scale=ss)
    assert_close(*_relerr(dx, golden["gcn_grad_x"]), what="gcn d x")
    dw = oracle.agg_bwd_w(g, x, g_agg * ds[:, None], src_scale=ss)
    assert_close(*_relerr(dw, golden["gcn_grad_w"]), what="gcn d w")


def test_sage_gradients_golden(oracle, golden):
    """GraphSAGE mean path under autograd (stag/zoo/graph_sage.py:70-75, 107-118), relu activation."""
    g, src, dst, n, ind, outd = _g(oracle, golden, "hub40")
    x, w = golden["zoo_x"], golden["zoo_w"]
    Ws, Wn = golden["sage_sd_fc_self.weight"].astype(np.float64), golden["sage_sd_fc_neigh.weight"].astype(np.float64)
    spec = oracle.make_spec("explicit", w)
    neigh = oracle.agg_fwd(g, x, spec, reduce=oracle.REDUCE_MEAN).astype(np.float64)
    pre = x @ Ws.T + neigh @ Wn.T + golden["sage_sd_bias"]
    gp = golden["sage_gout"].astype(np.float64) * (pre > 0)
    assert_close(*_relerr(gp.sum(0), golden["sage_grad_bias"]), what="sage d bias")
    assert_close(*_relerr(gp.T @ x, golden["sage_grad_fc_self.weight"]), what="sage d fc_self")
    assert_close(*_relerr(gp.T @ neigh, golden["sage_grad_fc_neigh.weight"]), what="sage d fc_neigh")
    g_neigh = (gp @ Wn).astype(np.float32)
    inv = (1.0 / np.maximum(ind, 1)).astype(np.float32)
    dx = oracle.agg_fwd(g.transpose(), g_neigh, spec, src_scale=inv).astype(np.float64) + gp @ Ws
    assert_close(*_relerr(dx, golden["sage_grad_x"]), what="sage d x")
    dw = oracle.agg_bwd_w(g, x, g_neigh * inv[:, None])
    assert_close(*_relerr(dw, golden["sage_grad_w"]), what="sage d w")


@pytest.mark.parametrize("tag", ["gat", "gat_last"])
def test_gat_gradients_golden(oracle, golden, tag):
    """oracle.gat_bwd (stag_gat_bwd_cpu) inside the layer's chain rule against the reference's autograd through
    stag/zoo/gat.py:93-141: d x, d edge_weight, d fc.weight, d attn_l, d attn_r, d bias."""
    g, *_ = _g(oracle, golden, "hub40")
    x, wh = golden["zoo_x"].astype(np.float64), golden[f"{tag}_w"]
    H, F = 3, 4
    Wfc = golden[f"{tag}_sd_fc.weight"].astype(np.float64)
    al, ar = golden[f"{tag}_sd_attn_l"].astype(np.float64).reshape(1, H, F), golden[f"{tag}_sd_attn_r"].astype(np.float64).reshape(1, H, F)
    ft = (x @ Wfc.T).reshape(-1, H, F)
    el, er = (ft * al).sum(-1), (ft * ar).sum(-1)
    gout = golden[f"{tag}_gout"].astype(np.float64)
    # rst + bias, then mean over the heads (last) or flatten (stag/zoo/gat.py:133-141)
    G = np.repeat(gout[:, None, :], H, 1) / H if tag == "gat_last" else gout.reshape(-1, H, F)
    d_el, d_er, d_ft, dw = oracle.gat_bwd(g, el, er, ft, G, 0.2, oracle.make_spec("explicit", wh), want_dw=True)
    assert_close(*_relerr(G.sum(0).reshape(-1), golden[f"{tag}_grad_bias"]), what=tag + " d bias")
    assert_close(*_relerr(dw, golden[f"{tag}_grad_w"]), what=tag + " d w")
    assert_close(*_relerr((d_el[:, :, None] * ft).sum(0), golden[f"{tag}_grad_attn_l"].reshape(H, F)), what=tag + " d attn_l")
    assert_close(*_relerr((d_er[:, :, None] * ft).sum(0), golden[f"{tag}_grad_attn_r"].reshape(H, F)), what=tag + " d attn_r")
    d_ft_all = (d_ft.astype(np.float64) + d_el[:, :, None] * al + d_er[:, :, None] * ar).reshape(-1, H * F)
    assert_close(*_relerr(d_ft_all.T @ x, golden[f"{tag}_grad_fc.weight"]), what=tag + " d fc.weight")
    assert_close(*_relerr(d_ft_all @ Wfc, golden[f"{tag}_grad_x"]), what=tag + " d x")


def test_gat_bwd_oracle_equals_float64_autograd(oracle):
    """stag_gat_bwd_cpu with every option the device kernels take — drawn weights, relu, in-norm, the attention-dropout
    mask — against torch autograd in float64 through the plain statement of stag/zoo/gat.py:114-126."""
    import torch
    rng = np.random.default_rng(11)
    n, E, H, F = 50, 600, 4, 8
    src = rng.integers(0, n, E)
    dst = np.concatenate([rng.integers(0, n - 1, E - 150), np.full(150, 3)])
    indptr, indices, eid, *_ = oracle.csr_build(src, dst, n, n)
    g = oracle.CsrGraph(indptr, indices, eid, n_src=n)
    el, er = rng.standard_normal((n, H)).astype(np.float32), rng.standard_normal((n, H)).astype(np.float32)
    ft, G = rng.standard_normal((n, H, F)).astype(np.float32), rng.standard_normal((n, H, F)).astype(np.float32)
    keep = (rng.random((E, H)) < 0.6).astype(np.float32)
    wx = rng.uniform(-0.5, 1.5, (E, H)).astype(np.float32)
    cases = [("none", oracle.make_spec("none"), None, False),
             ("explicit+relu", oracle.make_spec("explicit", wx, relu=True), None, True),
             ("normal", oracle.make_spec("normal", 1.0, 0.4, seed=5, offset=2, Dn=H, n_edges=E), keep, False),
             ("bernoulli+norm", oracle.make_spec("bernoulli", 0.7, None, in_norm=True, seed=5, offset=3, Dn=H, n_edges=E), keep, False)]
    S, Dt = torch.from_numpy(src).long(), torch.from_numpy(dst).long()
    for name, spec, kp, want_dw in cases:
        w_eff = oracle.noise_materialize(g, spec, H) if spec.kind != oracle.NOISE_NONE else np.ones((E, H), np.float32)
        tl, tr, tf = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (el, er, ft))
        tw = torch.tensor(wx if want_dw else w_eff, dtype=torch.float64, requires_grad=want_dw)
        e = (tw.relu() if want_dw else tw) * torch.nn.functional.leaky_relu(tl[S] + tr[Dt], 0.2)
        mx = torch.full((n, H), -float("inf"), dtype=torch.float64).scatter_reduce(0, Dt[:, None].expand(-1, H), e.detach(), "amax")
        ex = torch.exp(e - mx[Dt])
        a = ex / torch.zeros((n, H), dtype=torch.float64).index_add_(0, Dt, ex)[Dt]
        if kp is not None:
            a = a * torch.tensor(kp, dtype=torch.float64) / 0.6
        out = torch.zeros((n, H, F), dtype=torch.float64).index_add_(0, Dt, a[:, :, None] * tf[S])
        out.backward(torch.tensor(G, dtype=torch.float64))
        fwd = oracle.gat_fwd(g, el, er, ft, 0.2, spec, keep=kp, keep_prob=0.6)
        assert_close(fwd, out.detach().numpy(), what=name + " forward")
        d_el, d_er, d_ft, dw = oracle.gat_bwd(g, el, er, ft, G, 0.2, spec, keep=kp, keep_prob=0.6, want_dw=want_dw)
        assert_close(d_el, tl.grad.numpy(), what=name + " d el")
        assert_close(d_er, tr.grad.numpy(), what=name + " d er")
        assert_close(d_ft, tf.grad.numpy(), what=name + " d ft")
        if want_dw:
            assert_close(dw, tw.grad.numpy(), what=name + " d w")
