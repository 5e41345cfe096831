"""HIP path vs CPU oracle on the same seeded inputs (run on the GPU box: -m gpu).

Bit-exact: CSR arrays, raw Philox words, Uniform and Bernoulli draws.
Within |a-b| <= 1e-5 (1+|b|): Normal draws (hardware log/sin/cos vs libm) and every
aggregated feature.
"""
import os

import numpy as np
import pytest
import torch

from util import TOL, assert_close, assert_close_cond, assert_gat_grads_vs_oracle, hw_normals, oracle_graph, random_graph, scaled_err

pytestmark = pytest.mark.gpu
# the seeded sweeps run FUZZ_SCALE times their committed number of cases (the first ones are the same cases):
# STAG_FUZZ_SCALE=10 is the soak run whose result is kept in profiles/r02/fuzz_soak.txt
FUZZ_SCALE = max(1, int(os.environ.get("STAG_FUZZ_SCALE", "1")))


def _noise(g, dn, kind, p0, p1=None, **kw):
    import stag_amd
    from stag_amd import _lib
    k = {"normal": _lib.NOISE_NORMAL, "uniform": _lib.NOISE_UNIFORM, "bernoulli": _lib.NOISE_BERNOULLI}[kind]
    return stag_amd.EdgeNoise(g, dn, k, p0, p1, **kw)


def _np(p):
    return p.detach().cpu().numpy() if torch.is_tensor(p) else p


def _ospec(O, g, dn, kind, p0, p1=None, **kw):
    return O.make_spec(kind, _np(p0), _np(p1), Dn=dn, n_edges=g.number_of_edges(), **kw)


def test_library_loaded_is_in_tree():
    from stag_amd import _lib
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "stag_amd/libstag_hip.so" in maps


def test_torch_ops_equal_ctypes_binding(dev, monkeypatch):
    """The dispatcher ops (torch.ops.stag.agg_fwd / agg_bwd, csrc/torch_ext.cpp) and the ctypes binding are two
    front ends of the same library calls: outputs and gradients are bit-identical, plan or no plan, every noise
    kind, explicit weights, in-norm, vi gradients."""
    import stag_amd
    from stag_amd import _lib, _torch_ext, ops
    monkeypatch.setenv("STAG_TORCH_OPS", "1")
    # the same library calls on both sides: the dispatcher has agg_fwd / agg_bwd, so the ctypes side takes the
    # two-step parameter gradient too (stag_agg_bwd + stag_coldot), not stag_agg_bwd_dp
    monkeypatch.setattr(ops, "_AGG_BWD_DP_ONE_PASS", False)
    assert _torch_ext.available(), "the torch front end must be built and loaded on the GPU box"
    g = random_graph(400, 5000, seed=2, hub=900, device=dev)
    E, D = g.number_of_edges(), 48
    x = torch.randn(400, D, device=dev)
    gout = torch.randn(400, D, device=dev)
    loc = torch.rand(D, device=dev) + 0.5
    w = torch.rand(E, D, device=dev)

    def run(kind):
        xg = x.clone().requires_grad_(True)
        extra = []
        if kind == "explicit":
            wg = w.clone().requires_grad_(True); extra = [wg]
            out = ops.aggregate(g, xg, wg, reduce="mean")
        elif kind == "none":
            out = ops.aggregate(g, xg, None, src_scale=loc[:1].expand(400).contiguous())
        elif kind == "vi":
            a = loc.clone().requires_grad_(True); b = (loc * 0.3).requires_grad_(True); extra = [a, b]
            out = ops.aggregate(g, xg, stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, a, b, relu=True, in_norm=True, seed=4,
                                                          offset=1, differentiable=True), seg_len=32)
        else:
            k = {"normal": _lib.NOISE_NORMAL, "uniform": _lib.NOISE_UNIFORM, "bernoulli": _lib.NOISE_BERNOULLI}[kind]
            out = ops.aggregate(g, xg, stag_amd.EdgeNoise(g, D, k, 0.6 if kind == "bernoulli" else loc,
                                                          None if kind == "bernoulli" else loc + 1.0, seed=4, offset=1,
                                                          in_norm=(kind == "bernoulli")), seg_len=0 if kind == "uniform" else 64)
        out.backward(gout)
        return [out.detach(), xg.grad] + [t.grad for t in extra]

    for kind in ("none", "explicit", "normal", "uniform", "bernoulli", "vi"):
        monkeypatch.setenv("STAG_TORCH_OPS", "1")
        via_ops = run(kind)
        monkeypatch.delenv("STAG_TORCH_OPS")
        assert not _torch_ext.available()
        via_ctypes = run(kind)
        for a_, b_ in zip(via_ops, via_ctypes):
            assert torch.equal(a_, b_), kind


def test_torch_ops_of_round_4_equal_ctypes_binding(dev, monkeypatch):
    """stag::agg_fwd_mc, stag::agg_bwd_dp, stag::gat_fwd, stag::gat_bwd (csrc/torch_ext.cpp, round 4) against the ctypes
    binding of the same library calls: bit-identical outputs and gradients — Monte-Carlo batches (2, 3 and 4 samples, with
    and without in-norm), `vi=True` parameter gradients finished in the dx pass, the GAT layer step with noise, with explicit
    weights (dw), with in-norm and with attention dropout inside the kernels."""
    import stag_amd
    from stag_amd import _lib, _torch_ext, ops
    g = random_graph(500, 6000, seed=6, hub=1100, device=dev)
    E, D, H, F = g.number_of_edges(), 40, 4, 8
    n = g.number_of_nodes()
    x, gout = torch.randn(n, D, device=dev), torch.randn(n, D, device=dev)
    loc = torch.rand(D, device=dev) + 0.5
    el0, er0, ft0, G0 = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, F, device=dev), torch.randn(n, H, F, device=dev)
    wE = torch.rand(E, H, device=dev) + 0.5

    def run(kind):
        if kind.startswith("mc"):
            S = int(kind[2])
            nz = stag_amd.EdgeNoise(g, D, _lib.NOISE_BERNOULLI if "norm" in kind else _lib.NOISE_NORMAL,
                                    0.7 if "norm" in kind else 1.0, None if "norm" in kind else 0.4, seed=9, offset=2,
                                    in_norm="norm" in kind)
            with torch.no_grad():
                return [ops.aggregate_mc(g, x, nz, S, offset_stride=3, reduce="mean")]
        if kind == "vi_dp":
            xg = x.clone().requires_grad_(True)
            a = loc.clone().requires_grad_(True); b = (loc * 0.3).requires_grad_(True)
            out = ops.aggregate(g, xg, stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, a, b, relu=True, seed=4, offset=1,
                                                          differentiable=True))
            out.backward(gout)
            return [out.detach(), xg.grad, a.grad, b.grad]
        el, er, ft = (t.clone().requires_grad_(True) for t in (el0, er0, ft0))
        extra, drop, weight = [], None, None
        if kind == "gat_noise":
            weight = stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.3, seed=3, offset=1)
        elif kind == "gat_norm_drop":
            weight = stag_amd.EdgeNoise(g, H, _lib.NOISE_BERNOULLI, 0.8, None, seed=3, offset=1, in_norm=True)
            drop = (0.4, 11, 5)
        elif kind == "gat_w":
            weight = wE.clone().requires_grad_(True); extra = [weight]
        out = ops.gat_aggregate(g, el, er, ft, 0.2, weight, attn_drop=drop)
        out.backward(G0)
        return [out.detach(), el.grad, er.grad, ft.grad] + [t.grad for t in extra]

    for kind in ("mc2", "mc3", "mc4", "mc2norm", "vi_dp", "gat_none", "gat_noise", "gat_norm_drop", "gat_w"):
        monkeypatch.setenv("STAG_TORCH_OPS", "1")
        assert _torch_ext.available()
        via_ops = run(kind)
        monkeypatch.delenv("STAG_TORCH_OPS")
        assert not _torch_ext.available()
        via_ctypes = run(kind)
        assert len(via_ops) == len(via_ctypes)
        for a_, b_ in zip(via_ops, via_ctypes):
            assert torch.equal(a_, b_), kind


def test_dispatcher_ops_compile_with_fullgraph(dev):
    """The ops are ordinary custom ops to a compiled graph: `torch.compile(fullgraph=True)` of a function that calls
    torch.ops.stag.gat_fwd / agg_fwd between torch ops traces through their Meta kernels without a graph break and
    returns what eager returns.  (What is compiled is the op level.  A whole StagLayer step is not offered under
    torch.compile: every call takes a fresh Philox offset from host state, which a traced graph would freeze — the
    supported whole-step form is hipGraph capture with the device epoch)"""
    import stag_amd
    from stag_amd import _lib, _torch_ext, ops
    assert _torch_ext.loaded()
    g = random_graph(300, 4000, seed=8, hub=700, device=dev)
    n, H, F, D = g.number_of_nodes(), 4, 8, 32
    csrv = g.csr
    plan = csrv.plan(64, need=True)
    el, er, ft = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, F, device=dev)
    x = torch.randn(n, D, device=dev)
    gat_plan = ops._gat_plan_args(csrv, plan, dev, H * F)
    agg_plan = ops._plan_args(csrv, plan, 1, dev, width=D)
    nz_h = stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.3, seed=3, offset=1).torch_args()
    nz_d = stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, 1.0, 0.3, seed=3, offset=2).torch_args()

    def f(el, er, ft, x):
        out, stats = torch.ops.stag.gat_fwd(*csrv.torch_args(), *gat_plan, el * 2.0, er, ft, 0.2, *nz_h, None, [], [], None, True)
        agg, _ = torch.ops.stag.agg_fwd(*csrv.torch_args(), *agg_plan, x + 1.0, False, *nz_d, 0, None, None, False)
        return out, torch.relu(out).sum(-1) + stats[:, :H], agg * 0.5

    want = f(el, er, ft, x)
    got = torch.compile(f, fullgraph=True, backend="aot_eager")(el, er, ft, x)
    for a_, b_ in zip(got, want):
        assert torch.equal(a_, b_)
    with torch.no_grad():       # ... and the op is the library call the Python layer makes
        ref = ops.gat_aggregate(g, el * 2.0, er, ft, 0.2, stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.3, seed=3, offset=1))
        ref2 = ops.aggregate(g, x + 1.0, stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, 1.0, 0.3, seed=3, offset=2))
    assert torch.equal(got[0], ref) and torch.equal(got[2], ref2 * 0.5)


def test_philox_words_bit_exact(dev, oracle):
    from stag_amd import ops
    for seed, offset, pos0 in [(0, 0, 0), (0x5747A6, 3, 12345), (2**63 + 5, 2**40 + 9, 2**33 + 17)]:
        got = ops.philox_raw(seed, offset, pos0, 257, 5, dev).cpu().numpy().view(np.uint32)
        ref = oracle.philox_raw(seed, offset, pos0, 257, 5)
        assert np.array_equal(got, ref)


def test_normal_tables_exhaustive(dev):
    """The hardware functions a Normal draw is made of (v_log/v_sqrt, v_cos, v_sin) over ALL 2^23 inputs
    against fp64: the approximation the noise stream is defined with (include/stag_hip.h) is pinned
    entry by entry, not by sampling."""
    from stag_amd import ops
    t = ops.normal_tables(dev).cpu().numpy().astype(np.float64)
    m = np.arange(1 << 23, dtype=np.uint32)
    f12 = ((m & 0x7FFFFF) | 0x3F800000).view(np.float32).astype(np.float64)
    rad = np.sqrt(-2.0 * np.log(2.0 - f12))
    ang = 2.0 * np.pi * (f12 - 1.0)
    rel = np.abs(t[0] - rad)[1:] / rad[1:]                # m = 0: u1 = 1, radius exactly 0
    assert t[0][0] == 0.0
    e_c, e_s = np.abs(t[1] - np.cos(ang)), np.abs(t[2] - np.sin(ang))
    print(f"radius: max rel {rel.max():.3e} rms {np.sqrt((rel ** 2).mean()):.3e}; "
          f"cos: max abs {e_c.max():.3e} rms {np.sqrt((e_c ** 2).mean()):.3e}; "
          f"sin: max abs {e_s.max():.3e} rms {np.sqrt((e_s ** 2).mean()):.3e}")
    assert rel.max() <= 2e-7 and e_c.max() <= 2e-7 and e_s.max() <= 2e-7     # measured: 1.37e-7, 1.25e-7, 1.25e-7
    assert np.all(np.abs(t[1]) <= 1.0) and np.all(np.abs(t[2]) <= 1.0) and np.all(t[0] >= 0.0)


def test_normal_draws_bit_exact_from_tables(dev, oracle):
    """With the device's tables loaded, the oracle's normals ARE the kernel's normals: [E, Dn] fields
    (scalar / per-channel / per-edge parameters, relu) compare equal, not close."""
    from util import hw_normals
    g = random_graph(200, 3000, seed=9, hub=400, device=dev)
    E, dn = g.number_of_edges(), 12
    rng = np.random.default_rng(4)
    cases = [(1.0, 0.5, False), (0.2, 1.0, True),
             (torch.tensor(rng.uniform(0.5, 1.5, dn).astype(np.float32), device=dev),
              torch.tensor(rng.uniform(0.1, 1.0, dn).astype(np.float32), device=dev), False),
             (torch.tensor(rng.uniform(0.5, 1.5, (E, dn)).astype(np.float32), device=dev),
              torch.tensor(rng.uniform(0.1, 1.0, (E, dn)).astype(np.float32), device=dev), True)]
    og = oracle_graph(oracle, g)
    with hw_normals(oracle, dev):
        for p0, p1, relu in cases:
            got = _noise(g, dn, "normal", p0, p1, relu=relu, seed=77, offset=6).materialize().cpu().numpy()
            ref = oracle.noise_materialize(og, _ospec(oracle, g, dn, "normal", p0, p1, relu=relu, seed=77, offset=6), dn)
            assert np.array_equal(got, ref)
    # and without them the two differ by the hardware's approximation error only
    ref = oracle.noise_materialize(og, _ospec(oracle, g, dn, "normal", 1.0, 0.5, seed=77, offset=6), dn)
    got = _noise(g, dn, "normal", 1.0, 0.5, seed=77, offset=6).materialize().cpu().numpy()
    assert 0 < np.abs(got - ref).max() <= 1e-6


def test_csr_bit_exact(dev, oracle):
    g = random_graph(300, 4000, seed=3, hub=500, device=dev)
    src, dst = (t.cpu().numpy() for t in g.edges())
    indptr, indices, eid, ind, outd = oracle.csr_build(src, dst, 300, 300)
    assert np.array_equal(g.csr.indptr.cpu().numpy(), indptr)
    assert np.array_equal(g.csr.indices.cpu().numpy(), indices)
    assert np.array_equal(g.csr.eid.cpu().numpy(), eid)
    assert np.array_equal(g.in_degrees().cpu().numpy(), ind)
    assert np.array_equal(g.out_degrees().cpu().numpy(), outd)
    # transposed twin: nidx maps back to the forward position of the same edge
    t = g.csr_t
    assert np.array_equal(g.csr.eid.cpu().numpy()[t.nidx.cpu().numpy()], t.eid.cpu().numpy())


@pytest.mark.parametrize("dn", [1, 5, 16, 128, 130])
def test_uniform_and_bernoulli_draws_bit_exact(dev, oracle, dn):
    g = random_graph(64, 700, seed=dn, hub=90, device=dev)
    og = oracle_graph(oracle, g)
    for kind, p0, p1 in [("uniform", 0.25, 1.75), ("bernoulli", 0.7, None)]:
        for relu, norm in [(False, False), (True, kind == "bernoulli")]:
            kw = dict(relu=relu, in_norm=norm, seed=99, offset=4)
            got = _noise(g, dn, kind, p0, p1, **kw).materialize().cpu().numpy()
            ref = oracle.noise_materialize(og, _ospec(oracle, g, dn, kind, p0, p1, **kw), dn)
            if norm:   # division by a sum: same values up to the order of the sum
                assert_close(got, ref, what=f"{kind} in_norm dn={dn}")
            else:
                assert np.array_equal(got, ref), f"{kind} dn={dn}"


@pytest.mark.parametrize("dn", [4, 7, 128])
def test_normal_draws(dev, oracle, dn):
    g = random_graph(100, 3000, seed=dn + 1, device=dev)
    og = oracle_graph(oracle, g)
    got = _noise(g, dn, "normal", 1.0, 0.5, seed=5, offset=1).materialize().cpu().numpy()
    ref = oracle.noise_materialize(og, _ospec(oracle, g, dn, "normal", 1.0, 0.5, seed=5, offset=1), dn)
    assert np.abs(got - ref).max() <= 2e-6, np.abs(got - ref).max()


PARAM_CASES = ["scalar", "per_channel", "per_edge1", "per_edge"]


def _params(mode, E, D, rng):
    if mode == "scalar":
        return 1.0, 0.5
    if mode == "per_channel":
        return (torch.from_numpy(rng.uniform(0.5, 1.5, D).astype(np.float32)),
                torch.from_numpy(rng.uniform(0.1, 1.0, D).astype(np.float32)))
    if mode == "per_edge1":
        return (torch.from_numpy(rng.uniform(0.5, 1.5, (E, 1)).astype(np.float32)),
                torch.from_numpy(rng.uniform(0.1, 1.0, (E, 1)).astype(np.float32)))
    return (torch.from_numpy(rng.uniform(0.5, 1.5, (E, D)).astype(np.float32)),
            torch.from_numpy(rng.uniform(0.1, 1.0, (E, D)).astype(np.float32)))


@pytest.mark.parametrize("D", [1, 3, 4, 9, 16, 50, 64, 128, 256, 300, 1433])
@pytest.mark.parametrize("kind", ["none", "explicit", "normal", "uniform", "bernoulli"])
def test_agg_fwd_vs_oracle(dev, oracle, D, kind):
    from stag_amd import ops
    rng = np.random.default_rng(D * 7 + len(kind))
    n, e = (40, 300) if D > 512 else (200, 1500)
    g = random_graph(n, e, seed=D, hub=150, device=dev)
    E = g.number_of_edges()
    og = oracle_graph(oracle, g)
    x = rng.standard_normal((n, D)).astype(np.float32)
    ss = rng.uniform(0.5, 1.5, n).astype(np.float32)
    ds = rng.uniform(0.5, 1.5, n).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    for reduce in ("sum", "mean"):
        if kind == "none":
            w, spec = None, oracle.make_spec("none")
        elif kind == "explicit":
            wt = rng.uniform(-1, 2, (E, D)).astype(np.float32)
            w, spec = torch.from_numpy(wt).to(dev), oracle.make_spec("explicit", wt)
        else:
            p0, p1 = (0.7, None) if kind == "bernoulli" else ((0.2, 1.8) if kind == "uniform" else (1.0, 0.5))
            w = _noise(g, D, kind, p0, p1, seed=11, offset=2)
            spec = _ospec(oracle, g, D, kind, p0, p1, seed=11, offset=2)
        got = ops.aggregate(g, xd, w, reduce=reduce, src_scale=torch.from_numpy(ss).to(dev),
                            dst_scale=torch.from_numpy(ds).to(dev), seg_len=32)
        ref = oracle.agg_fwd(og, x, spec, reduce=oracle.REDUCE_MEAN if reduce == "mean" else oracle.REDUCE_SUM,
                             src_scale=ss, dst_scale=ds)
        assert_close(got, ref, what=f"agg {kind} D={D} {reduce}")


@pytest.mark.parametrize("mode", PARAM_CASES)
@pytest.mark.parametrize("kind", ["normal", "uniform", "bernoulli"])
@pytest.mark.parametrize("D", [6, 128])
def test_agg_param_modes_relu_innorm(dev, oracle, mode, kind, D):
    from stag_amd import ops
    rng = np.random.default_rng(len(mode) * 13 + D)
    g = random_graph(150, 1200, seed=D + 2, hub=200, device=dev)
    E = g.number_of_edges()
    og = oracle_graph(oracle, g)
    x = rng.standard_normal((150, D)).astype(np.float32)
    p0, p1 = _params(mode, E, D, rng)
    if kind == "bernoulli":
        p0, p1 = (p0 * 0.5 if torch.is_tensor(p0) else 0.6), None
    elif kind == "uniform" and torch.is_tensor(p0):
        p1 = p0 + p1
    elif kind == "uniform":
        p0, p1 = 0.5, 1.5
    for relu, norm in [(False, False), (True, False), (False, True)]:
        if norm and kind == "normal":
            continue   # sum of normals may pass arbitrarily close to 0: ill-conditioned by design
        kw = dict(relu=relu, in_norm=norm, seed=21, offset=7)
        w = _noise(g, D, kind, p0, p1, **kw)
        got = ops.aggregate(g, torch.from_numpy(x).to(dev), w, seg_len=48)
        ref = oracle.agg_fwd(og, x, _ospec(oracle, g, D, kind, p0, p1, **kw))
        assert_close(got, ref, what=f"{kind}/{mode} D={D} relu={relu} norm={norm}")
        # the materialised weights are the weights the fused kernel used
        wm = w.materialize()
        got2 = ops.aggregate(g, torch.from_numpy(x).to(dev), wm)
        assert_close(got2, ref, what=f"materialised {kind}/{mode}")


def test_empty_and_degenerate_graphs(dev, oracle):
    import stag_amd
    from stag_amd import ops
    # no edges at all
    g = stag_amd.Graph(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 5, device=dev)
    out = ops.aggregate(g, torch.randn(5, 8, device=dev), _noise(g, 8, "normal", 1.0, 1.0))
    assert out.shape == (5, 8) and float(out.abs().max()) == 0.0
    # single node, self loops only
    g = stag_amd.Graph(torch.zeros(3, dtype=torch.int64), torch.zeros(3, dtype=torch.int64), 1, device=dev)
    x = torch.ones(1, 4, device=dev)
    out = ops.aggregate(g, x, None)
    assert torch.allclose(out, torch.full((1, 4), 3.0, device=dev))
    # mean over a zero-in-degree node is 0 (DGL mean reducer)
    g = stag_amd.Graph(torch.tensor([0, 0]), torch.tensor([1, 1]), 3, device=dev)
    out = ops.aggregate(g, torch.ones(3, 4, device=dev), None, reduce="mean")
    assert torch.allclose(out, torch.tensor([[0.0] * 4, [1.0] * 4, [0.0] * 4], device=dev))


def test_seg_len_and_determinism(dev, oracle):
    from stag_amd import ops
    g = random_graph(500, 20000, seed=8, hub=5000, device=dev)
    x = torch.randn(500, 128, device=dev)
    w = _noise(g, 128, "normal", 1.0, 0.5, seed=3, offset=0)
    a = ops.aggregate(g, x, w, seg_len=64)
    b = ops.aggregate(g, x, w, seg_len=64)
    assert torch.equal(a, b), "same seed, same plan => same bits"
    # other segmentations are the same sum in another fp32 association: the 5000-edge hub
    # row drifts ~1e-5 when summed in runs of 1000+ (why the default plan cuts at 64)
    for sl in (16, 256, 1000, 0):
        c = ops.aggregate(g, x, w, seg_len=sl)
        assert scaled_err(c.cpu().numpy(), a.cpu().numpy()) <= 3 * TOL
    og = oracle_graph(oracle, g)
    ref = oracle.agg_fwd(og, x.cpu().numpy(), _ospec(oracle, g, 128, "normal", 1.0, 0.5, seed=3, offset=0))
    assert_close(a, ref, what="hub rows")


def test_backward_vs_oracle(dev, oracle):
    from stag_amd import ops
    rng = np.random.default_rng(5)
    n, D = 120, 20
    g = random_graph(n, 900, seed=4, hub=130, device=dev)
    E = g.number_of_edges()
    og, ogt = oracle_graph(oracle, g), oracle_graph(oracle, g, transposed=True)
    x = rng.standard_normal((n, D)).astype(np.float32)
    gout = rng.standard_normal((n, D)).astype(np.float32)
    ss = rng.uniform(0.5, 1.5, n).astype(np.float32)
    ds = rng.uniform(0.5, 1.5, n).astype(np.float32)
    # fused noise: dx on the transposed graph redraws the forward noise
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    w = _noise(g, D, "normal", 1.0, 0.5, seed=2, offset=9)
    out = ops.aggregate(g, xd, w, src_scale=torch.from_numpy(ss).to(dev), dst_scale=torch.from_numpy(ds).to(dev),
                        seg_len=32)
    out.backward(torch.from_numpy(gout).to(dev))
    wm = oracle.noise_materialize(og, _ospec(oracle, g, D, "normal", 1.0, 0.5, seed=2, offset=9), D)
    src, dst = (t.cpu().numpy() for t in g.edges())
    ref_dx = np.zeros((n, D))
    np.add.at(ref_dx, src, wm.astype(np.float64) * (gout * ds[:, None])[dst] * ss[src][:, None])
    assert_close(xd.grad, ref_dx, what="dx fused")
    # oracle statement of the same thing: aggregation over the transposed CSR with nidx
    ref2 = oracle.agg_fwd(ogt, gout, _ospec(oracle, g, D, "normal", 1.0, 0.5, seed=2, offset=9),
                          src_scale=ds, dst_scale=ss)
    assert_close(xd.grad, ref2, what="dx fused vs transposed oracle")
    # explicit weights: dx and dw
    wt = torch.from_numpy(rng.uniform(0.5, 1.5, (E, D)).astype(np.float32)).to(dev).requires_grad_(True)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = ops.aggregate(g, xd, wt, reduce="mean", src_scale=torch.from_numpy(ss).to(dev))
    out.backward(torch.from_numpy(gout).to(dev))
    deg = np.maximum(np.bincount(dst, minlength=n), 1)
    gs = gout / deg[:, None]
    ref_dw = oracle.agg_bwd_w(og, x, gs, src_scale=ss)
    assert_close(wt.grad, ref_dw, what="dw explicit")
    ref_dx = np.zeros((n, D))
    np.add.at(ref_dx, src, wt.detach().cpu().numpy().astype(np.float64) * gs[dst] * ss[src][:, None])
    assert_close(xd.grad, ref_dx, what="dx explicit")


@pytest.mark.parametrize("kind,relu", [("normal", False), ("normal", True), ("uniform", True)])
@pytest.mark.parametrize("D", [6, 40, 128, 300])
def test_agg_bwd_one_pass(dev, oracle, kind, relu, D):
    """stag_agg_bwd: dx and the two parameter-derivative aggregates from ONE pass equal three
    separate passes (spec.deriv = 0, 1, 2) bit for bit, and the oracle within the bar; hub row
    (segments + combine) and per-channel parameters included."""
    from stag_amd import _lib, ops
    rng = np.random.default_rng(17)
    n = 150
    g = random_graph(n, 1500, seed=6, hub=300, device=dev)
    ogt = oracle_graph(oracle, g, transposed=True)
    gout = rng.standard_normal((n, D)).astype(np.float32)
    gs = rng.uniform(0.5, 1.5, n).astype(np.float32)
    rs = rng.uniform(0.5, 1.5, n).astype(np.float32)
    p0 = torch.from_numpy(rng.uniform(0.2, 1.0, D).astype(np.float32)).to(dev)
    p1 = torch.from_numpy(rng.uniform(1.1, 1.8, D).astype(np.float32)).to(dev)
    gd, gsd, rsd = (torch.from_numpy(a).to(dev) for a in (gout, gs, rs))
    noise = _noise(g, D, kind, p0, p1, relu=relu, seed=3, offset=11)
    dx, t0, t1 = ops._agg_bwd_raw(g.csr_t, gd, D, noise.spec(), gsd, rsd, 32, True)
    only_dx, _, _ = ops._agg_bwd_raw(g.csr_t, gd, D, noise.spec(), gsd, rsd, 32, False)
    assert torch.equal(dx, only_dx)
    for deriv, got in ((0, dx), (1, t0), (2, t1)):
        noise.deriv = deriv
        sep, _ = ops._agg_raw(g.csr_t, gd, D, noise.spec(), _lib.REDUCE_SUM, gsd, rsd, 32)
        assert torch.equal(got, sep), f"deriv={deriv}: one pass != separate pass"
        ref = oracle.agg_fwd(ogt, gout, _ospec(oracle, g, D, kind, p0, p1, relu=relu, seed=3, offset=11, deriv=deriv),
                             src_scale=gs, dst_scale=rs)
        assert_close(got, ref, what=f"agg_bwd deriv={deriv}")
    noise.deriv = 0
    # argument validation
    cs = g.csr_t.struct()
    bern = _noise(g, D, "bernoulli", 0.5)
    import ctypes as C
    out = torch.empty(n, D, device=dev)
    rc = _lib.lib().stag_agg_bwd(C.byref(cs), None, _lib.ptr(gd), D, D, C.byref(bern.spec()), None, None,
                                 _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), D, None)
    assert rc == -22      # Bernoulli has no parameter derivative


@pytest.mark.parametrize("n,D", [(0, 8), (1, 1), (1000, 6), (20001, 40), (5000, 128), (3000, 300), (700, 1433)])
def test_coldot(dev, oracle, n, D):
    from stag_amd import ops
    rng = np.random.default_rng(n + D)
    x, t0, t1 = (rng.standard_normal((n, D)).astype(np.float32) for _ in range(3))
    xd, a, b = (torch.from_numpy(v).to(dev) for v in (x, t0, t1))
    o0, o1 = ops.coldot(xd, a, b)
    r0, r1 = oracle.coldot(x, t0, t1)
    scale = np.sqrt(max(n, 1))            # a column sum of n unit-variance products
    assert np.abs(o0.cpu().numpy() - r0).max() <= TOL * scale and np.abs(o1.cpu().numpy() - r1).max() <= TOL * scale
    only0, none = ops.coldot(xd, a)
    assert none is None and torch.equal(only0, o0)
    assert torch.equal(ops.coldot(xd, a, b)[1], o1)      # fixed summation order: repeatable


def test_node_linear_gradients(dev):
    """Split-K weight gradient of the dense transform == the plain GEMM's, to fp32 rounding."""
    from stag_amd import ops
    torch.manual_seed(0)
    for n, din, dout in ((169343, 128, 40), (5000, 16, 7), (100, 8, 8)):
        x = torch.randn(n, din, device=dev, requires_grad=True)
        w = torch.randn(din, dout, device=dev, requires_grad=True)
        g = torch.randn(n, dout, device=dev)
        y = ops.node_linear(x, w)
        y.backward(g)
        x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        y2 = x2 @ w2
        y2.backward(g)
        assert torch.equal(y, y2) and torch.allclose(x.grad, x2.grad, rtol=1e-5, atol=1e-5)
        ref = (x.detach().double().t() @ g.double())
        tol = 1e-5 * np.sqrt(n) * 4
        assert (w.grad.double() - ref).abs().max().item() <= tol
        assert (w.grad.double() - ref).abs().max().item() <= (w2.grad.double() - ref).abs().max().item() * 2 + 1e-6
    lin = torch.nn.Linear(12, 5, bias=False).to(dev)        # the nn.Linear form used by SAGE / GAT
    x = torch.randn(9000, 12, device=dev)
    ops.node_linear(x, lin.weight.t()).sum().backward()
    assert torch.allclose(lin.weight.grad, x.sum(0).expand(5, 12), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("kind", ["normal", "uniform", "bernoulli"])
@pytest.mark.parametrize("D", [6, 32, 128, 300])
def test_monte_carlo_samples_with_in_norm(dev, oracle, kind, D):
    """stag_agg_fwd_mc with in-norm (stag/layers.py:8-36; `norm=True`, scripts/arxiv_mle/gcn/run.py:70-74): every
    sample carries its own per-destination weight sums (two samples per pass), a hub row's segments included —
    bit-equal to separate launches, and against the oracle."""
    import copy
    from stag_amd import ops
    rng = np.random.default_rng(29)
    n = 200
    g = random_graph(n, 2000, seed=9, hub=500, device=dev)
    x = torch.randn(n, D, device=dev)
    ds = torch.rand(n, device=dev) + 0.5
    p0 = torch.from_numpy(rng.uniform(0.3, 0.9, D).astype(np.float32)).to(dev)
    p1 = None if kind == "bernoulli" else torch.from_numpy(rng.uniform(1.0, 1.6, D).astype(np.float32)).to(dev)
    noise = _noise(g, D, kind, p0, p1, relu=(kind == "normal"), in_norm=True, seed=4, offset=100)
    for S, stride in ((2, 1), (3, 5), (4, 3), (5, 2)):
        got = ops.aggregate_mc(g, x, noise, S, offset_stride=stride, dst_scale=ds, seg_len=32)
        assert got.shape == (S, n, D)
        for s in range(S):
            nz = copy.copy(noise)
            nz.offset = 100 + s * stride
            assert torch.equal(got[s], ops.aggregate(g, x, nz, dst_scale=ds, seg_len=32)), (S, s)
    og = oracle_graph(oracle, g)
    with hw_normals(oracle, dev):
        ref = oracle.agg_fwd_mc(og, x.cpu().numpy(), _ospec(oracle, g, D, kind, p0, p1, relu=(kind == "normal"), in_norm=True,
                                                             seed=4, offset=100), 4, offset_stride=3, dst_scale=ds.cpu().numpy())
    got = ops.aggregate_mc(g, x, noise, 4, offset_stride=3, dst_scale=ds)
    for s in range(4):
        assert_close(got[s], ref[s], what=f"mc + in-norm sample {s} vs oracle twin")


@pytest.mark.parametrize("kind", ["normal", "uniform", "bernoulli"])
@pytest.mark.parametrize("D", [6, 128, 300])
def test_monte_carlo_samples_one_pass(dev, oracle, kind, D):
    """stag_agg_fwd_mc: S samples from one pass over the gathered rows == S separate launches at
    offsets o + s * stride, bit for bit (hub row, per-channel parameters, relu, degree scalings)."""
    import copy
    from stag_amd import ops
    rng = np.random.default_rng(23)
    n = 200
    g = random_graph(n, 2000, seed=8, hub=400, device=dev)
    x = torch.randn(n, D, device=dev)
    ss, ds = torch.rand(n, device=dev) + 0.5, torch.rand(n, device=dev) + 0.5
    p0 = torch.from_numpy(rng.uniform(0.3, 0.9, D).astype(np.float32)).to(dev)
    p1 = None if kind == "bernoulli" else torch.from_numpy(rng.uniform(1.0, 1.6, D).astype(np.float32)).to(dev)
    noise = _noise(g, D, kind, p0, p1, relu=(kind == "normal"), seed=4, offset=100)
    for S, stride in ((1, 1), (2, 1), (3, 5), (4, 3), (7, 2)):
        got = ops.aggregate_mc(g, x, noise, S, offset_stride=stride, reduce="mean", src_scale=ss, dst_scale=ds, seg_len=32)
        assert got.shape == (S, n, D)
        for s in range(S):
            nz = copy.copy(noise)
            nz.offset = 100 + s * stride
            assert torch.equal(got[s], ops.aggregate(g, x, nz, reduce="mean", src_scale=ss, dst_scale=ds, seg_len=32)), (S, s)
    og = oracle_graph(oracle, g)
    ref = oracle.agg_fwd_mc(og, x.cpu().numpy(), _ospec(oracle, g, D, kind, p0, p1, relu=(kind == "normal"), seed=4, offset=100),
                            4, offset_stride=3, reduce=oracle.REDUCE_MEAN, src_scale=ss.cpu().numpy(),
                            dst_scale=ds.cpu().numpy())
    got = ops.aggregate_mc(g, x, noise, 4, offset_stride=3, reduce="mean", src_scale=ss, dst_scale=ds)
    for s in range(4):
        assert_close(got[s], ref[s], what=f"mc sample {s} vs oracle twin")


def test_segment_reduce(dev, oracle):
    from stag_amd import ops
    rng = np.random.default_rng(0)
    sizes = np.array([3, 0, 17, 1, 40, 5])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    x = rng.standard_normal((int(sizes.sum()), 37)).astype(np.float32)
    for red in ("sum", "mean"):
        got = ops.segment_reduce(torch.from_numpy(x).to(dev), torch.from_numpy(offs).to(dev), red)
        ref = oracle.segment_reduce(x, offs, oracle.REDUCE_MEAN if red == "mean" else oracle.REDUCE_SUM)
        assert_close(got, ref, what=f"segment {red}")
    # long segments (average > 256 rows) take the chunked two-stage form; a long, an empty, a
    # short and a chunk-aligned segment side by side, with the gradient
    sizes = np.array([4000, 0, 10, 512, 478])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    x = rng.standard_normal((int(sizes.sum()), 24)).astype(np.float32)
    for red in ("sum", "mean"):
        xd = torch.from_numpy(x).to(dev).requires_grad_(True)
        got = ops.segment_reduce(xd, torch.from_numpy(offs).to(dev), red)
        ref = oracle.segment_reduce(x, offs, oracle.REDUCE_MEAN if red == "mean" else oracle.REDUCE_SUM)
        assert np.abs(got.detach().cpu().numpy() - ref).max() <= TOL * (1 + np.abs(ref).max()) * 4, red
        got.sum().backward()
        want = np.repeat(1.0 / np.maximum(sizes, 1) if red == "mean" else np.ones(len(sizes)), sizes)
        assert np.allclose(xd.grad.cpu().numpy(), want[:, None], rtol=1e-6)


@pytest.mark.parametrize("H,F", [(3, 4), (8, 32), (4, 5), (1, 64), (4, 64), (2, 6)])
@pytest.mark.parametrize("kind", ["none", "explicit", "normal", "bernoulli"])
def test_gat_fwd_vs_oracle(dev, oracle, H, F, kind):
    from stag_amd import ops
    rng = np.random.default_rng(H * 31 + F)
    n = 150
    g = random_graph(n, 1200, seed=H + F, hub=300, device=dev)
    E = g.number_of_edges()
    og = oracle_graph(oracle, g)
    el = rng.standard_normal((n, H)).astype(np.float32)
    er = rng.standard_normal((n, H)).astype(np.float32)
    ft = rng.standard_normal((n, H, F)).astype(np.float32)
    if kind == "none":
        w, spec = None, oracle.make_spec("none")
    elif kind == "explicit":
        wt = rng.uniform(0.5, 1.5, (E, H)).astype(np.float32)
        w, spec = torch.from_numpy(wt).to(dev), oracle.make_spec("explicit", wt)
    else:
        p0, p1 = (0.7, None) if kind == "bernoulli" else (1.0, 0.5)
        norm = kind == "bernoulli"
        w = _noise(g, H, kind, p0, p1, seed=17, offset=3, in_norm=norm)
        spec = _ospec(oracle, g, H, kind, p0, p1, seed=17, offset=3, in_norm=norm)
    out, attn = ops.gat_aggregate(g, torch.from_numpy(el).to(dev), torch.from_numpy(er).to(dev),
                                  torch.from_numpy(ft).to(dev), 0.2, w, want_attn=True)
    ref, ref_attn = oracle.gat_fwd(og, el, er, ft, 0.2, spec, want_attn=True)
    assert_close(out, ref, what=f"gat out {kind} H={H} F={F}")
    assert_close(attn, ref_attn, what=f"gat attn {kind}")


def test_gat_hub_rows_and_seg_len(dev, oracle):
    """13k-ish hub row: per-segment softmax states merged by the last arriver == one pass."""
    from stag_amd import ops
    rng = np.random.default_rng(2)
    n, H, F = 400, 8, 32
    g = random_graph(n, 6000, seed=21, hub=9000, device=dev)
    og = oracle_graph(oracle, g)
    el = (3 * rng.standard_normal((n, H))).astype(np.float32)     # spread logits: exercises the max shift
    er = (3 * rng.standard_normal((n, H))).astype(np.float32)
    ft = rng.standard_normal((n, H, F)).astype(np.float32)
    w = _noise(g, H, "normal", 1.0, 0.5, seed=4, offset=1)
    ref, ref_attn = oracle.gat_fwd(og, el, er, ft, 0.2, _ospec(oracle, g, H, "normal", 1.0, 0.5, seed=4, offset=1),
                                   want_attn=True)
    args = [torch.from_numpy(t).to(dev) for t in (el, er, ft)]
    for sl in (64, 16, 1000, 0):
        out, attn = ops.gat_aggregate(g, *args, 0.2, w, want_attn=True, seg_len=sl)
        # one 9000-term fp32 sum (no plan) drifts past 1e-5; the default plan (64) must not
        tol = TOL if 0 < sl <= 64 else 2 * TOL
        assert_close(out, ref, tol=tol, what=f"gat hub seg_len={sl}")
        assert_close(attn, ref_attn, tol=tol, what=f"gat hub attn seg_len={sl}")
        out2 = ops.gat_aggregate(g, *args, 0.2, w, seg_len=sl)
        assert torch.equal(out2, out), "attention output must not change the result"
    # H*F > 256 is outside the fused kernel (the C entry point says STAG_ENOSYS); ops composes the
    # layer from the aggregation kernel instead (test_gat_outside_the_fused_kernel_limits)
    big = ops.gat_aggregate(g, torch.zeros(n, 8, device=dev), torch.zeros(n, 8, device=dev),
                            torch.ones(n, 8, 64, device=dev))
    has_in = (g.in_degrees() > 0).to(big.dtype).reshape(n, 1, 1)
    assert big.shape == (n, 8, 64) and torch.allclose(big, has_in.expand_as(big), atol=1e-6)


def test_device_csr_build_full_size(dev, oracle):
    """stag_csr_build (rocPRIM radix sort) at the BASELINE size == the oracle's counting sort."""
    import stag_amd
    from stag_amd import synthetic
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    indptr, indices, eid, ind, outd = oracle.csr_build(src, dst, n, n)
    assert np.array_equal(g.csr.indptr.cpu().numpy(), indptr)
    assert np.array_equal(g.csr.indices.cpu().numpy(), indices)
    assert np.array_equal(g.csr.eid.cpu().numpy(), eid)
    t = g.csr_t
    indptr_t, indices_t, eid_t, *_ = oracle.csr_build(dst, src, n, n)
    assert np.array_equal(t.indptr.cpu().numpy(), indptr_t) and np.array_equal(t.eid.cpu().numpy(), eid_t)
    # one destination only / one edge
    g1 = stag_amd.Graph(torch.arange(5), torch.zeros(5, dtype=torch.int64), 5, device=dev)
    assert g1.csr.indptr.tolist() == [0, 5, 5, 5, 5, 5] and g1.csr.eid.tolist() == [0, 1, 2, 3, 4]


def test_fuzz_agg_against_oracle(dev, oracle):
    """Seeded sweep over shapes x kinds x parameter modes x flags x plans: every combination the
    launcher can route (tiles, scalar tail lanes, per-edge parameters, segments, mean, scales)."""
    from stag_amd import ops
    rng = np.random.default_rng(20261003)
    kinds = ["none", "explicit", "normal", "uniform", "bernoulli"]
    for it in range(60 * FUZZ_SCALE):
        n = int(rng.integers(1, 400))
        e = int(rng.integers(0, 3000))
        hub = int(rng.choice([0, 0, 70, 400]))
        D = int(rng.choice([1, 2, 5, 8, 12, 31, 32, 64, 100, 128, 129, 200, 256, 260, 515]))
        kind = kinds[it % 5]
        seg_len = int(rng.choice([0, 8, 64, 64, 1000]))
        reduce = "mean" if rng.random() < 0.3 else "sum"
        g = random_graph(n, e, seed=1000 + it, hub=hub if n > 4 else 0, device=dev)
        E = g.number_of_edges()
        og = oracle_graph(oracle, g)
        x = rng.standard_normal((n, D)).astype(np.float32)
        ss = rng.uniform(0.5, 1.5, n).astype(np.float32) if rng.random() < 0.5 else None
        ds = rng.uniform(0.5, 1.5, n).astype(np.float32) if rng.random() < 0.5 else None
        relu = bool(rng.random() < 0.3)
        if kind == "none":
            w, spec = None, oracle.make_spec("none")
        elif kind == "explicit":
            wt = rng.uniform(-1, 2, (E, D)).astype(np.float32)
            w, spec = torch.from_numpy(wt).to(dev), oracle.make_spec("explicit", wt)
        else:
            mode = PARAM_CASES[int(rng.integers(0, 4))]
            p0, p1 = _params(mode, E, D, rng)
            norm = False
            if kind == "bernoulli":
                p0, p1 = (p0 * 0.5 if torch.is_tensor(p0) else 0.6), None
                norm = bool(rng.random() < 0.5)
            elif kind == "uniform":
                p1 = p0 + p1
            kw = dict(relu=relu, in_norm=norm, seed=int(rng.integers(0, 2**40)), offset=int(rng.integers(0, 99)))
            w = _noise(g, D, kind, p0, p1, **kw)
            spec = _ospec(oracle, g, D, kind, p0, p1, **kw)
        t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
        got = ops.aggregate(g, t(x), w, reduce=reduce, src_scale=t(ss), dst_scale=t(ds), seg_len=seg_len)
        red = oracle.REDUCE_MEAN if reduce == "mean" else oracle.REDUCE_SUM
        ref = oracle.agg_fwd(og, x, spec, reduce=red, src_scale=ss, dst_scale=ds)
        # sum |terms| of every output (util.assert_close_cond: graphs of a few nodes and hundreds of edges add the same
        # few rows over and over, and the one fp32 rounding of x[u] s[u] is shared by all of a row's repeats: soak
        # case 9410, n=4 E=494 without noise, 1.05e-5)
        absspec = oracle.make_spec("none") if kind == "none" else oracle.make_spec(
            "explicit", np.abs(wt if kind == "explicit" else oracle.noise_materialize(og, spec, D)))
        ab = oracle.agg_fwd(og, np.abs(x), absspec, reduce=red, src_scale=ss, dst_scale=ds)
        assert_close_cond(got, ref, ab, what=f"fuzz {it}: n={n} E={E} D={D} {kind} seg={seg_len} {reduce} relu={relu}")


def test_fuzz_gat_against_oracle_and_composed_backward(dev, oracle):
    """Seeded sweep over graphs x head shapes x weight kinds x plans for the GAT entry points: the forward
    (workgroup-cooperative kernel, 1 / 2 / 4 chunks per lane, or the one-unit-per-team fallback) against the
    oracle, the backward (stag_gat_bwd, stag_gat_bwd_two_pass, or the composed calls) against autograd through the composed statement
    of the same layer (ops._gat_composed: torch ops over [E, H] + the aggregation kernel)."""
    from stag_amd import ops
    rng = np.random.default_rng(20261004)
    shapes = [(1, 4), (2, 8), (3, 4), (4, 16), (8, 32), (8, 64), (16, 64), (5, 12), (4, 5), (2, 256), (16, 8), (6, 40)]
    kinds = ["none", "explicit", "normal", "uniform", "bernoulli"]
    for it in range(36 * FUZZ_SCALE):
        n = int(rng.integers(2, 500))
        e = int(rng.integers(1, 4000))
        hub = int(rng.choice([0, 0, 90, 700]))
        H, F = shapes[it % len(shapes)]
        kind = kinds[it % 5]
        seg_len = int(rng.choice([64, 64, 16, 256, 0]))
        g = random_graph(n, e, seed=3000 + it, hub=hub if n > 4 else 0, device=dev)
        E = g.number_of_edges()
        og = oracle_graph(oracle, g)
        el = rng.standard_normal((n, H)).astype(np.float32)
        er = rng.standard_normal((n, H)).astype(np.float32)
        ft = rng.standard_normal((n, H, F)).astype(np.float32)
        relu = bool(rng.random() < 0.3)
        if kind == "none":
            w, spec = None, oracle.make_spec("none")
        elif kind == "explicit":
            wt = rng.uniform(-0.5, 1.5, (E, H)).astype(np.float32)
            w, spec = torch.from_numpy(wt).to(dev), oracle.make_spec("explicit", wt)
        else:
            p0, p1 = {"normal": (1.0, 0.5), "uniform": (0.2, 1.7), "bernoulli": (0.7, None)}[kind]
            norm = kind == "bernoulli" and bool(rng.random() < 0.5)
            kw = dict(relu=relu, in_norm=norm, seed=int(rng.integers(0, 2**40)), offset=int(rng.integers(0, 99)))
            w = _noise(g, H, kind, p0, p1, **kw)
            spec = _ospec(oracle, g, H, kind, p0, p1, **kw)
        what = f"gat fuzz {it}: n={n} E={E} H={H} F={F} {kind} seg={seg_len} relu={relu}"
        t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
        out, attn = ops.gat_aggregate(g, *t, 0.2, w, want_attn=True, seg_len=seg_len)
        ref, ref_attn = oracle.gat_fwd(og, el, er, ft, 0.2, spec, want_attn=True)
        tol = TOL if 0 < seg_len <= 64 else 2 * TOL
        assert_close(out, ref, tol=tol, what=what + " out")
        assert_close(attn, ref_attn, tol=tol, what=what + " attn")
        G = torch.from_numpy(rng.standard_normal((n, H, F)).astype(np.float32)).to(dev)
        out.backward(G)
        # primary: the CPU twin of the backward (oracle.gat_bwd, float64); the composed statement under autograd below
        # is the second opinion
        with hw_normals(oracle, dev):
            ref_hw = oracle.gat_fwd(og, el, er, ft, 0.2, spec)
        assert_close(out, ref_hw, tol=tol, what=what + " out (device tables)")
        assert_gat_grads_vs_oracle(oracle, og, el, er, ft, G.cpu().numpy(), spec, [a_.grad for a_ in t], what=what, dev=dev, tol=tol)
        t2 = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
        out2 = ops.gat_aggregate(g, *t2, 0.2, w, seg_len=seg_len if seg_len else 64, attn_fn=lambda a_: a_)   # the composed path
        out2.backward(G)
        for a_, b_, nm in zip(t, t2, ("d el", "d er", "d ft")):
            sc = max(1.0, float(b_.grad.abs().max()))
            assert_close(a_.grad / sc, (b_.grad / sc).cpu().numpy(), what=what + " " + nm)
        # the two-gather form of the same backward (stag_gat_bwd_two_pass) stays covered
        t3 = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
        ops._GAT_BWD_ONE_GATHER = False
        try:
            ops.gat_aggregate(g, *t3, 0.2, w, seg_len=seg_len).backward(G)
        finally:
            ops._GAT_BWD_ONE_GATHER = True
        for a_, b_, nm in zip(t3, t2, ("d el", "d er", "d ft")):
            sc = max(1.0, float(b_.grad.abs().max()))
            assert_close(a_.grad / sc, (b_.grad / sc).cpu().numpy(), what=what + " two-pass " + nm)
        assert_gat_grads_vs_oracle(oracle, og, el, er, ft, G.cpu().numpy(), spec, [a_.grad for a_ in t3],
                                   what=what + " two-pass", dev=dev, tol=tol)


@pytest.mark.parametrize("kind,relu,logs", [("normal", False, False), ("normal", True, True), ("uniform", True, False)])
@pytest.mark.parametrize("D", [1, 4, 12, 50, 100, 128, 256])
def test_agg_bwd_edge_one_pass(dev, oracle, kind, relu, logs, D):
    """stag_agg_bwd_edge: dx and the gradients of [E, 1] parameters from ONE pass over the source-major CSR
    against the two-kernel statement (stag_agg_bwd for dx: bit for bit; stag_agg_bwd_w with reduce_k for the
    per-edge sums) and against the oracle's dw/dp summed over the channels.  Widths that leave lanes of the
    team idle (50, 100), a hub source row (segments), and the dx == NULL form included."""
    from stag_amd import _lib, ops
    import stag_amd
    rng = np.random.default_rng(23)
    n = 150
    g = random_graph(n, 1500, seed=6, hub=300, device=dev)
    E = g.number_of_edges()
    x = rng.standard_normal((n, D)).astype(np.float32)
    gout = rng.standard_normal((n, D)).astype(np.float32)
    gs = rng.uniform(0.5, 1.5, n).astype(np.float32)
    rs = rng.uniform(0.5, 1.5, n).astype(np.float32)
    p0 = rng.uniform(-0.3 if relu else 0.5, 1.0, (E, 1)).astype(np.float32)
    p1 = rng.uniform(0.3, 0.9, (E, 1)).astype(np.float32)
    if kind == "uniform":
        p1 = p0 + p1 + 0.5
    p1_in = np.log(p1) if logs else p1
    xd, gd, gsd, rsd, p0d, p1d = (torch.from_numpy(a).to(dev) for a in (x, gout, gs, rs, p0, p1_in))
    K = _lib.NOISE_NORMAL if kind == "normal" else _lib.NOISE_UNIFORM
    noise = stag_amd.EdgeNoise(g, D, K, p0d, p1d, relu=relu, seed=3, offset=11, p1_log=logs)
    spec = noise.spec()
    spec = spec if not isinstance(spec, tuple) else ops._targs_to_ctypes(spec)
    dx, e0, e1 = ops._agg_bwd_edge_raw(g.csr_t, gd, xd, D, spec, gsd, rsd, 32)
    ref_dx, _, _ = ops._agg_bwd_raw(g.csr_t, gd, D, noise.spec(), gsd, rsd, 32, False)
    assert_close(dx, ref_dx.cpu().numpy(), what="dx")
    none_dx, f0, f1 = ops._agg_bwd_edge_raw(g.csr_t, gd, xd, D, spec, gsd, rsd, 32, want_dx=False)
    assert none_dx is None and torch.equal(e0, f0) and torch.equal(e1, f1)
    # the two-kernel statement: gradient rows scaled on the host, destination-major pass
    gg = (gd * gsd.unsqueeze(1)).contiguous()
    r0, r1 = ops._bwd_w_raw(g.csr, xd, gg, D, rsd, spec=spec, reduce_k=True, both=True, seg_len=32)
    for got, ref, nm in ((e0, r0, "d p0"), (e1, r1, "d p1")):
        sc = max(1.0, float(ref.abs().max()))
        assert_close(got / sc, (ref / sc).cpu().numpy(), what=f"{nm} vs stag_agg_bwd_w")
    # the oracle: dw/dp_i [E, D] rows, summed over the channels here
    og = oracle_graph(oracle, g)
    for deriv, got in ((1, e0), (2, e1)):
        ospec = oracle.make_spec(kind, p0, p1_in, relu=relu, seed=3, offset=11, Dn=D, n_edges=E, deriv=deriv,
                                 **({"p1_log": True} if logs else {}))
        ref = oracle.agg_bwd_w(og, x, gout * gs[:, None], src_scale=rs, spec=ospec).astype(np.float64).sum(1, keepdims=True)
        sc = max(1.0, float(np.abs(ref).max()))
        assert_close(got / sc, ref / sc, what=f"deriv {deriv} vs oracle")
    # wider than one channel tile: no one-pass form
    assert ops._agg_bwd_edge_raw(g.csr_t, gd, xd, 300, spec, gsd, rsd, 32) is None


@pytest.mark.parametrize("kind,relu", [("normal", False), ("normal", True), ("uniform", True)])
@pytest.mark.parametrize("D", [1, 6, 40, 128, 300])
def test_agg_bwd_dp_one_pass(dev, oracle, kind, relu, D):
    """stag_agg_bwd_dp: dx and the FINISHED gradients of per-channel parameters from one pass over the source-major
    CSR, against the oracle's two-step statement (the aggregates T_i of spec.deriv = 1, 2, then sum_u x[u,k]
    T_i[u,k] in float64) and against stag_agg_bwd + stag_coldot; hub rows (segments), more than one channel tile
    (D = 300), no own row (x = None: the in-norm term), dx not wanted."""
    from stag_amd import ops
    rng = np.random.default_rng(29)
    n = 150
    g = random_graph(n, 1500, seed=6, hub=300, device=dev)
    ogt = oracle_graph(oracle, g, transposed=True)
    x = rng.standard_normal((n, D)).astype(np.float32)
    gout = rng.standard_normal((n, D)).astype(np.float32)
    gs = rng.uniform(0.5, 1.5, n).astype(np.float32)
    rs = rng.uniform(0.5, 1.5, n).astype(np.float32)
    p0 = torch.from_numpy(rng.uniform(0.2, 1.0, D).astype(np.float32)).to(dev)
    p1 = torch.from_numpy(rng.uniform(1.1, 1.8, D).astype(np.float32)).to(dev)
    xd, gd, gsd, rsd = (torch.from_numpy(a).to(dev) for a in (x, gout, gs, rs))
    noise = _noise(g, D, kind, p0, p1, relu=relu, seed=3, offset=11)
    spec = noise.spec()
    spec = spec if not isinstance(spec, tuple) else ops._targs_to_ctypes(spec)
    dx, c0, c1 = ops._agg_bwd_dp_raw(g.csr_t, gd, xd, D, spec, gsd, rsd, 32)
    ref_dx, t0, t1 = ops._agg_bwd_raw(g.csr_t, gd, D, noise.spec(), gsd, rsd, 32, True)
    assert_close(dx, ref_dx.cpu().numpy(), what="dx")
    k0, k1 = ops.coldot(xd, t0, t1)
    nodx, d0, d1 = ops._agg_bwd_dp_raw(g.csr_t, gd, xd, D, spec, gsd, rsd, 32, want_dx=False)
    assert nodx is None and torch.equal(c0, d0) and torch.equal(c1, d1)
    for deriv, got, two_step in ((1, c0, k0), (2, c1, k1)):
        T = oracle.agg_fwd(ogt, gout, _ospec(oracle, g, D, kind, p0, p1, relu=relu, seed=3, offset=11, deriv=deriv),
                           src_scale=gs, dst_scale=rs)
        ref = (x.astype(np.float64) * T.astype(np.float64)).sum(0)
        sc = max(1.0, float(np.abs(ref).max()))
        assert_close(got / sc, ref / sc, what=f"d p{deriv - 1} vs oracle")
        assert_close(got / sc, (two_step / sc).cpu().numpy(), what=f"d p{deriv - 1} vs stag_agg_bwd + stag_coldot")
    # no own row, no scales: the column sums of the aggregates
    _, n0, n1 = ops._agg_bwd_dp_raw(g.csr_t, gd, None, D, spec, None, None, 32, want_dx=False)
    _, u0, u1 = ops._agg_bwd_raw(g.csr_t, gd, D, noise.spec(), None, None, 32, True)
    for got, ref in ((n0, u0.double().sum(0)), (n1, u1.double().sum(0))):
        sc = max(1.0, float(ref.abs().max()))
        assert_close(got / sc, (ref / sc).cpu().numpy(), what="x = None")


@pytest.mark.parametrize("H,F,kind", [(8, 32, "normal"), (4, 16, "explicit"), (2, 256, "none"), (3, 8, "bernoulli"), (16, 64, "uniform"),
                                      (8, 40, "normal"), (5, 12, "explicit")])
def test_gat_attention_dropout_in_the_kernels(dev, oracle, H, F, kind):
    """Attention dropout (stag/zoo/gat.py:122; 0.6 in the reference's GAT scripts) inside the fused GAT kernels: the
    keep mask is a Bernoulli(keep_prob) stream of its own at (forward position, head) — the same words
    EdgeNoise(..., NOISE_BERNOULLI, keep_prob, seed, offset) materialises.  Forward against the oracle with that
    mask; backward (mask redrawn, never stored) against autograd through the composed statement with the mask as
    a tensor; hub rows, segments, explicit-weight gradients; eval-equivalent when nothing is dropped."""
    import stag_amd
    from stag_amd import _lib, ops
    rng = np.random.default_rng(41)
    n, p_drop, dseed, doff = 400, 0.6, 77, 5
    g = random_graph(n, 6000, seed=8, hub=900, device=dev)
    E = g.number_of_edges()
    og = oracle_graph(oracle, g)
    el = rng.standard_normal((n, H)).astype(np.float32)
    er = rng.standard_normal((n, H)).astype(np.float32)
    ft = rng.standard_normal((n, H, F)).astype(np.float32)
    if kind == "none":
        w, spec = None, oracle.make_spec("none")
    elif kind == "explicit":
        wt = rng.uniform(0.2, 1.5, (E, H)).astype(np.float32)
        w, spec = torch.from_numpy(wt).to(dev).requires_grad_(True), oracle.make_spec("explicit", wt)
    else:
        p0, p1 = {"normal": (1.0, 0.5), "uniform": (0.2, 1.7), "bernoulli": (0.7, None)}[kind]
        kw = dict(relu=(kind == "normal"), in_norm=(kind == "bernoulli"), seed=9, offset=3)
        w, spec = _noise(g, H, kind, p0, p1, **kw), _ospec(oracle, g, H, kind, p0, p1, **kw)
    # the kernels' threshold is fp32(1 - p) with the difference taken in double (ops._gat_drop_struct); fp32(1) - fp32(p)
    # is one ulp below it at p = 0.6, and a uniform on the grid point between them flips an edge (soak case 617)
    keep_prob = float(np.float32(1.0 - p_drop))
    keep = stag_amd.EdgeNoise(g, H, _lib.NOISE_BERNOULLI, keep_prob, seed=dseed, offset=doff).materialize()   # [E, H]
    frac = float(keep.mean())
    assert abs(frac - keep_prob) < 0.02
    t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
    out = ops.gat_aggregate(g, *t, 0.2, w, attn_drop=(p_drop, dseed, doff))
    ref = oracle.gat_fwd(og, el, er, ft, 0.2, spec, keep=keep.cpu().numpy(), keep_prob=keep_prob)
    assert_close(out, ref, what="forward vs oracle")
    G = torch.from_numpy(rng.standard_normal((n, H, F)).astype(np.float32)).to(dev)
    out.backward(G)
    got_dw = w.grad.clone() if kind == "explicit" else None
    if kind == "explicit":
        w.grad = None
    t2 = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
    out2 = ops.gat_aggregate(g, *t2, 0.2, w, attn_fn=lambda a_: a_ * keep / keep_prob)          # the composed path
    assert_close(out2, ref, what="composed forward vs oracle")
    out2.backward(G)
    assert_gat_grads_vs_oracle(oracle, og, el, er, ft, G.cpu().numpy(), spec, [a_.grad for a_ in t],
                               keep=keep.cpu().numpy(), keep_prob=keep_prob, got_dw=got_dw, what="dropout", dev=dev)
    for a_, b_, nm in zip(t, t2, ("d el", "d er", "d ft")):
        sc = max(1.0, float(b_.grad.abs().max()))
        assert_close(a_.grad / sc, (b_.grad / sc).cpu().numpy(), what=nm)
    if kind == "explicit":
        sc = max(1.0, float(w.grad.abs().max()))
        assert_close(got_dw / sc, (w.grad / sc).cpu().numpy(), what="dw")
    # the same call twice: the same bits (mask from counters); another offset: another mask
    again = ops.gat_aggregate(g, *[x_.detach() for x_ in t], 0.2, w.detach() if torch.is_tensor(w) else w,
                              attn_drop=(p_drop, dseed, doff))
    other = ops.gat_aggregate(g, *[x_.detach() for x_ in t], 0.2, w.detach() if torch.is_tensor(w) else w,
                              attn_drop=(p_drop, dseed, doff + 1))
    assert torch.equal(again, out.detach()) and not torch.equal(other, out.detach())
    # shapes without the cooperative form say so
    assert not ops.attn_drop_fusable(8, 10, 64) and ops.attn_drop_fusable(H, F, 64)


def test_fuzz_gat_attention_dropout(dev, oracle):
    """Seeded sweep over graphs x head shapes x weight kinds x drop rates for attention dropout inside the GAT
    kernels (stag/zoo/gat.py:122): forward against the oracle given the mask the same counters materialise, the
    one-gather backward (mask redrawn from the counters, its bit carried in the sign of a; the only form that takes
    dropout) against autograd through the composed statement with the mask as a tensor."""
    import stag_amd
    from stag_amd import _lib, ops
    rng = np.random.default_rng(20261006)
    shapes = [(1, 4), (2, 8), (3, 4), (4, 16), (8, 32), (8, 64), (16, 64), (5, 12), (2, 256), (16, 8), (6, 40), (8, 8)]
    kinds = ["none", "explicit", "normal", "uniform", "bernoulli"]
    for it in range(24 * FUZZ_SCALE):
        n = int(rng.integers(2, 500))
        e = int(rng.integers(1, 4000))
        hub = int(rng.choice([0, 0, 90, 700]))
        H, F = shapes[it % len(shapes)]
        kind = kinds[it % 5]
        p_drop = float(rng.choice([0.1, 0.5, 0.6, 0.9]))
        dseed, doff = int(rng.integers(0, 2**40)), int(rng.integers(0, 99))
        g = random_graph(n, e, seed=9000 + it, hub=hub if n > 4 else 0, device=dev)
        E = g.number_of_edges()
        og = oracle_graph(oracle, g)
        assert ops.attn_drop_fusable(H, F, 64)
        el = rng.standard_normal((n, H)).astype(np.float32)
        er = rng.standard_normal((n, H)).astype(np.float32)
        ft = rng.standard_normal((n, H, F)).astype(np.float32)
        if kind == "none":
            w, spec = None, oracle.make_spec("none")
        elif kind == "explicit":
            wt = rng.uniform(0.2, 1.5, (E, H)).astype(np.float32)
            w, spec = torch.from_numpy(wt).to(dev).requires_grad_(True), oracle.make_spec("explicit", wt)
        else:
            p0, p1 = {"normal": (1.0, 0.5), "uniform": (0.2, 1.7), "bernoulli": (0.7, None)}[kind]
            kw = dict(relu=bool(rng.random() < 0.3), in_norm=(kind == "bernoulli" and bool(rng.random() < 0.5)),
                      seed=int(rng.integers(0, 2**40)), offset=int(rng.integers(0, 99)))
            w, spec = _noise(g, H, kind, p0, p1, **kw), _ospec(oracle, g, H, kind, p0, p1, **kw)
        what = f"gat drop fuzz {it}: n={n} E={E} H={H} F={F} {kind} p={p_drop}"
        keep_prob = float(np.float32(1.0 - p_drop))           # the kernels' threshold: ops._gat_drop_struct
        keep = stag_amd.EdgeNoise(g, H, _lib.NOISE_BERNOULLI, keep_prob, seed=dseed, offset=doff).materialize()   # [E, H]
        t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
        out = ops.gat_aggregate(g, *t, 0.2, w, attn_drop=(p_drop, dseed, doff))
        ref = oracle.gat_fwd(og, el, er, ft, 0.2, spec, keep=keep.cpu().numpy(), keep_prob=keep_prob)
        assert_close(out, ref, what=what + " out")
        G = torch.from_numpy(rng.standard_normal((n, H, F)).astype(np.float32)).to(dev)
        out.backward(G)
        got_dw = None
        if kind == "explicit":
            got_dw, w.grad = w.grad.clone(), None
        t2 = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
        ops.gat_aggregate(g, *t2, 0.2, w, attn_fn=lambda a_: a_ * keep / keep_prob).backward(G)      # the composed path
        # d el / d er are sums of d s[e,h] = a' <G[v,h], ft[u,h]> - a <G[v,h], out[v,h]>: two F-term fp32 dot products
        # that cancel — exactly, when v has one in-edge and it is kept (a' = 1 / keep_prob, out = a' ft[u]).  Soak case
        # 1138: 20 edges, every destination of in-degree 1, p = 0.9: the true gradient is 0 everywhere, autograd's
        # softmax backward returns an exact 0, the kernels 36.6145 - 36.6145 = 1.7e-5.  So these two are measured
        # against the largest TERM of the sums as well as the largest result.
        with torch.no_grad():
            _, attn = ops.gat_aggregate(g, *[x_.detach() for x_ in t], 0.2, w.detach() if torch.is_tensor(w) else w, want_attn=True)
            src_, dst_ = g.edges()
            term = float((attn * keep / keep_prob * (G[dst_] * t[2].detach()[src_]).sum(-1)).abs().max()) if E else 0.0
        assert_gat_grads_vs_oracle(oracle, og, el, er, ft, G.cpu().numpy(), spec, [a_.grad for a_ in t],
                                   keep=keep.cpu().numpy(), keep_prob=keep_prob, got_dw=got_dw, what=what, dev=dev)
        for a_, b_, nm in zip(t, t2, ("d el", "d er", "d ft")):
            sc = max(1.0, float(b_.grad.abs().max()), term if nm != "d ft" else 0.0)
            assert_close(a_.grad / sc, (b_.grad / sc).cpu().numpy(), what=what + " " + nm)
        if kind == "explicit":
            sc = max(1.0, float(w.grad.abs().max()))
            assert_close(got_dw / sc, (w.grad / sc).cpu().numpy(), what=what + " dw")
            w.grad = None


def test_gat_layer_trains_with_attention_dropout_on_the_fused_path(dev):
    """zoo.GAT(attn_drop=0.6) in training mode — the reference's GAT scripts (scripts/citation_mle/gat/run.py:40) —
    stays on the fused kernels: every call takes one generator offset for its mask, gradients flow, eval mode is
    the plain kernel, and a fixed generator state replays the step bit for bit."""
    import stag_amd
    g = random_graph(300, 4000, seed=3, hub=500, device=dev)
    x = torch.randn(300, 32, device=dev)
    torch.manual_seed(1)
    layer = stag_amd.layers.StagLayer(stag_amd.zoo.GAT(32, 8, num_heads=4, attn_drop=0.6),
                                      q_a=torch.distributions.Normal(1.0, 0.5)).to(dev)
    layer.train()

    def run():
        stag_amd.manual_seed(21)
        o0 = stag_amd.random.default_generator.offset
        layer.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = layer(g, xg)
        y.square().mean().backward()
        used = stag_amd.random.default_generator.offset - o0
        return y.detach().clone(), xg.grad.clone(), layer.base_layer.attn_l.grad.clone(), used
    y1, dx1, da1, used = run()
    y2, dx2, da2, _ = run()
    assert used == 2                                   # one offset for the edge noise, one for the dropout mask
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and torch.equal(da1, da2)
    assert torch.isfinite(y1).all() and float(dx1.abs().sum()) > 0 and float(da1.abs().sum()) > 0
    layer.eval()
    stag_amd.manual_seed(21)
    with torch.no_grad():
        ye = layer(g, x)
    assert not torch.equal(ye, y1)                     # no mask in eval mode


@pytest.mark.parametrize("seg_len", [16, 64, 256])
def test_device_planner_equals_host_planner(dev, seg_len, monkeypatch):
    """stag_plan_device builds, from the device indptr, the plan stag_plan_count / stag_plan_fill build on the host:
    the same unit records in the same order (long rows first, most edges first, balanced segments; whole rows
    longest first; ties by row), long_rows, long_seg_ptr and counts — hubs, empty rows, a one-row graph, a graph
    without edges; and the block batches added on demand equal the host's."""
    import importlib
    import stag_amd
    G = importlib.import_module("stag_amd.graph")
    monkeypatch.setattr(G, "XCD_ORDER", "1")           # (auto would leave it out on these random graphs)
    rng = np.random.default_rng(seg_len)
    cases = [random_graph(300, 5000, seed=1, hub=700, device=dev), random_graph(50, 30, seed=2, device=dev),
             random_graph(2000, 30000, seed=3, hub=3000, device=dev),
             stag_amd.Graph(torch.zeros(200, dtype=torch.int64), torch.zeros(200, dtype=torch.int64), 1, device=dev),
             stag_amd.Graph(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 7, device=dev)]
    for g in cases:
        for view_name in ("csr", "csr_t"):
            view = getattr(g, view_name)
            assert G.DEVICE_PLANNER
            dplan = view.plan(seg_len, need=True)
            view._plans.clear()
            G.DEVICE_PLANNER = False
            try:
                hplan = view.plan(seg_len, need=True)
            finally:
                G.DEVICE_PLANNER = True
            view._plans.clear()
            for k in ("n_units", "n_long", "n_seg", "n_heavy", "n_blocks"):
                assert dplan[k] == hplan[k], (k, dplan[k], hplan[k])
            nu, nl = hplan["n_units"], hplan["n_long"]
            assert torch.equal(dplan["units"][:nu].cpu(), hplan["units"][:nu].cpu())
            assert torch.equal(dplan["long_rows"][:nl].cpu(), hplan["long_rows"][:nl].cpu())
            assert torch.equal(dplan["long_seg_ptr"][:nl + 1].cpu(), hplan["long_seg_ptr"][:nl + 1].cpu())
            assert torch.equal(dplan["block_ptr"].cpu(), hplan["block_ptr"].cpu())
            assert dplan["xcd_strides"] == hplan["xcd_strides"]          # stag_plan_xcd_device == stag_plan_xcd
            if nu:
                assert torch.equal(dplan["xcd"].cpu(), hplan["xcd"].cpu())
            else:
                assert dplan["xcd"] is None and hplan["xcd"] is None


def test_stripe_locality_device_equals_host(dev):
    """stag_stripe_locality (what "auto" decides the XCD-aware order by) counts what the host-side statement counts."""
    import stag_amd
    from stag_amd import synthetic
    s3, d3, sizes = synthetic.ppi_like(n_graphs=6, n_nodes=3000, n_edges=40000, seed=5)
    cases = [(torch.from_numpy(s3), torch.from_numpy(d3), int(sizes.sum()))]
    g = random_graph(2500, 30000, seed=3, hub=3000)
    cases.append((*g.edges(), 2500))
    cases.append((torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 7))
    for src, dst, n in cases:
        gd, gh = stag_amd.Graph(src, dst, n, device=dev), stag_amd.Graph(src, dst, n)
        for v in ("csr", "csr_t"):
            assert getattr(gd, v).stripe_locality() == pytest.approx(getattr(gh, v).stripe_locality(), abs=1e-6)
    assert stag_amd.Graph(cases[0][0], cases[0][1], cases[0][2], device=dev).csr.stripe_locality() > 0.5        # 6 graphs in 8 stripes


def _without_xcd_order(view, seg_len):
    p = view.plan(seg_len, need=True)
    p["xcd"], p["xcd_strides"], p["xcd_on"], p["xcd_decided"] = None, (0, 0), False, True
    p.pop("_structs", None)
    p.pop("_ints", None)


@pytest.mark.parametrize("D", [1, 8, 24, 50, 64, 128, 200, 256, 300])
def test_xcd_aware_unit_order_changes_no_bit(dev, oracle, D, monkeypatch):
    """stag_plan.xcd_order only changes WHICH workgroup walks a unit (workgroup b takes stripe b mod 8 of the
    destination rows): forward (every kind, in-norm, mean), the Monte-Carlo batch and the backward pass are
    bit-identical with and without it — hubs cut into segments, empty rows, a block-diagonal batch (the case it is for)
    and a graph with fewer units than stripes — and the forward equals the oracle."""
    import importlib
    import stag_amd
    from stag_amd import ops, synthetic
    monkeypatch.setattr(importlib.import_module("stag_amd.graph"), "XCD_ORDER", "1")
    s3, d3, sizes = synthetic.ppi_like(n_graphs=6, n_nodes=3000, n_edges=40000, seed=5)
    # "batch_graphs": the union KNOWS its graphs (batch_num_nodes): the stripes are whole graphs, bin-packed, and the wide
    # shapes walk one family of stripes (stag_plan_xcd_ranges; the budget lowered so that these small graphs make several
    # fine ranges and one of them is cut in two)
    monkeypatch.setattr(importlib.import_module("stag_amd.graph"), "XCD_RANGE_BYTES", 40_000 * max(1, min(D, 256)) // 64)
    graphs = [("hubs", lambda: random_graph(2500, 30000, seed=3, hub=3000, device=dev)),
              ("batch", lambda: stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), int(sizes.sum()), device=dev)),
              ("batch_graphs", lambda: stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), int(sizes.sum()),
                                                      batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)),
              ("tiny", lambda: random_graph(5, 12, seed=9, device=dev))]
    rng = np.random.default_rng(D)
    for name, mk in graphs:
        ga, gb = mk(), mk()
        for view in (gb.csr, gb.csr_t):
            _without_xcd_order(view, 64)
        assert ga.csr.plan(64, need=True)["xcd_on"] and not gb.csr.plan(64)["xcd_on"]
        if name == "batch_graphs":
            for view in (ga.csr, ga.csr_t):
                view.xcd_graphs = True          # (by default only for rows of 1 KB and up: here at every width)
            order, strides, tag = ga.csr.xcd_order(ga.csr.plan(64), D)
            assert tag == 1000 + min(D, 256) + (512 if D > 128 else 0) and (strides[0] == 0) == (D > 128)
            # the device builder (stable radix sort of the range keys) and the host one: the same ints
            gh = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), int(sizes.sum()), batch_num_nodes=torch.from_numpy(sizes))
            gh.csr.xcd_graphs = True
            oh, sh_, _ = gh.csr.xcd_order(gh.csr.plan(64, need=True), D)
            assert sh_ == strides and torch.equal(order.cpu(), oh), "device and host builders of the range-table order"
        og = oracle_graph(oracle, ga)
        n = ga.number_of_nodes()
        xh = rng.standard_normal((n, D)).astype(np.float32)
        x = torch.from_numpy(xh).to(dev)
        gout = torch.randn(n, D, device=dev)
        for kind, p0, p1 in (("none", None, None), ("normal", 1.0, 0.5), ("uniform", 0.2, 1.7), ("bernoulli", 0.6, None)):
            for in_norm, reduce in ((False, "sum"), (True, "mean")):
                if kind == "none" and in_norm:
                    continue
                what = f"{name} {kind} D={D} in_norm={in_norm}"
                relu = in_norm and kind == "normal"        # deg / sum(w) of signed weights is ill-conditioned: not this test's subject
                mk_noise = lambda g: None if kind == "none" else _noise(g, D, kind, p0, p1, seed=11, offset=3, in_norm=in_norm,
                                                                        relu=relu)
                ya = ops.aggregate(ga, x, mk_noise(ga), reduce=reduce)
                yb = ops.aggregate(gb, x, mk_noise(gb), reduce=reduce)
                assert torch.equal(ya, yb), what
                spec = oracle.make_spec("none") if kind == "none" else _ospec(oracle, ga, D, kind, p0, p1, seed=11, offset=3,
                                                                              in_norm=in_norm, relu=relu)
                ref = oracle.agg_fwd(og, xh, spec, reduce=oracle.REDUCE_MEAN if reduce == "mean" else oracle.REDUCE_SUM)
                assert_close(ya, ref, what=what)
                if kind != "none":
                    ma = ops.aggregate_mc(ga, x, mk_noise(ga), 4, reduce=reduce)
                    mb = ops.aggregate_mc(gb, x, mk_noise(gb), 4, reduce=reduce)
                    assert torch.equal(ma, mb) and torch.equal(ma[0], ya), what + " mc"
            if kind in ("normal", "uniform") and D <= 256:
                # (round 4) the other kernel families walk the same order: per-edge [E, 1] and [E, D] parameters forward,
                # the derivative outputs (stag_agg_bwd, three accumulator sets) and the [E, 1] parameter gradients
                # (stag_agg_bwd_edge) on the transposed view
                from stag_amd import _lib as L
                E = ga.number_of_edges()
                q0 = torch.tensor(rng.standard_normal((E, 1)).astype(np.float32) * 0.2 + 1.0, device=dev)
                q1 = torch.tensor(rng.random((E, 1)).astype(np.float32) * 0.5 + (0.2 if kind == "normal" else 1.5), device=dev)
                Q0, Q1 = q0.expand(E, D).contiguous() + 0.01, q1.expand(E, D).contiguous()
                k_ = L.NOISE_NORMAL if kind == "normal" else L.NOISE_UNIFORM
                for pa, pb, nm in ((q0, q1, "[E,1]"), (Q0, Q1, "[E,D]")):
                    ya = ops.aggregate(ga, x, stag_amd.EdgeNoise(ga, D, k_, pa, pb, seed=11, offset=3))
                    yb = ops.aggregate(gb, x, stag_amd.EdgeNoise(gb, D, k_, pa, pb, seed=11, offset=3))
                    assert torch.equal(ya, yb), f"{name} {kind} D={D} per-edge parameters {nm}"
                res = []
                for g in (ga, gb):
                    spec = ops._targs_or_c(ops._noise_spec(_noise(g, D, kind, p0, p1, seed=11, offset=3), in_norm=0))
                    res.append(ops._agg_bwd_raw(g.csr_t, gout, D, spec, None, None, 64, True))
                    se = ops._targs_or_c(ops._noise_spec(stag_amd.EdgeNoise(g, D, k_, q0, q1, seed=11, offset=3), in_norm=0))
                    res[-1] = res[-1] + ops._agg_bwd_edge_raw(g.csr_t, gout, x, D, se, None, None, 64)
                for a_, b_ in zip(*res):
                    assert torch.equal(a_, b_), f"{name} {kind} D={D}: derivative outputs / [E,1] gradients under the walk"
            outs = []                     # backward: dx on the transposed plan, the noise regenerated
            for g in (ga, gb):
                xr = x.clone().requires_grad_(True)
                ops.aggregate(g, xr, None if kind == "none" else _noise(g, D, kind, p0, p1, seed=11, offset=3)).backward(gout)
                outs.append(xr.grad)
            assert torch.equal(outs[0], outs[1]), f"{name} {kind} D={D} dx"


def test_fuzz_device_planner_equals_host_planner(dev, monkeypatch):
    """Seeded sweep of test_device_planner_equals_host_planner: random multigraphs (1 ... 3000 nodes, 0 ... 40,000
    edges, with and without hubs, a few nodes with thousands of parallel edges) x segment lengths, both CSR views,
    element for element."""
    import importlib
    G = importlib.import_module("stag_amd.graph")
    monkeypatch.setattr(G, "XCD_ORDER", "1")
    rng = np.random.default_rng(20261008)
    keys = ("units", "long_rows", "long_seg_ptr", "block_ptr")
    for it in range(20 * FUZZ_SCALE):
        n = int(rng.choice([1, 2, 3, 17, int(rng.integers(1, 3000))]))
        e = int(rng.choice([0, 1, int(rng.integers(0, 40000))]))
        hub = int(rng.choice([0, 0, 65, 1000, 5000])) if n > 4 else 0
        seg_len = int(rng.choice([8, 16, 64, 64, 100, 256]))
        monkeypatch.setattr(G, "XCD_FINE", int(rng.choice([0, 1, 3, 16])))      # finer row ranges inside the XCD stripes
        g = random_graph(n, e, seed=11000 + it, hub=hub, device=dev)
        for view_name in ("csr", "csr_t"):
            view = getattr(g, view_name)
            dplan = view.plan(seg_len, need=True)
            view._plans.clear()
            G.DEVICE_PLANNER = False
            try:
                hplan = view.plan(seg_len, need=True)
            finally:
                G.DEVICE_PLANNER = True
            view._plans.clear()
            what = f"plan fuzz {it}: n={n} E={g.number_of_edges()} seg={seg_len} {view_name}"
            for k in ("n_units", "n_long", "n_seg", "n_heavy", "n_blocks"):
                assert dplan[k] == hplan[k], (what, k, dplan[k], hplan[k])
            nu, nl = hplan["n_units"], hplan["n_long"]
            for k, m in zip(keys, (nu, nl, nl + 1, None)):
                assert torch.equal(dplan[k][:m].cpu(), hplan[k][:m].cpu()), (what, k)
            assert dplan["xcd_strides"] == hplan["xcd_strides"], what
            assert (dplan["xcd"] is None and hplan["xcd"] is None) or torch.equal(dplan["xcd"].cpu(), hplan["xcd"].cpu()), what


def test_fuzz_range_table_orders_host_equals_device(dev, monkeypatch):
    """Seeded sweep of the range-table XCD orders (stag_plan_xcd_ranges on host records, stag_plan_xcd_device_count_ranges +
    _fill on device records): random block-diagonal batches — 2 ... 40 graphs of 1 ... 900 nodes, some without edges, some
    with a hub, batches smaller than 8 graphs — x row widths x L2 budgets x heavy-merged or not: the same ints from both
    builders, every unit exactly once, and the aggregation over the order bit-identical to plan order."""
    import importlib
    import stag_amd
    from stag_amd import ops
    G = importlib.import_module("stag_amd.graph")
    monkeypatch.setattr(G, "XCD_ORDER", "1")
    rng = np.random.default_rng(20261104)
    for it in range(12 * FUZZ_SCALE):
        ng = int(rng.choice([2, 3, 7, 8, 9, 24, 40]))
        sizes = rng.integers(1, 900, ng)
        srcs, dsts, off = [], [], 0
        for k, n_k in enumerate(sizes):
            e_k = 0 if rng.random() < 0.15 else int(rng.integers(1, 12 * n_k + 2))
            s_ = rng.integers(0, n_k, e_k)
            d_ = rng.integers(0, n_k, e_k)
            if e_k and rng.random() < 0.3:
                d_[: e_k // 2] = int(rng.integers(0, n_k))          # a hub: segments of a long row inside one graph
            srcs.append(s_ + off); dsts.append(d_ + off); off += int(n_k)
        src, dst = np.concatenate(srcs), np.concatenate(dsts)
        if len(src) == 0:
            continue
        monkeypatch.setattr(G, "XCD_RANGE_BYTES", int(rng.choice([20_000, 200_000, 2_500_000])))
        width = int(rng.choice([24, 64, 128, 200, 256, 300]))
        bn = torch.from_numpy(sizes.astype(np.int64))
        gd = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), off, batch_num_nodes=bn.to(dev), device=dev)
        gh = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), off, batch_num_nodes=bn)
        what = f"range-table fuzz {it}: {ng} graphs, N={off}, E={len(src)}, width {width}"
        for name in ("csr", "csr_t"):
            vd, vh = getattr(gd, name), getattr(gh, name)
            vd.xcd_graphs = vh.xcd_graphs = True
            pd, ph = vd.plan(64, need=True), vh.plan(64, need=True)
            for drawn in (False, True):
                od, sd, td = vd.xcd_order(pd, width, drawn)
                oh, sh_, th = vh.xcd_order(ph, width, drawn)
                assert sd == sh_ and td == th and torch.equal(od.cpu(), oh), (what, name, drawn)
                rec = oh.numpy()[32:].reshape(-1, 4)
                real = rec[rec[:, 0] >= 0]
                units = ph["units"].numpy()[:ph["n_units"]]
                assert len(real) == len(units) and sorted(map(tuple, real)) == sorted(map(tuple, units)), (what, name)
        x = torch.randn(off, width, device=dev)
        gp = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), off, device=dev)
        for view in (gp.csr, gp.csr_t):
            _without_xcd_order(view, 64)
        for noise in (None, _noise(gd, width, "normal", 1.0, 0.5, seed=5, offset=it)):
            nz_p = None if noise is None else _noise(gp, width, "normal", 1.0, 0.5, seed=5, offset=it)
            assert torch.equal(ops.aggregate(gd, x, noise, reduce="mean"), ops.aggregate(gp, x, nz_p, reduce="mean")), what


def test_fuzz_backward_passes_against_oracle(dev, oracle):
    """Seeded sweep over graphs x widths x plans x kinds for the three backward entry points on the source-major
    CSR: stag_agg_bwd (dx + derivative aggregates), stag_agg_bwd_dp (dx + finished scalar / per-channel gradients),
    stag_agg_bwd_edge (dx + [E,1] gradients) — each against the oracle's aggregation with spec.deriv = 0, 1, 2 on
    the transposed graph (and, for the per-edge sums, its dw/dp rows summed over the channels)."""
    import stag_amd
    from stag_amd import _lib, ops
    rng = np.random.default_rng(20261005)
    widths = [1, 3, 4, 8, 20, 32, 50, 64, 100, 128, 200, 256, 300, 515]
    # relu's derivative is a step: a weight within an ulp of 0 (case 944 of the soak: -2.5e-8 from libm, +1.9e-8 from the
    # hardware functions) puts a whole term x g on one side or the other.  The oracle therefore draws from the device's
    # tables here (util.hw_normals: identical weights, bit for bit; the tables are pinned against libm exhaustively).
    from util import hw_normals
    with hw_normals(oracle, dev):
        for it in range(30 * FUZZ_SCALE):
            n = int(rng.integers(1, 400))
            e = int(rng.integers(0, 5000))
            hub = int(rng.choice([0, 0, 80, 600])) if n > 4 else 0
            D = int(widths[it % len(widths)])
            kind = ["normal", "uniform"][it % 2]
            relu = bool(rng.random() < 0.4)
            seg_len = int(rng.choice([64, 64, 16, 256, 0]))
            g = random_graph(n, e, seed=7000 + it, hub=hub, device=dev)
            E = g.number_of_edges()
            ogt, og = oracle_graph(oracle, g, transposed=True), oracle_graph(oracle, g)
            x = rng.standard_normal((n, D)).astype(np.float32)
            gout = rng.standard_normal((n, D)).astype(np.float32)
            gs = rng.uniform(0.5, 1.5, n).astype(np.float32)
            rs = rng.uniform(0.5, 1.5, n).astype(np.float32)
            xd, gd, gsd, rsd = (torch.from_numpy(a).to(dev) for a in (x, gout, gs, rs))
            what = f"bwd fuzz {it}: n={n} E={E} D={D} {kind} relu={relu} seg={seg_len}"
            seed, off = int(rng.integers(0, 2**40)), int(rng.integers(0, 99))
            # ---- per-channel parameters: stag_agg_bwd and stag_agg_bwd_dp ------------------------------------
            p0 = rng.uniform(0.2, 1.0, D).astype(np.float32)
            p1 = (p0 + rng.uniform(0.6, 1.2, D)).astype(np.float32) if kind == "uniform" else rng.uniform(0.3, 0.9, D).astype(np.float32)
            noise = _noise(g, D, kind, torch.from_numpy(p0).to(dev), torch.from_numpy(p1).to(dev), relu=relu, seed=seed, offset=off)
            T = [oracle.agg_fwd(ogt, gout, _ospec(oracle, g, D, kind, p0, p1, relu=relu, seed=seed, offset=off, deriv=dv),
                                src_scale=gs, dst_scale=rs) for dv in (0, 1, 2)]
            # sum |terms| of each of the three sums (the weights, or their derivatives, by edge id; util.assert_close_cond)
            A = [oracle.agg_fwd(ogt, np.abs(gout), oracle.make_spec("explicit", np.abs(oracle.noise_materialize(
                     ogt, _ospec(oracle, g, D, kind, p0, p1, relu=relu, seed=seed, offset=off, deriv=dv), D))),
                     src_scale=gs, dst_scale=rs) for dv in (0, 1, 2)]
            dx, t0, t1 = ops._agg_bwd_raw(g.csr_t, gd, D, noise.spec(), gsd, rsd, seg_len, True)
            for got, ref, ab, nm in ((dx, T[0], A[0], "dx"), (t0, T[1], A[1], "T0"), (t1, T[2], A[2], "T1")):
                assert_close_cond(got, ref, ab, tol=TOL if 0 < seg_len <= 64 else 2 * TOL, what=f"{what} stag_agg_bwd {nm}")
            spec = noise.spec()
            spec = spec if not isinstance(spec, tuple) else ops._targs_to_ctypes(spec)
            dx2, c0, c1 = ops._agg_bwd_dp_raw(g.csr_t, gd, xd, D, spec, gsd, rsd, seg_len)
            assert_close_cond(dx2, T[0], A[0], tol=TOL if 0 < seg_len <= 64 else 2 * TOL, what=f"{what} stag_agg_bwd_dp dx")
            for got, Ti, Ai, nm in ((c0, T[1], A[1], "d p0"), (c1, T[2], A[2], "d p1")):
                ref = (x.astype(np.float64) * Ti.astype(np.float64)).sum(0)
                sc = max(1.0, float(np.abs(ref).max()))
                # (the sums over a row's edges inside are T0 / T1 above: the same allowance off the default segment
                # length, and the same conditioning — soak case 1596: D = 1, four nodes, 3680 draws of both signs
                # summed into ONE number)
                assert_close_cond(got / sc, ref / sc, (np.abs(x).astype(np.float64) * Ai).sum(0) / sc,
                                  tol=TOL if 0 < seg_len <= 64 else 2 * TOL, what=f"{what} stag_agg_bwd_dp {nm}")
            # ---- [E, 1] parameters: stag_agg_bwd_edge (one channel tile) --------------------------------------
            if D <= 256 and E > 0:
                q0 = rng.uniform(0.2, 1.0, (E, 1)).astype(np.float32)
                q1 = (q0 + rng.uniform(0.6, 1.2, (E, 1))).astype(np.float32) if kind == "uniform" else rng.uniform(0.3, 0.9, (E, 1)).astype(np.float32)
                K = _lib.NOISE_NORMAL if kind == "normal" else _lib.NOISE_UNIFORM
                nz = stag_amd.EdgeNoise(g, D, K, torch.from_numpy(q0).to(dev), torch.from_numpy(q1).to(dev), relu=relu, seed=seed, offset=off)
                sp = nz.spec()
                sp = sp if not isinstance(sp, tuple) else ops._targs_to_ctypes(sp)
                dx3, e0, e1 = ops._agg_bwd_edge_raw(g.csr_t, gd, xd, D, sp, gsd, rsd, seg_len if seg_len else 64)
                ref_dx = oracle.agg_fwd(ogt, gout, oracle.make_spec(kind, q0, q1, relu=relu, seed=seed, offset=off, Dn=D, n_edges=E),
                                        src_scale=gs, dst_scale=rs)
                ab = oracle.agg_fwd(ogt, np.abs(gout), oracle.make_spec("explicit", np.abs(oracle.noise_materialize(
                    ogt, oracle.make_spec(kind, q0, q1, relu=relu, seed=seed, offset=off, Dn=D, n_edges=E), D))),
                    src_scale=gs, dst_scale=rs)
                assert_close_cond(dx3, ref_dx, ab, what=f"{what} stag_agg_bwd_edge dx")
                for dv, got in ((1, e0), (2, e1)):
                    osp = oracle.make_spec(kind, q0, q1, relu=relu, seed=seed, offset=off, Dn=D, n_edges=E, deriv=dv)
                    ref = oracle.agg_bwd_w(og, x, gout * gs[:, None], src_scale=rs, spec=osp).astype(np.float64).sum(1, keepdims=True)
                    sc = max(1.0, float(np.abs(ref).max()))
                    assert_close(got / sc, ref / sc, what=f"{what} stag_agg_bwd_edge d p{dv - 1}")


@pytest.mark.parametrize("H,F", [(8, 32), (3, 4), (4, 40)])
@pytest.mark.parametrize("kind,pmode,relu,drop", [("normal", "scalar", False, False), ("normal", "channel", True, True),
                                                  ("uniform", "channel", True, False), ("normal", "logscale", True, False)])
def test_gat_vi_parameter_gradients_in_the_kernels(dev, oracle, monkeypatch, H, F, kind, pmode, relu, drop):
    """vi=True through GAT (`rsample` weights scale the logits, stag/layers.py:123-124 + stag/zoo/gat.py:117-119): the
    draw stays in the kernels and the one-gather backward returns the FINISHED gradients of loc / scale (low / high)
    — stag_gat_bwd_dp — without an [E, H] tensor on either pass.  Against the oracle: its GAT backward with the
    materialised weights as explicit weights gives dL/dw [E, H]; dp_i[h] = sum_e dL/dw * dw/dp_i with z from the
    oracle's own standard draw.  Also d el / d er / d ft of the same call."""
    import stag_amd
    from stag_amd import _lib, ops
    rng = np.random.default_rng(H * 13 + F)
    n = 300
    g = random_graph(n, 3000, seed=H + F, hub=500, device=dev)
    E = g.number_of_edges()
    og = oracle_graph(oracle, g)
    el, er = rng.standard_normal((n, H)).astype(np.float32), rng.standard_normal((n, H)).astype(np.float32)
    ft, G = rng.standard_normal((n, H, F)).astype(np.float32), rng.standard_normal((n, H, F)).astype(np.float32)
    if pmode == "scalar":
        p0h, p1h = np.float32(0.9), np.float32(0.6)
        p0 = torch.tensor(0.9, device=dev, requires_grad=True)
        p1 = torch.tensor(0.6, device=dev, requires_grad=True)
    else:
        p0h = rng.uniform(0.2, 0.9, H).astype(np.float32)
        p1h = rng.uniform(1.0, 1.8, H).astype(np.float32) if kind == "uniform" else rng.uniform(0.3, 0.8, H).astype(np.float32)
        p0 = torch.tensor(p0h, device=dev, requires_grad=True)
        p1 = torch.tensor(np.log(p1h) if pmode == "logscale" else p1h, device=dev, requires_grad=True)
    K = _lib.NOISE_NORMAL if kind == "normal" else _lib.NOISE_UNIFORM
    # (a per-channel log-scale is exponentiated by the caller: the descriptor keeps the live exp() node)
    noise = stag_amd.EdgeNoise(g, H, K, p0, p1.exp() if pmode == "logscale" else p1, relu=relu, seed=31, offset=4,
                               differentiable=True)
    monkeypatch.setattr(stag_amd.EdgeNoise, "materialize", lambda self: (_ for _ in ()).throw(AssertionError("an [E, H] tensor was materialised")))
    attn_drop = (0.4, 77, 2) if drop else None
    t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
    out = ops.gat_aggregate(g, *t, 0.2, noise, attn_drop=attn_drop)
    out.backward(torch.from_numpy(G).to(dev))
    monkeypatch.undo()
    # ---- the oracle's statement --------------------------------------------------------------------------------
    with hw_normals(oracle, dev):
        std = oracle.noise_materialize(og, oracle.make_spec(kind, 0.0, 1.0, seed=31, offset=4, Dn=H, n_edges=E), H).astype(np.float64)
    raw = (p0h + p1h * std) if kind == "normal" else (p0h + (p1h - p0h) * std)             # [E, H] by edge id
    keep, keep_prob = None, 1.0
    if drop:
        keep_prob = float(np.float32(1.0 - 0.4))
        keep = stag_amd.EdgeNoise(g, H, _lib.NOISE_BERNOULLI, keep_prob, seed=77, offset=2)
        keep = ops.materialize_noise(g, keep).cpu().numpy()
    spec = oracle.make_spec("explicit", raw.astype(np.float32), relu=relu)
    ref_out = oracle.gat_fwd(og, el, er, ft, 0.2, spec, keep=keep, keep_prob=keep_prob)
    assert_close(out, ref_out, what="forward (weights drawn in the kernel) vs oracle with the materialised weights")
    d_el, d_er, d_ft, dw = oracle.gat_bwd(og, el, er, ft, G, 0.2, spec, keep=keep, keep_prob=keep_prob, want_dw=True)
    dw = dw.astype(np.float64)
    d0 = dw.sum(0) if kind == "normal" else (dw * (1.0 - std)).sum(0)
    d1 = (dw * std).sum(0)
    if pmode == "logscale":
        d1 = d1 * p1h                                  # d / d log(scale)
    if pmode == "scalar":
        d0, d1 = d0.sum(), d1.sum()
    for got, ref, nm in ((p0.grad, d0, "d p0"), (p1.grad, d1, "d p1")):
        ref = np.asarray(ref, np.float64)
        sc = max(1.0, float(np.abs(ref).max()), float(np.abs(dw).max()))
        assert_close(got.cpu().numpy().reshape(ref.shape) / sc, ref / sc, what=f"{nm} ({kind}, {pmode}, relu={relu}, drop={drop})")
    for got, ref, nm in ((t[0].grad, d_el, "d el"), (t[1].grad, d_er, "d er"), (t[2].grad, d_ft, "d ft")):
        sc = max(1.0, float(np.abs(ref).max()))
        assert_close(got / sc, ref.astype(np.float64) / sc, what=nm)
    # the materialised form (A/B switch) gives the same gradients
    ops._GAT_VI_FUSED = False
    try:
        q0, q1 = p0.detach().clone().requires_grad_(True), p1.detach().clone().requires_grad_(True)
        nz = stag_amd.EdgeNoise(g, H, K, q0, q1.exp() if pmode == "logscale" else q1, relu=relu, seed=31, offset=4, differentiable=True)
        t2 = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
        ops.gat_aggregate(g, *t2, 0.2, nz, attn_drop=attn_drop).backward(torch.from_numpy(G).to(dev))
    finally:
        ops._GAT_VI_FUSED = True
    for got, ref, nm in ((p0.grad, q0.grad, "d p0"), (p1.grad, q1.grad, "d p1")):
        sc = max(1.0, float(ref.abs().max()), float(np.abs(dw).max()))
        assert_close(got / sc, (ref / sc).cpu().numpy(), what=nm + " fused vs materialised")


def test_fuzz_gat_vi_parameter_gradients(dev, oracle):
    """Seeded sweep of stag_gat_bwd_dp over graphs x head shapes x Normal / Uniform x scalar / per-head x relu x dropout:
    the parameter gradients finished inside the kernels against the materialised form of the same layer (explicit
    [E, H] weights + autograd: a different code path end to end), and d el / d er / d ft against the oracle."""
    import stag_amd
    from stag_amd import _lib, ops
    rng = np.random.default_rng(20261011)
    shapes = [(1, 4), (2, 8), (3, 4), (4, 16), (8, 32), (8, 64), (16, 64), (5, 12), (2, 256), (16, 8), (6, 40), (8, 8)]
    for it in range(24 * FUZZ_SCALE):
        n = int(rng.integers(2, 400))
        g = random_graph(n, int(rng.integers(1, 3000)), seed=15000 + it, hub=int(rng.choice([0, 0, 90, 600])) if n > 4 else 0, device=dev)
        E = g.number_of_edges()
        og = oracle_graph(oracle, g)
        H, F = shapes[it % len(shapes)]
        kind = ["normal", "uniform"][it % 2]
        per_head, relu, drop = bool(rng.random() < 0.6), bool(rng.random() < 0.4), bool(rng.random() < 0.3)
        K = _lib.NOISE_NORMAL if kind == "normal" else _lib.NOISE_UNIFORM
        lo, hi = (0.2, 0.9), ((1.0, 1.8) if kind == "uniform" else (0.3, 0.8))
        mkp = lambda r: (torch.tensor(rng.uniform(*r, H).astype(np.float32), device=dev) if per_head
                         else torch.tensor(float(rng.uniform(*r)), device=dev))
        p0v, p1v = mkp(lo), mkp(hi)
        el, er = rng.standard_normal((n, H)).astype(np.float32), rng.standard_normal((n, H)).astype(np.float32)
        ft, G = rng.standard_normal((n, H, F)).astype(np.float32), rng.standard_normal((n, H, F)).astype(np.float32)
        seed, off = int(rng.integers(0, 2**40)), int(rng.integers(0, 99))
        attn_drop = (float(rng.choice([0.2, 0.6])), int(rng.integers(0, 2**30)), int(rng.integers(0, 50))) if drop else None
        what = f"gat vi fuzz {it}: n={n} E={E} H={H} F={F} {kind} per_head={per_head} relu={relu} drop={attn_drop}"
        res = []
        for fused in (True, False):
            ops._GAT_VI_FUSED = fused
            try:
                p0, p1 = p0v.clone().requires_grad_(True), p1v.clone().requires_grad_(True)
                nz = stag_amd.EdgeNoise(g, H, K, p0, p1, relu=relu, seed=seed, offset=off, differentiable=True)
                t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (el, er, ft)]
                out = ops.gat_aggregate(g, *t, 0.2, nz, attn_drop=attn_drop)
                out.backward(torch.from_numpy(G).to(dev))
                zero = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
                res.append((out.detach(), zero(p0), zero(p1), [a.grad for a in t]))
            finally:
                ops._GAT_VI_FUSED = True
        (o_f, d0_f, d1_f, g_f), (o_m, d0_m, d1_m, g_m) = res
        assert_close(o_f, o_m.cpu().numpy(), what=what + " out")
        with torch.no_grad():
            wm = stag_amd.EdgeNoise(g, H, K, p0v, p1v, relu=relu, seed=seed, offset=off).materialize() if E else torch.zeros(0, H, device=dev)
        keep, keep_prob = None, 1.0
        if drop and E:
            keep_prob = float(np.float32(1.0 - attn_drop[0]))
            keep = stag_amd.EdgeNoise(g, H, _lib.NOISE_BERNOULLI, keep_prob, seed=attn_drop[1], offset=attn_drop[2]).materialize().cpu().numpy()
        spec = oracle.make_spec("explicit", wm.cpu().numpy()) if E else oracle.make_spec("none")
        _, _, _, dw = oracle.gat_bwd(og, el, er, ft, G, 0.2, spec, keep=keep, keep_prob=keep_prob, want_dw=bool(E))
        sc_p = max(1.0, float(np.abs(dw).sum(0).max()) if E else 0.0)          # a parameter gradient sums E terms of d w
        assert_close(d0_f.reshape(-1) / sc_p, (d0_m.reshape(-1) / sc_p).cpu().numpy(), what=what + " d p0")
        assert_close(d1_f.reshape(-1) / sc_p, (d1_m.reshape(-1) / sc_p).cpu().numpy(), what=what + " d p1")
        if E:
            assert_gat_grads_vs_oracle(oracle, og, el, er, ft, G, spec, g_f, keep=keep, keep_prob=keep_prob, what=what)


@pytest.mark.parametrize("W", [1, 3, 8, 50, 128, 264])
def test_gather_rows_entry_point(dev, W):
    """stag_gather_rows (ABI v17): out[i, :] = x[idx[i], :] into an existing buffer — the send rows of the halo exchange.
    Bit-exact (it copies); widths that are not a multiple of 4 and strided rows take the scalar path; n = 0 is a no-op."""
    from stag_amd import _lib
    rng = np.random.default_rng(W)
    n_src, n = 1000, 3777
    ld = W + (4 if W % 2 else 0)                          # a row stride wider than the row
    xs = torch.randn(n_src, ld, device=dev)
    idx = torch.from_numpy(rng.integers(0, n_src, n).astype(np.int32)).to(dev)
    out = torch.full((n, W), float("nan"), device=dev)
    with _lib.on_device(dev):
        rc = _lib.lib().stag_gather_rows(_lib.ptr(xs), ld, _lib.ptr(idx), n, W, _lib.ptr(out), W, _lib.stream_of(dev))
        assert rc == 0
        assert torch.equal(out, xs[:, :W][idx.long()])
        assert _lib.lib().stag_gather_rows(_lib.ptr(xs), ld, _lib.ptr(idx), 0, W, _lib.ptr(out), W, _lib.stream_of(dev)) == 0
        assert _lib.lib().stag_gather_rows(_lib.ptr(xs), W - 1 if W > 1 else 0, _lib.ptr(idx), n, W, _lib.ptr(out), W, None) == -22
        assert _lib.lib().stag_gather_rows(None, ld, _lib.ptr(idx), n, W, _lib.ptr(out), W, None) == -22


def test_fuzz_monte_carlo_batches(dev, oracle):
    """Seeded sweep of stag_agg_fwd_mc over graphs x widths x kinds x in-norm x relu x reduce x S: every sample of the
    batched launch equals its own launch bit for bit (2 samples per pass with in-norm, 4 without), and one sample per
    case is held against the oracle."""
    import copy
    from stag_amd import ops
    rng = np.random.default_rng(20261012)
    for it in range(40 * FUZZ_SCALE):
        n = int(rng.integers(2, 400))
        g = random_graph(n, int(rng.integers(1, 4000)), seed=17000 + it, hub=int(rng.choice([0, 0, 100, 900])) if n > 4 else 0, device=dev)
        D = int(rng.choice([1, 3, 8, 16, 50, 64, 128, 260]))
        kind = ["normal", "uniform", "bernoulli"][it % 3]
        in_norm, relu = bool(rng.random() < 0.5), bool(rng.random() < 0.3)
        if kind == "normal" and in_norm:
            relu = True       # (in-norm divides by the sum of a row's weights: signed Normal weights can cancel there, and
                              #  the factor deg / sum then amplifies one rounding without bound — a conditioning matter
                              #  of the statement itself, stag/layers.py:24-28, not of any implementation)
        S, stride = int(rng.integers(2, 8)), int(rng.integers(1, 6))
        seg_len = int(rng.choice([64, 64, 16, 0]))
        reduce = "mean" if rng.random() < 0.3 else "sum"
        p0 = torch.from_numpy(rng.uniform(0.3, 0.9, D).astype(np.float32)).to(dev)
        p1 = None if kind == "bernoulli" else torch.from_numpy(rng.uniform(1.0, 1.6, D).astype(np.float32)).to(dev)
        kw = dict(relu=relu, in_norm=in_norm, seed=int(rng.integers(0, 2**40)), offset=int(rng.integers(0, 99)))
        noise = _noise(g, D, kind, p0, p1, **kw)
        x = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).to(dev)
        ds = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32)).to(dev)
        what = f"mc fuzz {it}: n={n} E={g.number_of_edges()} D={D} {kind} in_norm={in_norm} relu={relu} S={S} seg={seg_len} {reduce}"
        got = ops.aggregate_mc(g, x, noise, S, offset_stride=stride, reduce=reduce, dst_scale=ds, seg_len=seg_len)
        assert got.shape == (S, n, D), what
        for s in range(S):
            nz = copy.copy(noise)
            nz.offset = kw["offset"] + s * stride
            assert torch.equal(got[s], ops.aggregate(g, x, nz, reduce=reduce, dst_scale=ds, seg_len=seg_len)), (what, s)
        s = int(rng.integers(0, S))
        spec = _ospec(oracle, g, D, kind, p0, p1, **{**kw, "offset": kw["offset"] + s * stride})
        with hw_normals(oracle, dev):
            ref = oracle.agg_fwd(oracle_graph(oracle, g), x.cpu().numpy(), spec,
                                 reduce=oracle.REDUCE_MEAN if reduce == "mean" else oracle.REDUCE_SUM, dst_scale=ds.cpu().numpy())
        tol = TOL if 0 < seg_len <= 64 else 2 * TOL
        assert_close(got[s], ref, tol=tol, what=what + f" sample {s} vs oracle")


@pytest.mark.parametrize("kind", ["normal", "bernoulli", "none"])
@pytest.mark.parametrize("D", [72, 128])
def test_small_plan_twin_changes_no_bit(dev, kind, D):
    """Plain plan-order launches at 32 lanes per row take a twin instantiation when the plan holds at most
    STAG_SMALL_UNITS (49,152) units: heavy units on two edge slots (csrc/agg_kernel.hpp, heavy_slots_of).
    The same rows inside a graph that is NOT small — the graph with 30,000 edge-less rows appended: every edge keeps its
    CSR position, so its Philox counters — must come out bit for bit the same, hub rows (segments) and heavy rows included."""
    import stag_amd
    from stag_amd import _lib, ops
    rng = np.random.default_rng(29)
    n = 30000
    deg = np.minimum(rng.zipf(1.7, n), 3000)          # many rows of 17..64 edges (heavy units), a few long ones (segments)
    deg[:3] = (5000, 700, 65)
    dst = np.repeat(np.arange(n), deg)
    src = rng.integers(0, n, len(dst))
    x = torch.randn(n + 30000, D, generator=torch.Generator().manual_seed(3)).to(dev)
    outs = []
    for extra in (0, 30000):
        g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n + extra, device=dev)
        p = g.csr.plan(64)
        assert (p["n_units"] <= 49152) == (extra == 0) and p["n_heavy"] > 0 and p["n_long"] >= 3
        if kind == "none":
            noise = None
        else:
            k, p0, p1 = {"normal": (_lib.NOISE_NORMAL, 1.0, 0.5), "bernoulli": (_lib.NOISE_BERNOULLI, 0.5, None)}[kind]
            noise = stag_amd.EdgeNoise(g, D, k, p0, p1, seed=77, offset=5, in_norm=(kind == "bernoulli"))
        outs.append(ops.aggregate(g, x[:n + extra], noise)[:n].clone())
    assert torch.equal(outs[0], outs[1])
    assert bool(torch.isfinite(outs[0]).all()) and float(outs[0].abs().max()) > 0
