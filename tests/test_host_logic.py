"""Host-side logic on CPU (no kernel launch): graph build, launch plan, ABI surface,
noise descriptors, the distributions API, loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    """libstag_hip.so loads without a GPU and exports each function include/stag_hip.h declares."""
    from stag_amd import _lib
    lib = _lib.lib()
    header = open(os.path.join(ROOT, "include", "stag_hip.h")).read()
    names = set(re.findall(r"\b(stag_[a-z0-9_]+)\s*\(", header))
    assert {"stag_agg_fwd", "stag_noise_materialize", "stag_gat_fwd", "stag_plan_fill"} <= names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/stag_hip.h but not exported"
    assert lib.stag_abi_version() == 19
    assert lib.stag_strerror(-22) == b"invalid argument"


def test_torch_library_front_end_loads_and_traces():
    """stag_amd/_stag_torch.so (csrc/torch_ext.cpp): the TORCH_LIBRARY ops over the same C ABI load without a GPU,
    agree with the library on the ABI version, and have Meta kernels (shape inference for tracing / torch.compile)."""
    from stag_amd import _torch_ext
    assert os.path.exists(os.path.join(ROOT, "stag_amd", "_stag_torch.so")), "build with make -C stag_amd/csrc"
    assert _torch_ext.loaded() and not _torch_ext.available()      # eager mode keeps ctypes (it is faster)
    assert int(torch.ops.stag.abi_version()) == 19
    ip = torch.zeros(6, dtype=torch.int32, device="meta")
    ix = torch.zeros(9, dtype=torch.int32, device="meta")
    x = torch.zeros(5, 12, device="meta")
    noise = ([2, 0, 0, 1, 0, 0, 0, 0], [1, 2, 0], [1.0, 0.5], None, None, None)
    plan = (None, None, None, None, None, None, [0] * 8)
    out, ns = torch.ops.stag.agg_fwd(ip, ix, None, None, 5, *plan, x, False, *noise, 0, None, None, True)
    assert out.shape == (5, 12) and ns.shape == (5, 12) and out.device.type == "meta"
    dx, t0, t1 = torch.ops.stag.agg_bwd(ip, ix, None, None, 5, *plan, x, *noise, None, None, True)
    assert dx.shape == t0.shape == t1.shape == (5, 12)
    # (round 4) the rest of the hot surface: Monte-Carlo batches, one-pass parameter gradients, GAT forward and backward
    mc = torch.ops.stag.agg_fwd_mc(ip, ix, None, None, 5, *plan, x, *noise, 3, 1, 0, None, None)
    assert mc.shape == (3, 5, 12) and mc.device.type == "meta"
    dx, dp0, dp1 = torch.ops.stag.agg_bwd_dp(ip, ix, None, None, 5, *plan, x, x, *noise, None, None, True)
    assert dx.shape == (5, 12) and dp0.shape == dp1.shape == (12,)
    el, ft = torch.zeros(5, 2, device="meta"), torch.zeros(5, 2, 4, device="meta")
    out, stats = torch.ops.stag.gat_fwd(ip, ix, None, None, 5, *plan, el, el, ft, 0.2, *noise, None, [0.6], [1, 2], None, True)
    assert out.shape == (5, 2, 4) and stats.shape == (5, 4)
    plan_t = plan[:4] + plan[5:]
    d_el, d_er, d_ft, dw = torch.ops.stag.gat_bwd(ip, ix, None, None, 5, *plan, ip, ix, None, None, *plan_t, el, el, ft, stats,
                                                  ft, ft, 0.2, *noise, None, [], [], None, True)
    assert d_el.shape == d_er.shape == (5, 2) and d_ft.shape == (5, 2, 4) and dw.shape == (9, 2)
    # a CPU tensor is refused by the Python layer before the dispatcher would fail to find a CPU kernel
    import stag_amd
    from stag_amd._lib import StagHipError
    with pytest.raises(StagHipError):
        stag_amd.ops.aggregate(stag_amd.rand_graph(3, 9), torch.randn(3, 4), None)


def test_abi_struct_layouts_match_header():
    from stag_amd import _lib
    assert ctypes.sizeof(_lib.Csr) == 48
    assert ctypes.sizeof(_lib.NoiseSpec) == 88
    assert ctypes.sizeof(_lib.Plan) == 96 and _lib.Plan.xcd_order.offset == 80
    # stag_concat_job: graph._ConcatJobs lays these out with numpy — int64 columns 0..2, int32 columns 6 and 7
    assert ctypes.sizeof(_lib.ConcatJob) == 32 and _lib.ConcatJob.count.offset == 16 \
        and _lib.ConcatJob.add.offset == 24 and _lib.ConcatJob.kind.offset == 28
    assert _lib.NoiseSpec.deriv.offset == 40 and _lib.NoiseSpec.seed.offset == 48 and _lib.NoiseSpec.pos_base.offset == 64


def test_graph_csr_matches_oracle(oracle):
    import stag_amd
    rng = np.random.default_rng(0)
    src, dst = rng.integers(0, 70, 900), rng.integers(0, 70, 900)
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), 70)
    indptr, indices, eid, ind, outd = oracle.csr_build(src, dst, 70, 70)
    assert np.array_equal(g.csr.indptr.numpy(), indptr)
    assert np.array_equal(g.csr.indices.numpy(), indices)
    assert np.array_equal(g.csr.eid.numpy(), eid)
    assert np.array_equal(g.in_degrees().numpy(), ind) and np.array_equal(g.out_degrees().numpy(), outd)
    t = g.csr_t
    indptr_t, indices_t, eid_t, *_ = oracle.csr_build(dst, src, 70, 70)
    assert np.array_equal(t.indptr.numpy(), indptr_t) and np.array_equal(t.indices.numpy(), indices_t)
    assert np.array_equal(t.eid.numpy(), eid_t)
    assert np.array_equal(eid[t.nidx.numpy()], eid_t)     # nidx = forward position of the same edge
    og = oracle.CsrGraph(indptr, indices, eid, n_src=70).transpose()
    assert np.array_equal(og.nidx, t.nidx.numpy())


def test_graph_surface():
    import stag_amd
    g = stag_amd.rand_graph(3, 9)                      # stag/tests/test_layers.py:17
    assert g.number_of_nodes() == 3 and g.number_of_edges() == 9 and not g.is_block
    g.ndata["h"] = torch.ones(3, 2)
    h = g.local_var()
    h.ndata["h"] = torch.zeros(3, 2)
    h.edata["w"] = torch.ones(9)
    assert g.ndata["h"].sum() == 6 and "w" not in g.edata       # caller's frames untouched
    with g.local_scope():
        g.ndata["tmp"] = torch.zeros(3)
    assert "tmp" not in g.ndata
    assert h.csr is g.csr                                       # structure is shared and cached
    with pytest.raises(ValueError):
        stag_amd.Graph(torch.tensor([0, 5]), torch.tensor([1, 1]), 3)
    g2 = stag_amd.add_reverse_edges(stag_amd.add_self_loop(stag_amd.remove_self_loop(g)))
    assert g2.number_of_edges() == 2 * (int((g._src != g._dst).sum()) + 3)
    b = stag_amd.batch([stag_amd.rand_graph(4, 5), stag_amd.rand_graph(2, 3)])
    assert b.number_of_nodes() == 6 and b.batch_size == 2 and b.batch_num_nodes().tolist() == [4, 2]
    assert int(b._dst[5:].min()) >= 4


@pytest.mark.parametrize("seg_len", [1, 3, 64])
def test_plan_covers_every_edge_once(seg_len):
    import stag_amd
    rng = np.random.default_rng(seg_len)
    n = 300
    dst = np.concatenate([rng.integers(0, n - 5, 2000), np.full(700, 7), np.full(150, 9)])
    src = rng.integers(0, n, len(dst))
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    p = g.csr.plan(seg_len)
    units = p["units"].numpy()
    indptr = g.csr.indptr.numpy()
    deg = np.diff(indptr)
    assert p["n_units"] == len(units) == (deg <= seg_len).sum() + p["n_seg"]
    ns = p["n_seg"]
    assert (units[:ns, 3] == np.arange(ns)).all() and (units[ns:, 3] == -1).all(), "segments first"
    assert (np.diff(units[ns:, 2]) <= 0).all(), "whole rows sorted by length, longest first"
    assert units[:, 2].max() <= seg_len
    if ns:
        assert units[:ns, 2].min() >= (seg_len + 1) // 2, "segments are balanced"
    covered = np.zeros(len(dst), np.int32)
    long_rows, lsp = p["long_rows"].numpy(), p["long_seg_ptr"].numpy()
    for row, start, ln, slot in units:
        covered[start:start + ln] += 1
        if slot < 0:
            assert start == indptr[row] and ln == deg[row]
        else:
            v = long_rows[row]
            assert indptr[v] <= start and start + ln <= indptr[v + 1]
            assert lsp[row] <= slot < lsp[row + 1]
    assert (covered == 1).all()
    assert p["n_long"] == (deg > seg_len).sum()
    ld = deg[long_rows[:p["n_long"]]]
    assert (np.diff(ld) <= 0).all(), "hub rows first"
    assert set(units[units[:, 3] >= 0][:, 3]) == set(range(p["n_seg"]))


@pytest.mark.parametrize("seg_len,fine", [(3, 1), (64, 1), (64, 3), (16, 16)])
def test_xcd_order_is_a_stable_partition_of_the_plan(seg_len, fine, monkeypatch):
    """stag_plan_xcd (stag_plan.xcd_order): the plan's unit records grouped by the eighth of the CSR their first edge lies
    in (and inside the eighth by `fine` finer row ranges, walked one after the other) — heavy prefix and the rest
    separately, the plan's own order inside a fine stripe, every XCD stripe padded with null records to the longest one —
    with the stripe sizes, the two strides and `fine` in the header."""
    import importlib
    import stag_amd
    from stag_amd import _lib
    G = importlib.import_module("stag_amd.graph")
    rng = np.random.default_rng(seg_len)
    n = 3000
    dst = np.concatenate([rng.integers(0, n - 40, 20000), np.full(900, 7), np.full(200, 2500)])
    src = rng.integers(0, n, len(dst))
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    # "auto": only where an XCD's L2 would find its rows again — not on uniformly random sources (1/8 of the edges stay
    # inside their stripe), yes on a block-diagonal batch
    # ... and only for a view that keeps being launched (a fresh minibatch graph would pay more than it gains)
    many = lambda view: [view.plan(seg_len) for _ in range(G.XCD_AFTER_LAUNCHES + 1)][-1]
    assert G.XCD_ORDER == "auto" and abs(g.csr.stripe_locality() - 0.125) < 0.02 and many(g.csr)["xcd"] is None
    from stag_amd import synthetic
    s3, d3, sizes = synthetic.ppi_like(n_graphs=12, n_nodes=2400, n_edges=20000, seed=2)
    gb = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), int(sizes.sum()))
    assert gb.csr.stripe_locality() > 0.6 and gb.csr.plan(seg_len)["xcd"] is None and many(gb.csr)["xcd"] is not None
    monkeypatch.setattr(G, "XCD_ORDER", "1")
    monkeypatch.setattr(G, "XCD_FINE", fine)
    assert _lib.lib().stag_plan_xcd_fine(56944) == 4 and _lib.lib().stag_plan_xcd_fine(100) == 1 \
        and _lib.lib().stag_plan_xcd_fine(10 ** 7) == 16
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    p = g.csr.plan(seg_len)
    nu, nh, E = p["n_units"], p["n_heavy"], len(dst)
    units = p["units"].numpy()[:nu]
    xcd = p["xcd"].numpy()
    sh, sl = p["xcd_strides"]
    assert len(xcd) == _lib.XCD_HEADER + 4 * 8 * (sh + sl)
    ch, cl = xcd[0:8], xcd[8:16]                    # units per heavy stripe, per other stripe
    assert ch.sum() == nh and cl.sum() == nu - nh and (sh, sl) == (ch.max(), cl.max()) == (xcd[16], xcd[17])
    assert (xcd[19:_lib.XCD_HEADER] == 0).all() and xcd[18] == fine
    rec = xcd[_lib.XCD_HEADER:].reshape(8 * (sh + sl), 4)
    fstripe = np.minimum(units[:, 1].astype(np.int64) * (8 * fine) // E, 8 * fine - 1)     # fine stripe of every unit
    null = np.array([-1, 0, 0, -1], np.int32)
    for cnt, base, stride, lo, hi in ((ch, 0, sh, 0, nh), (cl, 8 * sh, sl, nh, nu)):
        for k in range(8):
            # the XCD stripe's fine stripes one after the other, the plan's order inside each
            want = np.concatenate([units[lo:hi][fstripe[lo:hi] == k * fine + f] for f in range(fine)])
            got = rec[base + k * stride: base + (k + 1) * stride]
            assert len(want) == cnt[k] and (got[:cnt[k]] == want).all() and (got[cnt[k]:] == null).all()
    # stripes are contiguous destination-row ranges: whole rows of stripe k lie below those of stripe k + 1
    tops = [rec[8 * sh + k * sl: 8 * sh + k * sl + cl[k], 0] for k in range(8)]
    for a, b in zip(tops[:-1], tops[1:]):
        if len(a) and len(b):
            assert a.max() < b.min()
    # a sub-plan carries its own
    keep = np.ones(nu, bool)
    keep[p["n_seg"]::3] = False
    sp = g.csr.subplan(seg_len, keep)
    srec = sp["xcd"].numpy()[_lib.XCD_HEADER:].reshape(-1, 4)
    srec = srec[srec[:, 0] >= 0]
    assert sorted(map(tuple, srec)) == sorted(map(tuple, units[keep]))


def test_batch_concatenates_the_parts_csr():
    """graph.batch (dgl.batch, scripts/ppi_mle/run.py:12-14): the union's CSR views are the parts' own views laid end to
    end — array for array what build_csr makes of the union's COO (edge ids and the transposed view's forward positions
    included), so the noise a batched graph draws does not depend on how its CSR came about; parts without edges, a
    one-node part; beyond BATCH_CONCAT_MAX_GRAPHS parts the union is sorted as before."""
    import importlib
    import stag_amd
    G = importlib.import_module("stag_amd.graph")
    rng = np.random.default_rng(0)
    parts = []
    for i in range(7):
        n = int(rng.integers(1, 40)) if i != 3 else 1
        e = int(rng.integers(0, 200)) if i not in (2, 3) else 0
        parts.append(stag_amd.Graph(torch.from_numpy(rng.integers(0, n, e)), torch.from_numpy(rng.integers(0, n, e)), n))
    b = stag_amd.batch(parts)
    assert b._csr is not None and b._csr_t is not None, "taken from the parts"
    ref = stag_amd.Graph(*b.edges(), b.number_of_nodes())
    for v in ("csr", "csr_t"):
        got, want = getattr(b, v), getattr(ref, v)
        for f in ("indptr", "indices", "eid", "nidx"):
            x, y = getattr(got, f), getattr(want, f)
            assert (x is None and y is None) or (x.dtype == torch.int32 and torch.equal(x, y)), (v, f)
    assert b.batch_num_nodes().tolist() == [g.number_of_nodes() for g in parts]
    old = G.BATCH_CONCAT_MAX_GRAPHS
    G.BATCH_CONCAT_MAX_GRAPHS = 3
    try:
        assert stag_amd.batch(parts)._csr is None
    finally:
        G.BATCH_CONCAT_MAX_GRAPHS = old


def test_batch_structure_cache():
    """graph.batch remembers the structure of a union by WHICH graphs were batched in WHICH order (a loader meets the same
    combination again; a validation set always): a hit shares CSR views and plans, gets its own frames, and never
    outlives the identity of its parts; another order is another union; BATCH_CACHE_SIZE = 0 switches it off."""
    import gc
    import importlib
    import stag_amd
    G = importlib.import_module("stag_amd.graph")
    G._batch_cache.clear()
    rng = np.random.default_rng(0)
    parts = [stag_amd.Graph(torch.from_numpy(rng.integers(0, 20, 50)), torch.from_numpy(rng.integers(0, 20, 50)), 20) for _ in range(3)]
    for p in parts:
        p.ndata["h"] = torch.randn(20, 4)
    a, b, c = stag_amd.batch(parts), stag_amd.batch(parts), stag_amd.batch(parts[::-1])
    assert a is not b and a.csr is b.csr and a.csr.plan(8, need=True) is b.csr.plan(8, need=True) and c.csr is not a.csr
    assert torch.equal(b.ndata["h"], torch.cat([p.ndata["h"] for p in parts])) and b.batch_num_nodes().tolist() == [20, 20, 20]
    b.ndata["x"] = torch.zeros(60, 1)
    assert "x" not in a.ndata and "x" not in stag_amd.batch(parts).ndata, "frames are the caller's, per call"
    assert stag_amd.batch([p.local_var() for p in parts]).csr is a.csr, "local_var() copies are the same graphs"
    others = [stag_amd.Graph(torch.from_numpy(rng.integers(0, 20, 50)), torch.from_numpy(rng.integers(0, 20, 50)), 20) for _ in range(3)]
    key_like = stag_amd.batch(others)
    assert len(G._batch_cache) == 3
    # what is kept is a frame-less structure: no entry holds a batch's feature tensors (a cached union used to
    # BE the first caller's graph, frames and all)
    assert all(not v[1].ndata and not v[1].edata for v in G._batch_cache.values())
    assert a.ndata and a._cache_owner() is not a and a._cache_owner() is b._cache_owner()
    del others
    gc.collect()
    assert len(G._batch_cache) == 2, "an entry goes as soon as one of its parts dies (weakref callback)"
    assert key_like.csr.n_edges == 150              # (the caller's graph is whole without the cache entry)
    # ... and the kept structures are bounded in BYTES, not only in number
    old_mb = G.BATCH_CACHE_MB
    G.BATCH_CACHE_MB = 1.5 * G._structure_bytes(a) / 2 ** 20
    try:
        stag_amd.batch(parts[1:] + parts[:1])
        assert len(G._batch_cache) == 1 and stag_amd.batch(parts[1:] + parts[:1]).csr is not a.csr
    finally:
        G.BATCH_CACHE_MB = old_mb
    with pytest.raises(ValueError, match="one device"):
        class _Elsewhere(stag_amd.Graph):
            device = property(lambda self: torch.device("meta"))
        stag_amd.batch([parts[0], _Elsewhere(parts[1]._src, parts[1]._dst, 20)])
    old = G.BATCH_CACHE_SIZE
    G.BATCH_CACHE_SIZE = 0
    try:
        assert stag_amd.batch(parts).csr is not a.csr
    finally:
        G.BATCH_CACHE_SIZE = old
    G._batch_cache.clear()


@pytest.mark.parametrize("fine", [1, 3])
def test_xcd_block_plan_deals_batches_to_stripes(fine, monkeypatch):
    """stag_plan_blocks_xcd: the cooperative GAT kernels' unit batches, dealt out so that batch b belongs to stripe
    b mod 8 of the destination rows — a permutation of the plan's units, every batch inside one fine stripe and inside
    the budget, the plan's order inside a fine stripe, empty batches where a stripe has run out."""
    import importlib
    import stag_amd
    from stag_amd import _lib
    G = importlib.import_module("stag_amd.graph")
    monkeypatch.setattr(G, "XCD_ORDER", "1")
    monkeypatch.setattr(G, "XCD_FINE", fine)
    rng = np.random.default_rng(fine)
    n = 3000
    dst = np.concatenate([rng.integers(0, n - 40, 20000), np.full(900, 7), np.full(200, 2500)])
    src = rng.integers(0, n, len(dst))
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    p = g.csr.plan(64, need=True)
    nu, E = p["n_units"], len(dst)
    units = p["units"].numpy()[:nu]
    U, bp, nb, f = g.csr.gat_blocks(p, 256)
    U, bp = U.numpy()[:nu], bp.numpy()
    assert f == fine and nb % 8 == 0 and len(bp) == nb + 1 and bp[0] == 0 and bp[-1] == nu and (np.diff(bp) >= 0).all()
    assert sorted(map(tuple, U)) == sorted(map(tuple, units))
    fstripe = lambda rec: np.minimum(rec[:, 1].astype(np.int64) * (8 * fine) // E, 8 * fine - 1)
    seen = [[] for _ in range(8)]
    for b in range(nb):
        rec = U[bp[b]:bp[b + 1]]
        if len(rec) == 0:
            continue
        fs = fstripe(rec)
        assert (fs == fs[0]).all() and fs[0] // fine == b % 8, "a batch lies in one fine stripe of stripe b mod 8"
        assert len(rec) <= _lib.BLOCK_UNITS and (rec[:, 2].sum() <= _lib.BLOCK_EDGES or len(rec) == 1)
        seen[b % 8].append(rec)
    for k in range(8):                       # a stripe's batches, in turn, restate its units: fine stripes in order, plan order inside
        got = np.concatenate(seen[k]) if seen[k] else np.zeros((0, 4), np.int32)
        fs = fstripe(units)
        want = np.concatenate([units[fs == k * fine + j] for j in range(fine)])
        assert (got == want).all()
    empty = int((np.diff(bp) == 0).sum())
    assert empty < 8 * 40, "only the tails of the shorter stripes are empty"


@pytest.mark.parametrize("width,merge", [(256, True), (50, False), (128, False)])
def test_xcd_order_keeps_whole_graphs_per_stripe(width, merge, monkeypatch):
    """Block-diagonal batches (scripts/ppi_mle/run.py:12-14): the XCD-aware order's stripes are whole GRAPHS, bin-packed
    by edge count (xcd_graph_ranges -> stag_plan_xcd_ranges): a graph never straddles two stripes, no fine range mixes
    part of a graph with another graph, the stripes carry near-equal edge counts, the records are a permutation of the
    plan's units in the plan's order inside every fine range — and for the wide shapes (no slotted loop for heavy units)
    there is ONE family of stripes, so an XCD passes over each fine range once."""
    import importlib
    import stag_amd
    from stag_amd import _lib, synthetic
    G = importlib.import_module("stag_amd.graph")
    monkeypatch.setattr(G, "XCD_ORDER", "1")
    monkeypatch.setattr(G, "XCD_RANGE_BYTES", 400_000)          # (small graphs: a small budget)
    s3, d3, sizes = synthetic.ppi_like(n_graphs=24, n_nodes=12000, n_edges=170000, seed=5)
    n = int(sizes.sum())
    g = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), n, batch_num_nodes=torch.from_numpy(sizes))
    view = g.csr
    assert view.xcd_ranges(128) is None and view.xcd_ranges(256) is not None, "by default: rows of 1 KB and up (GRAPHS_ABOVE)"
    view.xcd_graphs = True                     # every width, to test the builders
    view._part_cuts = None
    p = view.plan(64, need=True)
    nu, nh = p["n_units"], p["n_heavy"]
    units = p["units"].numpy()[:nu]
    node_off = np.concatenate([[0], np.cumsum(sizes)])
    indptr = view.indptr.numpy().astype(np.int64)
    edge_cuts = indptr[node_off]
    cuts, keys, fine = G.xcd_graph_ranges(edge_cuts, sizes, width)
    assert cuts[0] == 0 and cuts[-1] == len(s3) and (np.diff(cuts) >= 0).all() and keys.min() >= 0 and keys.max() < 8 * fine
    assert 1 <= fine <= _lib.XCD_FINE_MAX
    # graph of a CSR position, stripe of a graph
    graph_of = lambda pos: np.clip(np.searchsorted(edge_cuts, pos, side="right") - 1, 0, len(sizes) - 1)
    key_of = lambda pos: keys[np.clip(np.searchsorted(cuts, pos, side="right") - 1, 0, len(keys) - 1)]
    mids = (edge_cuts[:-1] + edge_cuts[1:]) // 2
    stripe_of_graph = key_of(mids) // fine
    for gi in range(len(sizes)):                 # every edge position of a graph has the graph's stripe
        pos = np.arange(edge_cuts[gi], edge_cuts[gi + 1], max(1, (edge_cuts[gi + 1] - edge_cuts[gi]) // 50))
        assert (key_of(pos) // fine == stripe_of_graph[gi]).all(), "a graph never straddles two stripes"
    load = np.bincount(stripe_of_graph, weights=np.diff(edge_cuts), minlength=8)
    assert load.max() <= 1.12 * load.mean(), f"stripes by edge count: {load}"
    # no fine range mixes a PART of a graph with another graph
    for k in np.unique(keys):
        gs = np.unique(graph_of(cuts[:-1][keys == k]))
        if len(gs) > 1:
            for gi in gs:
                assert (keys[(cuts[:-1] >= edge_cuts[gi]) & (cuts[:-1] < max(edge_cuts[gi + 1], edge_cuts[gi] + 1))] == k).all()
    # the library's order from that table
    order, (sh, sl), tag = view.xcd_order(p, width)
    assert tag == 1000 + min(width, 256) + (512 if merge else 0) and view.xcd_ranges(width)[2] == fine
    if merge:       # a launch that draws keeps the heavy units' own stripes
        o2, (sh2, _), tag2 = view.xcd_order(p, width, drawn=True)
        assert tag2 == 1000 + min(width, 256) and sh2 > 0 and o2 is not order
    o = order.numpy()
    rec = o[_lib.XCD_HEADER:].reshape(-1, 4)
    if merge:
        assert sh == 0 and list(o[:8]) == [0] * 8, "one family of stripes for the wide shapes"
    else:
        assert sh > 0 and o[:8].sum() == nh
    assert o[:16].sum() == nu and o[16] == sh and o[17] == sl and o[18] == fine
    real = rec[rec[:, 0] >= 0]
    assert sorted(map(tuple, real)) == sorted(map(tuple, units))
    heavy_idx = {tuple(u): i for i, u in enumerate(units)}
    for fam, (base, stride) in enumerate(((0, sh), (8 * sh, sl))):
        for k in range(8):
            blk = rec[base + k * stride: base + (k + 1) * stride]
            blk = blk[blk[:, 0] >= 0]
            if len(blk) == 0:
                continue
            assert (key_of(blk[:, 1].astype(np.int64)) // fine == k)[blk[:, 2] > 0].all(), "units lie in their graph's stripe"
            kk = key_of(blk[:, 1].astype(np.int64))
            assert (np.diff(kk) >= 0).all(), "fine ranges one after the other"
            for f in np.unique(kk):                     # the plan's order inside a fine range
                idx = [heavy_idx[tuple(u)] for u in blk[kk == f]]
                assert idx == sorted(idx)
                if not merge:
                    assert all((i < nh) == (fam == 0) for i in idx)
    # the cooperative GAT kernels' batches from the same table
    U, bp, nb, gtag = view.gat_blocks(p, width)
    U, bp = U.numpy()[:nu], bp.numpy()
    assert gtag == 1000 + min(width, 256) and nb % 8 == 0 and bp[-1] == nu and sorted(map(tuple, U)) == sorted(map(tuple, units))
    for b in range(nb):
        blk = U[bp[b]:bp[b + 1]]
        blk = blk[blk[:, 2] > 0]
        if len(blk):
            kk = key_of(blk[:, 1].astype(np.int64))
            assert (kk == kk[0]).all() and kk[0] // fine == b % 8
    # hundreds of small graphs per stripe (a molecule batch): contiguous runs, a few fine ranges
    s4, d4, sz4 = synthetic.molecules_like(1200)
    g4 = stag_amd.Graph(torch.from_numpy(s4), torch.from_numpy(d4), int(sz4.sum()), batch_num_nodes=torch.from_numpy(sz4))
    off4 = g4.csr.indptr.numpy().astype(np.int64)[np.concatenate([[0], np.cumsum(sz4)])]
    c4, k4, f4 = G.xcd_graph_ranges(off4, sz4, 128, range_bytes=200_000)
    assert f4 <= _lib.XCD_FINE_MAX and (np.diff(k4 // f4) >= 0).all() and len(np.unique(k4 // f4)) == 8


def test_fuzz_xcd_graph_ranges_invariants():
    """Seeded sweep of graph.xcd_graph_ranges (the range table of stag_plan_xcd_ranges): 2 ... 4096 graphs of 0 ... 60,000 rows,
    some without edges, widths 1 ... 1433, budgets from 1 KB up: at most STAG_XCD_FINE_MAX fine ranges per stripe, ascending cut
    points from 0 to E, keys inside [0, 8 * fine), and every graph that has edges lies in ONE stripe at every one of its
    positions (a graph without edges owns no position: round 4 found its range shadowing its neighbour's)."""
    import importlib
    from stag_amd import _lib
    G = importlib.import_module("stag_amd.graph")
    rng = np.random.default_rng(0)
    for it in range(400):
        ng = int(rng.choice([2, 3, 8, 9, 24, 100, 600, 4096]))
        rows = rng.integers(0 if rng.random() < 0.3 else 1, int(rng.choice([3, 50, 4000, 60000])), ng)
        e = (rows * rng.integers(0, 30, ng) * (rng.random(ng) > 0.2)).astype(np.int64)
        cuts_e = np.concatenate([[0], np.cumsum(e)])
        if cuts_e[-1] == 0:
            continue
        w = int(rng.choice([1, 9, 50, 128, 256, 1433]))
        cuts, keys, fine = G.xcd_graph_ranges(cuts_e, rows, w, range_bytes=int(rng.choice([1000, 100_000, 2_500_000])))
        assert 1 <= fine <= _lib.XCD_FINE_MAX and cuts[0] == 0 and cuts[-1] == cuts_e[-1] and (np.diff(cuts) >= 0).all()
        assert keys.min() >= 0 and keys.max() < 8 * fine and len(keys) == len(cuts) - 1
        for g in np.nonzero(e)[0]:
            pos = np.unique(np.clip(np.linspace(cuts_e[g], cuts_e[g + 1] - 1, 7).astype(np.int64), cuts_e[g], cuts_e[g + 1] - 1))
            k = keys[np.clip(np.searchsorted(cuts, pos, side="right") - 1, 0, len(keys) - 1)] // fine
            assert len(set(k.tolist())) == 1, (it, int(g), k.tolist(), fine)


@pytest.mark.parametrize("seg_len", [64, 16, 300])
def test_block_plan_batches_units(seg_len):
    """stag_plan_blocks: consecutive units of the plan in batches of at most STAG_BLOCK_EDGES edges and
    STAG_BLOCK_UNITS units (a unit longer than the budget gets a batch of its own) — what the workgroup-cooperative
    GAT kernels walk."""
    import stag_amd
    from stag_amd import _lib
    rng = np.random.default_rng(4)
    n = 2000
    dst = np.concatenate([rng.integers(0, n - 50, 9000), np.full(1500, 5), np.full(300, 77)])   # hubs + 50 empty rows
    src = rng.integers(0, n, len(dst))
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    plan = g.csr.plan(seg_len)
    bp = plan["block_ptr"].numpy()
    units = plan["units"].numpy()[:plan["n_units"]]
    assert bp[0] == 0 and bp[-1] == plan["n_units"] and len(bp) == plan["n_blocks"] + 1 and (np.diff(bp) > 0).all()
    for b0, b1 in zip(bp[:-1], bp[1:]):
        lens = units[b0:b1, 2]
        assert b1 - b0 <= _lib.BLOCK_UNITS
        assert lens.sum() <= _lib.BLOCK_EDGES or b1 - b0 == 1
        if b1 < plan["n_units"] and b1 - b0 < _lib.BLOCK_UNITS:           # closed because the next unit did not fit
            assert lens.sum() + units[b1, 2] > _lib.BLOCK_EDGES
    assert units[:, 2].sum() == g.number_of_edges()
    lib = _lib.lib()
    nb = ctypes.c_int32()
    assert lib.stag_plan_blocks(None, 3, 256, 32, None, ctypes.byref(nb)) == -22           # units missing
    assert lib.stag_plan_blocks(None, 0, 256, 32, None, ctypes.byref(nb)) == 0 and nb.value == 0


def test_subplan_partitions_the_units():
    """CsrView.subplan: complementary masks over the plan's units give two plans that cover every unit once,
    keep the plan's order, keep the segments together and the heavy units a prefix."""
    import stag_amd
    rng = np.random.default_rng(2)
    n = 300
    dst = np.concatenate([rng.integers(0, n, 2500), np.full(400, 9), np.full(70, 21)])
    src = rng.integers(0, n, len(dst))
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n)
    full = g.csr.plan(64)
    units = full["units"].numpy()[:full["n_units"]]
    keep = (units[:, 3] < 0) & (units[:, 0] % 3 == 0)
    a, b = g.csr.subplan(64, keep), g.csr.subplan(64, ~keep)
    ua, ub = a["units"].numpy()[:a["n_units"]], b["units"].numpy()[:b["n_units"]]
    assert np.array_equal(ua, units[keep]) and np.array_equal(ub, units[~keep])
    assert a["n_seg"] == 0 and a["n_long"] == 0 and b["n_seg"] == full["n_seg"] and b["n_long"] == full["n_long"]
    for p_, u in ((a, ua), (b, ub)):
        heavy = (u[:, 3] >= 0) | (u[:, 2] > 16)
        assert heavy[:p_["n_heavy"]].all() and not heavy[p_["n_heavy"]:].any()
    with pytest.raises(ValueError):
        g.csr.subplan(64, units[:, 1] % 2 == 0)          # splits the segments of a long row


def test_graph_transforms_keep_frames():
    """remove_self_loop / add_self_loop / add_reverse_edges keep node frames and the batch structure as DGL
    does; the reference's scripts read g.ndata after them (scripts/citation_mle/gcn/run.py:52-53,
    scripts/arxiv_mle/gcn/run.py:53-55)."""
    import stag_amd
    src = torch.tensor([0, 1, 1, 2, 3, 3])
    dst = torch.tensor([1, 1, 2, 0, 3, 0])
    g = stag_amd.Graph(src, dst, 4, batch_num_nodes=torch.tensor([3, 1]))
    g.ndata["feat"] = torch.arange(8.0).reshape(4, 2)
    g.ndata["train_mask"] = torch.tensor([True, False, True, False])
    g.edata["w"] = torch.arange(6.0).unsqueeze(1)
    r = stag_amd.remove_self_loop(g)
    assert r.number_of_edges() == 4 and torch.equal(r.ndata["feat"], g.ndata["feat"])
    assert torch.equal(r.edata["w"].squeeze(1), torch.tensor([0.0, 2.0, 3.0, 5.0]))     # rows of the surviving edges
    assert torch.equal(r.batch_num_nodes(), g.batch_num_nodes())
    a = stag_amd.add_self_loop(r)
    assert a.number_of_edges() == 8 and torch.equal(a.ndata["train_mask"], g.ndata["train_mask"])
    assert torch.equal(a.edata["w"].squeeze(1), torch.tensor([0.0, 2.0, 3.0, 5.0, 0.0, 0.0, 0.0, 0.0]))
    assert torch.equal(a.edges()[0][-4:], torch.arange(4)) and torch.equal(a.edges()[1][-4:], torch.arange(4))
    b = stag_amd.add_reverse_edges(a)
    assert b.number_of_edges() == 16 and "feat" in b.ndata and b.edata == {}         # DGL default: copy_edata=False
    b2 = stag_amd.add_reverse_edges(a, copy_edata=True)
    assert torch.equal(b2.edata["w"][8:], a.edata["w"])
    assert "feat" not in g.edata and g.number_of_edges() == 6                          # the input is untouched
    # the reference script's sequence (scripts/arxiv_mle/gcn/run.py:53-55)
    h = stag_amd.add_reverse_edges(stag_amd.add_self_loop(stag_amd.remove_self_loop(g)))
    assert torch.equal(h.ndata["feat"], g.ndata["feat"]) and h.number_of_nodes() == 4


def test_edge_noise_param_modes():
    import stag_amd
    from stag_amd import _lib
    g = stag_amd.rand_graph(5, 12)
    N = torch.distributions.Normal
    mk = lambda d, dn: stag_amd.EdgeNoise.from_distribution(g, dn, d)
    assert mk(N(1.0, 0.5), 8).param_mode == _lib.PARAM_SCALAR
    e = mk(N(torch.ones(8), torch.ones(8)), 8)
    assert e.param_mode == _lib.PARAM_PER_CHANNEL and e.p0.shape == (8,)
    e = mk(N(torch.ones(12, 1), torch.ones(12, 1)), 8)
    assert e.param_mode == _lib.PARAM_PER_EDGE1 and e.p1.shape == (12, 1)
    e = mk(N(torch.ones(12, 8), torch.ones(12, 1)), 8)          # mixed: widest mode wins
    assert e.param_mode == _lib.PARAM_PER_EDGE and e.p1.shape == (12, 8)
    assert e.shape == torch.Size([12, 8]) and e.unsqueeze(-1) is e
    assert mk(torch.distributions.Bernoulli(probs=0.9), 4).kind == _lib.NOISE_BERNOULLI
    assert mk(torch.distributions.Uniform(0.0, 2.0), 4).kind == _lib.NOISE_UNIFORM
    with pytest.raises(ValueError):
        mk(N(torch.ones(7), torch.ones(7)), 8)
    s = mk(N(1.0, 0.5), 8).spec()
    assert s.kind == _lib.NOISE_NORMAL and abs(s.p1_scalar - 0.5) < 1e-7


def test_noise_generator_offsets():
    from stag_amd import random as R
    gen = R.NoiseGenerator(seed=5)
    assert [gen.next_offset() for _ in range(3)] == [0, 1, 2]
    st = gen.get_state()
    gen.manual_seed(5)
    assert gen.next_offset() == 0
    gen.set_state(st)
    assert gen.next_offset() == 3


def test_no_cpu_fallback():
    """The product path fails loudly without a HIP device (no oracle, no eager fallback)."""
    import stag_amd
    from stag_amd._lib import StagHipError
    g = stag_amd.rand_graph(3, 9)
    layer = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(16, 32))
    with pytest.raises(StagHipError):
        layer(g, torch.randn(3, 16))
    with pytest.raises(StagHipError):
        stag_amd.ops.aggregate(g, torch.randn(3, 4), None)
    src = open(os.path.join(ROOT, "stag_amd", "ops.py")).read() + open(os.path.join(ROOT, "stag_amd", "layers.py")).read()
    assert "oracle" not in src


# ---- stag.distributions contracts (reference: stag/tests/test_distributions.py) ---------------
def test_parametrized_distribution_names_and_values(golden):
    from stag_amd.distributions import ParametrizedDistribution
    d = ParametrizedDistribution(torch.distributions.Normal(1.0, 0.5), vi=True)
    assert sorted(n for n, _ in d.named_parameters()) == list(golden["pd_vi_names"])
    assert np.allclose(d.loc.item(), golden["pd_vi_loc"]) and np.allclose(d.log_scale.item(), golden["pd_vi_log_scale"])
    d = ParametrizedDistribution(torch.distributions.Normal(1.0, 0.5))
    assert sorted(n for n, _ in d.named_buffers()) == list(golden["pd_buf_names"])
    assert len(list(d.parameters())) == 0
    half = 0.3 * np.sqrt(3.0)
    d = ParametrizedDistribution(torch.distributions.Uniform(1.0 - half, 1.0 + half, validate_args=False))
    assert sorted(n for n, _ in d.named_buffers()) == list(golden["pd_uniform_names"])
    assert np.allclose([d.low.item(), d.high.item()], golden["pd_uniform_low_high"])
    d = ParametrizedDistribution(torch.distributions.Bernoulli(probs=float(golden["pd_bernoulli_probs"][0])))
    assert sorted(n for n, _ in d.named_buffers()) == list(golden["pd_bernoulli_names"])   # no 'logits'
    assert isinstance(d.base_distribution, torch.distributions.Bernoulli)


def test_parametrized_distribution_expand_shapes(golden):
    from stag_amd.distributions import ParametrizedDistribution
    d = ParametrizedDistribution(torch.distributions.Normal(0, 1))
    assert d.expand(torch.Size([10, 8])).rsample().shape == (10, 8)
    d = ParametrizedDistribution(torch.distributions.Normal(torch.zeros(10, 8), torch.ones(10, 8)))
    assert list(d.expand(torch.Size([12, 11, 10, 8])).rsample().shape) == list(golden["pd_expand_shape"])
    assert d.batch_shape == (10, 8) and float(d.mean.sum()) == 0.0


def test_delta_distribution():
    from stag_amd.distributions import DeltaDistribution
    d = DeltaDistribution(0.0)
    assert d.sample() == 0.0 and d.rsample() == 0.0 and d.stddev == 0.0
    with pytest.raises(NotImplementedError):
        d.log_prob(torch.tensor(0.0))


@pytest.mark.parametrize("tag,of", [("re", 1), ("rec", 16)])
def test_amortized_distribution_condition_golden(golden, tag, of):
    """Same state_dict => same per-edge loc / log_scale as the reference computed."""
    import stag_amd
    from stag_amd.distributions import AmortizedDistribution
    q = AmortizedDistribution(16, of)
    sd = {k[len(f"amort_{tag}_sd_"):]: torch.from_numpy(golden[k]) for k in golden.files
          if k.startswith(f"amort_{tag}_sd_")}
    q.load_state_dict(sd)                                     # names are part of the contract
    g = stag_amd.Graph(torch.from_numpy(golden["hub40_src"]), torch.from_numpy(golden["hub40_dst"]), 40)
    q.condition(g, torch.from_numpy(golden[f"amort_{tag}_x"]))
    assert torch.allclose(q.new_parameters["loc"], torch.from_numpy(golden[f"amort_{tag}_loc"]), atol=1e-5)
    assert torch.allclose(q.new_parameters["log_scale"], torch.from_numpy(golden[f"amort_{tag}_log_scale"]), atol=1e-5)
    assert q.base_distribution.scale.shape == (g.number_of_edges(), of)
    init = AmortizedDistribution(4, 2, init_like=torch.distributions.Normal(1.0, 0.3))
    assert torch.allclose(init.parameters_mlp["loc"].bias, torch.ones(2))
    assert torch.allclose(init.parameters_mlp["log_scale"].bias, torch.full((2,), float(np.log(0.3))))


def test_oracle_amortized_parameters_and_kl_golden(golden, oracle):
    """The oracle's restatement of AmortizedDistribution.condition and of the Normal KL term against what the
    reference computed (fixtures amort_re, amort_rec, amort_kl: tests/golden/make_golden.py b', b'')."""
    src, dst = golden["hub40_src"], golden["hub40_dst"]
    for tag, pre in (("re", "amort_re_sd_"), ("rec", "amort_rec_sd_"), ("kl", "amort_kl_sd_q_a.")):
        sd = {k[len(pre):]: golden[k] for k in golden.files if k.startswith(pre)}
        got = oracle.amortized_parameters(
            src, dst, golden[f"amort_{tag}_x"], sd["embedding_mlp.0.weight"], sd["embedding_mlp.0.bias"],
            {n: (sd[f"parameters_mlp.{n}.weight"], sd[f"parameters_mlp.{n}.bias"]) for n in ("loc", "log_scale")})
        for n in ("loc", "log_scale"):
            ref = golden[f"amort_{tag}_{n}"]
            assert np.abs(got[n] - ref).max() <= 1e-5 * (1 + np.abs(ref).max()), (tag, n)
    kl = oracle.normal_kl_mean(golden["amort_kl_loc"], golden["amort_kl_log_scale"], golden["amort_kl_sd_p_a.loc"],
                               np.exp(golden["amort_kl_sd_p_a.log_scale"]))
    assert abs(kl - float(golden["amort_kl_value"][0])) <= 1e-5 * abs(kl)


def test_stag_layer_construction_and_state_dict():
    import stag_amd
    L = stag_amd.layers.StagLayer
    layer = L(stag_amd.zoo.GCN(16, 32))
    assert sorted(layer.state_dict()) == ["base_layer.bias", "base_layer.weight", "p_a.loc", "p_a.scale",
                                          "q_a.loc", "q_a.scale"]
    assert layer.kl_divergence() == 0.0 and layer.vi is False
    vi = L(stag_amd.zoo.GCN(16, 32), q_a=torch.distributions.Normal(1.0, 0.4), vi=True)
    assert {"q_a.loc", "q_a.log_scale", "p_a.loc", "p_a.log_scale"} <= set(dict(vi.named_parameters()))
    kl = vi.kl_divergence()        # closed form Normal||Normal (stag/layers.py:136-139)
    want = torch.distributions.kl_divergence(torch.distributions.Normal(1.0, 0.4), torch.distributions.Normal(1.0, 1.0))
    assert torch.allclose(kl, want)
    assert stag_amd.zoo.GAT(16, 4, num_heads=3).sample_dimension == 3
    for cls in (stag_amd.layers.FeatOnlyLayer, stag_amd.layers.SumNodes, stag_amd.layers.MeanNodes):
        assert cls.vi is False


def test_zoo_error_behaviour():
    import stag_amd
    from stag_amd.zoo._common import DGLError
    g = stag_amd.rand_graph(4, 6)
    with pytest.raises(AssertionError):     # stag/zoo/gcn.py:61
        stag_amd.zoo.GCN(4, 4)(g, torch.randn(4, 4), edge_weight=torch.ones(5, 4))
    with pytest.raises(DGLError):           # stag/zoo/gcn.py:77-81
        stag_amd.zoo.GCN(4, 4)(g, torch.randn(4, 4), weight=torch.ones(4, 4))
    with pytest.raises(KeyError):           # stag/zoo/graph_sage.py:101
        stag_amd.zoo.GraphSAGE(4, 4, aggregator_type="bogus")


def test_early_stopping():
    from stag_amd.utils import EarlyStopping
    es, m = EarlyStopping(patience=2), torch.nn.Linear(1, 1)
    assert es([1.0, 1.0], m) is False
    assert es([0.5, 0.9], m) is False and es.best_state is not None
    assert es([0.6, 0.8], m) is False and es.best_losses == [0.5, 0.8]
    assert es([0.7, 0.9], m) is False
    assert es([0.7, 0.9], m) is True


def test_synthetic_graph_shape():
    from stag_amd import synthetic
    src, dst = synthetic.arxiv_like(n_nodes=20000, n_edges=140000, max_in_degree=1500, seed=1)
    deg = np.bincount(dst, minlength=20000)
    assert len(src) == 140000 and deg.max() > 1000 and 0.02 < (deg == 0).mean() < 0.25
    s2, d2 = synthetic.arxiv_like(n_nodes=20000, n_edges=140000, max_in_degree=1500, seed=1)
    assert np.array_equal(src, s2) and np.array_equal(dst, d2)


def test_abi_argument_validation_without_gpu():
    """Every entry point rejects bad arguments BEFORE touching the device (errno-style codes,
    no exception across the ABI), so this runs without a GPU."""
    import ctypes as C
    from stag_amd import _lib
    lib = _lib.lib()
    EINVAL, ENOSYS = -22, -38
    indptr = np.array([0, 1, 2], np.int32)
    csr = _lib.Csr(2, 2, 2, indptr.ctypes.data, indptr.ctypes.data, None, None)   # never dereferenced
    spec = _lib.NoiseSpec()
    f = C.c_void_p(16)                       # a non-null, 16-B aligned dummy "device pointer"
    ok_args = lambda: [C.byref(csr), None, f, 4, 4, C.byref(spec), 0, None, None, f, 4, None, None]
    a = ok_args(); a[0] = None
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # no graph
    a = ok_args(); a[9] = None
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # no output
    a = ok_args(); a[3] = 2
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # ldx < D
    a = ok_args(); a[6] = 7
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # unknown reduce
    bad = _lib.NoiseSpec(); bad.kind = 9
    a = ok_args(); a[5] = C.byref(bad)
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # unknown noise kind
    bad = _lib.NoiseSpec(); bad.kind = _lib.NOISE_EXPLICIT
    a = ok_args(); a[5] = C.byref(bad)
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # explicit weights without p0
    bad = _lib.NoiseSpec(); bad.kind = _lib.NOISE_NORMAL; bad.param_mode = _lib.PARAM_PER_CHANNEL
    a = ok_args(); a[5] = C.byref(bad)
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # per-channel params without pointers
    bad = _lib.NoiseSpec(); bad.kind = _lib.NOISE_BERNOULLI; bad.deriv = 1
    a = ok_args(); a[5] = C.byref(bad)
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # Bernoulli has no reparameterised gradient
    bad = _lib.NoiseSpec(); bad.kind = _lib.NOISE_NORMAL; bad.deriv = 2; bad.in_norm = 1
    a = ok_args(); a[5] = C.byref(bad)
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # in-norm is not differentiated
    bad = _lib.NoiseSpec(); bad.kind = _lib.NOISE_NORMAL; bad.pos_base = (1 << 32) - 1
    a = ok_args(); a[5] = C.byref(bad)
    assert lib.stag_agg_fwd(*a) == ENOSYS                                   # launch would straddle 2^32 positions
    plan = _lib.Plan(64, 2, 1, 2, None, None, None, None, None, 0, 0, 0, None)
    a = ok_args(); a[1] = C.byref(plan)
    assert lib.stag_agg_fwd(*a) == EINVAL                                   # plan without units
    # GAT limits
    g = [C.byref(csr), None, f, f, f, 65, 2, 0.2, C.byref(spec), None, None, f, None, None]
    assert lib.stag_gat_fwd(*g) == ENOSYS                                   # H > 64
    g[5], g[6] = 8, 64
    assert lib.stag_gat_fwd(*g) == ENOSYS                                   # H*F > 256
    g[5], g[6] = 0, 4
    assert lib.stag_gat_fwd(*g) == EINVAL
    ns = _lib.NoiseSpec(); ns.in_norm = 1
    g = [C.byref(csr), None, f, f, f, 2, 4, 0.2, C.byref(ns), None, None, f, None, None]
    assert lib.stag_gat_fwd(*g) == EINVAL                                   # in-norm needs norm_scale
    drop = _lib.GatDrop(); drop.keep_prob = 0.0
    g = [C.byref(csr), None, f, f, f, 2, 4, 0.2, C.byref(spec), None, C.byref(drop), f, None, None]
    assert lib.stag_gat_fwd(*g) == EINVAL                                   # attention dropout that keeps nothing
    drop.keep_prob = 0.4
    assert lib.stag_gat_fwd(*g) == ENOSYS                                   # ... needs the cooperative kernel (a block plan)
    # planning on host arrays
    nu, nl, nsg = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.stag_plan_count(None, 2, 64, C.byref(nu), C.byref(nl), C.byref(nsg), None) == EINVAL
    assert lib.stag_plan_count(indptr.ctypes.data, 2, 0, C.byref(nu), C.byref(nl), C.byref(nsg), None) == EINVAL
    dec = np.array([0, 3, 1], np.int32)
    assert lib.stag_plan_count(dec.ctypes.data, 2, 64, C.byref(nu), C.byref(nl), C.byref(nsg), None) == EINVAL   # decreasing indptr
    assert lib.stag_plan_count(indptr.ctypes.data, 2, 64, C.byref(nu), C.byref(nl), C.byref(nsg), None) == 0 and nu.value == 2
    assert lib.stag_plan_workspace_bytes(0, 128, 0) == 0 and lib.stag_plan_workspace_bytes(3, 128, 1) == 3 * 128 * 2 * 4
    assert lib.stag_csr_build(None, None, 2, 2, 5, indptr.ctypes.data, None, None, None, None, 0, None) == EINVAL
    assert lib.stag_segment_reduce(f, 4, 4, None, 2, 0, f, 4, None) == EINVAL
    assert lib.stag_noise_materialize(C.byref(csr), None, C.byref(spec), 0, f, 4, None, None) == EINVAL
    assert lib.stag_philox_raw(0, 0, 0, 4, 0, f, None) == EINVAL


def test_hot_kernel_register_budget():
    """The aggregation kernels are occupancy-sensitive (one VGPR over a step costs a wave per SIMD
    and ~15 % of the launch): the compiler's resource report, saved by the Makefile next to the
    objects, must keep the wide shapes at >= 7 waves per SIMD (8 without noise)."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "stag_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "-j", "8"], check=True, stdout=subprocess.DEVNULL)

    def usage(kind_file):
        text = open(os.path.join(csrc, "_obj", f"{kind_file}.remarks")).read()
        out = {}
        for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", text, re.S):
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            out[name] = (int(m.group(2)), int(m.group(3)))
        return out
    normal, none = usage("agg_normal"), usage("agg_none")
    for lpe in (32, 64):         # D = 128 (the headline) and D >= 256
        for walk in ("false", "true"):       # the plan-order kernel and its XCD-aware twin
            # (last argument: SMALL — the two-slot twin of the plan-order kernel for shard-sized launches is not budgeted)
            v, occ = normal[f"void stag::agg_kernel<2, {lpe}, true, 0, 1, false, false, {walk}, false>(stag::AggArgs)"]
            assert occ >= 7 and v <= 72, f"Normal, LPE {lpe}, walk {walk}: {v} VGPRs, {occ} waves/SIMD"
            v, occ = none[f"void stag::agg_kernel<0, {lpe}, true, 0, 1, false, false, {walk}, false>(stag::AggArgs)"]
            assert occ >= 7 and v <= 72, f"no noise, LPE {lpe}, walk {walk}: {v} VGPRs, {occ} waves/SIMD"
    # the cooperative GAT forward asks for 40,000 B of LDS per workgroup = 4 waves per SIMD: its long-row merge may use
    # registers up to the 128 that occupancy allows (12 segment states per round trip), not more
    gat = usage("gat")
    fwd = {k: v for k, v in gat.items() if "gat_fwd_block_kernel<" in k and ", 1, 4>" in k}
    assert len(fwd) == 5 and all(v <= 128 and occ >= 4 for v, occ in fwd.values()), fwd


def test_round2_host_helpers_on_cpu():
    """Host-side pieces added late in round 2, checked without a GPU: the masked mean of the loss (index masks keep
    torch's indexing, boolean masks on CPU too), the GAT shape rules (lanes per head, which widths get padded,
    which shapes the cooperative kernels take), column_sum's CPU form, the amortised-parameter oracle's shapes."""
    import stag_amd
    from stag_amd import ops
    from stag_amd.models import _masked_mean
    v = torch.randn(30, 3)
    m = torch.rand(30) < 0.5
    assert torch.allclose(_masked_mean(v, m), v[m].mean()) and torch.allclose(_masked_mean(v, None), v.mean())
    idx = torch.tensor([0, 4, 9])
    assert torch.equal(_masked_mean(v, idx), v[idx].mean())
    assert [ops.gat_lanes_per_head(f) for f in (4, 8, 12, 32, 40, 124, 256)] == [1, 2, 4, 8, 16, 32, 64]
    assert ops.gat_cooperative_shape(8, 32, 64) and ops.gat_cooperative_shape(8, 40, 64) and ops.gat_cooperative_shape(4, 256, 64)
    assert not ops.gat_cooperative_shape(8, 10, 64) and not ops.gat_cooperative_shape(32, 8, 64)      # F % 4, H > 16
    assert not ops.gat_cooperative_shape(8, 256, 64) and not ops.gat_cooperative_shape(8, 32, 0)      # H*F > 1024, no plan
    GAT = stag_amd.zoo.GAT
    assert GAT._padded_width(8, 40) == 40 and GAT._padded_width(3, 7) == 8 and GAT._padded_width(4, 121) == 124
    assert GAT._padded_width(8, 32) == 32 and GAT._padded_width(64, 7) == 7                           # nothing fits: unchanged
    g = torch.randn(100, 7)
    assert torch.allclose(ops.column_sum(g), g.sum(0)) and torch.equal(ops.add_bias(g, torch.ones(7)), g + 1)
    assert ops.attn_drop_fusable(8, 32, 64) and not ops.attn_drop_fusable(8, 32, 64, want_attn=True)


def test_sage_lstm_aggregator_composed():
    """aggregator_type='lstm' (stag/zoo/graph_sage.py:97-99): DGL's degree-bucketed LSTM reducer over u_mul_e messages,
    composed from torch ops (no BASELINE config uses it; it must not raise).  Checked against a per-node loop with the
    same LSTM: mailbox in edge-id order, zero initial state, last hidden state, zeros for nodes without in-edges."""
    import stag_amd
    torch.manual_seed(3)
    n, D = 12, 5
    src = torch.tensor([0, 1, 2, 3, 4, 5, 6, 1, 2, 7, 8, 9, 3, 3])
    dst = torch.tensor([1, 1, 1, 2, 2, 4, 4, 4, 4, 5, 0, 0, 9, 1])       # node 3, 6, 7, 8, 10, 11: no in-edges
    g = stag_amd.Graph(src, dst, n)
    layer = stag_amd.zoo.GraphSAGE(D, 4, aggregator_type="lstm")
    assert {"lstm.weight_ih_l0", "fc_self.weight", "fc_neigh.weight", "bias"} <= set(layer.state_dict())
    x = torch.randn(n, D, requires_grad=True)
    w = torch.rand(len(src), D) + 0.5
    out = layer(g, x, edge_weight=w)
    neigh = torch.zeros(n, D)
    rows = []
    for v in range(n):
        ids = torch.nonzero(dst == v).flatten()
        if len(ids) == 0:
            rows.append(torch.zeros(D))
            continue
        box = (x[src[ids]] * w[ids]).unsqueeze(0)
        _, (h, _) = layer.lstm(box)
        rows.append(h.reshape(D))
    neigh = torch.stack(rows, 0)
    ref = layer.fc_self(x) + layer.fc_neigh(neigh) + layer.bias
    assert torch.allclose(out, ref, atol=1e-6)
    out.sum().backward()
    assert x.grad is not None and layer.lstm.weight_hh_l0.grad is not None


def test_tools_and_entry_points_compile():
    """Every script under tools/ (they run on the GPU box only) and the repo-root entry points at least parse."""
    import ast
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py")]
    assert len(files) > 20
    for f in files:
        ast.parse(open(f).read(), filename=f)
