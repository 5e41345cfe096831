"""BASELINE configs[4] as written, at FULL size, emulated on one GPU: the cfg5 GAT graph
(N = 169,343, E = 1,166,243, H = 8, F = 32) cut into 8 node-range shards, and the cfg2 aggregation cut the same way.

Every shard is the object the 8-GPU job builds (`partition.GraphShard`, both exchange layouts); only the collective is
replaced by the indexing it amounts to — `buffer = table[global id of every buffer row]` — whose autograd backward IS
the transposed exchange (a scatter-add of the buffer's gradient rows into the owners' rows).  Checked:
  * forward: the 8 shards' outputs, concatenated, are BIT-IDENTICAL to the whole-graph kernel launch — Philox
    positions in the millions (`pos_base`), the 13k-edge hub inside one shard's plan, `n_buf` remapping at 58 % halo,
    the two-table exchange (ft at H*F = 256, el), in-kernel attention dropout keyed by global position;
  * forward against the CPU oracle at 1e-5;
  * backward: per-shard `stag_gat_bwd` / transposed aggregation + the transposed exchange, summed over the shards,
    against the whole graph's `d ft / d el / d er` (`d x`) — a shard sums a source row's out-edges per shard and the
    exchange adds 8 partial sums, so this is 1e-5 relative, not bitwise.
The reference has no multi-GPU code (scripts/arxiv_mle/gcn/run.py:140-142: one process, `cuda:0`); the contract is
BASELINE.json north_star + SURVEY.md section 8e.
"""
import numpy as np
import pytest
import torch

from util import TOL, assert_close, hw_normals, oracle_graph

pytestmark = pytest.mark.gpu
WORLD = 8


@pytest.fixture(scope="module")
def arxiv(dev):
    import stag_amd
    from stag_amd import synthetic
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    return src, dst, n, g


@pytest.fixture(scope="module")
def shards(arxiv, dev):
    """The 8 shards of both exchange layouts + the global id behind every buffer row (-1: padding)."""
    from stag_amd.partition import GraphShard
    src, dst, n, _ = arxiv
    out = {}
    for exchange in ("halo", "allgather"):
        lst = []
        for r in range(WORLD):
            sh = GraphShard(src, dst, n, r, WORLD, device=dev, exchange=exchange)
            if exchange == "halo":
                gid = torch.cat([torch.arange(sh.row_lo, sh.row_hi), torch.from_numpy(sh.recv_ids)])
            else:
                gid = torch.full((sh.n_buf,), -1, dtype=torch.int64)
                for q in range(WORLD):
                    a, b = int(sh.bounds[q]), int(sh.bounds[q + 1])
                    gid[q * sh.max_rows:q * sh.max_rows + (b - a)] = torch.arange(a, b)
            assert gid.shape[0] == sh.n_buf
            lst.append((sh, gid.to(dev)))
        out[exchange] = lst
    return out


def _exchange(table, gid):
    """What the collective delivers to a shard: the table's row behind every buffer row (padding rows: zeros).
    Differentiable: its backward scatter-adds the buffer's gradient into the owners' rows = the transposed exchange."""
    rows = table[gid.clamp(min=0)]
    return rows * (gid >= 0).to(rows.dtype).unsqueeze(1) if bool((gid < 0).any()) else rows


def _rel(got, ref):
    sc = max(1.0, float(ref.abs().max()))
    return got / sc, (ref / sc).cpu().numpy()


def test_shards_cover_the_graph(arxiv, shards):
    src, dst, n, g = arxiv
    for exchange, lst in shards.items():
        assert [int(sh.bounds[r]) for r, (sh, _) in enumerate(lst)] == sorted(int(sh.bounds[r]) for r, (sh, _) in enumerate(lst))
        assert sum(sh.n_rows for sh, _ in lst) == n
        assert sum(sh.number_of_edges() for sh, _ in lst) == g.number_of_edges()
        # edge-balanced: no shard holds more than the hub row above E/8
        emax = max(sh.number_of_edges() for sh, _ in lst)
        assert emax <= g.number_of_edges() / WORLD + 14_000
        assert lst[-1][0].pos_base > 1_000_000          # positions in the millions reach the kernels
        if exchange == "halo":
            frac = np.mean([sum(sh.out_splits) / max(n - sh.n_rows, 1) for sh, _ in lst])
            assert 0.4 < frac < 0.75                     # ~58 % of the remote rows at P = 8 on this graph


@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_cfg5_gat_eight_shards_forward_and_backward(dev, oracle, arxiv, shards, exchange):
    import stag_amd
    from stag_amd import _lib, ops
    src, dst, n, g = arxiv
    H, F = 8, 32
    gen = torch.Generator().manual_seed(55)
    el0 = torch.randn(n, H, generator=gen).to(dev)
    er0 = torch.randn(n, H, generator=gen).to(dev)
    ft0 = torch.randn(n, H, F, generator=gen).to(dev)
    G = torch.randn(n, H, F, generator=gen).to(dev)
    mk = lambda graph: stag_amd.EdgeNoise(graph, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=0x5747A6, offset=7)
    drop = (0.6, 1234, 5)
    for attn_drop in (None, drop):
        el, er, ft = (t.clone().requires_grad_(True) for t in (el0, er0, ft0))
        whole = ops.gat_aggregate(g, el, er, ft, 0.2, mk(g), attn_drop=attn_drop)
        whole.backward(G)
        els, ers, fts = (t.clone().requires_grad_(True) for t in (el0, er0, ft0))
        parts = []
        for sh, gid in shards[exchange]:
            sh = sh.local_var()
            # the two tables of the exchange step (ft, el), each delivered as the collective would deliver it
            sh.halo_gather_multi = lambda ts, gid=gid: [_exchange(fts.reshape(n, H * F), gid).reshape(-1, H, F),
                                                        _exchange(els, gid)]
            lo, hi = sh.row_lo, sh.row_hi
            parts.append(sh.gat_aggregate(els[lo:hi], ers[lo:hi], fts[lo:hi], 0.2, mk(sh), attn_drop=attn_drop))
        got = torch.cat(parts, 0)
        assert torch.equal(got, whole), f"{exchange}, attn_drop={attn_drop}: 8 shards != whole graph"
        got.backward(G)
        for a, b, nm in ((fts, ft, "d ft"), (els, el, "d el"), (ers, er, "d er")):
            assert_close(*_rel(a.grad, b.grad), what=f"cfg5 8 x {exchange} shards, attn_drop={attn_drop}: {nm}")
        if attn_drop is None and exchange == "halo":
            og = oracle_graph(oracle, g)
            with hw_normals(oracle, dev):
                ref = oracle.gat_fwd(og, el0.cpu().numpy(), er0.cpu().numpy(), ft0.cpu().numpy(), 0.2,
                                     oracle.make_spec("normal", 1.0, 0.5, seed=0x5747A6, offset=7, Dn=H,
                                                      n_edges=g.number_of_edges()))
            assert_close(got.reshape(n, -1), ref.reshape(n, -1), what="cfg5 8 shards vs oracle")


@pytest.mark.parametrize("exchange", ["halo", "allgather"])
@pytest.mark.parametrize("kind", ["normal", "bernoulli_norm"])
def test_cfg2_aggregation_eight_shards_forward_and_backward(dev, oracle, arxiv, shards, exchange, kind):
    """cfg2 (D = 128) on the same 8 shards, with GCN's 'both' degree scalings (stag/zoo/gcn.py:67-75, 100-108): the
    source scale indexes buffer rows (`out_degrees()` of a shard = the GLOBAL out-degree behind each of them)."""
    import stag_amd
    from stag_amd import _lib, ops
    src, dst, n, g = arxiv
    D = 128
    gen = torch.Generator().manual_seed(56)
    x0 = torch.randn(n, D, generator=gen).to(dev)
    G = torch.randn(n, D, generator=gen).to(dev)
    if kind == "normal":
        mk = lambda graph: stag_amd.EdgeNoise(graph, D, _lib.NOISE_NORMAL, 1.0, 0.5, seed=0x5747A6, offset=0)
        ospec = lambda: oracle.make_spec("normal", 1.0, 0.5, seed=0x5747A6, offset=0, Dn=D, n_edges=g.number_of_edges())
    else:       # scripts/arxiv_mle/gcn/run.py:70-74: Bernoulli + norm=True
        mk = lambda graph: stag_amd.EdgeNoise(graph, D, _lib.NOISE_BERNOULLI, 0.9, seed=0x5747A6, offset=0, in_norm=True)
        ospec = lambda: oracle.make_spec("bernoulli", 0.9, None, in_norm=True, seed=0x5747A6, offset=0, Dn=D,
                                         n_edges=g.number_of_edges())
    ss = g.out_degrees().clamp(min=1).float() ** -0.5
    ds = g.in_degrees().clamp(min=1).float() ** -0.5
    x = x0.clone().requires_grad_(True)
    whole = ops.aggregate(g, x, mk(g), src_scale=ss, dst_scale=ds)
    whole.backward(G)
    xs = x0.clone().requires_grad_(True)
    parts = []
    for sh, gid in shards[exchange]:
        noise = mk(sh)
        noise.pos_base = sh.pos_base
        ss_buf = sh.out_degrees().clamp(min=1).float() ** -0.5
        ds_loc = sh.in_degrees().clamp(min=1).float() ** -0.5
        parts.append(ops.aggregate(sh, _exchange(xs, gid), noise, src_scale=ss_buf, dst_scale=ds_loc, _gathered=True))
    got = torch.cat(parts, 0)
    assert torch.equal(got, whole), f"{exchange} {kind}: 8 shards != whole graph"
    got.backward(G)
    assert_close(*_rel(xs.grad, x.grad), what=f"cfg2 8 x {exchange} shards {kind}: d x")
    if exchange == "halo":
        og = oracle_graph(oracle, g)
        with hw_normals(oracle, dev):
            ref = oracle.agg_fwd(og, x0.cpu().numpy(), ospec(), src_scale=ss.cpu().numpy(), dst_scale=ds.cpu().numpy())
        assert_close(got, ref, what=f"cfg2 8 shards {kind} vs oracle")


def test_cfg2_partitioned_backward_pieces_at_full_size(dev, arxiv, shards):
    """The partitioned backward of `_ShardAggregate` at P = 8 and full size, its collective emulated: every shard reduces
    its REMOTE buffer rows and its own rows in two launches over the sub-plans of the source-major twin
    (`plan_split_t`), the rows the peers computed for a rank are placed behind its buffer rows in send-list order
    (what the transposed all-to-all delivers), and ONE launch over `_combined_csr` adds own + received.  The result is
    the whole graph's d x at 1e-5, equals — bit for bit — the same adds done with torch in the kernel's fixed order (own
    row first, then the peers in rank order, two terms at a time), and does not depend on how the two sub-plan launches
    are split."""
    import stag_amd
    from stag_amd import _lib, ops
    src, dst, n, g = arxiv
    D = 128
    gen = torch.Generator().manual_seed(57)
    x0 = torch.randn(n, D, generator=gen).to(dev)
    G = torch.randn(n, D, generator=gen).to(dev)
    mk = lambda graph: stag_amd.EdgeNoise(graph, D, _lib.NOISE_NORMAL, 1.0, 0.5, seed=0x5747A6, offset=3)
    x = x0.clone().requires_grad_(True)
    ops.aggregate(g, x, mk(g)).backward(G)
    lst = shards["halo"]
    dx_bufs = []
    for sh, gid in lst:
        noise = mk(sh)
        noise.pos_base = sh.pos_base
        spec = ops._targs_or_c(ops._noise_spec(noise, in_norm=0))
        gl = G[sh.row_lo:sh.row_hi].contiguous()
        p_first, p_second = sh.plan_split_t(64)
        assert p_first["n_units"] > 0 and p_second["n_units"] > 0
        buf = torch.full((sh.n_buf, D), float("nan"), device=dev)
        ops._agg_raw(sh.csr_t, gl, D, spec, _lib.REDUCE_SUM, None, None, 64, out=buf, plan_t=p_first)
        assert not torch.isnan(buf[sh.n_rows:]).any(), "the remote rows are complete after the first launch"
        ops._agg_raw(sh.csr_t, gl, D, spec, _lib.REDUCE_SUM, None, None, 64, out=buf, plan_t=p_second)
        whole_launch, _ = ops._agg_raw(sh.csr_t, gl, D, spec, _lib.REDUCE_SUM, None, None, 64)
        assert torch.equal(buf, whole_launch)
        dx_bufs.append(buf)
    # position of every global row in every shard's buffer (-1: not there)
    where = []
    for sh, gid in lst:
        w = torch.full((n,), -1, dtype=torch.int64, device=dev)
        w[gid] = torch.arange(sh.n_buf, device=dev)
        where.append(w)
    got = []
    for r, (sh, gid) in enumerate(lst):
        n_send = int(sh.send_idx.shape[0])
        T = torch.empty((sh.n_buf + n_send, D), device=dev)
        T[:sh.n_buf] = dx_bufs[r]
        # the transposed all-to-all: peer q returns, in the order of my send list to q, the gradient of those rows
        off = 0
        terms = [[dx_bufs[r][:sh.n_rows], torch.ones(sh.n_rows, dtype=torch.bool, device=dev)]]   # (values, present)
        for q, cnt in enumerate(sh.in_splits):
            if cnt == 0:
                continue
            rows_local = sh.send_idx[off:off + cnt]
            pos = where[q][rows_local + sh.row_lo]
            assert bool((pos >= 0).all())
            T[sh.n_buf + off:sh.n_buf + off + cnt] = dx_bufs[q][pos]
            val = torch.zeros((sh.n_rows, D), device=dev)
            has = torch.zeros(sh.n_rows, dtype=torch.bool, device=dev)
            val[rows_local] = dx_bufs[q][pos]                       # (a row is sent to a peer at most once)
            has[rows_local] = True
            terms.append([val, has])
            off += cnt
        assert off == n_send
        # the same adds with torch, in the kernel's order: a row's terms — its own gradient, then the peers' in rank
        # order — are taken two at a time, each pair summed from zero, the pair sums added up in order
        vals = torch.stack([t[0] for t in terms], 1)                # [n_rows, K, D]
        has = torch.stack([t[1] for t in terms], 1)                 # [n_rows, K]
        K = vals.shape[1]
        rank_in_row = torch.cumsum(has.long(), 1) - 1               # position of a present term in the row's list
        seq = torch.zeros((sh.n_rows, D), device=dev)
        for pair in range((K + 1) // 2):
            blk = torch.zeros((sh.n_rows, D), device=dev)
            for j in (2 * pair, 2 * pair + 1):
                pick = has & (rank_in_row == j)                     # the row's j-th term, whichever peer it came from
                blk = blk + (vals * pick.unsqueeze(-1)).sum(1)      # at most one non-zero term: exact
            seq = seq + blk
        dx, _ = ops._agg_raw(sh._combined_csr(), T, D, ops._targs_or_c(ops._none_spec()), _lib.REDUCE_SUM, None, None, 64)
        assert torch.equal(dx, seq), f"rank {r}: own row first, then the peers in rank order, summed in the kernel's pairs"
        got.append(dx)
    assert_close(*_rel(torch.cat(got, 0), x.grad), what="cfg2 8 shards: d x through the partitioned backward's pieces")


def test_cfg5_staged_gat_backward_equals_the_single_call_at_full_size(dev, arxiv, shards):
    """stag_gat_bwd_stages (ABI v19) at BASELINE configs[4]'s size: on the whole graph the source pass cut into two random
    complementary sub-plans (all 18k segments of the long rows riding in the first), and on shard 3 of 8 cut the way
    `_ShardGat` cuts it (remote buffer rows first, then this rank's own) — row dots, two source passes, d er — write what
    ONE stag_gat_bwd call writes, bit for bit; the first pass leaves its own rows complete (they are what the transposed
    exchange would already be sending)."""
    import stag_amd
    from stag_amd import _lib, ops
    src, dst, n, g = arxiv
    H, F = 8, 32
    gen = torch.Generator().manual_seed(77)
    mk = lambda graph: stag_amd.EdgeNoise(graph, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=0x5747A6, offset=11)
    cases = [("whole graph", g, g.csr, g.csr_t, None)]
    sh, gid = shards["halo"][3]
    cases.append(("shard 3 of 8", sh, sh.csr, sh.csr_t, sh))
    for name, graph, csrv, csrt, shard in cases:
        n_dst, n_src = csrv.n_dst, csrt.n_dst
        el = torch.randn(n_src, H, generator=gen).to(dev)
        er = torch.randn(n_dst, H, generator=gen).to(dev)
        ft = torch.randn(n_src, H, F, generator=gen).to(dev)
        G = torch.randn(n_dst, H, F, generator=gen).to(dev)
        noise = mk(graph)
        if shard is not None:
            noise.pos_base = shard.pos_base
        spec = noise.spec()
        drop = (0.6, 1234, 5)
        dstruct = ops._gat_drop_struct(drop)
        out = torch.empty(n_dst, H, F, device=dev)
        stats = torch.empty(n_dst, 2 * H, device=dev)
        ops._gat_fwd_into(csrv, csrv.plan(64, need=True), el, er, ft, H, F, 0.2, spec, None, dstruct, out, stats, dev)
        d_el, d_er, d_ft, _ = ops._gat_bwd_fused(csrv, csrt, el, er, ft, stats, G, out, H, F, 0.2, spec, None, False, 64, dev, drop)
        if shard is None:
            full_t = csrt.plan(64, need=True)
            units_t = full_t["units"].cpu().numpy()[:full_t["n_units"]]
            keep = (units_t[:, 3] < 0) & (np.random.default_rng(5).random(len(units_t)) < 0.5)
            first, second = csrt.subplan(64, ~keep), csrt.subplan(64, keep)
            rows_first = torch.from_numpy(units_t[~keep & (units_t[:, 3] < 0)][:, 0].astype(np.int64)).to(dev)
        else:
            first, second = shard.plan_split_t(64)
            rows_first = torch.arange(shard.n_rows, shard.n_buf, device=dev)
        assert first["n_units"] > 0 and second["n_units"] > 0
        T_ft = torch.full((n_src + 3, H * F), float("nan"), device=dev)
        T_el = torch.full((n_src + 3, H), float("nan"), device=dev)
        e_r = torch.full((n_dst, H), float("nan"), device=dev)
        st = ops._GatBwdStages(csrv, csrt, el, er, ft, stats, G, out, H, F, 0.2, spec, None, drop, 64, T_el, e_r, T_ft, dev)
        st.rowdot()
        st.source(first)
        assert torch.equal(T_ft[rows_first], d_ft.reshape(n_src, -1)[rows_first]), f"{name}: the first pass's rows are complete"
        assert torch.equal(T_el[rows_first], d_el[rows_first])
        st.source(second)
        st.der()
        assert torch.equal(T_ft[:n_src], d_ft.reshape(n_src, -1)), f"{name}: d ft"
        assert torch.equal(T_el[:n_src], d_el) and torch.equal(e_r, d_er), f"{name}: d el / d er"
