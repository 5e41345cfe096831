"""Node-range partitioning of one graph over the GPUs of a node (BASELINE north_star;
new functionality — the reference is single-process).

    rank g owns the contiguous destination rows [bounds[g], bounds[g+1]) — cut so the
    ranks hold equal EDGE counts (in-degree is skewed) — the matching rows of `x` and
    of `out`, and the CSR of exactly those rows;
    per layer, ONE RCCL collective over xGMI brings the source features together
    (`halo_gather`).  exchange="halo" (default): an all-to-all of exactly the remote
    rows this rank's edges reference (at 8 ranks a random-source graph needs 58 % of
    them, so 42 % less traffic than gathering everything; xGMI is point-to-point, and
    an all-to-all drives all 7 links at once).  exchange="allgather": equal-size padded
    shards, one all_gather_into_tensor (what a denser halo degenerates to);
    the local CSR's column ids are pre-mapped to rows of the exchanged buffer
    [my rows | rows from rank 0 | from rank 1 | ...], so the aggregation kernel is the
    single-GPU kernel, unchanged;
    the rows whose sources are ALL local form their own unit list (`plan_split`): their
    launch overlaps the collective, the remaining rows are launched behind it;
    Philox counters are keyed by the GLOBAL CSR position (`pos_base` = first global
    position of the shard), and every row is reduced by exactly one rank in the
    single-GPU order, so 1/2/4/8-GPU outputs are bit-identical;
    backward = the transposed exchange (reduce-scatter of dx).

Two constructors with identical results:
    GraphShard(src, dst, n, rank, world)         every rank holds the whole COO list
                                                 (O(E log E) per rank, no communication);
    GraphShard.from_edge_slices(src_part, ...)   every rank starts from ANY E/P slice of the
                                                 edge list — the form for graphs that do not
                                                 fit one GPU: two all-reduced degree vectors,
                                                 one all-to-all that routes every edge to the
                                                 owner of its destination, a local stable sort,
                                                 one all-to-all of the needed-row ids.

A shard duck-types the graph surface the layers touch (`local_var`, `in_degrees`,
`out_degrees`, `number_of_edges`, ...): `StagLayer(zoo.GCN | GraphSAGE | GIN | GAT)` runs on
it with `feat` = this rank's rows; ops.aggregate / ops.gat_aggregate route through the shard.

Channel sharding (`ChannelShard`) is the alternative for graphs that FIT one GPU (arxiv: 5 MB
of CSR): every rank keeps the whole CSR and D/P of the feature channels.  The channels of the
aggregation are independent, so the step needs NO exchange at all, and the only traffic of a
GCN layer is the all-to-all that turns the channel-sharded result into row shards for the
dense transform (`to_row_shards`, N*D/P floats per rank; the halo exchange moves up to
N*D*(P-1)/P).  What does not shrink with P is the per-edge part (index loads, address
arithmetic), so this mode stops scaling at about D/P = 32.  Philox
counters are keyed by the GLOBAL channel (`chunk_base`), so outputs are again bit-identical.
"""
import numpy as np
import torch
import torch.distributed as dist

from .graph import DEFAULT_SEG_LEN, CsrView, build_csr


def edge_balanced_bounds(indptr, world):
    """Row cut points [world+1]: rank g gets rows whose CSR positions straddle
    [g*E/world, (g+1)*E/world).  indptr: numpy int array [N+1]."""
    n = len(indptr) - 1
    E = int(indptr[-1])
    targets = (np.arange(1, world, dtype=np.float64) * E / world)
    cuts = np.searchsorted(indptr[1:], targets, side="left") + 1 if n else np.zeros(world - 1, int)
    bounds = np.concatenate([[0], np.minimum(cuts, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


class _HaloExchange(torch.autograd.Function):
    """The halo exchange of one or several row tables as ONE autograd node:
        forward  x_t [n_rows, W_t] on every rank  ->  the exchanged buffer [n_buf, W_t] of every table
                 (`GraphShard.halo_start_multi`: local rows in place, no concatenation);
        backward the TRANSPOSED exchange of the buffers' gradient rows, added to the owners' rows in a FIXED order —
                 a row's own contribution first, then the peers' in rank order (`GraphShard._segsum_back`: a
                 segmented sum over a CSR of the send list on the aggregation kernel, no atomics) — so the
                 gradients of a partitioned step are the same bits on every run."""

    @staticmethod
    def forward(ctx, shard, *xs):
        ctx.shard = shard._origin
        bufs, work = shard.halo_start_multi(list(xs), persistent=False)
        if work is not None:
            work.wait()
        return tuple(bufs)

    @staticmethod
    def backward(ctx, *gs):
        return (None,) + tuple(ctx.shard.halo_transpose_multi([g.contiguous() for g in gs]))


def _wait_all(works):
    class _W:
        def wait(self_inner):
            for w in works:
                if w is not None:
                    w.wait()
    return _W() if any(w is not None for w in works) else None


class _ShardAggregate(torch.autograd.Function):
    """One partitioned layer-forward AND its backward with the collective overlapped on both sides.

    forward : start the exchange; launch the units whose sources are all local rows; wait; launch the rest.
    backward: launch the transposed aggregation of the REMOTE buffer rows (their gradient has to travel), start
              the transposed exchange on them, launch the local rows while it is in flight, wait, then ONE launch
              adds every local row's own gradient and the rows its peers sent, own first, peers in rank order
              (`GraphShard._combined_csr`): fixed order, no atomics — the same bits on every run.
    Covers weight = None | an EdgeNoise without parameter gradients (everything `vi=False` runs); the others take
    halo_gather + ops.aggregate."""

    @staticmethod
    def forward(ctx, x_local, shard, weight, reduce, src_scale, dst_scale, seg_len, overlap):
        from . import ops
        o = shard._origin
        x_local = ops._f32c(x_local)
        D = x_local.shape[1]
        buf, work = shard.halo_start(x_local, persistent=True)
        spec = ops._targs_or_c(ops._noise_spec(weight)) if weight is not None else ops._targs_or_c(ops._none_spec())
        need_dx = ctx.needs_input_grad[0]
        ns = (torch.empty((shard.n_rows, D), dtype=torch.float32, device=x_local.device)
              if (need_dx and spec.in_norm) else None)
        out = torch.empty((shard.n_rows, D), dtype=torch.float32, device=x_local.device)
        p_loc, p_rem = shard.plan_split(seg_len)
        red = ops._REDUCE[reduce]
        if overlap and work is not None and p_loc["n_units"]:
            ops._agg_raw(shard._csr, buf, D, spec, red, src_scale, dst_scale, seg_len, out=out, plan_t=p_loc, ns_out=ns)
            if work is not None:
                work.wait()
            if p_rem["n_units"]:
                ops._agg_raw(shard._csr, buf, D, spec, red, src_scale, dst_scale, seg_len, out=out, plan_t=p_rem, ns_out=ns)
        else:
            if work is not None:
                work.wait()
            ops._agg_raw(shard._csr, buf, D, spec, red, src_scale, dst_scale, seg_len, out=out,
                         plan_t=shard._csr.plan(seg_len, need=True), ns_out=ns)
        ctx.shard, ctx.weight, ctx.reduce, ctx.seg_len, ctx.D, ctx.overlap = o, weight, red, seg_len, D, overlap
        ctx.save_for_backward(src_scale, dst_scale, ns)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import _lib, ops
        if not ctx.needs_input_grad[0]:
            return (None,) * 8
        src_scale, dst_scale, ns = ctx.saved_tensors
        sh, weight, D, seg_len = ctx.shard, ctx.weight, ctx.D, ctx.seg_len
        g = ops._f32c(grad_out)
        if ns is not None:
            g = g * ns
        dvec = dst_scale
        if ctx.reduce == _lib.REDUCE_MEAN:
            inv = 1.0 / sh._csr.degrees.clamp(min=1).to(torch.float32)
            dvec = inv if dvec is None else dvec * inv
        spec = (ops._targs_or_c(ops._noise_spec(weight, in_norm=0)) if weight is not None
                else ops._targs_or_c(ops._none_spec()))
        csr_t = sh.csr_t
        n_send = int(sh.send_idx.shape[0])
        halo = sh.exchange == "halo" and sh.world > 1
        # T = [gradient of every buffer row | rows received from the peers]: one allocation, so ONE launch can add a
        # local row's own gradient and what its peers sent
        T = torch.empty((sh.n_buf + (n_send if halo else 0), D), dtype=torch.float32, device=g.device)
        dx_buf = T[:sh.n_buf]
        p_first, p_second = sh.plan_split_t(seg_len)
        run = lambda plan: ops._agg_raw(csr_t, g, D, spec, _lib.REDUCE_SUM, dvec, src_scale, seg_len, out=dx_buf, plan_t=plan)
        if not halo:
            run(csr_t.plan(seg_len, need=True))
            if sh.world == 1:
                return (dx_buf[:sh.n_rows].clone(),) + (None,) * 7
            return (sh.halo_transpose(dx_buf),) + (None,) * 7
        if ctx.overlap and p_first["n_units"] and p_second["n_units"]:
            run(p_first)
            work = sh._transpose_start(dx_buf[sh.n_rows:], T[sh.n_buf:])
            run(p_second)
        else:
            run(csr_t.plan(seg_len, need=True))
            work = sh._transpose_start(dx_buf[sh.n_rows:], T[sh.n_buf:])
        if work is not None:
            work.wait()
        dx, _ = ops._agg_raw(sh._combined_csr(), T, D, ops._targs_or_c(ops._none_spec()), _lib.REDUCE_SUM, None, None, seg_len)
        return (dx,) + (None,) * 7


class _ShardGat(torch.autograd.Function):
    """One partitioned GAT layer-forward (BASELINE configs[4]; stag/zoo/gat.py:109-126) AND its backward, with the
    exchange overlapped on both sides — `_ShardAggregate`'s shape for the two-table step.

    forward : start the exchange of ft [n, H, F] and el [n, H] (two tables, one step); launch the unit batches whose
              sources are all local rows; wait; launch the rest (two stag_gat_fwd calls over sub-plans, same `out`).
    backward: stag_gat_bwd one stage at a time (include/stag_hip.h v19): the row dots, then the source pass over the
              REMOTE buffer rows (their d ft / d el have to travel; the segments of long rows ride with them), the
              transposed exchange of both tables starts on them, the source pass over this rank's own rows runs while
              it is in flight, then d er; ONE launch per table adds every local row's own gradient and what its peers
              sent, own first, peers in rank order (`GraphShard._combined_csr`): no atomics, the same bits every run.
    Covers weight = None | an EdgeNoise without parameter gradients, with or without in-kernel attention dropout, on
    the workgroup-cooperative shapes; everything else takes halo_gather_multi + ops.gat_aggregate."""

    @staticmethod
    def forward(ctx, el_local, er_local, ft_local, shard, noise, neg_slope, seg_len, attn_drop, overlap):
        from . import _lib, ops
        o = shard._origin
        el, er, ft = ops._f32c(el_local), ops._f32c(er_local), ops._f32c(ft_local)
        H, F = ft.shape[1], ft.shape[2]
        dev = _lib.require_device(el, er, ft, shard._csr.indptr)
        (ft_buf, el_buf), work = shard.halo_start_multi([ft, el], persistent=False)
        spec = noise.spec() if noise is not None else ops._targs_or_c(ops._none_spec())
        spec.pos_base = int(shard.pos_base)        # noise and the dropout mask are the whole graph's
        csrv = shard._csr
        nscale = ops._gat_norm_scale(csrv, noise, H, seg_len, dev) if spec.in_norm else None
        need_grad = any(ctx.needs_input_grad[:3])
        out = torch.empty((shard.n_rows, H, F), dtype=torch.float32, device=dev)
        stats = torch.empty((shard.n_rows, 2 * H), dtype=torch.float32, device=dev) if need_grad else None
        drop = ops._gat_drop_struct(attn_drop)
        p_loc, p_rem = shard.plan_split(seg_len)
        run = lambda plan: ops._gat_fwd_into(csrv, plan, el_buf, er, ft_buf, H, F, neg_slope, spec, nscale, drop, out, stats, dev)
        if overlap and work is not None and p_loc["n_units"] and p_rem["n_units"]:
            run(p_loc)
            work.wait()
            run(p_rem)
        else:
            if work is not None:
                work.wait()
            run(csrv.plan(seg_len, need=True))
        if need_grad:
            ctx.shard, ctx.noise, ctx.neg_slope, ctx.seg_len, ctx.attn_drop, ctx.overlap = o, noise, float(neg_slope), seg_len, attn_drop, overlap
            ctx.save_for_backward(el_buf, er, ft_buf, stats, out, nscale)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import _lib, ops
        el_buf, er, ft_buf, stats, out, nscale = ctx.saved_tensors
        sh, noise, seg_len = ctx.shard, ctx.noise, ctx.seg_len
        H, F = ft_buf.shape[1], ft_buf.shape[2]
        HF, dev = H * F, ft_buf.device
        G = ops._f32c(grad_out)
        spec = noise.spec() if noise is not None else ops._targs_or_c(ops._none_spec())
        spec.pos_base = int(sh.pos_base)
        n_send = int(sh.send_idx.shape[0])
        halo = sh.exchange == "halo" and sh.world > 1
        # T_* = [gradient of every buffer row | rows received back from the peers]: one allocation per table
        extra = n_send if halo else 0
        T_ft = torch.empty((sh.n_buf + extra, HF), dtype=torch.float32, device=dev)
        T_el = torch.empty((sh.n_buf + extra, H), dtype=torch.float32, device=dev)
        d_er = torch.empty((sh.n_rows, H), dtype=torch.float32, device=dev)
        st = ops._GatBwdStages(sh._csr, sh.csr_t, el_buf, er, ft_buf, stats, G, out, H, F, ctx.neg_slope, spec, nscale,
                               ctx.attn_drop, seg_len, T_el, d_er, T_ft, dev)
        st.rowdot()
        p_first, p_second = sh.plan_split_t(seg_len)
        if halo and ctx.overlap and p_first["n_units"] and p_second["n_units"]:
            st.source(p_first)
            work = sh._transpose_start_multi([T_ft[sh.n_rows:sh.n_buf], T_el[sh.n_rows:sh.n_buf]],
                                             [T_ft[sh.n_buf:], T_el[sh.n_buf:]])
            st.source(p_second)
            st.der()
        else:
            st.source()
            st.der()
            work = (sh._transpose_start_multi([T_ft[sh.n_rows:sh.n_buf], T_el[sh.n_rows:sh.n_buf]],
                                              [T_ft[sh.n_buf:], T_el[sh.n_buf:]]) if halo else None)
        if not halo:
            if sh.world == 1:
                d_ft = T_ft[sh.loc_off:sh.loc_off + sh.n_rows].clone()
                d_el = T_el[sh.loc_off:sh.loc_off + sh.n_rows].clone()
            else:
                d_ft, d_el = sh.halo_transpose_multi([T_ft, T_el])
        else:
            if work is not None:
                work.wait()
            none = ops._targs_or_c(ops._none_spec())
            comb = sh._combined_csr()
            d_ft, _ = ops._agg_raw(comb, T_ft, HF, none, _lib.REDUCE_SUM, None, None, seg_len)
            d_el, _ = ops._agg_raw(comb, T_el, H, none, _lib.REDUCE_SUM, None, None, seg_len)
        ni = ctx.needs_input_grad
        return (d_el if ni[0] else None, d_er if ni[1] else None, d_ft.reshape(-1, H, F) if ni[2] else None,
                None, None, None, None, None, None)


class NativeComm:
    """An RCCL communicator owned by libstag_hip.so (include/stag_hip.h: stag_comm_*): the halo exchange
    as calls into the C ABI on a HIP stream, without torch.distributed on the data path.  torch.distributed
    (any backend) is used once, to hand rank 0's 128-byte id to the other ranks.  One process per GPU."""

    def __init__(self, rank, world, device, group=None):
        import ctypes as C
        from . import _lib
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        lib = _lib.lib()
        buf = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(lib.stag_comm_unique_id(buf), "stag_comm_unique_id")
        box = [bytes(buf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.stag_comm_init(box[0], self.rank, self.world, C.byref(self._handle)), "stag_comm_init")
        self._side = torch.cuda.Stream(device=self.device)
        self._marshalled = {}

    def close(self):
        if self._handle:
            from . import _lib
            _lib.lib().stag_comm_destroy(self._handle)
            self._handle = None

    def allgather(self, x_pad, out=None):
        """[rows, D] on every rank -> [world * rows, D] (stag_halo_allgather, on the current stream).  x_pad may BE
        this rank's slot of `out` (RCCL's in-place all-gather: no staging copy)."""
        from . import _lib
        x_pad = x_pad.contiguous()
        if out is None:
            out = torch.empty((self.world * x_pad.shape[0],) + tuple(x_pad.shape[1:]), dtype=torch.float32,
                              device=x_pad.device)
        _lib.check(_lib.lib().stag_halo_allgather(self._handle, x_pad.data_ptr(), x_pad.numel(), out.data_ptr(),
                                                  _lib.stream_of(x_pad.device)), "stag_halo_allgather")
        return out

    def exchange_async(self, send, in_rows, recv, out_rows, width):
        """All-to-all-v of rows on a side stream (stag_halo_exchange): -> an object whose wait() orders the
        current stream behind it, so kernels launched in between overlap the transfer."""
        return self.exchange_multi_async([send], in_rows, [recv], out_rows, [width])

    def exchange_multi_async(self, sends, in_rows, recvs, out_rows, widths):
        """The same for several row tables bound for the same peers (GAT: ft and el), ONE RCCL group
        (stag_halo_exchange_multi)."""
        import ctypes as C
        from . import _lib
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)                       # the send rows are produced on the current stream
        n = len(sends)
        # the marshalled arguments of a step that repeats (persistent buffers, the shard's fixed row counts) are built
        # once: five ctypes arrays per step were host time on a step whose local kernel is ~13 us at 8 ranks
        key = (tuple(t.data_ptr() if t.numel() else 0 for t in sends), tuple(t.data_ptr() if t.numel() else 0 for t in recvs),
               tuple(int(w) for w in widths), tuple(in_rows), tuple(out_rows))
        args = self._marshalled.get(key)
        if args is None:
            if len(self._marshalled) > 64:
                self._marshalled.clear()
            args = self._marshalled[key] = (
                (C.c_void_p * n)(*[p or None for p in key[0]]), (C.c_void_p * n)(*[p or None for p in key[1]]),
                (C.c_int32 * n)(*key[2]), (C.c_int64 * self.world)(*[int(r) for r in in_rows]),
                (C.c_int64 * self.world)(*[int(r) for r in out_rows]))
        sp, rp, wd, sr, rr = args
        with torch.cuda.stream(self._side):
            rc = _lib.lib().stag_halo_exchange_multi(self._handle, n, sp, rp, wd, sr, rr, self._side.cuda_stream)
        _lib.check(rc, "stag_halo_exchange_multi")
        for t in list(sends) + list(recvs):
            t.record_stream(self._side)
        side = self._side

        class _Work:
            def wait(self_inner):
                torch.cuda.current_stream(side.device).wait_stream(side)
        return _Work()


def _owner_of(ids, bounds_t, world):
    """Rank owning each node id (ids: int64 tensor; bounds_t: [world+1] int64 tensor on its device)."""
    return torch.bucketize(ids, bounds_t[1:-1], right=True).clamp_(max=world - 1)


def _split_sorted(ids, bounds_t, world):
    """A sorted id tensor cut at the rank bounds -> list of `world` tensors."""
    cuts = torch.searchsorted(ids, bounds_t[1:-1]).tolist() if world > 1 else []
    return list(torch.tensor_split(ids, cuts)) if world > 1 else [ids]


class GraphShard:
    """The rows of one rank, shaped like a stag_amd.Graph for the layers, ops.aggregate and EdgeNoise.

    Every step is a collective — forward and, under autograd, backward (the transposed exchange): all ranks must run the
    same steps, a rank whose loss does not reach its shard's outputs (e.g. a cut that left it without rows — more ranks
    than rows with in-edges) included; give such a rank a zero-weight term over its outputs, or use fewer ranks."""

    is_block = False
    is_shard = True

    def __init__(self, src, dst, n_nodes, rank, world, device=None, group=None, exchange="halo"):
        """Every rank holds the whole COO list (arxiv: 19 MB).  O(E log E) per rank — one stable sort
        and one pass over the edges that leave this rank's node range — and no communication."""
        src = torch.as_tensor(np.asarray(src) if not torch.is_tensor(src) else src).to(torch.int64)
        dst = torch.as_tensor(np.asarray(dst) if not torch.is_tensor(dst) else dst).to(torch.int64)
        self._init_common(n_nodes, rank, world, device, group, exchange)
        N, world, rank = self.n_global, self.world, self.rank
        E = int(src.shape[0])
        # global stable dst-major order (same as graph.build_csr => same global positions)
        order = torch.sort(dst, stable=True).indices
        counts = torch.bincount(dst, minlength=N)
        indptr = torch.zeros(N + 1, dtype=torch.int64, device=src.device)
        indptr[1:] = torch.cumsum(counts, 0)
        out_deg = torch.bincount(src, minlength=N)
        self._set_bounds(indptr)
        lo, hi = self.row_lo, self.row_hi
        p_lo, p_hi = int(indptr[lo]), int(indptr[hi])
        mine = order[p_lo:p_hi]
        g_src = src[mine]
        # what the peers need from me: edges that start in my node range and end elsewhere
        bt = self._bounds_t.to(src.device)
        m = (src >= lo) & (src < hi)
        s_m, d_owner = src[m], _owner_of(dst[m], bt, world)
        away = d_owner != rank
        keys = torch.unique(d_owner[away] * N + s_m[away])                # sorted by (peer, id)
        send_owner = torch.div(keys, N, rounding_mode="floor") if N else keys
        in_splits = torch.bincount(send_owner, minlength=world).tolist()
        self._finish(indptr[lo:hi + 1] - p_lo, g_src, mine, p_lo, E, counts[lo:hi], out_deg,
                     send_ids=keys - send_owner * N, in_splits=in_splits)

    @classmethod
    def from_edge_slices(cls, src_part, dst_part, eid_base, n_nodes, rank, world, device=None, group=None,
                         exchange="halo"):
        """Distributed construction: this rank contributes the edges with global ids
        [eid_base, eid_base + len(src_part)) — any slice of the COO list — and ends up with the
        shard `GraphShard(src, dst, ...)` would have built from the whole list.  Per-rank work and
        memory are O(E / world + N): two all-reduced degree vectors (N ints), one all-to-all that
        routes each edge to the owner of its destination, a local stable sort, and one all-to-all of
        the ids of the rows each rank needs from each peer."""
        self = cls.__new__(cls)
        self._init_common(n_nodes, rank, world, device, group, exchange)
        N = self.n_global
        src_part = torch.as_tensor(src_part).to(torch.int64)
        dst_part = torch.as_tensor(dst_part).to(torch.int64)
        cdev = src_part.device
        deg = torch.stack([torch.bincount(dst_part, minlength=N), torch.bincount(src_part, minlength=N)])
        if world > 1:
            dist.all_reduce(deg, group=group)
        counts, out_deg = deg[0], deg[1]
        indptr = torch.zeros(N + 1, dtype=torch.int64, device=cdev)
        indptr[1:] = torch.cumsum(counts, 0)
        E = int(indptr[-1])
        self._set_bounds(indptr)
        lo, hi = self.row_lo, self.row_hi
        bt = self._bounds_t.to(cdev)
        eid = torch.arange(eid_base, eid_base + src_part.shape[0], dtype=torch.int64, device=cdev)
        # route every edge to the rank that owns its destination row
        d_owner = _owner_of(dst_part, bt, world)
        by_owner = torch.sort(d_owner, stable=True).indices
        payload = torch.stack([src_part[by_owner], dst_part[by_owner], eid[by_owner]], 1).contiguous()   # [n, 3]
        n_to = torch.bincount(d_owner, minlength=world)
        if world > 1:
            n_from = torch.empty_like(n_to)
            dist.all_to_all_single(n_from, n_to, group=group)
            recv = torch.empty((int(n_from.sum()), 3), dtype=torch.int64, device=cdev)
            dist.all_to_all_single(recv, payload, n_from.tolist(), n_to.tolist(), group=group)
        else:
            recv = payload
        # stable destination-major order of my rows: ascending (dst, original edge id)
        order = torch.sort(recv[:, 2], stable=True).indices
        order = order[torch.sort(recv[order, 1], stable=True).indices]
        g_src, eid_g = recv[order, 0].contiguous(), recv[order, 2].contiguous()
        p_lo = int(indptr[lo])
        self._finish(indptr[lo:hi + 1] - p_lo, g_src, eid_g, p_lo, E, counts[lo:hi], out_deg,
                     send_ids=None, in_splits=None)
        return self

    # ---- construction helpers ----------------------------------------------------------------
    def _init_common(self, n_nodes, rank, world, device, group, exchange):
        if exchange not in ("halo", "allgather"):
            raise ValueError("exchange must be 'halo' or 'allgather'")
        self.exchange = exchange if world > 1 else "allgather"
        self.rank, self.world, self.group = int(rank), int(world), group
        self.n_global = int(n_nodes)
        self._device = torch.device(device) if device is not None else torch.device("cpu")
        self.ndata, self.edata = {}, {}
        self._csr_t = None
        self._plan_split = {}
        self._origin = self
        self.native_comm = None      # a NativeComm: the no-grad exchange then goes through stag_halo_* (C ABI)

    def _set_bounds(self, indptr):
        self.bounds = edge_balanced_bounds(indptr.cpu().numpy(), self.world)
        self._bounds_t = torch.from_numpy(self.bounds)
        self.max_rows = int(np.max(np.diff(self.bounds))) if self.world else 0
        self.row_lo, self.row_hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])

    def _finish(self, indptr_rel, g_src, eid_g, p_lo, n_edges_global, in_deg_local, out_deg_global,
                send_ids, in_splits):
        """Column ids -> rows of the exchanged buffer; the send / receive lists; device tensors."""
        rank, world, dev = self.rank, self.world, self._device
        lo, hi = self.row_lo, self.row_hi
        self.pos_base, self.n_edges_global = int(p_lo), int(n_edges_global)
        self.n_rows = hi - lo
        bt = self._bounds_t.to(g_src.device)
        owner = _owner_of(g_src, bt, world)
        if self.exchange == "allgather":
            buf_row = owner * self.max_rows + (g_src - bt[owner])
            self.n_buf = world * self.max_rows
            self.loc_off = rank * self.max_rows          # this rank's rows sit at [loc_off, loc_off + n_rows)
            gid = torch.full((self.n_buf,), -1, dtype=torch.int64, device=g_src.device)
            for q in range(world):
                a, b = int(self.bounds[q]), int(self.bounds[q + 1])
                gid[q * self.max_rows:q * self.max_rows + (b - a)] = torch.arange(a, b, device=g_src.device)
            self.recv_ids = np.zeros(0, np.int64)
            self.out_splits = self.in_splits = [0] * world
            self.send_idx = torch.zeros(0, dtype=torch.int64, device=dev)
        else:
            # buffer = [my rows | rows needed from rank 0 | from rank 1 | ...], each peer's part sorted by
            # global id: the receive lists follow from this rank's own edges alone
            need = torch.unique(g_src[owner != rank])                           # sorted
            recv_lists = _split_sorted(need, bt, world)
            self.out_splits = [int(l.shape[0]) for l in recv_lists]              # rows I receive from each peer
            if send_ids is None:        # distributed construction: tell every peer which rows it must send
                n_need = torch.tensor(self.out_splits, dtype=torch.int64, device=need.device)
                n_send = torch.empty_like(n_need)
                dist.all_to_all_single(n_send, n_need, group=self.group)
                in_splits = n_send.tolist()
                send_ids = torch.empty(int(n_send.sum()), dtype=torch.int64, device=need.device)
                dist.all_to_all_single(send_ids, need, in_splits, self.out_splits, group=self.group)
            self.in_splits = [int(v) for v in in_splits]                        # rows I send to each peer
            self.recv_ids = need.cpu().numpy()                                   # global ids, buffer order
            self.send_idx = (send_ids - lo).to(dev)
            buf_row = torch.where(owner == rank, g_src - lo,
                                  (hi - lo) + torch.searchsorted(need, g_src))
            self.n_buf = (hi - lo) + int(need.shape[0])
            self.loc_off = 0
            gid = torch.cat([torch.arange(lo, hi, device=g_src.device), need])
        if self.n_buf >= 2 ** 31 or g_src.shape[0] >= 2 ** 31:
            raise ValueError("a shard holds at most 2^31-1 edges / buffer rows: use more ranks")
        self.local_indptr = indptr_rel.to(torch.int32).to(dev)
        self.local_indices = buf_row.to(torch.int32).to(dev)
        self.local_eid_global = eid_g.to(dev)
        self.send_idx32 = self.send_idx.to(torch.int32)
        self._bufs, self._sendbufs = {}, {}
        self._csr = CsrView(self.n_rows, self.n_buf, self.local_indptr, self.local_indices, None)
        self._in_deg = in_deg_local.to(torch.int64).to(dev)
        # global out-degree of the node behind every buffer row (GCN's source scaling indexes columns)
        self._out_deg_buf = torch.where(gid >= 0, out_deg_global[gid.clamp(min=0)],
                                        torch.zeros_like(gid)).to(dev)
        # rows whose sources are all local can be aggregated while the exchange is in flight
        E_loc = int(g_src.shape[0])
        if E_loc:
            rows = torch.repeat_interleave(torch.arange(self.n_rows, device=g_src.device),
                                           (indptr_rel[1:] - indptr_rel[:-1]))
            remote = torch.zeros(self.n_rows, dtype=torch.int64, device=g_src.device)
            remote.index_add_(0, rows, (owner != rank).to(torch.int64))      # (not "buf_row >= n_rows": the all-gather
                                                                             # layout keeps rank 0's rows at the front)
            self._row_is_local = (remote == 0).cpu().numpy()
        else:
            self._row_is_local = np.ones(self.n_rows, bool)

    # ---- Graph-like surface used by the layers / ops / EdgeNoise -----------------------------
    device = property(lambda self: self._device)
    csr = property(lambda self: self._csr)
    # endpoints of this rank's edges as rows of the exchanged buffer, by local edge id (ops.edge_mlp reads them)
    _src = property(lambda self: self.edge_endpoints()[0])
    _dst = property(lambda self: self.edge_endpoints()[1])
    srcdata = property(lambda self: self.ndata)
    dstdata = property(lambda self: self.ndata)

    def number_of_edges(self):
        return self._csr.n_edges

    def number_of_nodes(self):
        return self.n_rows

    number_of_dst_nodes = num_dst_nodes = num_nodes = number_of_nodes
    num_edges = number_of_edges

    def number_of_src_nodes(self):
        return self.n_buf

    def in_degrees(self):
        """In-degree of this rank's rows (all their in-edges live on this rank)."""
        return self._in_deg

    def out_degrees(self):
        """GLOBAL out-degree of the node behind every row of the exchanged buffer [n_buf]: source-side
        scalings (GCN norm='both', stag/zoo/gcn.py:67-75) are indexed by column id."""
        return self._out_deg_buf

    def local_var(self):
        g = GraphShard.__new__(GraphShard)
        g.__dict__.update(self.__dict__)
        g.ndata, g.edata = dict(self.ndata), dict(self.edata)
        return g

    def _cache_owner(self):
        return self._origin

    def edges(self):
        raise NotImplementedError("a shard keeps its edges in CSR form over the exchanged buffer: "
                                  "edge_endpoints() gives their endpoints as buffer rows")

    @property
    def csr_t(self):
        """Source-major twin over the gathered buffer rows (backward).  nidx = LOCAL forward position;
        the library adds spec.pos_base (include/stag_hip.h: stag_csr.nidx)."""
        o = self._origin
        if o._csr_t is None:
            E = self._csr.n_edges
            rows = torch.repeat_interleave(
                torch.arange(self.n_rows, dtype=torch.int32, device=self._device),
                (self.local_indptr[1:] - self.local_indptr[:-1]).long())
            indptr, indices, eid = build_csr(rows, self.local_indices, self.n_rows, self.n_buf)
            # edge data (explicit weights, GAT's de / a) is indexed by local edge id = forward position
            o._csr_t = CsrView(self.n_buf, self.n_rows, indptr, indices, eid if E else None, eid if E else None)
        return o._csr_t

    def plan_split(self, seg_len=DEFAULT_SEG_LEN):
        """(local plan, remote plan): the launch plan of `csr` cut into the rows every source of which
        is a local row — their launch needs no exchanged data — and the rest.  Unit order inside each
        part is the plan's (longest first), each row is still reduced whole, in the same order, by the
        same kernel: results do not change."""
        o = self._origin
        if seg_len not in o._plan_split:
            full = self._csr.plan(seg_len, need=True)
            units = full["units"].cpu().numpy()[:full["n_units"]]
            whole = units[:, 3] < 0
            loc = whole & self._row_is_local[np.where(whole, units[:, 0], 0)]
            o._plan_split[seg_len] = (self._csr.subplan(seg_len, loc), self._csr.subplan(seg_len, ~loc))
        return o._plan_split[seg_len]

    def plan_split_t(self, seg_len=DEFAULT_SEG_LEN):
        """(first, second) for the BACKWARD: the plan of the source-major twin `csr_t` (its rows = buffer rows) cut
        into the rows whose gradient has to travel — the remote buffer rows; all segments of long rows ride with
        them, a sub-plan takes them together — and this rank's own rows, launched while the transposed exchange
        is in flight."""
        o = self._origin
        key = ("t", seg_len)
        if key not in o._plan_split:
            csr_t = self.csr_t
            full = csr_t.plan(seg_len, need=True)
            units = full["units"].cpu().numpy()[:full["n_units"]]
            whole = units[:, 3] < 0
            row = np.where(whole, units[:, 0], 0)
            local = whole & (row >= self.loc_off) & (row < self.loc_off + self.n_rows)
            o._plan_split[key] = (csr_t.subplan(seg_len, ~local), csr_t.subplan(seg_len, local))
        return o._plan_split[key]

    # ---- the exchange step ---------------------------------------------------------
    def pad_rows(self, x_local):
        if x_local.shape[0] != self.n_rows:
            raise ValueError(f"rank {self.rank} owns {self.n_rows} rows, got {x_local.shape[0]}")
        if self.n_rows == self.max_rows:
            return x_local
        pad = torch.zeros((self.max_rows - self.n_rows,) + tuple(x_local.shape[1:]),
                          dtype=x_local.dtype, device=x_local.device)
        return torch.cat([x_local, pad], 0)

    def exchange_bytes(self, D, itemsize=4):
        """Bytes this rank receives / sends per exchange of D-wide rows."""
        if self.world == 1:
            return 0, 0
        if self.exchange == "halo":
            return sum(self.out_splits) * D * itemsize, sum(self.in_splits) * D * itemsize
        return (self.world - 1) * self.max_rows * D * itemsize, (self.world - 1) * self.max_rows * D * itemsize

    def exchange_buffer(self, tail, dtype=torch.float32, device=None):
        """The PERSISTENT exchange buffer [n_buf, *tail] of this shard for rows of that shape (one per shape and
        dtype, zero-filled once: padding rows of the all-gather layout stay zero).  `local_rows(tail)` is the view
        a producer writes this rank's rows into — a layer output that already lives there is not copied again."""
        o = self._origin
        tail = (tail,) if isinstance(tail, int) else tuple(tail)
        device = self._device if device is None else torch.device(device)
        key = (tail, dtype, str(device))
        buf = o._bufs.get(key)
        if buf is None:
            buf = o._bufs[key] = torch.zeros((self.n_buf,) + tail, dtype=dtype, device=device)
        return buf

    def local_rows(self, tail, dtype=torch.float32, device=None):
        """This rank's rows INSIDE the persistent exchange buffer: fill them (e.g. as the `out=` of the op that
        produces the layer's input) and hand them to `aggregate` / `halo_start`: no copy into the buffer."""
        return self.exchange_buffer(tail, dtype, device)[self.loc_off:self.loc_off + self.n_rows]

    def _send_buffer(self, tail, dtype, device):
        o = self._origin
        key = (tuple(tail), dtype, str(device))
        sb = o._sendbufs.get(key)
        if sb is None:
            sb = o._sendbufs[key] = torch.empty((int(self.send_idx.shape[0]),) + tuple(tail), dtype=dtype, device=device)
        return sb

    def _fill_send(self, x_local, send):
        """send[j] = x_local[send_idx[j]]: into an existing buffer (stag_gather_rows on the device)."""
        if send.shape[0] == 0:
            return send
        if x_local.is_cuda and x_local.dtype == torch.float32 and x_local.is_contiguous():
            from . import _lib
            W = int(np.prod(x_local.shape[1:])) if x_local.dim() > 1 else 1
            with _lib.on_device(x_local.device):
                rc = _lib.lib().stag_gather_rows(_lib.ptr(x_local), W, _lib.ptr(self.send_idx32), send.shape[0], W,
                                                 _lib.ptr(send), W, _lib.stream_of(x_local.device))
            _lib.check(rc, "stag_gather_rows")
        else:
            torch.index_select(x_local, 0, self.send_idx, out=send)
        return send

    def halo_start(self, x_local, persistent=False):
        """Begin the exchange: -> (buffer [n_buf, ...] whose local rows are filled, work | None).  The
        remote rows are valid after `work.wait()`, which orders the CURRENT stream behind the
        collective (RCCL runs it on its own stream): kernels launched in between overlap it.
        persistent=True: the shard's own buffer of this row shape (overwritten by the next exchange of that shape —
        for callers that keep nothing of it); else a fresh tensor."""
        bufs, work = self.halo_start_multi([x_local], persistent)
        return bufs[0], work

    def halo_start_multi(self, xs, persistent=False):
        """halo_start for several row tables bound for the same peers (GAT: ft [n, H, F] and el [n, H]): each gets
        its own buffer — no packed copy on either side — and, through the native communicator, they travel in ONE
        RCCL group; through torch.distributed it is one collective per table, back to back."""
        bufs, works, sends = [], [], []
        for x in xs:
            if x.shape[0] != self.n_rows:
                raise ValueError(f"rank {self.rank} owns {self.n_rows} rows, got {x.shape[0]}")
        native = self.native_comm if all(x.is_cuda and x.dtype == torch.float32 for x in xs) else None
        for x in xs:
            tail = tuple(x.shape[1:])
            buf = (self.exchange_buffer(tail, x.dtype, x.device) if persistent
                   else torch.empty((self.n_buf,) + tail, dtype=x.dtype, device=x.device))
            loc = buf[self.loc_off:self.loc_off + self.n_rows]
            if loc.data_ptr() != x.data_ptr() or not x.is_contiguous():
                loc.copy_(x)
            bufs.append(buf)
            if self.world == 1:
                if not persistent and self.n_buf > self.n_rows:
                    buf[self.n_rows:].zero_()
                continue
            if self.exchange == "halo":
                send = (self._send_buffer(tail, x.dtype, x.device) if persistent
                        else torch.empty((int(self.send_idx.shape[0]),) + tail, dtype=x.dtype, device=x.device))
                sends.append(self._fill_send(x.contiguous(), send))
                if native is None:
                    works.append(dist.all_to_all_single(buf[self.n_rows:], send, self.out_splits, self.in_splits,
                                                        group=self.group, async_op=True))
            else:
                slot = buf[self.rank * self.max_rows:(self.rank + 1) * self.max_rows]
                if not persistent and self.n_rows < self.max_rows:
                    slot[self.n_rows:].zero_()
                if native is not None:
                    native.allgather(slot, out=buf)          # in place: the slot IS this rank's part of buf
                else:
                    # (gloo stages through the output: hand it a copy of the slot; RCCL takes the slot in place)
                    src = slot if x.is_cuda else slot.clone()
                    works.append(dist.all_gather_into_tensor(buf, src, group=self.group, async_op=True))
        if self.world > 1 and self.exchange == "halo" and native is not None:
            widths = [int(np.prod(x.shape[1:])) if x.dim() > 1 else 1 for x in xs]
            works.append(native.exchange_multi_async(sends, self.in_splits, [b[self.n_rows:] for b in bufs],
                                                     self.out_splits, widths))
        return bufs, _wait_all(works)

    # ---- the transposed exchange (backward) ----------------------------------------------------------------
    def _transpose_start(self, g_remote, back):
        """Start sending the gradient rows of the remote buffer rows [n_buf - n_rows, W] back to their owners; `back`
        [n_send, W] receives what the peers computed for MY rows, in send-list order.  -> work | None."""
        if self.world == 1 or self.exchange != "halo":
            return None
        native = self.native_comm if (g_remote.is_cuda and g_remote.dtype == torch.float32) else None
        if native is not None:
            W = int(np.prod(g_remote.shape[1:])) if g_remote.dim() > 1 else 1
            return native.exchange_async(g_remote, self.out_splits, back, self.in_splits, W)
        return dist.all_to_all_single(back, g_remote, self.in_splits, self.out_splits, group=self.group, async_op=True)

    def _transpose_start_multi(self, g_remotes, backs):
        """_transpose_start for several tables bound for the same peers (GAT: d ft and d el): ONE RCCL group through
        the native communicator, back-to-back collectives through torch.distributed.  -> work | None."""
        if self.world == 1 or self.exchange != "halo":
            return None
        native = self.native_comm if all(g.is_cuda and g.dtype == torch.float32 for g in g_remotes) else None
        if native is not None:
            widths = [int(np.prod(g.shape[1:])) if g.dim() > 1 else 1 for g in g_remotes]
            return native.exchange_multi_async(list(g_remotes), self.out_splits, list(backs), self.in_splits, widths)
        return _wait_all([self._transpose_start(g, b) for g, b in zip(g_remotes, backs)])

    def _send_csr(self):
        """CSR over the send list: row i of this rank -> the positions j with send_idx[j] == i, ascending (peers in
        rank order).  A segmented sum over it on the aggregation kernel is the deterministic scatter-add."""
        o = self._origin
        if getattr(o, "_send_csr_view", None) is None:
            idx = self.send_idx
            order = torch.sort(idx, stable=True).indices
            cnt = torch.bincount(idx, minlength=self.n_rows)
            indptr = torch.zeros(self.n_rows + 1, dtype=torch.int64, device=idx.device)
            indptr[1:] = torch.cumsum(cnt, 0)
            o._send_csr_view = CsrView(self.n_rows, int(idx.shape[0]), indptr.to(torch.int32),
                                       order.to(torch.int32).contiguous(), None)
        return o._send_csr_view

    def _combined_csr(self):
        """CSR over T = [buffer rows | rows received back]: row i -> its own buffer row first, then the rows its
        peers sent for it, in rank order.  One launch finishes the backward of a partitioned aggregation."""
        o = self._origin
        if getattr(o, "_combined_csr_view", None) is None:
            idx = self.send_idx
            n, n_send, dev = self.n_rows, int(idx.shape[0]), idx.device
            order = torch.sort(idx, stable=True).indices
            cnt = torch.bincount(idx, minlength=n)
            indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            indptr[1:] = torch.cumsum(cnt + 1, 0)
            indices = torch.empty(n + n_send, dtype=torch.int64, device=dev)
            indices[indptr[:-1]] = self.loc_off + torch.arange(n, device=dev)
            if n_send:
                rows = idx[order]
                first = (torch.cumsum(cnt, 0) - cnt)[rows]                 # first sorted position of the row's group
                indices[indptr[rows] + 1 + (torch.arange(n_send, device=dev) - first)] = self.n_buf + order
            o._combined_csr_view = CsrView(n, self.n_buf + n_send, indptr.to(torch.int32),
                                           indices.to(torch.int32).contiguous(), None)
        return o._combined_csr_view

    def _segsum_back(self, back):
        """dx[i] = sum over j with send_idx[j] == i of back[j], j ascending — fixed order.  Device: the aggregation
        kernel over `_send_csr` (each row summed by one team, in order); CPU tensors (the gloo tests): index_add_,
        which is sequential there."""
        tail = tuple(back.shape[1:])
        if back.is_cuda and back.dtype == torch.float32:
            from . import _lib, ops
            W = int(np.prod(tail)) if tail else 1
            out, _ = ops._agg_raw(self._send_csr(), back.reshape(back.shape[0], W).contiguous(), W,
                                  ops._targs_or_c(ops._none_spec()), _lib.REDUCE_SUM, None, None, DEFAULT_SEG_LEN)
            return out.reshape((self.n_rows,) + tail)
        dx = torch.zeros((self.n_rows,) + tail, dtype=back.dtype, device=back.device)
        return dx.index_add_(0, self.send_idx, back)

    def halo_transpose(self, g_buf):
        return self.halo_transpose_multi([g_buf])[0]

    def halo_transpose_multi(self, gs):
        """The adjoint of the exchange: gradient of the buffers [n_buf, ...] -> gradient of this rank's rows
        [n_rows, ...]: own rows + (halo) what the peers computed for them, added in a fixed order, or (all-gather) the
        reduce-scatter of the padded shards."""
        outs = []
        if self.world == 1:
            return [g[self.loc_off:self.loc_off + self.n_rows].clone() for g in gs]
        if self.exchange == "halo":
            backs, works = [], []
            for g in gs:
                back = torch.empty((int(self.send_idx.shape[0]),) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
                works.append(self._transpose_start(g[self.n_rows:], back))
                backs.append(back)
            for w in works:
                if w is not None:
                    w.wait()
            for g, back in zip(gs, backs):
                outs.append(g[:self.n_rows] + self._segsum_back(back))
            return outs
        for g in gs:
            out = torch.empty((self.max_rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, g.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
            outs.append(out[:self.n_rows])
        return outs

    def halo_gather(self, x_local):
        """[n_rows, ...] on every rank -> [n_buf, ...] source rows this rank's CSR indexes
        (one collective: RCCL over xGMI on GPUs, gloo in the CPU tests).  Differentiable (`_HaloExchange`)."""
        if not (torch.is_grad_enabled() and x_local.requires_grad):
            buf, work = self.halo_start(x_local)
            if work is not None:
                work.wait()
            return buf
        return _HaloExchange.apply(self, x_local)[0]

    def halo_gather_multi(self, xs):
        """halo_gather for several row tables bound for the same peers, one autograd node (GAT: ft and el)."""
        if not (torch.is_grad_enabled() and any(x.requires_grad for x in xs)):
            bufs, work = self.halo_start_multi(list(xs))
            if work is not None:
                work.wait()
            return bufs
        return list(_HaloExchange.apply(self, *xs))

    def scatter_rows(self, x_global):
        """This rank's rows of a replicated [N, D] tensor (test / setup helper)."""
        return x_global[self.row_lo:self.row_hi]

    def buffer_scale(self, s_local):
        """A value per LOCAL node [n_rows] -> the value behind every buffer row [n_buf]: one exchange (every rank
        enters it, whatever it receives itself).  Keep the result: source-side scales are constants of a graph."""
        if s_local.shape[0] != self.n_rows:
            raise ValueError(f"rank {self.rank} owns {self.n_rows} rows, got {s_local.shape[0]}")
        with torch.no_grad():
            return self.halo_gather(s_local.detach().reshape(-1, 1)).reshape(-1).contiguous()

    def aggregate(self, x_local, weight=None, reduce="sum", src_scale_local=None,
                  dst_scale_local=None, seg_len=None, overlap=True, src_scale_buf=None):
        """One partitioned layer-forward: halo exchange + the single-GPU fused kernel on this rank's
        rows.  `weight`: None or an EdgeNoise built on this shard (its pos_base is forced to the
        shard's global offset).  Source-side scale: `src_scale_buf` [n_buf] indexes buffer rows (`out_degrees()`-
        derived; what ops.aggregate hands over), `src_scale_local` [n_rows] is exchanged first (`buffer_scale`: a
        collective — every rank must pass it).  The rows whose sources are all local are launched while the
        collective is in flight (`overlap`), forward and — under autograd — backward (`_ShardAggregate`)."""
        from . import ops
        from .noise import EdgeNoise
        seg_len = DEFAULT_SEG_LEN if seg_len is None else seg_len
        if isinstance(weight, EdgeNoise):
            weight.pos_base = self.pos_base
        if src_scale_local is not None:
            if src_scale_buf is not None:
                raise ValueError("one of src_scale_local / src_scale_buf")
            src_scale_buf = self.buffer_scale(src_scale_local)
        if src_scale_buf is not None and src_scale_buf.shape[0] != self.n_buf:
            raise ValueError(f"src_scale_buf indexes the {self.n_buf} rows of the exchanged buffer, got {src_scale_buf.shape[0]}")
        live_params = (isinstance(weight, EdgeNoise) and weight.grad_params is not None and torch.is_grad_enabled()
                       and any(torch.is_tensor(p_) and p_.requires_grad for p_ in weight.grad_params))
        fusable = (weight is None or (isinstance(weight, EdgeNoise) and weight.n_samples == 1 and not live_params))
        if fusable and x_local.is_cuda and seg_len > 0 and x_local.dim() == 2:
            if isinstance(weight, EdgeNoise) and weight.dn != x_local.shape[1]:
                raise ValueError(f"noise width {weight.dn} != feature width {x_local.shape[1]}")
            return _ShardAggregate.apply(x_local, self, weight, reduce, ops._f32c(src_scale_buf),
                                         ops._f32c(dst_scale_local), seg_len, bool(overlap))
        x_full = self.halo_gather(x_local)
        return ops.aggregate(self, x_full, weight, reduce=reduce, src_scale=src_scale_buf,
                             dst_scale=dst_scale_local, seg_len=seg_len, _gathered=True)

    def gat_aggregate(self, el_local, er_local, ft_local, neg_slope=0.2, weight=None, seg_len=None,
                      want_attn=False, attn_drop=None, overlap=True):
        """Partitioned GAT layer-forward (BASELINE cfg5): ft [n, H, F] and el [n, H] of the referenced source rows
        travel as two tables of ONE exchange step (one RCCL group through the native communicator) — no packed
        [ft | el] copy before, no column slices after —, then the single-GPU fused kernel runs on this
        rank's rows.  `weight`: None or an EdgeNoise(dn=H) built on this shard.  The unit batches whose sources are all
        local are launched while the exchange is in flight, and the backward sends the remote rows' gradients while the
        local ones are computed (`_ShardGat`; `overlap=False`: everything behind the collective, same bits)."""
        from . import ops
        from .noise import EdgeNoise
        seg_len = DEFAULT_SEG_LEN if seg_len is None else seg_len
        if isinstance(weight, EdgeNoise):
            weight.pos_base = self.pos_base
        live = (isinstance(weight, EdgeNoise) and weight.grad_params is not None and torch.is_grad_enabled()
                and any(torch.is_tensor(p_) and p_.requires_grad for p_ in weight.grad_params))
        H, F = (ft_local.shape[1], ft_local.shape[2]) if ft_local.dim() == 3 else (0, 0)
        if ((weight is None or (isinstance(weight, EdgeNoise) and weight.n_samples == 1 and not live)) and not want_attn
                and ft_local.is_cuda and self._csr.n_edges > 0 and ops.gat_cooperative_shape(H, F, seg_len)
                and (self.world == 1 or dist.is_initialized())      # (a caller that supplies its own exchange: below)
                and ops._GAT_BWD_FUSED and (attn_drop is None or ops.attn_drop_fusable(H, F, seg_len))):
            if isinstance(weight, EdgeNoise) and weight.dn != H:
                raise ValueError(f"noise width {weight.dn} != number of heads {H}")
            return _ShardGat.apply(el_local, er_local, ft_local, self, weight, float(neg_slope), seg_len, attn_drop,
                                   bool(overlap))
        ft_full, el_full = self.halo_gather_multi([ft_local, el_local])
        if self.n_rows == 0:
            # a cut that left this rank without rows: nothing to compute, but the exchange (and, under autograd, its
            # transposed twin in the backward) is a collective every rank enters
            empty = ft_full.new_zeros((0,) + tuple(ft_local.shape[1:])) + 0.0 * (ft_full.sum() + el_full.sum() + er_local.sum())
            return (empty, el_full.new_zeros((0, el_local.shape[1]))) if want_attn else empty
        # attn_drop: the mask is keyed by GLOBAL forward position (pos_base), so shards draw the whole graph's mask
        return ops.gat_aggregate(self, el_full, er_local, ft_full, neg_slope, weight, want_attn=want_attn,
                                 seg_len=seg_len, _gathered=True, attn_drop=attn_drop)

    # ---- per-edge endpoints over the buffer (AmortizedDistribution on a shard) ---------------------------------
    def edge_endpoints(self):
        """(src_buf, dst_buf) int32 [E_local], by LOCAL edge id (= forward CSR position): the buffer row of each
        edge's source and of its destination.  The per-edge MLP of an AmortizedDistribution
        (stag/distributions.py:221-233) reads its two projected node rows through these."""
        o = self._origin
        if getattr(o, "_edge_endpoints", None) is None:
            rows = torch.repeat_interleave(torch.arange(self.n_rows, dtype=torch.int32, device=self._device),
                                           (self.local_indptr[1:] - self.local_indptr[:-1]).long())
            o._edge_endpoints = (self.local_indices, (rows + self.loc_off).contiguous())
        return o._edge_endpoints


# ------------------------------------------------------------------------------------- #
def channel_bounds(D, world):
    """Column cut points [world+1], multiples of 4 (one Philox block = 4 channels), as even
    as that allows."""
    blocks = (D + 3) // 4
    cuts = [min(D, 4 * ((blocks * r) // world)) for r in range(world)] + [D]
    return cuts


class ChannelShard:
    """Rank `rank`'s channels [c_lo, c_hi) of a graph every rank holds whole.

    shard.aggregate(x_cols, noise) == ops.aggregate(graph, x, noise)[:, c_lo:c_hi] bit for bit;
    `noise` is an EdgeNoise of dn = c_hi - c_lo built on `graph` (per-channel / per-edge
    parameters are the shard's own columns)."""

    def __init__(self, graph, D, rank, world, group=None):
        self.graph, self.D, self.rank, self.world, self.group = graph, int(D), int(rank), int(world), group
        self.bounds = channel_bounds(self.D, self.world)
        self.c_lo, self.c_hi = self.bounds[rank], self.bounds[rank + 1]
        self.dn = self.c_hi - self.c_lo

    def scatter_cols(self, x_global):
        return x_global[:, self.c_lo:self.c_hi].contiguous()

    def aggregate(self, x_cols, weight=None, reduce="sum", src_scale=None, dst_scale=None, seg_len=None):
        from . import ops
        from .noise import EdgeNoise
        if x_cols.shape[1] != self.dn:
            raise ValueError(f"rank {self.rank} owns {self.dn} channels, got {x_cols.shape[1]}")
        if isinstance(weight, EdgeNoise):
            weight.chunk_base = self.c_lo // 4
        return ops.aggregate(self.graph, x_cols, weight, reduce=reduce, src_scale=src_scale,
                             dst_scale=dst_scale,
                             seg_len=DEFAULT_SEG_LEN if seg_len is None else seg_len)

    # ---- layout exchange for the dense transform that follows --------------------------
    def row_bounds(self):
        n = self.graph.number_of_dst_nodes() if hasattr(self.graph, "number_of_dst_nodes") else self.graph.number_of_nodes()
        return [(n * r) // self.world for r in range(self.world + 1)]

    def to_row_shards(self, out_cols):
        """[N, dn] channel shard -> [N/P, D] row shard: ONE all-to-all (RCCL over xGMI)."""
        if self.world == 1:
            return out_cols
        rb = self.row_bounds()
        n_mine = rb[self.rank + 1] - rb[self.rank]
        widths = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        in_splits = [(rb[r + 1] - rb[r]) * self.dn for r in range(self.world)]
        out_splits = [n_mine * w for w in widths]
        recv = torch.empty(sum(out_splits), dtype=out_cols.dtype, device=out_cols.device)
        dist.all_to_all_single(recv, out_cols.contiguous().reshape(-1), out_splits, in_splits,
                               group=self.group)
        parts = recv.split(out_splits)
        return torch.cat([p.reshape(n_mine, w) for p, w in zip(parts, widths)], 1)

    def to_channel_shards(self, y_rows):
        """[N/P, D] row shard -> [N, dn] channel shard: the inverse all-to-all."""
        if self.world == 1:
            return y_rows
        rb = self.row_bounds()
        n_mine = rb[self.rank + 1] - rb[self.rank]
        widths = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        send = torch.cat([y_rows[:, self.bounds[r]:self.bounds[r + 1]].reshape(-1)
                          for r in range(self.world)])
        in_splits = [n_mine * w for w in widths]
        out_splits = [(rb[r + 1] - rb[r]) * self.dn for r in range(self.world)]
        recv = torch.empty(sum(out_splits), dtype=y_rows.dtype, device=y_rows.device)
        dist.all_to_all_single(recv, send, out_splits, in_splits, group=self.group)
        return recv.reshape(-1, self.dn)
