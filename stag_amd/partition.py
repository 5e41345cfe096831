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
    the local CSR's column ids are pre-mapped to rows of the exchanged buffer, so the
    aggregation kernel is the single-GPU kernel, unchanged;
    Philox counters are keyed by the GLOBAL CSR position (`pos_base` = first global
    position of the shard), so 1/2/4/8-GPU outputs are bit-identical;
    backward = the transposed exchange (reduce-scatter of dx).

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

from .graph import CsrView, build_csr


def edge_balanced_bounds(indptr, world):
    """Row cut points [world+1]: rank g gets rows whose CSR positions straddle
    [g*E/world, (g+1)*E/world).  indptr: numpy int array [N+1]."""
    n = len(indptr) - 1
    E = int(indptr[-1])
    targets = (np.arange(1, world, dtype=np.float64) * E / world)
    cuts = np.searchsorted(indptr[1:], targets, side="left") + 1 if n else np.zeros(world - 1, int)
    bounds = np.concatenate([[0], np.minimum(cuts, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


class _HaloAllToAll(torch.autograd.Function):
    """Send each peer the local rows it references; backward = the transposed all-to-all,
    scatter-added into the local rows."""

    @staticmethod
    def forward(ctx, x_local, send_idx, in_splits, out_splits, group):
        ctx.group, ctx.in_splits, ctx.out_splits, ctx.n_local = group, in_splits, out_splits, x_local.shape[0]
        ctx.save_for_backward(send_idx)
        send = x_local.index_select(0, send_idx)
        recv = torch.empty((sum(out_splits),) + tuple(x_local.shape[1:]), dtype=x_local.dtype,
                           device=x_local.device)
        dist.all_to_all_single(recv, send, out_splits, in_splits, group=group)
        return recv

    @staticmethod
    def backward(ctx, g):
        (send_idx,) = ctx.saved_tensors
        back = torch.empty((sum(ctx.in_splits),) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.all_to_all_single(back, g.contiguous(), ctx.in_splits, ctx.out_splits, group=ctx.group)
        dx = torch.zeros((ctx.n_local,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dx.index_add_(0, send_idx, back)
        return dx, None, None, None, None


class _HaloGather(torch.autograd.Function):
    """all-gather of padded row shards; backward = reduce-scatter (sum) of the gradient."""

    @staticmethod
    def forward(ctx, x_pad, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = torch.empty((world * x_pad.shape[0],) + tuple(x_pad.shape[1:]), dtype=x_pad.dtype,
                          device=x_pad.device)
        dist.all_gather_into_tensor(out, x_pad.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        world = dist.get_world_size(ctx.group)
        out = torch.empty((g.shape[0] // world,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g.contiguous(), op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None


class GraphShard:
    """The rows of one rank, shaped like a stag_amd.Graph for ops.aggregate / EdgeNoise."""

    is_block = False

    def __init__(self, src, dst, n_nodes, rank, world, device=None, group=None, exchange="halo"):
        if exchange not in ("halo", "allgather"):
            raise ValueError("exchange must be 'halo' or 'allgather'")
        self.exchange = exchange if world > 1 else "allgather"
        src = np.asarray(src, dtype=np.int64)
        dst = np.asarray(dst, dtype=np.int64)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.n_global = int(n_nodes)
        E = len(src)
        # global stable dst-major order (same as graph.build_csr => same global positions)
        order = np.argsort(dst, kind="stable")
        counts = np.bincount(dst, minlength=n_nodes)
        indptr = np.zeros(n_nodes + 1, dtype=np.int64)
        indptr[1:] = np.cumsum(counts)
        self.bounds = edge_balanced_bounds(indptr, world)
        self.max_rows = int(np.max(np.diff(self.bounds))) if world else 0
        lo, hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.row_lo, self.row_hi = lo, hi
        p_lo, p_hi = int(indptr[lo]), int(indptr[hi])
        self.pos_base = p_lo
        self.n_edges_global = E
        g_src = src[order[p_lo:p_hi]]
        owner = np.searchsorted(self.bounds, g_src, side="right") - 1
        owner = np.minimum(owner, world - 1)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self._device = dev
        if self.exchange == "allgather":
            buf_row = owner * self.max_rows + (g_src - self.bounds[owner])
            self.n_buf = world * self.max_rows
        else:
            # buffer = [my rows | rows needed from rank 0 | from rank 1 | ...], each peer's part
            # sorted by global id.  Every rank derives every pair's list from the same global
            # CSR, so no negotiation round is needed.
            sorted_src = src[order]
            src_owner = np.minimum(np.searchsorted(self.bounds, sorted_src, side="right") - 1, world - 1)
            edge_rank = np.minimum(np.searchsorted(indptr[self.bounds], np.arange(E), side="right") - 1,
                                   world - 1)            # rank owning each CSR position
            def needed(r, q):    # global ids rank r needs from rank q
                m = (edge_rank == r) & (src_owner == q)
                return np.unique(sorted_src[m])
            recv_lists = [needed(rank, q) if q != rank else np.zeros(0, np.int64) for q in range(world)]
            send_lists = [needed(r, rank) if r != rank else np.zeros(0, np.int64) for r in range(world)]
            self.recv_ids = np.concatenate(recv_lists) if world > 1 else np.zeros(0, np.int64)   # global ids, buffer order
            self.out_splits = [len(l) for l in recv_lists]      # rows I receive from each peer
            self.in_splits = [len(l) for l in send_lists]       # rows I send to each peer
            self.send_idx = torch.from_numpy(
                (np.concatenate(send_lists) - lo).astype(np.int64) if sum(self.in_splits) else np.zeros(0, np.int64)).to(dev)
            offs = np.concatenate([[0], np.cumsum(self.out_splits)])[:-1] + (hi - lo)
            buf_row = np.empty(len(g_src), dtype=np.int64)
            mine = owner == rank
            buf_row[mine] = g_src[mine] - lo
            for q in range(world):
                if q == rank or not self.out_splits[q]:
                    continue
                m = owner == q
                buf_row[m] = offs[q] + np.searchsorted(recv_lists[q], g_src[m])
            self.n_buf = (hi - lo) + int(sum(self.out_splits))
        self.local_indptr = torch.from_numpy((indptr[lo:hi + 1] - p_lo).astype(np.int32)).to(dev)
        self.local_indices = torch.from_numpy(buf_row.astype(np.int32)).to(dev)
        self.local_eid_global = torch.from_numpy(order[p_lo:p_hi].astype(np.int64)).to(dev)
        self.n_rows = hi - lo
        self._csr = CsrView(self.n_rows, self.n_buf, self.local_indptr, self.local_indices, None)
        self._csr_t = None
        self._in_deg = torch.from_numpy(counts[lo:hi].astype(np.int64)).to(dev)

    # ---- Graph-like surface used by ops / EdgeNoise ---------------------------------
    device = property(lambda self: self._device)
    csr = property(lambda self: self._csr)

    def number_of_edges(self):
        return self._csr.n_edges

    def number_of_nodes(self):
        return self.n_rows

    def in_degrees(self):
        return self._in_deg

    def _cache_owner(self):
        return self

    @property
    def csr_t(self):
        """Source-major twin over the gathered buffer rows (backward)."""
        if self._csr_t is None:
            E = self._csr.n_edges
            rows = torch.repeat_interleave(
                torch.arange(self.n_rows, dtype=torch.int32, device=self._device),
                (self.local_indptr[1:] - self.local_indptr[:-1]).long())
            indptr, indices, eid = build_csr(rows, self.local_indices, self.n_rows, self.n_buf)
            # local position p is the forward position; its global noise index adds pos_base
            nidx = (eid.long() + self.pos_base).to(torch.int32) if E else eid
            self._csr_t = CsrView(self.n_buf, self.n_rows, indptr, indices, None, nidx)
        return self._csr_t

    # ---- the exchange step ---------------------------------------------------------
    def pad_rows(self, x_local):
        if x_local.shape[0] != self.n_rows:
            raise ValueError(f"rank {self.rank} owns {self.n_rows} rows, got {x_local.shape[0]}")
        if self.n_rows == self.max_rows:
            return x_local
        pad = torch.zeros((self.max_rows - self.n_rows,) + tuple(x_local.shape[1:]),
                          dtype=x_local.dtype, device=x_local.device)
        return torch.cat([x_local, pad], 0)

    def halo_gather(self, x_local):
        """[n_rows, D] on every rank -> [n_buf, D] source features this rank's CSR indexes
        (one collective: RCCL over xGMI on GPUs, gloo in the CPU tests)."""
        if self.exchange == "halo":
            if x_local.shape[0] != self.n_rows:
                raise ValueError(f"rank {self.rank} owns {self.n_rows} rows, got {x_local.shape[0]}")
            if not (torch.is_grad_enabled() and x_local.requires_grad):
                # inference: the collective writes straight behind the local rows (no concatenation copy)
                buf = torch.empty((self.n_buf,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
                buf[:self.n_rows].copy_(x_local)
                dist.all_to_all_single(buf[self.n_rows:], x_local.index_select(0, self.send_idx),
                                       self.out_splits, self.in_splits, group=self.group)
                return buf
            recv = _HaloAllToAll.apply(x_local, self.send_idx, self.in_splits, self.out_splits, self.group)
            return torch.cat([x_local, recv], 0)
        x_pad = self.pad_rows(x_local)
        if self.world == 1:
            return x_pad
        return _HaloGather.apply(x_pad, self.group)

    def scatter_rows(self, x_global):
        """This rank's rows of a replicated [N, D] tensor (test / setup helper)."""
        return x_global[self.row_lo:self.row_hi]

    def aggregate(self, x_local, weight=None, reduce="sum", src_scale_local=None,
                  dst_scale_local=None, seg_len=None):
        """One partitioned layer-forward: halo all-gather + the single-GPU fused kernel on
        this rank's rows.  `weight`: None or an EdgeNoise built on this shard (its
        pos_base is forced to the shard's global offset)."""
        from . import ops
        from .graph import DEFAULT_SEG_LEN
        from .noise import EdgeNoise
        x_full = self.halo_gather(x_local)
        if isinstance(weight, EdgeNoise):
            weight.pos_base = self.pos_base
        src_scale = None
        if src_scale_local is not None:
            src_scale = self.halo_gather(src_scale_local.reshape(-1, 1)).reshape(-1)
        return ops.aggregate(self, x_full, weight, reduce=reduce, src_scale=src_scale,
                             dst_scale=dst_scale_local,
                             seg_len=DEFAULT_SEG_LEN if seg_len is None else seg_len)

    def gat_aggregate(self, el_local, er_local, ft_local, neg_slope=0.2, weight=None, seg_len=None):
        """Partitioned GAT layer-forward (BASELINE cfg5): ONE exchange carries [ft | el] of the
        referenced source rows (H*F + H columns), then the single-GPU fused kernel runs on this
        rank's rows.  `weight`: None or an EdgeNoise(dn=H) built on this shard."""
        from . import ops
        from .graph import DEFAULT_SEG_LEN
        from .noise import EdgeNoise
        n, H, F = ft_local.shape
        packed = torch.cat([ft_local.reshape(n, H * F), el_local], 1)
        full = self.halo_gather(packed)
        ft_full = full[:, :H * F].reshape(-1, H, F)
        el_full = full[:, H * F:]
        er_rows = er_local
        if isinstance(weight, EdgeNoise):
            weight.pos_base = self.pos_base
        return ops.gat_aggregate(self, el_full, er_rows, ft_full, neg_slope, weight,
                                 seg_len=DEFAULT_SEG_LEN if seg_len is None else seg_len)


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
        from .graph import DEFAULT_SEG_LEN
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
