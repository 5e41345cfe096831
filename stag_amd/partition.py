"""Node-range partitioning of one graph over the GPUs of a node (BASELINE north_star;
new functionality — the reference is single-process).

    rank g owns the contiguous destination rows [bounds[g], bounds[g+1]) — cut so the
    ranks hold equal EDGE counts (in-degree is skewed) — the matching rows of `x` and
    of `out`, and the CSR of exactly those rows;
    per layer, ONE RCCL all-gather over xGMI brings the source features together
    (`halo_gather`: equal-size padded shards, so a single all_gather_into_tensor);
    the local CSR's column ids are pre-mapped to rows of that gathered buffer, so the
    aggregation kernel is the single-GPU kernel, unchanged;
    Philox counters are keyed by the GLOBAL CSR position (`pos_base` = first global
    position of the shard), so 1/2/4/8-GPU outputs are bit-identical;
    backward = the transposed exchange (reduce-scatter of dx).
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

    def __init__(self, src, dst, n_nodes, rank, world, device=None, group=None):
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
        buf_row = owner * self.max_rows + (g_src - self.bounds[owner])
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self._device = dev
        self.local_indptr = torch.from_numpy((indptr[lo:hi + 1] - p_lo).astype(np.int32)).to(dev)
        self.local_indices = torch.from_numpy(buf_row.astype(np.int32)).to(dev)
        self.local_eid_global = torch.from_numpy(order[p_lo:p_hi].astype(np.int64)).to(dev)
        self.n_rows = hi - lo
        self.n_buf = world * self.max_rows
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
        """[n_rows, D] on every rank -> [world*max_rows, D] gathered source features
        (one all_gather_into_tensor: RCCL over xGMI on GPUs, gloo in the CPU tests)."""
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
