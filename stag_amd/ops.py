"""Autograd-aware operators over libstag_hip.so.

`aggregate` is the hot path: one call = noise draw + relu + in-norm + degree
scaling + `update_all(u_mul_e, sum|mean)` of the reference (stag/layers.py:84-113,
stag/zoo/gcn.py:67-108), as one fused HIP pass.  Its backward walks the
source-major CSR and REDRAWS the forward's noise from the Philox counters
(`csr_t.nidx`), so no [E, D] tensor is saved for autograd.
"""
import ctypes as C

import torch

from . import _lib, _torch_ext
from .graph import DEFAULT_SEG_LEN
from .noise import EdgeNoise

_REDUCE = {"sum": _lib.REDUCE_SUM, "mean": _lib.REDUCE_MEAN}


def _owner(graph):
    """The graph object that owns the cached structure.  Autograd contexts keep THIS, never a
    `local_var()` copy: a copy's frames hold the step's tensors, whose grad_fn holds the context
    — a cycle through the autograd graph that kept an [E, D] tensor alive per training step."""
    return graph._cache_owner() if hasattr(graph, "_cache_owner") else graph


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# A noise description travels to the library in one of two forms: the ctypes `stag_noise_spec`, or — when
# the TORCH_LIBRARY front end is loaded (_torch_ext) — the argument tuple of torch.ops.stag.* (ints, 64-bit
# patterns, floats, the parameter tensors themselves).  _agg_raw / _agg_bwd_raw take either.
def _none_spec():
    if _torch_ext.available():
        return _NONE_ARGS
    s = _lib.NoiseSpec()
    s.kind = _lib.NOISE_NONE
    return s


def _explicit_spec(w, relu=False, in_norm=False, group=0):
    if _torch_ext.available():
        return ([_lib.NOISE_EXPLICIT, 0, int(relu), int(in_norm), 0, int(group), 0, 0], [0, 0, 0], [0.0, 0.0], w, None, None)
    s = _lib.NoiseSpec()
    s.kind = _lib.NOISE_EXPLICIT
    s.p0 = w.data_ptr()
    s.relu, s.in_norm, s.group = int(relu), int(in_norm), int(group)
    return s


def _noise_spec(noise, in_norm=None):
    """The spec of an EdgeNoise (in_norm optionally overridden: the backward passes redraw the raw weights)."""
    if _torch_ext.available():
        return noise.torch_args(in_norm=in_norm)
    s = noise.spec()
    if in_norm is not None:
        s.in_norm = int(in_norm)
    return s


def _targs_or_c(spec):
    """The ctypes form, whichever form came in (entry points bound through ctypes only)."""
    return _targs_to_ctypes(spec) if isinstance(spec, tuple) else spec


def _spec_in_norm(spec):
    return bool(spec[0][3]) if isinstance(spec, tuple) else bool(spec.in_norm)


def _plan_struct(csrv, seg_len, tiles, nbytes, dev, plan_t=None, width=None, gat_width=None, drawn=False):
    """ctypes stag_plan for csrv (None when planning is off), plus the tensors it points into.
    plan_t: a sub-plan of csrv (CsrView.subplan) instead of its whole plan.  width: the row width of an aggregation
    launch — it may walk the plan's XCD-aware order (stag_plan.xcd_order).  gat_width: H * F of a cooperative GAT
    launch — its unit batches may be the XCD-aware ones (stag_plan_blocks_xcd).  The other entry points use neither."""
    if plan_t is None:
        plan_t = csrv.plan(seg_len)
    if plan_t is None:
        return None, None
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev) if nbytes else None
    # arrival counters of the long rows: one set per (channel tiles, stream) — two launches that may
    # be in flight together (different streams) must not share them; zeroed once, every completed
    # launch leaves them zero again
    key = (tiles, _lib.stream_of(dev))
    counters = plan_t["counters"].get(key)
    if counters is None:
        counters = torch.zeros(max(plan_t["n_long"], 1) * tiles, dtype=torch.int32, device=dev)
        plan_t["counters"][key] = counters
    order, strides, fine = csrv.xcd_order(plan_t, width, drawn) if width and plan_t.get("xcd_on") else (None, (0, 0), 0)
    units, block_ptr, n_blocks = plan_t["units"], plan_t["block_ptr"], plan_t["n_blocks"]
    if gat_width and plan_t.get("xcd_on"):
        blocks = csrv.gat_blocks(plan_t, gat_width)
        if blocks is not None:
            units, block_ptr, n_blocks, fine = blocks[0], blocks[1], blocks[2], -blocks[3]
            order = units       # (non-NULL xcd_order: "these batches are XCD-local" — the GAT forward keeps two rows in flight)
    # the struct is kept per (tiles, stream, order): only the workspace changes from call to call (host time of a
    # call matters on launch-bound graphs).  Safe to reuse: the library reads it during the call only.
    plan_c = plan_t.setdefault("_structs", {}).get(key + (fine,))
    if plan_c is None:
        plan_c = _lib.Plan(plan_t["seg_len"], plan_t["n_units"], plan_t["n_long"], plan_t["n_seg"],
                           _lib.ptr(units), _lib.ptr(plan_t["long_rows"]),
                           _lib.ptr(plan_t["long_seg_ptr"]), _lib.ptr(counters), None, 0,
                           plan_t["n_heavy"] if units is plan_t["units"] else 0, n_blocks, _lib.ptr(block_ptr),
                           _lib.ptr(order), *strides)
        plan_t["_structs"][key + (fine,)] = plan_c
    plan_c.workspace, plan_c.workspace_bytes = _lib.ptr(ws), nbytes
    return plan_c, (ws, counters)


_NONE_ARGS = ([_lib.NOISE_NONE, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0], [0.0, 0.0], None, None, None)


def _plan_args(csrv, plan_t, tiles, dev, width=None, drawn=False):
    """(units, long_rows, long_seg_ptr, block_ptr, xcd, counters, plan_ints) of torch.ops.stag.*"""
    if plan_t is None:
        return (None, None, None, None, None, None, [0, 0, 0, 0, 0, 0, 0, 0])
    key = (tiles, _lib.stream_of(dev))
    counters = plan_t["counters"].get(key)
    if counters is None:
        counters = torch.zeros(max(plan_t["n_long"], 1) * tiles, dtype=torch.int32, device=dev)
        plan_t["counters"][key] = counters
    order, strides, _ = csrv.xcd_order(plan_t, width, drawn) if width and plan_t.get("xcd_on") else (None, (0, 0), 0)
    ints = [plan_t["seg_len"], plan_t["n_units"], plan_t["n_long"], plan_t["n_seg"], plan_t["n_heavy"], plan_t["n_blocks"],
            *strides]
    return (plan_t["units"], plan_t["long_rows"], plan_t["long_seg_ptr"], plan_t["block_ptr"], order, counters, ints)


def _agg_fwd_torch(csrv, x, noise_args, reduce, src_scale, dst_scale, seg_len, want_norm_scale, broadcast_x):
    """stag_agg_fwd through the dispatcher op (csrc/torch_ext.cpp): same library call, no ctypes."""
    dev = _lib.require_device(x, csrv.indptr, src_scale, dst_scale)
    D = x.numel() if broadcast_x else x.shape[1]
    plan_t = csrv.plan(seg_len)
    out, ns = torch.ops.stag.agg_fwd(*csrv.torch_args(), *_plan_args(csrv, plan_t, (D + 255) // 256, dev, width=D,
                                                                     drawn=noise_args[0][0] >= _lib.NOISE_NORMAL), x,
                                     broadcast_x, *noise_args, reduce, src_scale, dst_scale, want_norm_scale)
    return out, (ns if want_norm_scale else None)


def _agg_raw(csrv, x, D, spec, reduce, src_scale, dst_scale, seg_len, want_norm_scale=False,
             broadcast_x=False, out=None, plan_t=None, ns_out=None):
    """One stag_agg_fwd launch on csrv (a CsrView). x: [n_src, D] fp32 contiguous.
    out / plan_t: write the rows of a sub-plan's units into an existing [n_dst, D] tensor (ns_out: and their
    in-norm factors into an existing one)."""
    if isinstance(spec, tuple):
        if out is None and plan_t is None:
            return _agg_fwd_torch(csrv, x, spec, reduce, src_scale, dst_scale, seg_len, want_norm_scale, broadcast_x)
        spec = _targs_to_ctypes(spec)
    dev = _lib.require_device(x, csrv.indptr, src_scale, dst_scale)
    if out is None:
        out = torch.empty((csrv.n_dst, D), dtype=torch.float32, device=dev)
    ns = ns_out if ns_out is not None else (
        torch.empty((csrv.n_dst, D), dtype=torch.float32, device=dev) if want_norm_scale else None)
    if csrv.n_dst == 0:          # no row: nothing to launch (a node-range shard whose cut left it without rows)
        return out, ns
    if plan_t is None:
        plan_t = csrv.plan(seg_len)
    nbytes = (_lib.lib().stag_plan_workspace_bytes(plan_t["n_seg"], D, int(spec.in_norm))
              if plan_t is not None else 0)
    plan_c, _keep = _plan_struct(csrv, seg_len, (D + 255) // 256, nbytes, dev, plan_t, width=D,
                                 drawn=spec.kind >= _lib.NOISE_NORMAL)
    # (spec is the ctypes form from here on)
    cs = csrv.struct()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_agg_fwd(
            C.byref(cs), C.byref(plan_c) if plan_c is not None else None, _lib.ptr(x),
            0 if broadcast_x else x.stride(0), D, C.byref(spec), reduce, _lib.ptr(src_scale),
            _lib.ptr(dst_scale), _lib.ptr(out), D, _lib.ptr(ns), _lib.stream_of(dev))
    _lib.check(rc, "stag_agg_fwd")
    return out, ns


def _targs_to_ctypes(t):
    """The torch-op argument tuple as a ctypes stag_noise_spec (for the entry points bound through ctypes)."""
    ni, nu, nf, p0, p1, epoch = t
    s = _lib.NoiseSpec()
    s.kind, s.param_mode, s.relu, s.in_norm, s.deriv, s.group, s.chunk_base, s.p1_log = ni
    s.seed, s.offset, s.pos_base = nu[0] & ((1 << 64) - 1), nu[1] & ((1 << 64) - 1), nu[2]
    s.p0_scalar, s.p1_scalar = nf
    s.p0, s.p1, s.epoch = _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(epoch)
    return s


def _agg_bwd_raw(csrv_t, g, D, spec, g_scale, row_scale, seg_len, want_dp):
    """One stag_agg_bwd launch on the source-major CSR: dx and, if want_dp, the two per-row
    parameter-derivative aggregates (same gather, same Philox block)."""
    if isinstance(spec, tuple):
        dev = _lib.require_device(g, csrv_t.indptr, g_scale, row_scale)
        plan_t = csrv_t.plan(seg_len)
        dx, t0, t1 = torch.ops.stag.agg_bwd(*csrv_t.torch_args(), *_plan_args(csrv_t, plan_t, (D + 255) // 256, dev, width=D,
                                                                               drawn=spec[0][0] >= _lib.NOISE_NORMAL),
                                            g, *spec, g_scale, row_scale, want_dp)
        return dx, (t0 if want_dp else None), (t1 if want_dp else None)
    dev = _lib.require_device(g, csrv_t.indptr, g_scale, row_scale)
    dx = torch.empty((csrv_t.n_dst, D), dtype=torch.float32, device=dev)
    t0 = torch.empty_like(dx) if want_dp else None
    t1 = torch.empty_like(dx) if want_dp else None
    plan_t = csrv_t.plan(seg_len)
    nbytes = (_lib.lib().stag_plan_workspace_bytes(plan_t["n_seg"], (3 if want_dp else 1) * D, 0)
              if plan_t is not None else 0)
    plan_c, _keep = _plan_struct(csrv_t, seg_len, (D + 255) // 256, nbytes, dev, plan_t=plan_t, width=D,
                                 drawn=spec.kind >= _lib.NOISE_NORMAL)
    cs = csrv_t.struct()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_agg_bwd(
            C.byref(cs), C.byref(plan_c) if plan_c is not None else None, _lib.ptr(g), g.stride(0), D,
            C.byref(spec), _lib.ptr(g_scale), _lib.ptr(row_scale), _lib.ptr(dx), _lib.ptr(t0),
            _lib.ptr(t1), D, _lib.stream_of(dev))
    _lib.check(rc, "stag_agg_bwd")
    return dx, t0, t1


def _agg_bwd_edge_raw(csrv_t, g, x, D, spec, g_scale, row_scale, seg_len, want_dx=True):
    """One stag_agg_bwd_edge launch on the source-major CSR: dx (if wanted) and the gradients of the
    [E, 1] parameter pair of every edge, by edge id.  None when the shape has no one-pass form
    (D > 256: the caller takes stag_agg_bwd + stag_agg_bwd_w)."""
    if D > 256:
        return None
    dev = _lib.require_device(g, x, csrv_t.indptr, g_scale, row_scale)
    dx = torch.empty((csrv_t.n_dst, D), dtype=torch.float32, device=dev) if want_dx else None
    e0 = torch.empty((csrv_t.n_edges, 1), dtype=torch.float32, device=dev)
    e1 = torch.empty_like(e0)
    plan_t = csrv_t.plan(seg_len)
    nbytes = _lib.lib().stag_plan_workspace_bytes(plan_t["n_seg"], D, 0) if plan_t is not None else 0
    plan_c, _keep = _plan_struct(csrv_t, seg_len, 1, nbytes, dev, plan_t=plan_t, width=D, drawn=True)
    cs = csrv_t.struct()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_agg_bwd_edge(
            C.byref(cs), C.byref(plan_c) if plan_c is not None else None, _lib.ptr(g), g.stride(0), D,
            C.byref(spec), _lib.ptr(g_scale), _lib.ptr(row_scale), _lib.ptr(x), x.stride(0),
            _lib.ptr(dx), D, _lib.ptr(e0), _lib.ptr(e1), _lib.stream_of(dev))
    _lib.check(rc, "stag_agg_bwd_edge")
    return dx, e0, e1


def _agg_bwd_dp_raw(csrv_t, g, x, D, spec, g_scale, row_scale, seg_len, want_dx=True):
    """One stag_agg_bwd_dp launch on the source-major CSR: dx (if wanted) and the FINISHED gradients of scalar /
    per-channel parameters, dp0 [D], dp1 [D] (x = None: ones in its place)."""
    dev = _lib.require_device(g, x, csrv_t.indptr, g_scale, row_scale)
    if isinstance(spec, tuple):     # the dispatcher op (csrc/torch_ext.cpp)
        plan_t = csrv_t.plan(seg_len)
        dx, dp0, dp1 = torch.ops.stag.agg_bwd_dp(*csrv_t.torch_args(), *_plan_args(csrv_t, plan_t, (D + 255) // 256, dev),
                                                 g, x, *spec, g_scale, row_scale, bool(want_dx))
        return (dx if want_dx else None), dp0, dp1
    dx = torch.empty((csrv_t.n_dst, D), dtype=torch.float32, device=dev) if want_dx else None
    dp0 = torch.empty(D, dtype=torch.float32, device=dev)
    dp1 = torch.empty(D, dtype=torch.float32, device=dev)
    plan_t = csrv_t.plan(seg_len)
    nbytes = _lib.lib().stag_plan_workspace_bytes(plan_t["n_seg"], D, 0) if plan_t is not None else 0
    plan_c, _keep = _plan_struct(csrv_t, seg_len, (D + 255) // 256, nbytes, dev, plan_t=plan_t, width=D)
    n_units = plan_t["n_units"] if plan_t is not None else csrv_t.n_dst
    wbytes = _lib.lib().stag_agg_bwd_dp_workspace_bytes(n_units, D)
    ws = torch.empty(max(wbytes // 4, 1), dtype=torch.float32, device=dev)
    cs = csrv_t.struct()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_agg_bwd_dp(
            C.byref(cs), C.byref(plan_c) if plan_c is not None else None, _lib.ptr(g), g.stride(0), D,
            C.byref(spec), _lib.ptr(g_scale), _lib.ptr(row_scale), _lib.ptr(x), x.stride(0) if x is not None else 0,
            _lib.ptr(dx), D, _lib.ptr(dp0), _lib.ptr(dp1), _lib.ptr(ws), wbytes, _lib.stream_of(dev))
    _lib.check(rc, "stag_agg_bwd_dp")
    return dx, dp0, dp1


_AGG_BWD_DP_ONE_PASS = True     # stag_agg_bwd_dp | stag_agg_bwd + stag_coldot (A/B, and the torch-op argument form)


def coldot(x, t0, t1=None):
    """out_i[k] = sum_n x[n,k] * t_i[n,k]  (stag_coldot; no autograd — a backward-pass helper)."""
    dev = _lib.require_device(x, t0, t1)
    x, t0, t1 = _f32c(x), _f32c(t0), _f32c(t1)
    n, D = x.shape
    if n == 0:          # empty tensors have no device pointer to hand over
        z = torch.zeros(D, dtype=torch.float32, device=dev)
        return z, (z.clone() if t1 is not None else None)
    nbytes = _lib.lib().stag_coldot_workspace_bytes(D)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    o0 = torch.empty(D, dtype=torch.float32, device=dev)
    o1 = torch.empty(D, dtype=torch.float32, device=dev) if t1 is not None else None
    with _lib.on_device(dev):
        rc = _lib.lib().stag_coldot(_lib.ptr(x), x.stride(0), _lib.ptr(t0), _lib.ptr(t1), t0.stride(0), n, D,
                                    _lib.ptr(o0), _lib.ptr(o1), _lib.ptr(ws), nbytes, _lib.stream_of(dev))
    _lib.check(rc, "stag_coldot")
    return o0, o1


_COLSUM_OFFSETS = {}


def column_sum(g):
    """sum over the rows of g [N, D] — a bias gradient.  torch's reduction takes a slow path when D is not a multiple
    of 4 (575 us for [56,944, 121] on MI355X against 13 us for [56,944, 256]); those widths go through the readout
    kernel over 512 row chunks and a sum of the 512 partial rows (fixed order)."""
    n, D = g.shape
    if not g.is_cuda or D % 4 == 0 or n < 8192:
        return g.sum(0)
    g = _f32c(g)
    key = (n, g.device)
    offs = _COLSUM_OFFSETS.get(key)
    if offs is None:
        step = (n + 511) // 512
        offs = torch.clamp(torch.arange(513, dtype=torch.int64) * step, max=n).to(torch.int32).to(g.device)
        if len(_COLSUM_OFFSETS) > 64:
            _COLSUM_OFFSETS.clear()
        _COLSUM_OFFSETS[key] = offs
    return _segment_reduce_raw(g, offs, _lib.REDUCE_SUM, g.device).sum(0)


class _BiasAdd(torch.autograd.Function):
    """x + bias with the bias gradient by column_sum (autograd's own reduction for a broadcast add is the slow one)."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.needs_input_grad[0] else None), (column_sum(g).reshape(-1) if ctx.needs_input_grad[1] else None)


def add_bias(x, bias):
    """x [N, D] + bias [D]; the bias gradient avoids torch's slow column reduction for D % 4 != 0."""
    if x.dim() != 2 or not x.is_cuda or bias.dim() != 1 or x.shape[1] % 4 == 0:
        return x + bias
    return _BiasAdd.apply(x, bias)


class _NodeLinear(torch.autograd.Function):
    """y = x @ w for a tall x [N, in] (the dense transform after an aggregation,
    stag/zoo/gcn.py:97-98).  Forward and dx are plain rocBLAS/hipBLASLt GEMMs; the weight gradient
    x^T g is a K = N reduction that the library runs as ONE tile column (403 us at N = 169,343,
    128 x 128, MI355X) — here it is split over K into a batched GEMM + sum (67 us)."""

    SPLIT = 64

    @staticmethod
    def forward(ctx, x, w, bias=None, add=None):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.has_add = add is not None
        # the bias — or a whole addend [rows, out]: GraphSAGE's other branch — rides in the GEMM's epilogue (one
        # pass over the result instead of two)
        if add is not None:
            return torch.addmm(add if bias is None else add + bias, x, w)
        return torch.addmm(bias, x, w) if bias is not None else x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        # g @ w^T as an NN product on a transposed copy of w (64 KB): 75 against 92 us for the NT form at
        # N = 169,343, 128 x 128 (tools/gemm_probe.py)
        # (and when the output is wider than the input — GAT's fc, 128 -> 8 x 32 — the NT form on a contiguous w
        # wins instead: 119 against 158 us; the library's kernel choice is what differs)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (torch.nn.functional.linear(g, w.contiguous()) if g.shape[1] > w.shape[0]
                  else g @ w.t().contiguous())
        db = column_sum(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dw = None
        if ctx.needs_input_grad[1]:
            n, S = x.shape[0], _NodeLinear.SPLIT
            n1 = (n // S) * S
            if n1 >= 64 * S:
                g = g.contiguous()
                dw = torch.bmm(x[:n1].view(S, n1 // S, x.shape[1]).transpose(1, 2),
                               g[:n1].view(S, n1 // S, g.shape[1])).sum(0)
                if n1 < n:
                    dw = dw + x[n1:].t() @ g[n1:]
            else:
                dw = x.t() @ g
        return dx, dw, db, (g if (ctx.has_add and ctx.needs_input_grad[3]) else None)


def node_linear(x, w, bias=None, add=None):
    """x [N, in] @ w [in, out] (+ bias, + add [N, out], both in the GEMM's epilogue) with a split-K weight gradient
    (see _NodeLinear)."""
    if x.dim() == 3 and x.is_contiguous() and add is None:
        # [S, N, in]: the Monte-Carlo samples of one layer (aggregate_mc) — ONE product over S*N rows, so the weight
        # gradient keeps its split-K form instead of a K = S*N single-tile-column reduction
        S, n = x.shape[0], x.shape[1]
        return _NodeLinear.apply(x.view(S * n, x.shape[2]), w, bias).view(S, n, w.shape[1])
    if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        y = x @ w if bias is None else x @ w + bias
        return y if add is None else y + add
    # (rows may be strided — the [:, :D] view of a padded aggregation result: the GEMMs take the leading dimension,
    # the split-K weight gradient views the rows in groups, neither needs a packed copy)
    if add is not None and (add.shape != (x.shape[0], w.shape[1]) or add.dtype != x.dtype):
        return _NodeLinear.apply(x, w, bias) + add
    return _NodeLinear.apply(x, w, bias, add)


class _GatherRows(torch.autograd.Function):
    """x[src] or x[dst] as an [E, D] tensor by edge id.  Forward is an index_select; the backward —
    a scatter-add of E rows into N — is the aggregation kernel with the incoming gradient as
    explicit edge weights over one broadcast row of ones (torch's index backward takes 3.9 ms
    for [1.17M, 128] on MI355X, this 0.2 ms)."""

    @staticmethod
    def forward(ctx, x, graph, which):
        if getattr(graph, "is_shard", False):     # sources are buffer rows, destinations this rank's rows
            src, dst = graph.edge_endpoints()
            if which == "dst":
                dst = dst - graph.loc_off
        else:
            src, dst = graph.edges()
        ctx.graph, ctx.which = _owner(graph), which
        return x.index_select(0, (src if which == "src" else dst).long())

    @staticmethod
    def backward(ctx, g):
        graph = ctx.graph
        g = _f32c(g)
        csrv = graph.csr_t if ctx.which == "src" else graph.csr
        D = g.shape[1]
        ones = torch.ones(D, dtype=torch.float32, device=g.device)
        out, _ = _agg_raw(csrv, ones, D, _explicit_spec(g), _lib.REDUCE_SUM, None, None, DEFAULT_SEG_LEN,
                          broadcast_x=True)
        return out, None, None


def gather_rows(graph, x, which):
    """x[u] (which="src") or x[v] (which="dst") for every edge u -> v, rows by edge id."""
    if (x.dim() != 2 or not x.is_cuda) and not getattr(graph, "is_shard", False):
        src, dst = graph.edges()
        return x[(src if which == "src" else dst).long()]
    return _GatherRows.apply(x, graph, which)


# ---- amortised per-edge parameters with narrow heads (csrc/amort.hip) ---------------------------------------
# MI355X, N = 169,343, K = 128, forward + backward: 122 | 125 | 137 | 238 us for 2 | 4 | 8 | 16 columns against
# 193-273 us for the library GEMMs: the one-pass kernels take up to 8 columns (the library builds them for 16)
NARROW_MAX_HIDDEN, NARROW_MAX_PAR, NARROW_MAX_COLS = 4, 4, 8


def _amort_ws(n_values, dev):
    nbytes = _lib.lib().stag_amort_workspace_bytes(n_values)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=dev), nbytes


class _NodeProject(torch.autograd.Function):
    """y = x [N, K] . w [K, C] + b for C <= 16 columns: one pass over x forward, one pass backward (dx, dw, db
    together) — the library GEMMs take 75 us per call at N = 169,343, K = 128, C = 1 and run one such call per
    projected column and per gradient (stag_node_project_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, w, b):
        dev = _lib.require_device(x, w, b)
        x, w, b = _f32c(x), _f32c(w), _f32c(b)
        n, K = x.shape
        C_ = w.shape[1]
        y = torch.empty((n, C_), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_node_project_fwd(_lib.ptr(x), x.stride(0), n, K, _lib.ptr(w), _lib.ptr(b), C_,
                                                  _lib.ptr(y), _lib.stream_of(dev))
        _lib.check(rc, "stag_node_project_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        dev = x.device
        gy = _f32c(gy)
        n, K = x.shape
        C_ = w.shape[1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if dx is None and not want_w and not want_b:
            return None, None, None
        dw = torch.empty_like(w) if want_w else None
        db = torch.empty(C_, dtype=torch.float32, device=dev) if want_b else None
        ws, nbytes = _amort_ws((K + 1) * C_, dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_node_project_bwd(_lib.ptr(x), x.stride(0), n, K, _lib.ptr(w), C_, _lib.ptr(gy),
                                                  _lib.ptr(dx), K, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), nbytes,
                                                  _lib.stream_of(dev))
        _lib.check(rc, "stag_node_project_bwd")
        return dx, dw, db


def node_project(x, w, b=None):
    """x [N, K] @ w [K, C] (+ b) for a handful of output columns (C <= 8; wider: node_linear)."""
    if w.shape[1] > NARROW_MAX_COLS or not x.is_cuda or x.dim() != 2:
        return node_linear(x, w, b)
    return _NodeProject.apply(x, w, b)


class _HeadDot(torch.autograd.Function):
    """(ft * attn_l).sum(-1) and (ft * attn_r).sum(-1) — GAT's el / er (stag/zoo/gat.py:109-110) — from one pass over
    ft [N, H, F], and d ft, d attn_l, d attn_r from one pass back (stag_head_dot_fwd / _bwd).  As elementwise
    multiply + reduce: 2 x 242 us at cfg5 and as much again backward; as a GEMM with a block-diagonal right side:
    53 + 168 + 47 us in the library."""

    @staticmethod
    def forward(ctx, ft, al, ar):
        dev = _lib.require_device(ft, al, ar)
        n, H, F = ft.shape
        ft = _f32c(ft)
        w = torch.stack([al.reshape(H * F), ar.reshape(H * F)], 0).float().contiguous()
        y = torch.empty((2, n, H), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_head_dot_fwd(_lib.ptr(ft), H * F, n, H, F, _lib.ptr(w), 2, _lib.ptr(y),
                                              _lib.stream_of(dev))
        _lib.check(rc, "stag_head_dot_fwd")
        ctx.save_for_backward(ft, w)
        ctx.wshape = (al.shape, ar.shape)
        return y[0], y[1]

    @staticmethod
    def backward(ctx, gl, gr):
        ft, w = ctx.saved_tensors
        dev = ft.device
        n, H, F = ft.shape
        gy = torch.stack([_f32c(gl), _f32c(gr)], 0)
        dft = torch.empty_like(ft) if ctx.needs_input_grad[0] else None
        want_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dw = torch.empty_like(w) if want_w else None
        ws, nbytes = _amort_ws(2 * H * F, dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_head_dot_bwd(_lib.ptr(ft), H * F, n, H, F, _lib.ptr(w), 2, _lib.ptr(gy), _lib.ptr(dft),
                                              H * F, _lib.ptr(dw), _lib.ptr(ws), nbytes, _lib.stream_of(dev))
        _lib.check(rc, "stag_head_dot_bwd")
        dal = dw[0].reshape(ctx.wshape[0]) if (want_w and ctx.needs_input_grad[1]) else None
        dar = dw[1].reshape(ctx.wshape[1]) if (want_w and ctx.needs_input_grad[2]) else None
        return dft, dal, dar


def head_dot(ft, attn_l, attn_r):
    """el, er [N, H] = (ft * attn_l).sum(-1), (ft * attn_r).sum(-1) for ft [N, H, F], attn_* [1, H, F] | [H, F].
    None when the shape has no one-pass form (F % 4 != 0 or F > 256): the caller takes the GEMM form."""
    if ft.dim() != 3 or not ft.is_cuda or ft.shape[0] == 0:
        return None
    F = ft.shape[2]
    if F < 4 or F > 256 or F % 4 != 0:
        return None
    return _HeadDot.apply(ft, attn_l, attn_r)


class _EdgeMlp(torch.autograd.Function):
    """par_c[e] = b_c + sum_j SiLU(P[src_e, j] + P[dst_e, hidden + j]) wh[j, c]: the heads of an
    AmortizedDistribution with narrow hidden / output widths (stag/distributions.py:178-191, 225-231) on the two
    projected node tables, a thread per edge (stag_edge_mlp_fwd / _bwd).  Returns one [E, 1] tensor per head."""

    @staticmethod
    def forward(ctx, graph, P, wh, bh):
        dev = _lib.require_device(P, wh, bh)
        P, wh, bh = _f32c(P), _f32c(wh), _f32c(bh)
        hidden, n_par = wh.shape
        g = _owner(graph)
        src, dst = g._src, g._dst
        E = src.shape[0]
        par = torch.empty((n_par, E), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_edge_mlp_fwd(_lib.ptr(src), _lib.ptr(dst), E, _lib.ptr(P), _lib.ptr(P[:, hidden:]),
                                              P.stride(0), hidden, _lib.ptr(wh), _lib.ptr(bh), n_par, _lib.ptr(par),
                                              _lib.stream_of(dev))
        _lib.check(rc, "stag_edge_mlp_fwd")
        ctx.graph = g
        ctx.has_bias = bh is not None
        ctx.save_for_backward(P, wh)
        return tuple(par[c].unsqueeze(1) for c in range(n_par))

    @staticmethod
    def backward(ctx, *gs):
        P, wh = ctx.saved_tensors
        g = ctx.graph
        dev = P.device
        hidden, n_par = wh.shape
        src, dst = g._src, g._dst
        E = src.shape[0]
        gpar = torch.stack([_f32c(t).reshape(E) for t in gs], 0)
        dpre = torch.empty((E, hidden), dtype=torch.float32, device=dev)
        dwh = torch.empty_like(wh) if ctx.needs_input_grad[2] else None
        dbh = torch.empty(n_par, dtype=torch.float32, device=dev) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        ws, nbytes = _amort_ws(36, dev)      # 8 hidden x 4 parameters + 4: the library's largest shape
        with _lib.on_device(dev):
            rc = _lib.lib().stag_edge_mlp_bwd(_lib.ptr(src), _lib.ptr(dst), E, _lib.ptr(P), _lib.ptr(P[:, hidden:]),
                                              P.stride(0), hidden, _lib.ptr(wh), n_par, _lib.ptr(gpar), _lib.ptr(dpre),
                                              _lib.ptr(dwh), _lib.ptr(dbh), _lib.ptr(ws), nbytes, _lib.stream_of(dev))
        _lib.check(rc, "stag_edge_mlp_bwd")
        dP = None
        if ctx.needs_input_grad[1]:
            # d P[u, j] = sum over the out-edges of u, d P[v, hidden + j] = sum over the in-edges of v of dpre[e, j]:
            # the aggregation kernel over explicit rows and one broadcast row of ones, written side by side
            ones = torch.ones(hidden, dtype=torch.float32, device=dev)
            spec = _explicit_spec(dpre)
            d_src, _ = _agg_raw(g.csr_t, ones, hidden, spec, _lib.REDUCE_SUM, None, None, DEFAULT_SEG_LEN, broadcast_x=True)
            d_dst, _ = _agg_raw(g.csr, ones, hidden, spec, _lib.REDUCE_SUM, None, None, DEFAULT_SEG_LEN, broadcast_x=True)
            if getattr(g, "is_shard", False):      # P is the exchanged buffer: destinations are this rank's rows of it
                full = torch.zeros_like(d_src)
                full[g.loc_off:g.loc_off + g.n_rows] = d_dst
                d_dst = full
            dP = torch.cat([d_src, d_dst], 1)
        return None, dP, dwh, dbh


def edge_mlp(graph, P, wh, bh=None):
    """The per-edge heads of a narrow AmortizedDistribution: P [N, 2 hidden] (source half | destination half),
    wh [hidden, n_par], bh [n_par]  ->  n_par tensors [E, 1] by edge id."""
    return _EdgeMlp.apply(graph, P, wh, bh)


class _NormalKlMean(torch.autograd.Function):
    """mean over all elements of KL(N(loc, exp(log_scale)) || N(p_loc, p_scale)) with one-element prior parameters
    (torch.distributions.kl._kl_normal_normal; stag/layers.py:132-145): one pass forward, one backward, against
    ~25 elementwise passes over the [E, out] tensors (stag_normal_kl_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, loc, log_scale, p_loc, p_scale):
        dev = _lib.require_device(loc, log_scale, p_loc, p_scale)
        loc, log_scale = _f32c(loc), _f32c(log_scale)
        p_loc, p_scale = _f32c(p_loc.reshape(1)), _f32c(p_scale.reshape(1))
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ws, nbytes = _amort_ws(2, dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_normal_kl_fwd(_lib.ptr(loc), _lib.ptr(log_scale), loc.numel(), _lib.ptr(p_loc),
                                               _lib.ptr(p_scale), _lib.ptr(out), _lib.ptr(ws), nbytes, _lib.stream_of(dev))
        _lib.check(rc, "stag_normal_kl_fwd")
        ctx.save_for_backward(loc, log_scale, p_loc, p_scale)
        ctx.shapes = (loc.shape, log_scale.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        loc, log_scale, p_loc, p_scale = ctx.saved_tensors
        dev = loc.device
        g = _f32c(g.reshape(1))
        need = ctx.needs_input_grad
        dloc = torch.empty_like(loc) if need[0] else None
        dls = torch.empty_like(log_scale) if need[1] else None
        dpl = torch.empty(1, dtype=torch.float32, device=dev) if need[2] else None
        dps = torch.empty(1, dtype=torch.float32, device=dev) if need[3] else None
        ws, nbytes = _amort_ws(2, dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_normal_kl_bwd(_lib.ptr(loc), _lib.ptr(log_scale), loc.numel(), _lib.ptr(p_loc),
                                               _lib.ptr(p_scale), _lib.ptr(g), _lib.ptr(dloc), _lib.ptr(dls),
                                               _lib.ptr(dpl), _lib.ptr(dps), _lib.ptr(ws), nbytes, _lib.stream_of(dev))
        _lib.check(rc, "stag_normal_kl_bwd")
        return dloc, dls, dpl, dps


def normal_kl_mean(loc, log_scale, p_loc, p_scale):
    """KL(N(loc, exp(log_scale)) || N(p_loc, p_scale)).mean(), loc / log_scale of one shape, the prior's two
    parameters one-element tensors (their original shapes get their gradients back)."""
    if loc.shape != log_scale.shape or p_loc.numel() != 1 or p_scale.numel() != 1 or loc.numel() == 0:
        raise ValueError("normal_kl_mean: same-shape loc / log_scale and a one-element prior")
    return _NormalKlMean.apply(loc, log_scale, p_loc, p_scale)


def _bwd_w_raw(csrv, x, g, D, src_scale, broadcast_x=False, spec=None, reduce_k=False, both=False,
               seg_len=DEFAULT_SEG_LEN):
    """stag_agg_bwd_w over the plan's units.  both=True: (d/dp0, d/dp1) of a Normal | Uniform spec
    from one pass; else the single tensor selected by spec.deriv (or the plain dw)."""
    dev = _lib.require_device(x, g)
    cols = 1 if reduce_k else D
    dw = torch.empty((csrv.n_edges, cols), dtype=torch.float32, device=dev)
    dw1 = torch.empty_like(dw) if both else None
    plan_c, _keep = _plan_struct(csrv, seg_len, 1, 0, dev)
    cs = csrv.struct()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_agg_bwd_w(C.byref(cs), C.byref(plan_c) if plan_c is not None else None,
                                       _lib.ptr(x), 0 if broadcast_x else x.stride(0),
                                       _lib.ptr(g), g.stride(0), D, _lib.ptr(src_scale),
                                       C.byref(spec) if spec is not None else None, int(reduce_k),
                                       _lib.ptr(dw), _lib.ptr(dw1), cols, _lib.stream_of(dev))
    _lib.check(rc, "stag_agg_bwd_w")
    return (dw, dw1) if both else dw


class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, graph, noise, reduce, src_scale, dst_scale, seg_len, broadcast_x):
        x = _f32c(x)
        D = x.shape[1]
        csrv = graph.csr
        if noise is not None:
            spec = _noise_spec(noise)  # ctx.noise keeps the parameter tensors alive
        elif w is not None:
            w = _f32c(w)
            spec = _explicit_spec(w)
        else:
            spec = _none_spec()
        # the in-norm factor is only kept (one more [N, D] store) when a backward can follow
        want_ns = _spec_in_norm(spec) and any(ctx.needs_input_grad[:2])
        out, ns = _agg_raw(csrv, x, D, spec, reduce, src_scale, dst_scale, seg_len,
                           want_norm_scale=want_ns, broadcast_x=broadcast_x)
        ctx.graph, ctx.noise, ctx.reduce, ctx.seg_len = _owner(graph), noise, reduce, seg_len
        ctx.broadcast_x = broadcast_x
        ctx.D = D
        need_x = w is not None and ctx.needs_input_grad[1]   # x is only read by dw
        ctx.save_for_backward(x if need_x else None, w, src_scale, dst_scale, ns)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, w, src_scale, dst_scale, ns = ctx.saved_tensors
        graph, noise = ctx.graph, ctx.noise
        D = ctx.D
        g = _f32c(grad_out)
        if ns is not None:                      # in-norm factor is a constant of the backward
            g = g * ns
        dvec = dst_scale
        if ctx.reduce == _lib.REDUCE_MEAN:
            inv = 1.0 / graph.csr.degrees.clamp(min=1).to(torch.float32)
            dvec = inv if dvec is None else dvec * inv
        dx = dw = None
        if ctx.needs_input_grad[0] and not ctx.broadcast_x:
            if noise is not None:
                spec = _noise_spec(noise, in_norm=0)
            elif w is not None:
                spec = _explicit_spec(w)
            else:
                spec = _none_spec()
            # dx[u,:] = ss[u] * sum_{p: src_p = u} w[p,:] * dvec[v_p] * g[v_p,:]
            dx, _ = _agg_raw(graph.csr_t, g, D, spec, _lib.REDUCE_SUM, dvec, src_scale, ctx.seg_len)
        if w is not None and ctx.needs_input_grad[1]:
            gg = g if dvec is None else g * dvec.unsqueeze(1)
            dw = _bwd_w_raw(graph.csr, x, gg.contiguous(), D, src_scale, broadcast_x=ctx.broadcast_x,
                            seg_len=ctx.seg_len)
        return dx, dw, None, None, None, None, None, None, None


class _AggregateVI(torch.autograd.Function):
    """Fused aggregation whose noise parameters carry gradients (`vi=True`, reparameterised
    draw w = p0 + p1 * z | low + (high-low) u; stag/layers.py:123-124).  Nothing [E, D]-sized
    is saved: the backward (stag_agg_bwd) redraws z from the counters and returns dx together with
    the two parameter-derivative aggregates from one pass."""

    @staticmethod
    def forward(ctx, x, p0, p1, graph, noise, reduce, src_scale, dst_scale, seg_len):
        x = _f32c(x)
        D = x.shape[1]
        spec = _noise_spec(noise)
        out, ns = _agg_raw(graph.csr, x, D, spec, reduce, src_scale, dst_scale, seg_len,
                           want_norm_scale=_spec_in_norm(spec))
        ctx.graph, ctx.noise, ctx.reduce, ctx.seg_len, ctx.D = _owner(graph), noise, reduce, seg_len, D
        ctx.shapes = (p0.shape, p1.shape)
        # in-norm (stag/layers.py:8-36) is differentiated too: its factor s = indeg / sum_in(w) and the
        # output are kept ([N, D] each; nothing [E, D]-sized)
        ctx.save_for_backward(x, src_scale, dst_scale, ns, out if ns is not None else None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, src_scale, dst_scale, ns, out = ctx.saved_tensors
        graph, noise, D = ctx.graph, ctx.noise, ctx.D
        g = _f32c(grad_out)
        dvec = dst_scale
        if ctx.reduce == _lib.REDUCE_MEAN:
            inv = 1.0 / graph.csr.degrees.clamp(min=1).to(torch.float32)
            dvec = inv if dvec is None else dvec * inv
        spec = _noise_spec(noise, in_norm=0)     # the backward redraws the RAW weights; the factor rides in g
        spec_c = spec if not isinstance(spec, tuple) else _targs_to_ctypes(spec)   # stag_agg_bwd_w goes through ctypes
        q = None
        if ns is not None:
            # out = dv * s * A, A = sum_e w x', s = indeg / W, W = sum_e w  =>  with g' = g * s (dv applied by
            # the kernel):  dL/dw[e,k] = dv g'[v,k] (x'[u,k] - A/W),  A/W = out / (dv * indeg).
            # The x' part is the usual pass with g'; the other part is the same aggregate of dw/dp with
            #   q[v,k] = dv g' A/W = g[v,k] s[v,k] out[v,k] / indeg[v]   in the place of g and no x.
            deg = graph.csr.degrees.clamp(min=1).to(torch.float32).unsqueeze(1)
            q = (g * ns * out / deg).contiguous()
            g = (g * ns).contiguous()
        dx = dp0 = dp1 = None
        per_edge = noise.param_mode >= _lib.PARAM_PER_EDGE1
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if not per_edge and need_p and _AGG_BWD_DP_ONE_PASS:
            # scalar / per-channel parameters: ONE transposed pass yields dx AND the finished gradients
            #   dp_i[k] = sum_e dw/dp_i[e,k] dv g'[v,k] s_u x[u,k]   (x[u] is the unit's own row there)
            dx, c0, c1 = _agg_bwd_dp_raw(graph.csr_t, g, x, D, spec, dvec, src_scale, ctx.seg_len,
                                         want_dx=ctx.needs_input_grad[0])
            if q is not None:       # in-norm: minus sum_e dw/dp_i[e,k] q[v,k], the same pass over q without x
                _, n0, n1 = _agg_bwd_dp_raw(graph.csr_t, q, None, D, spec, None, None, ctx.seg_len, want_dx=False)
                c0, c1 = c0 - n0, c1 - n1
            rows = {1: c0, 2: c1}
        elif not per_edge:
            # the two-step form: the two derivative aggregates T_i[u,k] = ss[u] sum_p dw/dp_i[p,k] g'[v_p,k], then
            # dp_i[k] = sum_u x[u,k] T_i[u,k]   (sum_e D[e,k] s_u x[u,k] g'[v,k] regrouped by u)
            dx, t0, t1 = _agg_bwd_raw(graph.csr_t, g, D, spec, dvec, src_scale, ctx.seg_len, need_p)
            if not ctx.needs_input_grad[0]:
                dx = None
            if need_p:      # dp_i[k] = sum_u x[u,k] T_i[u,k], both in one pass over x
                c0, c1 = coldot(x, t0, t1)
                if q is not None:   # in-norm: minus sum_e dw/dp_i[e,k] q[v,k], the same pass over q
                    _, n0, n1 = _agg_bwd_raw(graph.csr_t, q, D, spec, None, None, ctx.seg_len, True)
                    c0, c1 = c0 - n0.sum(0), c1 - n1.sum(0)
                rows = {1: c0, 2: c1}
        elif need_p and q is None and noise.param_mode == _lib.PARAM_PER_EDGE1 and D <= 256:
            # [E, 1] (amortised) parameters: dx and both per-edge gradients from ONE transposed pass — the
            # row of x an edge's gradient needs is the unit's own row there
            dx, e0, e1 = _agg_bwd_edge_raw(graph.csr_t, g, x, D, spec_c, dvec, src_scale, ctx.seg_len,
                                           want_dx=ctx.needs_input_grad[0])
            rows = {1: e0, 2: e1}
            need_p = False
        elif ctx.needs_input_grad[0]:
            dx, _, _ = _agg_bwd_raw(graph.csr_t, g, D, spec, dvec, src_scale, ctx.seg_len, False)
        if per_edge and need_p:
            # per-edge (amortised) parameters: both derivatives from ONE pass over the edges
            gg = (g if dvec is None else g * dvec.unsqueeze(1)).contiguous()
            rk = noise.param_mode == _lib.PARAM_PER_EDGE1
            e0, e1 = _bwd_w_raw(graph.csr, x, gg, D, src_scale, spec=spec_c, reduce_k=rk, both=True,
                                seg_len=ctx.seg_len)
            if q is not None:
                ones = torch.ones(D, dtype=torch.float32, device=q.device)
                m0, m1 = _bwd_w_raw(graph.csr, ones, q, D, None, broadcast_x=True, spec=spec_c, reduce_k=rk,
                                    both=True, seg_len=ctx.seg_len)
                e0, e1 = e0 - m0, e1 - m1
            rows = {1: e0, 2: e1}
        for which, need in ((1, ctx.needs_input_grad[1]), (2, ctx.needs_input_grad[2])):
            if not need:
                continue
            d = rows[which]
            shape = ctx.shapes[which - 1]
            d = d.sum_to_size(shape) if d.dim() >= len(shape) and shape != d.shape else d.reshape(shape)
            if which == 1:
                dp0 = d
            else:
                dp1 = d
        return dx, dp0, dp1, None, None, None, None, None, None


# ---- constant inputs of a width that is not a multiple of 4 (PPI's 50 input features: BASELINE configs[2], layer 1) ------
# With D % 4 != 0 the kernels take their scalar row forms (4 bounds-checked stores per lane, more registers: 83 against 79
# VGPRs at 16 lanes per row = 5 against 6 waves per SIMD).  A tensor that is a CONSTANT of the run — it carries no gradient
# and the same object comes back unchanged, as dataset features do on every epoch — is zero-padded to the next multiple
# of 4 ONCE (on its second sighting: a fresh tensor per step, e.g. a freshly batched minibatch's features, would pay the
# copy the vector forms save — measured in round 1) and the launch runs at the padded width: the first D channels of the
# result are the same bits (a Philox block covers 4 channels either way; the padded channels gather zeros), returned as a
# view [:, :D] of the padded result, which the dense transform behind it reads through its row stride (node_linear).
PAD_CONSTANT_INPUTS = True
PAD_MIN_WIDTH = 32        # narrower rows (molhiv's 9 atom features) are launch-bound: the slice of the padded result and the
                          # second descriptor cost the host more than the vector forms save the kernel (D = 9: 13 -> 24 us)
_const_pads = {}          # id(tensor) -> [weakref, _version, data_ptr, sightings, padded | None]


def _padded_constant(x):
    """The cached zero-padded copy [N, ceil4(D)] of a constant x [N, D], or None (first sighting, or it changed)."""
    import weakref
    key = id(x)
    ent = _const_pads.get(key)
    if ent is None or ent[0]() is not x or ent[1] != x._version or ent[2] != x.data_ptr():
        if len(_const_pads) >= 64:
            _const_pads.clear()
        _const_pads[key] = [weakref.ref(x, lambda _r, key=key: _const_pads.pop(key, None)), x._version, x.data_ptr(), 1, None]
        return None
    ent[3] += 1
    if ent[4] is None:
        D = x.shape[1]
        Dp = (D + 3) // 4 * 4
        xp = torch.zeros((x.shape[0], Dp), dtype=torch.float32, device=x.device)
        xp[:, :D].copy_(x)
        ent[4] = xp
    return ent[4]


_row_pads = {}            # id(row) -> (weakref, _version, {Dp: padded row})


def _padded_noise(noise, Dp):
    """The descriptor for a launch at the padded width: scalar parameters as they are; per-channel rows [D] extended to
    [Dp] with their last entry (the padded channels' draws are multiplied by zeros), kept per row tensor — the rows a
    layer hands over come from noise._expanded's own cache, so this costs nothing from the second call on.  None when
    the descriptor has per-edge parameters (the plain path then)."""
    if noise is None or noise.param_mode == _lib.PARAM_SCALAR:
        return noise
    if noise.param_mode != _lib.PARAM_PER_CHANNEL:
        return None
    import copy
    import weakref
    out = copy.copy(noise)
    out.dn = Dp
    for name in ("p0", "p1"):
        row = getattr(noise, name)
        if row is None:
            continue
        ent = _row_pads.get(id(row))
        if ent is None or ent[0]() is not row or ent[1] != row._version:
            if len(_row_pads) > 256:
                _row_pads.clear()
            ent = _row_pads[id(row)] = (weakref.ref(row, lambda _r, k=id(row): _row_pads.pop(k, None)), row._version, {})
        pad = ent[2].get(Dp)
        if pad is None:
            pad = ent[2][Dp] = torch.cat([row, row[-1:].expand(Dp - row.shape[0])]).contiguous()
        setattr(out, name, pad)
    return out


def aggregate_into(csrv, x, out, weight, reduce, src_scale, dst_scale, plan_t):
    """The rows of ONE sub-plan of csrv (CsrView.subplan) written into `out` [n_dst, D]; no autograd.
    How a node-range shard launches its local-source rows while the halo exchange is in flight and
    the other rows behind it (partition.GraphShard.aggregate); every row is reduced exactly as
    a whole-plan launch would reduce it."""
    x = _f32c(x)
    if weight is None:
        spec = _none_spec()
    elif isinstance(weight, EdgeNoise):
        if weight.dn != x.shape[1]:
            raise ValueError(f"noise width {weight.dn} != feature width {x.shape[1]}")
        spec = _noise_spec(weight)
    else:
        raise TypeError("aggregate_into takes None or an EdgeNoise")
    _agg_raw(csrv, x, x.shape[1], spec, _REDUCE[reduce], _f32c(src_scale), _f32c(dst_scale),
             plan_t["seg_len"], out=out, plan_t=plan_t)
    return out


def aggregate(graph, x, weight=None, reduce="sum", src_scale=None, dst_scale=None,
              seg_len=DEFAULT_SEG_LEN, _broadcast_x=False, _gathered=False):
    """out[v,:] = dscale[v] * sum|mean_{e=(u->v)} w[e,:] * sscale[u] * x[u,:]

    weight: None (plain copy_u), a tensor [E, D] indexed by edge id (explicit
    `edge_weight`, stag/zoo/gcn.py:60-63), or an EdgeNoise (fused sampling).
    On a node-range shard (partition.GraphShard) x holds this rank's rows and the call includes
    the halo exchange."""
    if x.dim() != 2:
        raise ValueError("aggregate expects x of shape [N, D]")
    if getattr(graph, "is_shard", False) and not _gathered and not _broadcast_x:
        # (a source-side scale indexes COLUMNS: on a shard, the rows of the exchanged buffer — shard.out_degrees())
        return graph.aggregate(x, weight, reduce=reduce, src_scale_buf=src_scale,
                               dst_scale_local=dst_scale, seg_len=seg_len)
    noise = weight if isinstance(weight, EdgeNoise) else None
    w = weight if torch.is_tensor(weight) else None
    D = x.shape[1]
    if w is not None:
        if w.shape[0] != graph.number_of_edges():
            raise AssertionError("edge_weight.shape[0] != number_of_edges")   # zoo/gcn.py:61
        if w.dim() == 1:
            w = w.unsqueeze(1)
        if w.shape[1] != D:
            w = w.expand(w.shape[0], D)
    if noise is not None and noise.dn != D:
        raise ValueError(f"noise width {noise.dn} != feature width {D}")
    if noise is not None and noise.n_samples > 1:
        if _broadcast_x:
            raise ValueError("Monte-Carlo batching does not apply to a broadcast row")
        import copy
        one = copy.copy(noise)
        one.n_samples = 1
        return aggregate_mc(graph, x, one, noise.n_samples, noise.offset_stride, reduce=reduce,
                            src_scale=src_scale, dst_scale=dst_scale, seg_len=seg_len, _gathered=_gathered)
    if noise is not None and noise.grad_params is not None and torch.is_grad_enabled():
        p0, p1 = (torch.as_tensor(p, dtype=torch.float32, device=x.device) for p in noise.grad_params)
        if p0.requires_grad or p1.requires_grad:
            return _AggregateVI.apply(x, p0, p1, graph, noise, _REDUCE[reduce], _f32c(src_scale),
                                      _f32c(dst_scale), seg_len)
    if not (torch.is_grad_enabled() and (x.requires_grad or (w is not None and w.requires_grad))):
        # nothing to differentiate: straight to the library (no autograd node; host time of a call matters on
        # launch-bound graphs)
        xin, x = x, _f32c(x)
        if (PAD_CONSTANT_INPUTS and D % 4 and D >= PAD_MIN_WIDTH and w is None and not _broadcast_x and x is xin and x.is_cuda
                and not x.requires_grad and (noise is None or noise.param_mode <= _lib.PARAM_PER_CHANNEL)):
            xp = _padded_constant(x)
            noise_p = _padded_noise(noise, (D + 3) // 4 * 4) if xp is not None else None
            if xp is not None and (noise is None or noise_p is not None):
                spec_p = _noise_spec(noise_p) if noise_p is not None else _none_spec()
                return _agg_raw(graph.csr, xp, xp.shape[1], spec_p, _REDUCE[reduce], _f32c(src_scale), _f32c(dst_scale),
                                seg_len)[0][:, :D]
        if noise is not None:
            spec = _noise_spec(noise)
        elif w is not None:
            spec = _explicit_spec(_f32c(w))
        else:
            spec = _none_spec()
        return _agg_raw(graph.csr, x, D, spec, _REDUCE[reduce], _f32c(src_scale), _f32c(dst_scale), seg_len,
                        broadcast_x=_broadcast_x)[0]
    return _Aggregate.apply(x, w, graph, noise, _REDUCE[reduce], _f32c(src_scale),
                            _f32c(dst_scale), seg_len, _broadcast_x)


def aggregate_max(graph, x, weight=None):
    """out[v,:] = max_{e=(u->v)} w[e,:] * x[u,:], 0 for rows without in-edges (DGL's `fn.max`;
    GraphSAGE 'pool', stag/zoo/graph_sage.py:90-93).  Not a fused path: the messages are formed
    ([E, D]: a kernel-backed gather, the noise materialised) and reduced with scatter-amax."""
    if x.dim() != 2:
        raise ValueError("aggregate_max expects x of shape [N, D]")
    m = gather_rows(graph, x, "src")
    if isinstance(weight, EdgeNoise):
        weight = weight.materialize()
    if weight is not None:
        if weight.shape[0] != graph.number_of_edges():
            raise AssertionError("edge_weight.shape[0] != number_of_edges")
        m = m * (weight if weight.dim() == 2 else weight.unsqueeze(1))
    _, dst = graph.edges()
    n = graph.number_of_dst_nodes() if hasattr(graph, "number_of_dst_nodes") else graph.number_of_nodes()
    out = torch.zeros((n, x.shape[1]), dtype=m.dtype, device=m.device)
    return out.scatter_reduce(0, dst.long().unsqueeze(1).expand(-1, x.shape[1]), m, reduce="amax", include_self=False)


def aggregate_mc(graph, x, noise, n_samples, offset_stride=1, reduce="sum", src_scale=None,
                 dst_scale=None, seg_len=DEFAULT_SEG_LEN, _gathered=False):
    """[n_samples, N, D]: sample s == aggregate(graph, x, noise at offset + s * offset_stride), bit for
    bit, from one pass over the gathered rows per 4 samples (2 with in-norm: every sample carries its own weight
    sums) (stag_agg_fwd_mc).  The reference's Monte-Carlo loops — inference, stag/models.py:45-55, and training,
    stag/models.py:67-68 with `--n_samples_training` > 1 — on a layer whose input is the same for every sample.
    It needs no backward of its own when neither x nor the noise parameters carry a gradient (every `*_mle` script:
    the first layer's input is the data, the noise is fixed): the dense transform that follows differentiates
    through the S outputs.  When something here does need a gradient, or for noise the fused sampler does not
    cover, it is the stack of n_samples ordinary calls."""
    def one(s):
        import copy
        nz = copy.copy(noise)
        nz.offset = noise.offset + s * offset_stride
        return aggregate(graph, x, nz, reduce=reduce, src_scale=src_scale, dst_scale=dst_scale, seg_len=seg_len,
                         _gathered=_gathered)
    if getattr(graph, "is_shard", False) and not _gathered:
        return torch.stack([one(s) for s in range(n_samples)], 0)
    fusable_mc = (isinstance(noise, EdgeNoise) and noise.kind >= _lib.NOISE_NORMAL
                  and noise.param_mode <= _lib.PARAM_PER_CHANNEL)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or (noise is not None and noise.grad_params is not None
                                                                  and any(torch.is_tensor(p) and p.requires_grad
                                                                          for p in noise.grad_params)))
    if not fusable_mc or n_samples == 1:
        return torch.stack([one(s) for s in range(n_samples)], 0)
    if noise.dn != x.shape[1]:
        raise ValueError(f"noise width {noise.dn} != feature width {x.shape[1]}")
    if needs_grad:
        # the batched FORWARD under autograd (x or the noise parameters carry a gradient): the gathers of the S samples
        # are shared, the backward is the S per-sample transposed passes the loop would run (_AggregateMC).  Not with
        # in-norm (its factor has a derivative of its own: ops._AggregateVI) or explicit / per-edge parameters.
        live = noise.grad_params is not None and any(torch.is_tensor(p) and p.requires_grad for p in noise.grad_params)
        if noise.in_norm or (live and noise.kind not in (_lib.NOISE_NORMAL, _lib.NOISE_UNIFORM)):
            return torch.stack([one(s) for s in range(n_samples)], 0)
        if live:
            p0, p1 = (torch.as_tensor(p, dtype=torch.float32, device=x.device) for p in noise.grad_params)
        else:
            p0 = p1 = None
        return _AggregateMC.apply(x, p0, p1, graph, noise, int(n_samples), int(offset_stride), _REDUCE[reduce],
                                  _f32c(src_scale), _f32c(dst_scale), seg_len)
    return _agg_fwd_mc_raw(graph.csr, _f32c(x), noise, n_samples, offset_stride, _REDUCE[reduce], _f32c(src_scale),
                           _f32c(dst_scale), seg_len)


def _agg_fwd_mc_raw(csrv, x, noise, n_samples, offset_stride, reduce, src_scale, dst_scale, seg_len):
    """One stag_agg_fwd_mc call: [n_samples, n_dst, D]."""
    D = x.shape[1]
    dev = _lib.require_device(x, csrv.indptr, src_scale, dst_scale)
    if _torch_ext.available():      # the dispatcher op (csrc/torch_ext.cpp): same library call, visible to a compiled graph
        plan_t = csrv.plan(seg_len)
        return torch.ops.stag.agg_fwd_mc(*csrv.torch_args(), *_plan_args(csrv, plan_t, (D + 255) // 256, dev, width=D, drawn=True),
                                         x, *noise.torch_args(), int(n_samples), int(offset_stride), reduce, src_scale, dst_scale)
    out = torch.empty((n_samples, csrv.n_dst, D), dtype=torch.float32, device=dev)
    plan_t = csrv.plan(seg_len)
    nbytes = _lib.lib().stag_plan_workspace_bytes(plan_t["n_seg"], 4 * D, 0) if plan_t is not None else 0
    plan_c, _keep = _plan_struct(csrv, seg_len, (D + 255) // 256, nbytes, dev, plan_t=plan_t, width=D)
    cs, spec = csrv.struct(), noise.spec()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_agg_fwd_mc(
            C.byref(cs), C.byref(plan_c) if plan_c is not None else None, _lib.ptr(x), x.stride(0), D,
            C.byref(spec), n_samples, offset_stride, reduce, _lib.ptr(src_scale), _lib.ptr(dst_scale),
            _lib.ptr(out), D, csrv.n_dst * D, _lib.stream_of(dev))
    _lib.check(rc, "stag_agg_fwd_mc")
    return out


class _AggregateMC(torch.autograd.Function):
    """S Monte-Carlo samples of one aggregation (stag/models.py:67-68 on a layer whose input is the same for every
    sample) with gradients: the forward is ONE batched pass (stag_agg_fwd_mc: the gathers are shared), the backward
    the per-sample transposed passes of the sequential loop — d x summed over the samples, and for a reparameterised
    draw (`vi=True`) the finished parameter gradients of every sample (stag_agg_bwd_dp), summed.  Sample s draws at
    offset + s * stride, so values and gradients are those of the loop."""

    @staticmethod
    def forward(ctx, x, p0, p1, graph, noise, n_samples, stride, reduce, src_scale, dst_scale, seg_len):
        x = _f32c(x)
        out = _agg_fwd_mc_raw(graph.csr, x, noise, n_samples, stride, reduce, src_scale, dst_scale, seg_len)
        ctx.graph, ctx.noise, ctx.S, ctx.stride, ctx.reduce, ctx.seg_len = _owner(graph), noise, n_samples, stride, reduce, seg_len
        ctx.pshapes = None if p0 is None else (p0.shape, p1.shape)
        ctx.save_for_backward(x, src_scale, dst_scale)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        import copy
        x, src_scale, dst_scale = ctx.saved_tensors
        graph, D = ctx.graph, x.shape[1]
        g_all = _f32c(grad_out)
        dvec = dst_scale
        if ctx.reduce == _lib.REDUCE_MEAN:
            inv = 1.0 / graph.csr.degrees.clamp(min=1).to(torch.float32)
            dvec = inv if dvec is None else dvec * inv
        need_x = ctx.needs_input_grad[0]
        need_p = ctx.pshapes is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dx = d0 = d1 = None
        for s_i in range(ctx.S):
            nz = copy.copy(ctx.noise)
            nz.offset = ctx.noise.offset + s_i * ctx.stride
            spec = _targs_or_c(_noise_spec(nz, in_norm=0))
            g = g_all[s_i]
            if need_p:
                dxs, c0, c1 = _agg_bwd_dp_raw(graph.csr_t, g, x, D, spec, dvec, src_scale, ctx.seg_len, want_dx=need_x)
                d0 = c0 if d0 is None else d0 + c0
                d1 = c1 if d1 is None else d1 + c1
            else:
                dxs, _ = _agg_raw(graph.csr_t, g, D, spec, _lib.REDUCE_SUM, dvec, src_scale, ctx.seg_len)
            if need_x:
                dx = dxs if dx is None else dx + dxs
        dp = [None, None]
        if need_p:
            for i, d in enumerate((d0, d1)):
                if ctx.needs_input_grad[1 + i]:
                    shape = ctx.pshapes[i]
                    dp[i] = d.sum().reshape(shape) if d.numel() != int(torch.Size(shape).numel()) else d.reshape(shape)
        return dx, dp[0], dp[1], None, None, None, None, None, None, None, None


def materialize_noise(graph, noise, seg_len=DEFAULT_SEG_LEN):
    """The [E, Dn] tensor the reference would have sampled (rows by edge id)."""
    csrv = graph.csr
    dev = _lib.require_device(csrv.indptr)
    dn = noise.dn
    w = torch.empty((csrv.n_edges, dn), dtype=torch.float32, device=dev)
    ns = torch.empty((csrv.n_dst, dn), dtype=torch.float32, device=dev) if noise.in_norm else None
    plan_t = csrv.plan(seg_len)
    nbytes = (_lib.lib().stag_plan_workspace_bytes(plan_t["n_seg"], dn, 1)
              if (plan_t is not None and noise.in_norm) else 0)
    plan_c, _keep = _plan_struct(csrv, seg_len, (dn + 255) // 256, nbytes, dev)
    spec, cs = noise.spec(), csrv.struct()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_noise_materialize(C.byref(cs), C.byref(plan_c) if plan_c is not None else None,
                                               C.byref(spec), dn, _lib.ptr(w), dn, _lib.ptr(ns),
                                               _lib.stream_of(dev))
    _lib.check(rc, "stag_noise_materialize")
    return w


def philox_raw(seed, offset, pos0, n_pos, n_chunk, device):
    out = torch.empty((n_pos, n_chunk, 4), dtype=torch.int32, device=device)
    dev = _lib.require_device(out)
    with _lib.on_device(dev):
        rc = _lib.lib().stag_philox_raw(seed, offset, pos0, n_pos, n_chunk, _lib.ptr(out),
                                        _lib.stream_of(dev))
    _lib.check(rc, "stag_philox_raw")
    return out


def normal_tables(device):
    """(rad, cos, sin): the hardware functions of a Normal draw over all 2^23 mantissas, [3, 2^23] fp32
    (stag_normal_tables; test hook — a CPU checker can redraw the device's normals from them)."""
    t = torch.empty((3, 1 << 23), dtype=torch.float32, device=device)
    dev = _lib.require_device(t)
    with _lib.on_device(dev):
        rc = _lib.lib().stag_normal_tables(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.stream_of(dev))
    _lib.check(rc, "stag_normal_tables")
    return t


def _segment_reduce_raw(x, offsets, reduce, dev):
    B, D = offsets.shape[0] - 1, x.shape[1]
    out = torch.empty((B, D), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.lib().stag_segment_reduce(_lib.ptr(x), x.stride(0), D, _lib.ptr(offsets), B,
                                            reduce, _lib.ptr(out), D, _lib.stream_of(dev))
    _lib.check(rc, "stag_segment_reduce")
    return out


_SEG_CHUNK = 256


class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, offsets, reduce):
        x = _f32c(x)
        dev = _lib.require_device(x, offsets)
        B, n = offsets.shape[0] - 1, x.shape[0]
        if B > 0 and n > _SEG_CHUNK * B:
            # long segments (a team walks a segment's rows: 15.8 ms for one 169k-row graph): sum
            # chunks of <= 256 rows first — many teams —, then each segment's chunk sums (0.2 ms).
            # The chunk table is built on the device: no host read-back of the segment lengths.
            R = _SEG_CHUNK
            lens = (offsets[1:] - offsets[:-1]).long()
            cptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
            cptr[1:] = torch.cumsum((lens + R - 1) // R, 0)
            n_max = (n + R - 1) // R + B                      # host-side bound on the chunk count
            c = torch.arange(n_max, dtype=torch.int64, device=dev)
            seg = torch.searchsorted(cptr[1:], c, right=True).clamp(max=B - 1)
            start = offsets[:-1].long()[seg] + (c - cptr[seg]) * R
            start = torch.where(c < cptr[-1], start, torch.full_like(start, n))     # padding chunks: empty
            off1 = torch.cat([start, torch.full((1,), n, dtype=torch.int64, device=dev)]).to(torch.int32)
            # a chunk ends where the next one starts, except the last chunk of a segment, which
            # ends with the segment; segments are contiguous, so "next start" is right in both cases
            part = _segment_reduce_raw(x, off1, _lib.REDUCE_SUM, dev)
            out = _segment_reduce_raw(part, cptr.to(torch.int32), _lib.REDUCE_SUM, dev)
            if reduce == _lib.REDUCE_MEAN:
                out = out / lens.clamp(min=1).to(out.dtype).unsqueeze(1)
        else:
            out = _segment_reduce_raw(x, offsets, reduce, dev)
        ctx.reduce, ctx.n = reduce, n
        ctx.save_for_backward(offsets)
        return out

    @staticmethod
    def backward(ctx, g):
        (offsets,) = ctx.saved_tensors
        sizes = (offsets[1:] - offsets[:-1]).long()
        if ctx.reduce == _lib.REDUCE_MEAN:
            g = g / sizes.clamp(min=1).to(g.dtype).unsqueeze(1)
        return torch.repeat_interleave(g, sizes, dim=0, output_size=ctx.n), None, None


def segment_reduce(x, offsets, reduce="sum"):
    """Per-graph readout over a batched graph (dgl.sum_nodes / mean_nodes)."""
    return _SegmentReduce.apply(x, offsets.to(torch.int32).contiguous(), _REDUCE[reduce])


def _gat_norm_scale(csrv, noise, H, seg_len, dev):
    """in-norm factor [N, H] of H-wide weights (stag/layers.py:8-36): row sums on the aggregation
    kernel (a broadcast row of ones, same noise, in-norm off), then indeg / sum."""
    ones = torch.ones(1, H, dtype=torch.float32, device=dev)
    sums, _ = _agg_raw(csrv, ones, H, _noise_spec(noise, in_norm=0), _lib.REDUCE_SUM, None, None, seg_len,
                       broadcast_x=True)
    deg = csrv.degrees.to(torch.float32).unsqueeze(1)
    return torch.where(sums != 0, deg / sums, torch.ones_like(sums)).contiguous()


def _gat_drop_struct(attn_drop):
    """(p, seed, offset[, epoch tensor]) -> stag_gat_drop, or None"""
    if attn_drop is None:
        return None
    d = _lib.GatDrop()
    d.keep_prob = 1.0 - float(attn_drop[0])
    d.seed, d.offset = int(attn_drop[1]) & ((1 << 64) - 1), int(attn_drop[2]) & ((1 << 64) - 1)
    d.epoch = _lib.ptr(attn_drop[3]) if len(attn_drop) > 3 else None
    return d


def gat_lanes_per_head(F):
    """Lanes a head takes in the cooperative GAT kernels: F / 4 rounded up to a power of two (their head sums are
    butterflies; lanes past a head's channels idle)."""
    l = 1
    while l * 4 < F:
        l *= 2
    return l


def gat_cooperative_shape(H, F, seg_len):
    """The workgroup-cooperative GAT kernels (forward, one-gather backward, in-kernel dropout) take this shape."""
    return (F % 4 == 0 and F >= 4 and H <= 16 and H * F <= 1024 and gat_lanes_per_head(F) <= 64
            and H * gat_lanes_per_head(F) <= 256 and seg_len is not None and 0 < seg_len <= _lib.BLOCK_EDGES)


def attn_drop_fusable(H, F, seg_len, want_attn=False):
    """Attention dropout rides in the workgroup-cooperative GAT kernels (forward and one-gather backward)."""
    return not want_attn and _GAT_BWD_FUSED and _GAT_BWD_ONE_GATHER and gat_cooperative_shape(H, F, seg_len)


class _GatAggregate(torch.autograd.Function):
    """p0, p1 (optional): the live parameter tensors of a reparameterised `noise` (vi=True): the draw stays in the
    kernels, forward and backward, and the backward returns their finished gradients (stag_gat_bwd_dp)."""

    @staticmethod
    def forward(ctx, el, er, ft, w, graph, noise, neg_slope, want_attn, seg_len, attn_drop=None, p0=None, p1=None):
        el, er, ft = _f32c(el), _f32c(er), _f32c(ft)
        ctx.pshapes = None if p0 is None else (p0.shape, p1.shape)
        H, F = ft.shape[1], ft.shape[2]
        csrv = graph.csr
        dev = _lib.require_device(el, er, ft, csrv.indptr)
        if noise is not None:
            spec = noise.spec()
        elif w is not None:
            w = _f32c(w)
            spec = _targs_or_c(_explicit_spec(w))
        else:
            spec = _targs_or_c(_none_spec())
        if attn_drop is not None and noise is None:
            spec.pos_base = int(getattr(graph, "pos_base", 0))    # a shard's dropout mask is the whole graph's
        nscale = _gat_norm_scale(csrv, noise, H, seg_len, dev) if spec.in_norm else None
        need_grad = any(ctx.needs_input_grad[:4]) or any(ctx.needs_input_grad[10:12])
        out = torch.empty((csrv.n_dst, H, F), dtype=torch.float32, device=dev)
        # softmax statistics per row [N, 2H]: the backward and the attention values are computed
        # from them (storing a[E, H] from the forward kernel cost it 250 us at cfg5)
        stats = (torch.empty((csrv.n_dst, 2 * H), dtype=torch.float32, device=dev)
                 if (want_attn or need_grad) else None)
        plan_t = csrv.plan(seg_len, need=True)       # the cooperative kernels want the plan's unit batches
        if _torch_ext.available() and not want_attn and plan_t is not None:
            # the dispatcher op (csrc/torch_ext.cpp: stag::gat_fwd): same library call, visible to a compiled graph
            targs = noise.torch_args() if noise is not None else (_explicit_spec(w) if w is not None else _NONE_ARGS)
            if attn_drop is not None and noise is None:
                targs = (targs[0], [targs[1][0], targs[1][1], int(getattr(graph, "pos_base", 0))]) + tuple(targs[2:])
            out, stats_t = torch.ops.stag.gat_fwd(*csrv.torch_args(), *_gat_plan_args(csrv, plan_t, dev, H * F), el, er, ft,
                                                  float(neg_slope), *targs, nscale, *_gat_drop_args(attn_drop), bool(need_grad))
            if need_grad:
                ctx.graph, ctx.noise, ctx.neg_slope, ctx.seg_len = _owner(graph), noise, float(neg_slope), seg_len
                ctx.attn_drop = attn_drop
                ctx.save_for_backward(el, er, ft, w, stats_t, out, nscale)
            return out
        nbytes = _lib.lib().stag_gat_workspace_bytes(plan_t["n_seg"], H, F) if plan_t is not None else 0
        plan_c, _keep = _plan_struct(csrv, seg_len, 1, nbytes, dev, plan_t=plan_t, gat_width=H * F)
        cs = csrv.struct()
        attn = None
        drop = _gat_drop_struct(attn_drop)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_gat_fwd(C.byref(cs), C.byref(plan_c) if plan_c is not None else None,
                                         _lib.ptr(el), _lib.ptr(er), _lib.ptr(ft), H, F,
                                         float(neg_slope), C.byref(spec), _lib.ptr(nscale),
                                         C.byref(drop) if drop is not None else None,
                                         _lib.ptr(out), _lib.ptr(stats), _lib.stream_of(dev))
            _lib.check(rc, "stag_gat_fwd")
            if want_attn:
                attn = torch.empty((csrv.n_edges, H), dtype=torch.float32, device=dev)
                rc = _lib.lib().stag_gat_attn(C.byref(cs), C.byref(plan_c) if plan_c is not None else None,
                                              _lib.ptr(el), _lib.ptr(er), H, float(neg_slope), C.byref(spec),
                                              _lib.ptr(nscale), _lib.ptr(stats), _lib.ptr(attn),
                                              _lib.stream_of(dev))
                _lib.check(rc, "stag_gat_attn")
        if need_grad:
            ctx.graph, ctx.noise, ctx.neg_slope, ctx.seg_len = _owner(graph), noise, float(neg_slope), seg_len
            ctx.attn_drop = attn_drop
            ctx.save_for_backward(el, er, ft, w, stats, out, nscale)
        if want_attn:
            ctx.mark_non_differentiable(attn)
            return out, attn
        return out

    @staticmethod
    def backward(ctx, grad_out, *unused):
        el, er, ft, w, stats, out, nscale = ctx.saved_tensors
        graph, noise = ctx.graph, ctx.noise
        H, F = ft.shape[1], ft.shape[2]
        HF = H * F
        csrv, csrt = graph.csr, graph.csr_t
        dev = ft.device
        G = _f32c(grad_out)
        if noise is not None:
            spec = noise.spec()
        elif w is not None:
            spec = _targs_or_c(_explicit_spec(w))
        else:
            spec = _targs_or_c(_none_spec())
        want_dw = w is not None and ctx.needs_input_grad[3]
        if ctx.attn_drop is not None and noise is None:
            spec.pos_base = int(getattr(graph, "pos_base", 0))
        want_dp = ctx.pshapes is not None and any(ctx.needs_input_grad[10:12])
        if csrv.n_edges == 0:        # no edge, no gradient
            zp = lambda i: (torch.zeros(ctx.pshapes[i], dtype=torch.float32, device=dev)
                            if (want_dp and ctx.needs_input_grad[10 + i]) else None)
            return (torch.zeros_like(el) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(er) if ctx.needs_input_grad[1] else None,
                    torch.zeros_like(ft) if ctx.needs_input_grad[2] else None,
                    torch.zeros_like(w) if want_dw else None, None, None, None, None, None, None, zp(0), zp(1))
        spec_tensors = ((noise.p0, noise.p1, noise.epoch) if noise is not None else (w, None, None))
        fused = _gat_bwd_fused(csrv, csrt, el, er, ft, stats, G, out, H, F, ctx.neg_slope, spec, nscale,
                               want_dw, ctx.seg_len, dev, ctx.attn_drop, want_dp=want_dp, spec_tensors=spec_tensors)
        if fused is not None:
            d_el, d_er, d_ft, dw = fused[:4]
            dps = [None, None]
            if want_dp:
                for i in range(2):
                    if ctx.needs_input_grad[10 + i]:
                        d, shape = fused[4 + i], ctx.pshapes[i]
                        dps[i] = d.sum().reshape(shape) if len(shape) == 0 or d.numel() != int(torch.Size(shape).numel()) else d.reshape(shape)
            return (d_el if ctx.needs_input_grad[0] else None, d_er if ctx.needs_input_grad[1] else None,
                    d_ft if ctx.needs_input_grad[2] else None, dw, None, None, None, None, None, None, dps[0], dps[1])
        if want_dp:
            raise NotImplementedError("vi=True GAT parameter gradients need the one-gather backward (stag_gat_bwd_dp)")
        if ctx.attn_drop is not None:
            raise NotImplementedError("attention dropout needs the one-gather GAT backward (stag_gat_bwd)")
        de = torch.empty((csrv.n_edges, H), dtype=torch.float32, device=dev)
        dw = torch.empty((csrv.n_edges, H), dtype=torch.float32, device=dev) if want_dw else None
        # a[E, H], a by-product of the edge pass: the weights of the d ft aggregation
        attn = (torch.empty((csrv.n_edges, H), dtype=torch.float32, device=dev)
                if ctx.needs_input_grad[2] else None)
        plan_c, _keep = _plan_struct(csrv, ctx.seg_len, 1, 0, dev)
        cs = csrv.struct()
        with _lib.on_device(dev):
            rc = _lib.lib().stag_gat_bwd_edge(
                C.byref(cs), C.byref(plan_c) if plan_c is not None else None, _lib.ptr(el),
                _lib.ptr(er), _lib.ptr(ft), _lib.ptr(stats), _lib.ptr(G), _lib.ptr(out), H, F,
                ctx.neg_slope, C.byref(spec), _lib.ptr(nscale), _lib.ptr(de), _lib.ptr(dw),
                _lib.ptr(attn), _lib.stream_of(dev))
        if rc == -38:
            raise NotImplementedError("GAT backward needs out_feats % 4 == 0 and out_feats / 4 a power of two")
        _lib.check(rc, "stag_gat_bwd_edge")
        ones = torch.ones(1, H, dtype=torch.float32, device=dev)
        d_el = d_er = d_ft = None
        if ctx.needs_input_grad[1]:      # d er[v,h] = sum over in-edges of de
            d_er, _ = _agg_raw(csrv, ones, H, _explicit_spec(de), _lib.REDUCE_SUM, None, None,
                               ctx.seg_len, broadcast_x=True)
        if ctx.needs_input_grad[0]:      # d el[u,h] = sum over out-edges of de
            d_el, _ = _agg_raw(csrt, ones, H, _explicit_spec(de), _lib.REDUCE_SUM, None, None,
                               ctx.seg_len, broadcast_x=True)
        if ctx.needs_input_grad[2]:      # d ft[u,h,:] = sum over out-edges of a[e,h] * G[v,h,:]
            d_ft, _ = _agg_raw(csrt, G.reshape(-1, HF), HF, _explicit_spec(attn, group=F), _lib.REDUCE_SUM, None, None,
                               ctx.seg_len)
            d_ft = d_ft.reshape(-1, H, F)
        return d_el, d_er, d_ft, dw, None, None, None, None, None, None, None, None


def _gat_plan_args(csrv, plan_t, dev, gat_width, transposed=False):
    """The plan arguments of torch.ops.stag.gat_fwd / gat_bwd: (units, long_rows, long_seg_ptr, block_ptr, xcd, counters,
    plan_ints), the unit batches the XCD-aware ones when the plan has them (transposed: the second plan of gat_bwd has no
    xcd slot)."""
    key = (1, _lib.stream_of(dev))
    counters = plan_t["counters"].get(key)
    if counters is None:
        counters = torch.zeros(max(plan_t["n_long"], 1), dtype=torch.int32, device=dev)
        plan_t["counters"][key] = counters
    units, block_ptr, n_blocks, n_heavy = plan_t["units"], plan_t["block_ptr"], plan_t["n_blocks"], plan_t["n_heavy"]
    local = None
    if plan_t.get("xcd_on"):
        blocks = csrv.gat_blocks(plan_t, gat_width)
        if blocks is not None:
            units, block_ptr, n_blocks, n_heavy = blocks[0], blocks[1], blocks[2], 0
            local = units       # (non-NULL xcd slot: "these batches are XCD-local", as in _plan_struct)
    ints = [plan_t["seg_len"], plan_t["n_units"], plan_t["n_long"], plan_t["n_seg"], n_heavy, n_blocks, 0, 0]
    if transposed:
        return (units, plan_t["long_rows"], plan_t["long_seg_ptr"], block_ptr, counters, ints)
    return (units, plan_t["long_rows"], plan_t["long_seg_ptr"], block_ptr, local, counters, ints)


def _gat_drop_args(attn_drop):
    """(drop_floats, drop_u64, drop_epoch) of torch.ops.stag.gat_*: what _gat_drop_struct puts into a stag_gat_drop."""
    if attn_drop is None:
        return [], [], None
    s64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v
    seed, off = int(attn_drop[1]) & ((1 << 64) - 1), int(attn_drop[2]) & ((1 << 64) - 1)
    return [1.0 - float(attn_drop[0])], [s64(seed), s64(off)], (attn_drop[3] if len(attn_drop) > 3 else None)


def _ctypes_to_targs(s, tensors):
    """A ctypes stag_noise_spec back as the argument tuple of torch.ops.stag.* (tensors: (p0, p1, epoch) it points into)."""
    s64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v
    p0, p1, epoch = tensors
    return ([s.kind, s.param_mode, s.relu, s.in_norm, s.deriv, s.group, s.chunk_base, s.p1_log],
            [s64(s.seed), s64(s.offset), s.pos_base], [s.p0_scalar, s.p1_scalar], p0, p1, epoch)


def _gat_bwd_fused(csrv, csrt, el, er, ft, stats, G, out, H, F, neg_slope, spec, nscale, want_dw, seg_len, dev,
                   attn_drop=None, want_dp=False, spec_tensors=(None, None, None)):
    """stag_gat_bwd: the whole backward on the workgroup-cooperative kernels (one gather of the [H*F] rows; the
    two-gather form stag_gat_bwd_two_pass stays for A/B); None when the shape or the plans are outside what it
    covers (the caller then composes the older kernels)."""
    lph = F // 4
    E = csrv.n_edges
    one = _GAT_BWD_ONE_GATHER or attn_drop is not None or want_dp
    if not (_GAT_BWD_FUSED and E > 0 and gat_cooperative_shape(H, F, seg_len)
            and (one or (lph & (lph - 1)) == 0)):      # the two-pass form wants F / 4 a power of two
        return None
    plan_f, plan_b = csrv.plan(seg_len, need=True), csrt.plan(seg_len, need=True)
    if _torch_ext.available() and not want_dp and (_GAT_BWD_ONE_GATHER or attn_drop is not None):
        # the dispatcher op (csrc/torch_ext.cpp: stag::gat_bwd): same library call, visible to a compiled graph
        if isinstance(spec, tuple):
            targs = spec
        else:       # (this function is handed the ctypes form: back to the argument tuple, the tensors from the caller's)
            targs = _ctypes_to_targs(spec, spec_tensors)
        try:
            d_el, d_er, d_ft, dw = torch.ops.stag.gat_bwd(
                *csrv.torch_args(), *_gat_plan_args(csrv, plan_f, dev, H * F),
                csrt.indptr, csrt.indices, csrt.eid, csrt.nidx, *_gat_plan_args(csrt, plan_b, dev, H * F, transposed=True),
                el, er, ft, stats, G, out, float(neg_slope), *targs, nscale, *_gat_drop_args(attn_drop), bool(want_dw))
        except RuntimeError as exc:
            if "rc=-38" in str(exc):        # STAG_ENOSYS: a shape / plan outside the cooperative kernels
                return None
            raise
        return d_el, d_er, d_ft, (dw if want_dw else None)
    nbytes = _lib.lib().stag_gat_bwd_workspace_bytes(plan_f["n_seg"], plan_b["n_seg"], H, F)
    pf, _k1 = _plan_struct(csrv, seg_len, 1, nbytes, dev, plan_t=plan_f, gat_width=H * F)
    pb, _k2 = _plan_struct(csrt, seg_len, 1, 0, dev, plan_t=plan_b, gat_width=H * F)
    d_el = torch.empty((csrt.n_dst, H), dtype=torch.float32, device=dev)
    d_er = torch.empty((csrv.n_dst, H), dtype=torch.float32, device=dev)
    d_ft = torch.empty((csrt.n_dst, H, F), dtype=torch.float32, device=dev)
    dw = torch.empty((E, H), dtype=torch.float32, device=dev) if want_dw else None
    scratch = torch.empty(_lib.lib().stag_gat_bwd_scratch_bytes(csrv.n_dst, E, H) // 4, dtype=torch.float32, device=dev)
    cs, ct = csrv.struct(), csrt.struct()
    drop = _gat_drop_struct(attn_drop)
    if want_dp:        # the one-gather backward that also finishes the gradients of scalar / per-head noise parameters
        dp0 = torch.empty(H, dtype=torch.float32, device=dev)
        dp1 = torch.empty(H, dtype=torch.float32, device=dev)
        wbytes = _lib.lib().stag_gat_bwd_dp_workspace_bytes(max(pb.n_blocks, pf.n_blocks), H)   # (the structs': XCD batches count more)
        ws = torch.empty(max(wbytes // 4, 1), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            rc = _lib.lib().stag_gat_bwd_dp(C.byref(cs), C.byref(pf), C.byref(ct), C.byref(pb), _lib.ptr(el), _lib.ptr(er),
                                            _lib.ptr(ft), _lib.ptr(stats), _lib.ptr(G), _lib.ptr(out), H, F, neg_slope,
                                            C.byref(spec), _lib.ptr(nscale), C.byref(drop) if drop is not None else None,
                                            _lib.ptr(d_el), _lib.ptr(d_er), _lib.ptr(d_ft), _lib.ptr(dp0), _lib.ptr(dp1),
                                            _lib.ptr(scratch), _lib.ptr(ws), wbytes, _lib.stream_of(dev))
        if rc == -38:
            return None
        _lib.check(rc, "stag_gat_bwd_dp")
        return d_el, d_er, d_ft, None, dp0, dp1
    fn = _lib.lib().stag_gat_bwd if (_GAT_BWD_ONE_GATHER or attn_drop is not None) else _lib.lib().stag_gat_bwd_two_pass
    with _lib.on_device(dev):
        rc = fn(C.byref(cs), C.byref(pf), C.byref(ct), C.byref(pb), _lib.ptr(el), _lib.ptr(er),
                _lib.ptr(ft), _lib.ptr(stats), _lib.ptr(G), _lib.ptr(out), H, F, neg_slope,
                C.byref(spec), _lib.ptr(nscale), C.byref(drop) if drop is not None else None,
                _lib.ptr(d_el), _lib.ptr(d_er), _lib.ptr(d_ft), _lib.ptr(dw), _lib.ptr(scratch), _lib.stream_of(dev))
    if rc == -38:
        return None
    _lib.check(rc, "stag_gat_bwd")
    return d_el, d_er, d_ft, dw


def _gat_fwd_into(csrv, plan_t, el, er, ft, H, F, neg_slope, spec, nscale, drop, out, stats, dev):
    """One stag_gat_fwd launch over the units of `plan_t` (the whole plan of csrv or a sub-plan of it: a shard's
    all-local rows, then the rest — partition._ShardGat) into existing out [n_dst, H, F] / stats [n_dst, 2H]."""
    nbytes = _lib.lib().stag_gat_workspace_bytes(plan_t["n_seg"], H, F)
    plan_c, _keep = _plan_struct(csrv, plan_t["seg_len"], 1, nbytes, dev, plan_t=plan_t, gat_width=H * F)
    cs = csrv.struct()
    with _lib.on_device(dev):
        rc = _lib.lib().stag_gat_fwd(C.byref(cs), C.byref(plan_c), _lib.ptr(el), _lib.ptr(er), _lib.ptr(ft), H, F,
                                     float(neg_slope), C.byref(spec), _lib.ptr(nscale),
                                     C.byref(drop) if drop is not None else None,
                                     _lib.ptr(out), _lib.ptr(stats), _lib.stream_of(dev))
    _lib.check(rc, "stag_gat_fwd")


class _GatBwdStages:
    """stag_gat_bwd one stage at a time (include/stag_hip.h, v19) with the arguments every stage shares held once:
    `rowdot()`, `source(sub-plan of csr_t)` as often as there are sub-plans, `der()`.  d_ft [csr_t.n_dst, H, F] and
    d_el [csr_t.n_dst, H] may be the head of larger allocations (the caller's)."""

    def __init__(self, csrv, csrt, el, er, ft, stats, G, out, H, F, neg_slope, spec, nscale, attn_drop, seg_len,
                 d_el, d_er, d_ft, dev):
        lib = _lib.lib()
        self.csrv, self.csrt, self.dev, self.H, self.F, self.seg_len = csrv, csrt, dev, H, F, seg_len
        self.plan_f, self.plan_b = csrv.plan(seg_len, need=True), csrt.plan(seg_len, need=True)
        nbytes = lib.stag_gat_bwd_workspace_bytes(self.plan_f["n_seg"], self.plan_b["n_seg"], H, F)
        # ONE forward-plan struct (and so one workspace) for every stage: the segment partials of step 2 and step 3
        self.pf, self._k1 = _plan_struct(csrv, seg_len, 1, nbytes, dev, plan_t=self.plan_f)
        self.scratch = torch.empty(lib.stag_gat_bwd_scratch_bytes(csrv.n_dst, csrv.n_edges, H) // 4, dtype=torch.float32,
                                   device=dev)
        self.drop = _gat_drop_struct(attn_drop)
        self.t = (el, er, ft, stats, G, out, nscale, d_el, d_er, d_ft)
        self.neg_slope, self.spec = float(neg_slope), spec
        self._keep = []

    def _call(self, plan_b, stages):
        pb, k = _plan_struct(self.csrt, self.seg_len, 1, 0, self.dev, plan_t=plan_b)
        self._keep.append(k)
        el, er, ft, stats, G, out, nscale, d_el, d_er, d_ft = self.t
        cs, ct = self.csrv.struct(), self.csrt.struct()
        with _lib.on_device(self.dev):
            rc = _lib.lib().stag_gat_bwd_stages(
                C.byref(cs), C.byref(self.pf), C.byref(ct), C.byref(pb), _lib.ptr(el), _lib.ptr(er), _lib.ptr(ft),
                _lib.ptr(stats), _lib.ptr(G), _lib.ptr(out), self.H, self.F, self.neg_slope, C.byref(self.spec),
                _lib.ptr(nscale), C.byref(self.drop) if self.drop is not None else None, _lib.ptr(d_el), _lib.ptr(d_er),
                _lib.ptr(d_ft), _lib.ptr(self.scratch), stages, _lib.stream_of(self.dev))
        _lib.check(rc, "stag_gat_bwd_stages")

    def rowdot(self):
        self._call(self.plan_b, _lib.GAT_BWD_ROWDOT)

    def source(self, plan_b=None):
        self._call(self.plan_b if plan_b is None else plan_b, _lib.GAT_BWD_SOURCE)

    def der(self):
        self._call(self.plan_b, _lib.GAT_BWD_DER)


_GAT_VI_FUSED = True       # vi=True parameter gradients inside the GAT kernels (stag_gat_bwd_dp) | materialised [E, H] weights
_GAT_BWD_FUSED = True      # tools/bench_configs.py --gat-old-bwd flips it for A/B runs
_GAT_BWD_ONE_GATHER = True  # stag_gat_bwd (one gather of [H*F] rows) | stag_gat_bwd_two_pass; --gat-two-pass for A/B


def gat_aggregate(graph, el, er, ft, neg_slope=0.2, weight=None, want_attn=False,
                  seg_len=DEFAULT_SEG_LEN, _gathered=False, attn_fn=None, attn_drop=None):
    """Fused noisy-logit edge softmax + aggregation (stag/zoo/gat.py:114-126).
    el: [N,H], er: [N,H], ft: [N,H,F]; weight: None | [E,H] tensor | EdgeNoise(dn=H).
    attn_drop: (p, seed, offset[, epoch]) — attention dropout (stag/zoo/gat.py:122) inside the kernels, its mask
    from its own Philox stream (attn_drop_fusable() says whether the shape has that form); attn_fn: any function of
    a[E, H] between the softmax and the sum, on the composed path.
    On a node-range shard the inputs are this rank's rows and the call includes the exchange."""
    if getattr(graph, "is_shard", False) and not _gathered:
        if attn_fn is not None:
            raise NotImplementedError("a function of a[E, H] (attn_fn) is not partitioned: pass attn_drop")
        return graph.gat_aggregate(el, er, ft, neg_slope, weight, seg_len=seg_len, want_attn=want_attn,
                                   attn_drop=attn_drop)
    noise = weight if isinstance(weight, EdgeNoise) else None
    w = weight if torch.is_tensor(weight) else None
    if w is not None and w.shape[0] != graph.number_of_edges():
        raise AssertionError("edge_weight.shape[0] != number_of_edges")
    H, F = ft.shape[1], ft.shape[2]
    live = None
    if (noise is not None and noise.grad_params is not None and torch.is_grad_enabled()
            and any(torch.is_tensor(p) and p.requires_grad for p in noise.grad_params)):
        # vi=True (stag/layers.py:123-124 through stag/zoo/gat.py:117-119).  Scalar / per-head parameters without
        # in-norm stay in the kernels: the forward draws from the descriptor, the one-gather backward redoes the draw
        # with its derivatives and returns the finished parameter gradients (stag_gat_bwd_dp) — no [E, H] tensor.
        # Otherwise the H-wide weights are formed from the kernel's standard draw and the live parameters ([E, H]:
        # 37 MB at cfg5), enter as explicit weights, the edge pass returns dw[E, H] and autograd does the affine map.
        # (the preconditions of stag_gat_bwd_dp mirrored here — channel shards' chunk_base, the plan's segment length
        # inside gat_cooperative_shape — so that a shape it refuses takes the materialised route below instead of
        # failing in the middle of a backward pass)
        if (_GAT_VI_FUSED and not noise.in_norm and noise.param_mode <= _lib.PARAM_PER_CHANNEL and ft.is_cuda
                and gat_cooperative_shape(H, F, seg_len) and _GAT_BWD_FUSED and not want_attn
                and int(getattr(noise, "chunk_base", 0) or 0) == 0):
            live = tuple(torch.as_tensor(p, dtype=torch.float32, device=ft.device) for p in noise.grad_params)
        else:
            w, noise = noise.materialize(), None
    lph = F // 4
    # the fused kernels: H*F <= 256 always; up to 1024 channels on the workgroup-cooperative forms (F % 4 == 0,
    # H <= 16, the default plan); the backward wants F/4 a power of two (<= 64)
    wide_ok = gat_cooperative_shape(H, F, seg_len)
    bwd_ok = wide_ok and _GAT_BWD_FUSED and (_GAT_BWD_ONE_GATHER or (lph & (lph - 1)) == 0)
    old_bwd_ok = F % 4 == 0 and lph <= 64 and (lph & (lph - 1)) == 0 and H * F <= 256     # stag_gat_bwd_edge + aggregations
    if (attn_fn is not None or H > 64 or H * F > (1024 if wide_ok else 256)
            or torch.is_grad_enabled() and not (bwd_ok or old_bwd_ok)):
        if getattr(graph, "is_shard", False):
            raise NotImplementedError("the composed GAT path (attention dropout, H > 64 or H*F > 256) is not partitioned")
        if attn_drop is not None:
            raise NotImplementedError("attn_drop rides in the fused kernels only: check attn_drop_fusable()")
        return _gat_composed(graph, el, er, ft, neg_slope, noise, w, want_attn, seg_len, attn_fn)
    if attn_drop is not None and not attn_drop_fusable(H, F, seg_len, want_attn):
        raise NotImplementedError("attn_drop rides in the fused kernels only: check attn_drop_fusable()")
    if live is not None:
        return _GatAggregate.apply(el, er, ft, None, graph, noise, neg_slope, want_attn, seg_len, attn_drop, *live)
    return _GatAggregate.apply(el, er, ft, w, graph, noise, neg_slope, want_attn, seg_len, attn_drop)


def _gat_composed(graph, el, er, ft, neg_slope, noise, w, want_attn, seg_len, attn_fn=None):
    """The same layer outside the fused kernel's shape limits (one wave spans the H*F row: H <= 64,
    H*F <= 256; its backward wants F/4 a power of two): logits and the edge softmax as torch ops over
    [E, H] (segment max by scatter-reduce), the sums and the weighted aggregation on the aggregation
    kernel, which tiles any width.  Device-only like everything else; several passes instead of one."""
    N, H, F = ft.shape
    _, dst = graph.edges()
    dst = dst.long()
    e = torch.nn.functional.leaky_relu(gather_rows(graph, el, "src") + gather_rows(graph, er, "dst"), neg_slope)
    if noise is not None:
        w = noise.materialize()              # [E, H], relu / in-norm applied (stag/zoo/gat.py:117-119)
    if w is not None:
        e = e * (w if w.dim() == 2 else w.reshape(w.shape[0], -1))
    m = torch.full((N, H), float("-inf"), dtype=e.dtype, device=e.device)
    m = m.scatter_reduce(0, dst.unsqueeze(1).expand(-1, H), e.detach(), reduce="amax", include_self=True)
    p = torch.exp(e - m[dst])
    l = aggregate(graph, torch.ones(1, H, device=e.device), p, seg_len=seg_len, _broadcast_x=True)   # sum_in p
    a = p / gather_rows(graph, l, "dst")
    if attn_fn is not None:                  # e.g. attention dropout (stag/zoo/gat.py:122)
        a = attn_fn(a)
    out = aggregate(graph, ft.reshape(N, H * F), a.repeat_interleave(F, dim=1), seg_len=seg_len)
    out = out.reshape(N, H, F)
    return (out, a.detach()) if want_attn else out
