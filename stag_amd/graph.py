"""Graph object that stands where DGL's graph stands in stag.

DGL has no ROCm build, so the drop-in boundary owns this type.  It offers exactly
the surface stag touches: `local_var`, `local_scope`,
`number_of_edges/nodes`, `in_degrees/out_degrees`, `ndata/edata/srcdata/dstdata`,
`update_all`, `apply_edges`, `is_block`, `.to()`; edge frames are indexed by the
ORIGINAL edge id, as in DGL (stag/zoo/gcn.py:61-63).

Device layout (all int32, built once, kept resident in HBM):
    coo      src[E], dst[E]                        original edge order
    csr      indptr[N+1], indices[E], eid[E]       destination-major, stable
    csr_t    indptr[N+1], indices[E], eid[E], nidx[E]
             source-major twin for the backward pass; nidx = forward CSR position
             of the edge, so the backward redraws the forward's noise
    plan     long-row segment lists (stag_plan_*)  per seg_len
"""
import collections
import contextlib
import ctypes as C

import os
import weakref

import numpy as np
import torch

from . import _lib
from . import function as fn

DEFAULT_SEG_LEN = 64


DEVICE_PLANNER = True        # plans of device-resident graphs come from stag_plan_device (False: the host planner)
# Plans carry the XCD-aware unit order (stag_plan.xcd_order): "auto" = once a view has been launched XCD_AFTER_LAUNCHES
# times (a static graph: a freshly batched minibatch graph would pay 0.2-0.4 ms for what saves it 5-30 us a launch) AND
# at least XCD_MIN_LOCALITY of its edges have their source in the same eighth of the rows as their destination (a
# block-diagonal batch: 0.86-1.0; uniformly random sources: 0.125, where the striped order has nothing to offer and
# costs 1 % — tools/xcd_stripe_probe.py); "1" always, with the plan; "0" never.
XCD_ORDER = os.environ.get("STAG_XCD_ORDER", "auto")
XCD_MIN_LOCALITY = 0.25
XCD_FINE = int(os.environ.get("STAG_XCD_FINE", "0"))   # finer row ranges inside an XCD's stripe; 0: CsrView.xcd_fine_for(width)
XCD_RANGE_BYTES = 2_500_000
XCD_AFTER_LAUNCHES = 16
# block-diagonal batches: whole graphs per XCD (xcd_graph_ranges) | equal eighths of the CSR, which cut graphs (A/B)
WHOLE_GRAPHS_PER_XCD = os.environ.get("STAG_XCD_GRAPHS", "1") != "0"
MERGE_HEAVY_WIDE = os.environ.get("STAG_XCD_MERGE_HEAVY", "1") != "0"
# Measured on the PPI batch (tools/xcd_graphs_probe.py, profiles/r04/xcd_graphs_*.txt; us, eighths | graphs | graphs merged):
#   D = 128  no draw 35.2 | 36.2 | 42.0   Normal  62.3 |  64.0 |  74.0      (an eighth of the table fits an L2 either way)
#   D = 256  no draw 77.6 | 74.6 | 71.8   Normal 121.7 | 122.7 | 126.1      (fabric traffic 402 -> 170 MB without a draw)
#   GAT forward H*F = 256: 108.1 | 96.3,  4 x 256: 379 | 354
# so whole graphs per XCD are for rows of more than GRAPHS_ABOVE floats (1 KB and up: a stripe's share of the table no longer
# fits the 4 MB L2), and the one-family order for launches that do not draw (a drawing launch is the VALU's either way).
GRAPHS_ABOVE = int(os.environ.get("STAG_XCD_GRAPHS_ABOVE", "128"))         # floats per row
MERGE_HEAVY_ABOVE = int(os.environ.get("STAG_XCD_MERGE_ABOVE", "128"))     # floats per row
PLAN_AFTER_LAUNCHES = 16     # launches a short-row view runs without a plan before one is built for it


def xcd_graph_ranges(edge_cuts, rows, width, range_bytes=None, stripes=None, fine_max=None):
    """The range table of stag_plan_xcd_ranges for a block-diagonal batch: whole GRAPHS per XCD.

    edge_cuts [G + 1]: CSR position where every graph's rows begin (and the edge count at the end); rows [G]: its row
    count; width: floats per gathered row.  -> (cuts int64 [R + 1], keys int32 [R], fine):
      1. graphs go to the 8 stripes by edge count — longest-processing-time first (the lightest stripe takes the next
         largest graph; 24 PPI graphs: the stripes differ by a few per cent, where 8 contiguous runs differ by 20-30) — or,
         for hundreds of small graphs per stripe (a molecule batch), as 8 contiguous runs;
      2. a stripe's graphs, in batch order, are packed into fine ranges of at most `range_bytes` of gathered rows — what an
         XCD's 4 MB L2 holds beside the streams passing through it — which the XCD walks one after the other; a graph
         larger than that is cut into ranges of its own by edge position;
      3. more than STAG_XCD_FINE_MAX ranges in a stripe: the budget doubles until they fit.
    A graph never straddles two XCDs, and no fine range mixes part of one graph with another."""
    stripes = _lib.XCD_STRIPES if stripes is None else stripes
    fine_max = _lib.XCD_FINE_MAX if fine_max is None else fine_max
    rb = float(XCD_RANGE_BYTES if range_bytes is None else range_bytes)
    edge_cuts = np.asarray(edge_cuts, np.int64)
    rows = np.asarray(rows, np.int64)
    G, E = len(rows), int(edge_cuts[-1])
    e = np.diff(edge_cuts)
    if G >= 64 * stripes:
        stripe = np.minimum(stripes - 1, ((edge_cuts[:-1] + e // 2) * stripes) // max(E, 1)).astype(np.int64)
    else:
        stripe = np.zeros(G, np.int64)
        load = np.zeros(stripes, np.int64)
        for g in np.argsort(-e, kind="stable"):
            k = int(np.argmin(load))
            stripe[g] = k
            load[k] += max(int(e[g]), 1)
    row_bytes = 4 * min(max(int(width), 1), 256)          # (wider rows are tiled at 256 floats)
    # (a budget below a sixteenth of the heaviest stripe cannot fit STAG_XCD_FINE_MAX ranges: start there instead of
    # doubling up to it — a stripe of huge graphs under a tiny budget would otherwise be cut into millions of pieces first)
    per_stripe = np.bincount(stripe, weights=rows.astype(np.float64) * row_bytes * (e > 0), minlength=stripes)
    rb = max(rb, float(per_stripe.max()) / fine_max)
    while True:
        out, fine = [], 1
        for k in range(stripes):
            f, acc = 0, 0.0
            for g in np.nonzero(stripe == k)[0]:
                if e[g] == 0:        # a graph without edges owns no CSR position (its rows' empty units ride with a neighbour)
                    continue
                b = float(rows[g]) * row_bytes
                if b > rb:
                    if acc > 0:
                        f, acc = f + 1, 0.0
                    nsub = int(-(-b // rb))
                    for j in range(nsub):
                        out.append((int(edge_cuts[g] + e[g] * j // nsub), k, f))
                        f += 1
                else:
                    if acc > 0 and acc + b > rb:
                        f, acc = f + 1, 0.0
                    out.append((int(edge_cuts[g]), k, f))
                    acc += b
            fine = max(fine, f + (1 if acc > 0 else 0))
        if fine <= fine_max:
            break
        rb *= 2
    out.sort(key=lambda t: t[0])         # (positions are distinct: every piece holds at least one edge ... except a large
    if not out:                          #  graph cut into more pieces than it has edges: equal positions, any order)
        out = [(0, 0, 0)]
    cuts = np.array([t[0] for t in out] + [E], np.int64)
    cuts[0] = 0
    keys = np.array([t[1] * fine + t[2] for t in out], np.int32)
    return cuts, keys, int(fine)


class CsrView:
    """Tensors of one CSR plus the ctypes struct the library takes."""

    def __init__(self, n_dst, n_src, indptr, indices, eid=None, nidx=None):
        self.n_dst, self.n_src = int(n_dst), int(n_src)
        self._short = None
        self._locality = None
        self._plan_requests = 0
        self.indptr, self.indices, self.eid, self.nidx = indptr, indices, eid, nidx
        self.n_edges = int(indices.shape[0])
        self._plans = {}
        self._degrees = None
        self._struct = None
        self.xcd_graphs = self.xcd_merge = None     # per-view overrides of WHOLE_GRAPHS_PER_XCD / MERGE_HEAVY_WIDE (A/B tools)
        self.part_sizes = None       # rows of every graph of a block-diagonal batch (tensor / array; Graph sets it): the
        self._part_cuts = None       # XCD-aware order then keeps whole graphs per XCD (xcd_ranges)

    def struct(self):
        """The ctypes stag_csr (built once: the tensors it points into live as long as this view)."""
        if self._struct is None:
            self._struct = _lib.Csr(self.n_dst, self.n_src, self.n_edges, _lib.ptr(self.indptr),
                                    _lib.ptr(self.indices), _lib.ptr(self.eid), _lib.ptr(self.nidx))
        return self._struct

    def torch_args(self):
        """(indptr, indices, eid, nidx, n_src): the graph arguments of torch.ops.stag.*"""
        return (self.indptr, self.indices, self.eid, self.nidx, self.n_src)

    @property
    def degrees(self):
        if self._degrees is None:
            self._degrees = (self.indptr[1:] - self.indptr[:-1])
        return self._degrees

    def stripe_locality(self):
        """Fraction of the edges whose source row lies in the same stripe as their destination row — the stripes being the
        STAG_XCD_STRIPES contiguous row ranges that hold an eighth of the edges each (what stag_plan_xcd cuts the units
        into).  What one XCD's L2 can hope to find again when it walks one stripe.  One device reduction, kept."""
        if self._locality is None:
            E = self.n_edges
            if E == 0 or self.n_src != self.n_dst:
                self._locality = 0.0
            elif self.indptr.is_cuda:
                # (the library's own kernel: a chain of torch ops costs 0.3 s of lazy kernel loading on first use)
                dev, same = self.indptr.device, C.c_int64(0)
                ws = torch.empty(8, dtype=torch.uint8, device=dev)
                with _lib.on_device(dev):
                    _lib.check(_lib.lib().stag_stripe_locality(_lib.ptr(self.indptr), _lib.ptr(self.indices), self.n_dst, E,
                                                               C.byref(same), _lib.ptr(ws), _lib.stream_of(dev)),
                               "stag_stripe_locality")
                self._locality = same.value / E
            else:
                pos = torch.arange(E, device=self.indptr.device, dtype=torch.int64)
                src_pos = self.indptr[:-1][self.indices.long()].long()      # where the source's own row starts
                same = (pos * _lib.XCD_STRIPES // E) == torch.clamp(src_pos * _lib.XCD_STRIPES // E, max=_lib.XCD_STRIPES - 1)
                self._locality = float(same.float().mean())
        return self._locality

    def gat_blocks(self, plan, width):
        """(units, block_ptr, n_blocks, fine) for the cooperative GAT kernels on rows of `width` floats when `plan` has the
        XCD-aware order switched on (stag_plan_blocks_xcd: batch b belongs to stripe b mod 8), else None — also while a
        hipGraph is being captured and the arrays for this width do not exist yet (they are built on the host)."""
        if not plan.get("xcd_on"):
            return None
        ranges = self.xcd_ranges(width, build=not (plan["units"].is_cuda and torch.cuda.is_current_stream_capturing()))
        fine = self.xcd_fine_for(width, cap=None) if ranges is None else ranges[2]
        okey = fine if ranges is None else ("graphs", min(int(width), 256))
        got = plan.setdefault("gat_orders", {}).get(okey)
        if got is None:
            if plan["units"].is_cuda and torch.cuda.is_current_stream_capturing():
                return None
            lib, nu, dev = _lib.lib(), plan["n_units"], plan["units"].device
            units_h = np.ascontiguousarray(plan["units"][:nu].cpu().numpy(), dtype=np.int32)
            nb = C.c_int32(0)
            if ranges is None:
                fn_, args = lib.stag_plan_blocks_xcd, (units_h.ctypes.data, nu, self.n_edges, fine, _lib.BLOCK_EDGES, _lib.BLOCK_UNITS)
            else:       # whole graphs per XCD (stag_plan_blocks_xcd_ranges)
                cuts, keys = ranges[0], ranges[1]
                fn_, args = lib.stag_plan_blocks_xcd_ranges, (units_h.ctypes.data, nu, cuts.ctypes.data, keys.ctypes.data,
                                                              len(keys), fine, _lib.BLOCK_EDGES, _lib.BLOCK_UNITS)
            _lib.check(fn_(*args, None, None, C.byref(nb)), "stag_plan_blocks_xcd")
            out = np.zeros((max(nu, 1), 4), np.int32)
            ptr = np.zeros(nb.value + 1, np.int32)
            _lib.check(fn_(*args, out.ctypes.data, ptr.ctypes.data, C.byref(nb)), "stag_plan_blocks_xcd")
            # (the struct cache of ops._plan_struct is keyed by the last element: one value per order)
            tag = fine if ranges is None else 1000 + min(int(width), 256)
            got = plan["gat_orders"][okey] = (torch.from_numpy(out).to(dev), torch.from_numpy(ptr).to(dev), nb.value, tag)
        return got

    def xcd_ranges(self, width, build=True):
        """(cuts, keys, fine, device cuts, device keys) of xcd_graph_ranges for launches that gather rows of `width`
        floats, or None when this view is not a batch of several graphs (then the stripes are equal eighths of the CSR).
        One read-back of G + 1 row pointers, kept; the table is kept per row-byte class."""
        if self.part_sizes is None or not (WHOLE_GRAPHS_PER_XCD if self.xcd_graphs is None else self.xcd_graphs):
            return None
        if min(max(int(width), 1), 256) <= GRAPHS_ABOVE and self.xcd_graphs is None:
            return None
        if self._part_cuts is None:
            if not build:          # (a hipGraph is being captured: no read-back, no upload — what exists is used)
                return None
            sizes = (self.part_sizes.detach().cpu().numpy() if torch.is_tensor(self.part_sizes) else
                     np.asarray(self.part_sizes)).astype(np.int64)
            if len(sizes) < 2 or int(sizes.sum()) != self.n_dst or self.n_edges == 0:
                self.part_sizes = None
                return None
            node_off = np.concatenate([[0], np.cumsum(sizes)])
            at = torch.from_numpy(node_off).to(self.indptr.device)
            self._part_cuts = (sizes, self.indptr[at].cpu().numpy().astype(np.int64), {})
        sizes, edge_cuts, tables = self._part_cuts
        cls = min(max(int(width), 1), 256)
        got = tables.get(cls)
        if got is None:
            if not build:
                return None
            cuts, keys, fine = xcd_graph_ranges(edge_cuts, sizes, cls)
            dev = self.indptr.device
            got = tables[cls] = (cuts, keys, fine, torch.from_numpy(cuts).to(dev), torch.from_numpy(keys).to(dev))
        return got

    def xcd_fine_for(self, width, cap=256):
        """Finer row ranges inside an XCD's stripe for launches that gather rows of `width` floats: the rows one range
        gathers should fit the XCD's 4 MB L2 beside the streams that pass through it — 2.5 MB measured best on the PPI
        batch (D = 50 | 128 | 256: 1 | 2 | 3 ranges; tools/xcd_stripe_probe.py).  STAG_XCD_FINE overrides."""
        if XCD_FINE > 0:
            return XCD_FINE
        rows = -(-self.n_dst // _lib.XCD_STRIPES)
        nbytes = rows * 4 * (min(max(int(width), 1), cap) if cap else max(int(width), 1))   # (wider rows are tiled at 256)
        return int(min(_lib.XCD_FINE_MAX, max(1, -(-nbytes // XCD_RANGE_BYTES))))

    def _build_xcd_order(self, plan, fine, ranges=None, merge_heavy=False):
        """stag_plan.xcd_order for `plan` (stag_plan_xcd on host records, stag_plan_xcd_device_* on device records: the same
        ints): the units grouped by the eighth of the CSR their rows lie in — inside it by `fine` finer ranges — for
        workgroup b to take stripe b mod 8.  ranges (xcd_ranges): the stripes and fine ranges are whole graphs of a
        block-diagonal batch instead (stag_plan_xcd_ranges); merge_heavy: no separate heavy stripes (the wide shapes have
        no slotted loop for them) — an XCD then passes over each fine range once instead of twice."""
        nu, nh = plan["n_units"], (0 if merge_heavy else plan["n_heavy"])
        lib, units, dev = _lib.lib(), plan["units"], plan["units"].device
        st = (C.c_int32 * 2)()
        if units.is_cuda:
            nbytes = lib.stag_plan_xcd_device_workspace_bytes(nu)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            with _lib.on_device(dev):
                if ranges is None:
                    _lib.check(lib.stag_plan_xcd_device_count(_lib.ptr(units), nu, nh, self.n_edges, fine, st, _lib.ptr(ws),
                                                              nbytes, _lib.stream_of(dev)), "stag_plan_xcd_device_count")
                else:
                    _lib.check(lib.stag_plan_xcd_device_count_ranges(_lib.ptr(units), nu, nh, _lib.ptr(ranges[3]),
                                                                     _lib.ptr(ranges[4]), len(ranges[1]), fine, st, _lib.ptr(ws),
                                                                     nbytes, _lib.stream_of(dev)), "stag_plan_xcd_device_count_ranges")
                order = torch.empty(lib.stag_plan_xcd_ints(st[0], st[1]), dtype=torch.int32, device=dev)
                _lib.check(lib.stag_plan_xcd_device_fill(_lib.ptr(units), nu, st, fine, _lib.ptr(order), _lib.ptr(ws), nbytes,
                                                         _lib.stream_of(dev)), "stag_plan_xcd_device_fill")
        else:
            units_h = np.ascontiguousarray(units[:nu].numpy(), dtype=np.int32)
            if ranges is None:
                call = lambda out: lib.stag_plan_xcd(units_h.ctypes.data, nu, nh, self.n_edges, fine, out, st)
            else:
                call = lambda out: lib.stag_plan_xcd_ranges(units_h.ctypes.data, nu, nh, ranges[0].ctypes.data,
                                                            ranges[1].ctypes.data, len(ranges[1]), fine, out, st)
            _lib.check(call(None), "stag_plan_xcd")
            order_h = np.zeros(lib.stag_plan_xcd_ints(st[0], st[1]), np.int32)
            _lib.check(call(order_h.ctypes.data), "stag_plan_xcd")
            order = torch.from_numpy(order_h)
        return order, (int(st[0]), int(st[1]))

    def xcd_order(self, plan, width, drawn=False):
        """(order, strides, tag) of `plan` for launches of this row width, or (None, (0, 0), 0): the plan has none (see
        XCD_ORDER), or the one for this width is not built yet and a hipGraph is being captured (it reads counts back).
        drawn: the launch draws noise (it then keeps the heavy units' own stripes: see GRAPHS_ABOVE).
        tag: one value per distinct order (ops._plan_struct keys its struct cache by it)."""
        if not plan.get("xcd_on"):
            return None, (0, 0), 0
        capturing = plan["units"].is_cuda and torch.cuda.is_current_stream_capturing()
        ranges = self.xcd_ranges(width, build=not capturing)
        if ranges is None:
            fine, merge = self.xcd_fine_for(width), False
            okey = tag = fine
        else:
            cls = min(max(int(width), 1), 256)
            # one family of stripes where a row takes a whole wave (LPE 64: no slotted loop for heavy units, and the PPI
            # batch at D = 256 without noise reads 71.6 against 74.5 us; at D = 128 the split order wins, 36.0 against 41.6)
            fine = ranges[2]
            merge = ((MERGE_HEAVY_WIDE and not drawn) if self.xcd_merge is None else self.xcd_merge) and cls > MERGE_HEAVY_ABOVE
            okey, tag = ("graphs", cls, merge), 1000 + cls + (512 if merge else 0)
        got = plan["xcd_orders"].get(okey)
        if got is None:
            if capturing:
                return None, (0, 0), 0
            got = plan["xcd_orders"][okey] = self._build_xcd_order(plan, fine, ranges, merge)
        return got[0], got[1], tag

    def _add_xcd_order(self, plan):
        """Switch the XCD-aware order on for `plan`; plan["xcd"] / ["xcd_strides"] show the one for 128-wide rows."""
        plan["xcd_decided"] = True
        if plan["n_units"] <= 0:
            return
        plan["xcd_on"], plan["xcd_orders"] = True, {}
        plan["xcd"], plan["xcd_strides"], _ = self.xcd_order(plan, 128)
        plan.pop("_structs", None)
        plan.pop("_ints", None)

    def _xcd_policy(self, plan):
        """Called on every request for `plan`: see XCD_ORDER."""
        if plan.get("xcd_decided") or XCD_ORDER == "0":
            return
        if XCD_ORDER == "auto":
            plan["uses"] = plan.get("uses", 0) + 1
            if plan["uses"] <= XCD_AFTER_LAUNCHES:
                return
            if self.indptr.is_cuda and torch.cuda.is_current_stream_capturing():
                return                           # (read-backs: not inside a hipGraph capture; the next eager launch decides)
            if self.stripe_locality() < XCD_MIN_LOCALITY:
                plan["xcd_decided"] = True
                return
        self._add_xcd_order(plan)

    def _short_rows(self):
        """Every row has at most HEAVY_LEN edges (one device reduction and a one-number read-back, kept)."""
        if self._short is None:
            self._short = bool(self.indptr.is_cuda and self.n_dst > 0 and self.n_edges > 0 and
                               int((self.indptr[1:] - self.indptr[:-1]).max()) <= _lib.HEAVY_LEN)
        return self._short

    def plan(self, seg_len=DEFAULT_SEG_LEN, need=False):
        """Launch plan (stag_plan_count / stag_plan_fill on the host, uploaded once): units
        sorted by length, long rows cut into segments of seg_len edges.
        None when planning is off (seg_len) — or, unless the caller needs one (need=True: the cooperative GAT
        kernels want unit batches), when no row has more than HEAVY_LEN edges: no segments, no heavy units, the
        plan would only restate the row order, and building it (indptr to the host, four uploads: 0.5 ms) is
        what a freshly batched minibatch graph — molecules, superpixels — would pay on every step
        (Graph + both CSRs + both plans 1.43 ms -> 0.39 ms for the 4096-molecule batch).  A view that is launched
        more than PLAN_AFTER_LAUNCHES times gets its plan after all."""
        if seg_len is None or seg_len <= 0:
            return None
        if seg_len not in self._plans and self.indptr.is_cuda and torch.cuda.is_current_stream_capturing():
            # building a plan reads counts back from the device: not inside a hipGraph capture.  A view that was
            # launched before the capture has its plan already (or runs plan-less, which gives the same bits)
            if need:
                raise RuntimeError("the launch plan of this graph must exist before a hipGraph capture: run one "
                                   "eager step first")
            return None
        if seg_len not in self._plans and not need and self._short_rows():
            # ... but a view that keeps being launched (a static graph) earns its plan: unit records instead of
            # two dependent indptr loads per row are worth 7 us of a 40 us launch on the molecule batch
            self._plan_requests += 1
            if self._plan_requests <= PLAN_AFTER_LAUNCHES:
                return None
        if seg_len not in self._plans and self.indptr.is_cuda and DEVICE_PLANNER:
            self._plans[seg_len] = self._plan_on_device(seg_len)
        if seg_len not in self._plans:
            lib = _lib.lib()
            indptr_h = np.ascontiguousarray(self.indptr.detach().cpu().numpy(), dtype=np.int32)
            nu, nl, ns, nh = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
            _lib.check(lib.stag_plan_count(indptr_h.ctypes.data, self.n_dst, seg_len, C.byref(nu),
                                           C.byref(nl), C.byref(ns), C.byref(nh)), "stag_plan_count")
            nu, nl, ns, nh = nu.value, nl.value, ns.value, nh.value
            units = np.zeros((max(nu, 1), 4), np.int32)
            long_rows = np.zeros(max(nl, 1), np.int32)
            long_seg_ptr = np.zeros(nl + 1, np.int32)
            _lib.check(lib.stag_plan_fill(indptr_h.ctypes.data, self.n_dst, seg_len,
                                          units.ctypes.data, long_rows.ctypes.data,
                                          long_seg_ptr.ctypes.data), "stag_plan_fill")
            dev = self.indptr.device
            self._plans[seg_len] = dict(
                seg_len=seg_len, n_units=nu, n_long=nl, n_seg=ns, n_heavy=nh,
                units=torch.from_numpy(units).to(dev),
                long_rows=torch.from_numpy(long_rows).to(dev),
                long_seg_ptr=torch.from_numpy(long_seg_ptr).to(dev),
                counters={}, xcd=None, xcd_strides=(0, 0), **_block_plan(units, nu, dev))
        plan = self._plans[seg_len]
        self._xcd_policy(plan)
        if need and plan.get("block_ptr") is None:
            # the unit batches of the cooperative GAT kernels: a greedy pass over the unit records on the host
            units_h = plan["units"][:max(plan["n_units"], 1)].cpu().numpy()
            plan.update(_block_plan(units_h, plan["n_units"], self.indptr.device))
            plan.pop("_structs", None)
            plan.pop("_ints", None)
        return plan

    def _plan_on_device(self, seg_len):
        """stag_plan_device: the plan from the device indptr, no host pass (block batches are added on demand)."""
        lib, dev = _lib.lib(), self.indptr.device
        n, E = self.n_dst, self.n_edges
        ucap, lcap = n + E // seg_len + 1, E // (seg_len + 1) + 1
        units = torch.empty((max(ucap, 1), 4), dtype=torch.int32, device=dev)
        long_rows = torch.empty(max(lcap, 1), dtype=torch.int32, device=dev)
        long_seg_ptr = torch.zeros(lcap + 1, dtype=torch.int32, device=dev)
        nbytes = lib.stag_plan_device_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        counts = (C.c_int32 * 4)()
        with _lib.on_device(dev):
            rc = lib.stag_plan_device(_lib.ptr(self.indptr), n, E, seg_len, _lib.ptr(units), ucap, _lib.ptr(long_rows),
                                      _lib.ptr(long_seg_ptr), lcap + 1, counts, _lib.ptr(ws), nbytes, _lib.stream_of(dev))
        _lib.check(rc, "stag_plan_device")
        nu, nl, ns, nh = (int(v) for v in counts)
        return dict(seg_len=seg_len, n_units=nu, n_long=nl, n_seg=ns, n_heavy=nh, units=units, long_rows=long_rows,
                    long_seg_ptr=long_seg_ptr, counters={}, n_blocks=0, block_ptr=None, xcd=None, xcd_strides=(0, 0))

    def subplan(self, seg_len, keep):
        """The plan of `seg_len` restricted to the units keep[i] is True for (a host bool array over
        the plan's units; the segments of a long row must be kept or dropped together).  Order is the
        plan's, so segments still come first and the heavy units stay a prefix."""
        full = self.plan(seg_len, need=True)
        units = full["units"].cpu().numpy()[:full["n_units"]]
        keep = np.asarray(keep, bool)
        sel = units[keep]
        n_seg = int((sel[:, 3] >= 0).sum()) if len(sel) else 0
        if n_seg not in (0, full["n_seg"]):
            raise ValueError("a sub-plan takes all segments of the long rows or none")
        dev = self.indptr.device
        buf = np.zeros((max(len(sel), 1), 4), np.int32)
        buf[:len(sel)] = sel
        n_heavy = n_seg + int((sel[n_seg:, 2] > _lib.HEAVY_LEN).sum()) if len(sel) else 0
        sub = dict(seg_len=seg_len, n_units=int(len(sel)), n_long=full["n_long"] if n_seg else 0, n_seg=n_seg,
                   n_heavy=n_heavy, units=torch.from_numpy(buf).to(dev), long_rows=full["long_rows"],
                   long_seg_ptr=full["long_seg_ptr"], counters={}, xcd=None, xcd_strides=(0, 0),
                   **_block_plan(buf, int(len(sel)), dev))
        if XCD_ORDER == "1":
            self._add_xcd_order(sub)
        return sub


def _block_plan(units_host, n_units, dev):
    """block_ptr of a unit list (stag_plan_blocks): the workgroup batches of the cooperative GAT kernels."""
    lib = _lib.lib()
    units_host = np.ascontiguousarray(units_host, dtype=np.int32)
    nb = C.c_int32(0)
    args = (units_host.ctypes.data, n_units, _lib.BLOCK_EDGES, _lib.BLOCK_UNITS)
    _lib.check(lib.stag_plan_blocks(*args, None, C.byref(nb)), "stag_plan_blocks")
    ptr = np.zeros(nb.value + 1, np.int32)
    _lib.check(lib.stag_plan_blocks(*args, ptr.ctypes.data, C.byref(nb)), "stag_plan_blocks")
    return dict(n_blocks=nb.value, block_ptr=torch.from_numpy(ptr).to(dev))


def build_csr(src, dst, n_src, n_dst):
    """Stable destination-major CSR from COO (position order inside a row = ascending
    original edge id).  On a HIP device: stag_csr_build (rocPRIM radix sort + scan in the
    library); on the CPU (host-logic tests, partition setup): torch sort/bincount."""
    E = int(src.shape[0])
    dev = src.device
    if src.is_cuda and E > 0:
        lib = _lib.lib()
        src32, dst32 = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
        indptr = torch.empty(n_dst + 1, dtype=torch.int32, device=dev)
        indices = torch.empty(E, dtype=torch.int32, device=dev)
        eid = torch.empty(E, dtype=torch.int32, device=dev)
        nbytes = lib.stag_csr_build_workspace_bytes(n_dst, E)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = lib.stag_csr_build(_lib.ptr(src32), _lib.ptr(dst32), n_src, n_dst, E, _lib.ptr(indptr),
                                    _lib.ptr(indices), _lib.ptr(eid), None, _lib.ptr(ws), nbytes,
                                    _lib.stream_of(dev))
        _lib.check(rc, "stag_csr_build")
        return indptr, indices, eid
    if E == 0:
        z = torch.zeros(0, dtype=torch.int32, device=dev)
        return torch.zeros(n_dst + 1, dtype=torch.int32, device=dev), z, z.clone()
    order = torch.sort(dst.long(), stable=True).indices
    counts = torch.bincount(dst.long(), minlength=n_dst)
    indptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(counts, 0)
    return indptr.to(torch.int32), src[order].to(torch.int32).contiguous(), order.to(torch.int32)


def _edge_rows(g, x, which):
    """x[src] | x[dst] per edge; on the device through ops.gather_rows, whose backward is the
    aggregation kernel instead of torch's sort-based index backward (4-7 ms at cfg2 size)."""
    if torch.is_tensor(x) and x.is_cuda and x.dim() >= 2 and x.is_floating_point():
        from . import ops
        return ops.gather_rows(g, x.reshape(x.shape[0], -1), which).reshape((-1,) + tuple(x.shape[1:]))
    idx = (g._src if which == "src" else g._dst).long()
    return x[idx]


class _EdgeBatch:
    """What a Python `apply_edges` callable receives (stag/distributions.py:225-227)."""

    def __init__(self, g):
        self.src = {k: _edge_rows(g, v, "src") for k, v in g.srcdata.items()}
        self.dst = {k: _edge_rows(g, v, "dst") for k, v in g.dstdata.items()}
        self.data = g.edata


class Graph:
    is_block = False

    def __init__(self, src, dst, num_nodes=None, batch_num_nodes=None, device=None, _trusted=False):
        src = torch.as_tensor(src)
        dst = torch.as_tensor(dst)
        if device is not None:
            src, dst = src.to(device), dst.to(device)
        if src.shape != dst.shape or src.dim() != 1:
            raise ValueError("src and dst must be 1-D tensors of equal length")
        if num_nodes is None:
            num_nodes = int(torch.maximum(src.max(), dst.max()).item()) + 1 if src.numel() else 0      # (one read-back)
        elif not _trusted and src.numel() and int(torch.maximum(src.max(), dst.max()).item()) >= num_nodes:
            raise ValueError("node id out of range")      # (_trusted: a union of checked graphs — two read-backs saved)
        if src.numel() >= 2 ** 31:
            raise ValueError("more than 2^31-1 edges: shard the graph (stag_amd.partition)")
        self._src = src.to(torch.int32).contiguous()
        self._dst = dst.to(torch.int32).contiguous()
        self._n = int(num_nodes)
        self._csr = None
        self._csr_t = None
        self._in_deg = None
        self._out_deg = None
        self._batch_num_nodes = batch_num_nodes
        self.ndata, self.edata = {}, {}

    # ---- frames -----------------------------------------------------------------
    srcdata = property(lambda self: self.ndata)
    dstdata = property(lambda self: self.ndata)

    def _share_structure(self):
        g = Graph.__new__(Graph)
        g.__dict__.update(self.__dict__)
        return g

    def local_var(self):
        """Same structure (CSR, plans stay shared and cached), private frames."""
        g = self._share_structure()
        g.ndata, g.edata = dict(self.ndata), dict(self.edata)
        g._origin = getattr(self, "_origin", self)
        return g

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata.clear(); self.ndata.update(nd)
            self.edata.clear(); self.edata.update(ed)

    def _cache_owner(self):
        return getattr(self, "_origin", self)

    def _array_ptrs(self):
        """Device addresses of (src, dst, csr: indptr, indices, eid, csr_t: indptr, indices, eid, nidx) — what batch()
        lays end to end; kept, so that a batch of thousands of graphs does not make thousands of calls to ask for them."""
        a, b = self.csr, self.csr_t
        got = self.__dict__.get("_ptrs")
        # (kept WITH the views they were read from: a view that has been rebuilt or replaced is noticed, and a view that is
        # still referenced here cannot have been freed — its arrays are write-once)
        if got is None or got[0] is not a or got[1] is not b:
            got = self._ptrs = (a, b, np.array([t.data_ptr() for t in (self._src, self._dst, a.indptr, a.indices, a.eid,
                                                                       b.indptr, b.indices, b.eid, b.nidx)], np.int64))
        return got[2]

    # ---- structure --------------------------------------------------------------
    @property
    def device(self):
        return self._src.device

    def number_of_edges(self): return int(self._src.shape[0])
    def number_of_nodes(self): return self._n
    def number_of_src_nodes(self): return self._n
    def number_of_dst_nodes(self): return self._n
    num_edges, num_nodes = number_of_edges, number_of_nodes
    num_src_nodes, num_dst_nodes = number_of_src_nodes, number_of_dst_nodes

    def edges(self):
        return self._src.long(), self._dst.long()

    def in_degrees(self):
        o = self._cache_owner()
        if o._in_deg is None:
            o._in_deg = torch.bincount(self._dst.long(), minlength=self._n)
        return o._in_deg

    def out_degrees(self):
        o = self._cache_owner()
        if o._out_deg is None:
            o._out_deg = torch.bincount(self._src.long(), minlength=self._n)
        return o._out_deg

    @property
    def batch_size(self):
        return 1 if self._batch_num_nodes is None else int(self._batch_num_nodes.shape[0])

    def batch_num_nodes(self):
        if self._batch_num_nodes is None:
            return torch.tensor([self._n], device=self.device)
        return self._batch_num_nodes

    @property
    def csr(self):
        """Destination-major CSR (forward pass)."""
        o = self._cache_owner()
        if o._csr is None:
            indptr, indices, eid = build_csr(self._src, self._dst, self._n, self._n)
            o._csr = CsrView(self._n, self._n, indptr, indices, eid)
        if o._csr.part_sizes is None and o._csr._part_cuts is None and self._batch_num_nodes is not None:
            o._csr.part_sizes = self._batch_num_nodes
        return o._csr

    @property
    def csr_t(self):
        """Source-major CSR of the same edges; nidx[q] = forward CSR position of the edge
        at transposed position q (backward pass regenerates the forward noise)."""
        o = self._cache_owner()
        if o._csr_t is None:
            fwd = self.csr
            indptr, indices, eid = build_csr(self._dst, self._src, self._n, self._n)
            E = self.number_of_edges()
            pos_of_eid = torch.empty(E, dtype=torch.int32, device=self.device)
            if E:
                pos_of_eid[fwd.eid.long()] = torch.arange(E, dtype=torch.int32, device=self.device)
                nidx = pos_of_eid[eid.long()].contiguous()
            else:
                nidx = pos_of_eid
            o._csr_t = CsrView(self._n, self._n, indptr, indices, eid, nidx)
        if o._csr_t.part_sizes is None and o._csr_t._part_cuts is None and self._batch_num_nodes is not None:
            o._csr_t.part_sizes = self._batch_num_nodes      # (block-diagonal: both orientations share the boundaries)
        return o._csr_t

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        g = Graph(self._src.to(device), self._dst.to(device), self._n,
                  None if self._batch_num_nodes is None else self._batch_num_nodes.to(device))
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        g.edata = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.edata.items()}
        return g

    def __repr__(self):
        return f"Graph(num_nodes={self._n}, num_edges={self.number_of_edges()}, device={self.device})"

    # ---- message passing --------------------------------------------------------
    def update_all(self, message_func, reduce_func):
        """`update_all(u_mul_e | copy_u | copy_e, sum | mean)` on the HIP path.

        This is the call the reference's zoo layers make (stag/zoo/gcn.py:94-96,
        stag/zoo/graph_sage.py:71-73, stag/layers.py:12-15)."""
        from . import ops
        from .noise import EdgeNoise
        if not isinstance(message_func, fn.Message) or not isinstance(reduce_func, fn.Reduce):
            raise TypeError("update_all takes stag_amd.function builtins")
        kind = message_func.kind
        if reduce_func.kind == "max":          # composed, not fused (ops.aggregate_max)
            if kind not in ("copy_u", "u_mul_e"):
                raise NotImplementedError(f"update_all max with message {kind}")
            x = self.srcdata[message_func.fields[0]]
            w = self.edata[message_func.fields[1]] if kind == "u_mul_e" else None
            if torch.is_tensor(w):
                w = w.reshape(w.shape[0], -1)
            out = ops.aggregate_max(self, x.reshape(x.shape[0], -1), w)
            self.dstdata[reduce_func.out] = out.reshape((self._n,) + tuple(x.shape[1:]))
            return
        if kind == "copy_u":
            x, w = self.srcdata[message_func.fields[0]], None
        elif kind == "u_mul_e":
            x, w = self.srcdata[message_func.fields[0]], self.edata[message_func.fields[1]]
        elif kind == "copy_e":
            w = self.edata[message_func.fields[0]]
            x = None
        else:
            raise NotImplementedError(f"update_all with message {kind}")
        if x is None:   # sum of edge data: gather a broadcast row of ones
            shape = w.shape
            w2 = w.reshape(shape[0], -1)
            ones = torch.ones(1, w2.shape[1], dtype=w2.dtype, device=w2.device)
            out = ops.aggregate(self, ones, w2, reduce=reduce_func.kind, _broadcast_x=True)
            self.dstdata[reduce_func.out] = out.reshape((self._n,) + tuple(shape[1:]))
            return
        shape = x.shape
        x2 = x.reshape(shape[0], -1)
        if torch.is_tensor(w):
            if tuple(w.shape[1:]) != tuple(shape[1:]):   # DGL broadcasting, e.g. [E,H,1] x [N,H,F]
                while w.dim() < x.dim():
                    w = w.unsqueeze(-1)
                w = w.expand((w.shape[0],) + tuple(shape[1:]))
            w = w.reshape(w.shape[0], -1)
        elif w is not None and not isinstance(w, EdgeNoise):
            raise TypeError("edge weight must be a tensor or an EdgeNoise")
        out = ops.aggregate(self, x2, w, reduce=reduce_func.kind)
        self.dstdata[reduce_func.out] = out.reshape((self._n,) + tuple(shape[1:]))

    def apply_edges(self, func):
        if isinstance(func, fn.Message):
            if func.kind == "u_add_v":
                self.edata[func.fields[2]] = (_edge_rows(self, self.srcdata[func.fields[0]], "src")
                                              + _edge_rows(self, self.dstdata[func.fields[1]], "dst"))
            elif func.kind == "copy_u":
                self.edata[func.fields[1]] = _edge_rows(self, self.srcdata[func.fields[0]], "src")
            else:
                raise NotImplementedError(f"apply_edges with message {func.kind}")
        else:
            self.edata.update(func(_EdgeBatch(self)))


# ---- constructors / transforms (dgl.* free functions the scripts use) --------------
def graph(data, num_nodes=None, device=None):
    src, dst = data
    return Graph(src, dst, num_nodes=num_nodes, device=device)


def rand_graph(num_nodes, num_edges, device=None, generator=None):
    """`dgl.rand_graph` of the reference's tests (stag/tests/test_layers.py:17)."""
    src = torch.randint(0, num_nodes, (num_edges,), generator=generator)
    dst = torch.randint(0, num_nodes, (num_edges,), generator=generator)
    return Graph(src, dst, num_nodes, device=device)


def _carry_frames(g, new, edata):
    """Node frames and the batch structure survive an edge transform, as in DGL (the reference's scripts
    read g.ndata['feat'] / ['label'] / ['train_mask'] after remove_self_loop + add_self_loop,
    scripts/citation_mle/gcn/run.py:52-53)."""
    new.ndata = dict(g.ndata)
    new._batch_num_nodes = g._batch_num_nodes
    new.edata = edata
    return new


def remove_self_loop(g):
    """`dgl.remove_self_loop`: edge frames keep the rows of the surviving edges."""
    keep = g._src != g._dst
    edata = {k: (v[keep] if torch.is_tensor(v) and v.shape[:1] == keep.shape else v) for k, v in g.edata.items()}
    return _carry_frames(g, Graph(g._src[keep], g._dst[keep], g._n), edata)


def add_self_loop(g):
    """`dgl.add_self_loop`: one loop per node appended after the existing edges; their rows of every edge
    frame are zero, as in DGL."""
    loop = torch.arange(g._n, dtype=torch.int32, device=g.device)
    edata = {k: torch.cat([v, v.new_zeros((g._n,) + tuple(v.shape[1:]))], 0) for k, v in g.edata.items()
             if torch.is_tensor(v)}
    return _carry_frames(g, Graph(torch.cat([g._src, loop]), torch.cat([g._dst, loop]), g._n), edata)


def add_reverse_edges(g, copy_ndata=True, copy_edata=False):
    """`dgl.add_reverse_edges` (defaults as DGL's: node frames kept, edge frames dropped unless copy_edata,
    in which case a reverse edge carries its original's row)."""
    edata = ({k: torch.cat([v, v], 0) for k, v in g.edata.items() if torch.is_tensor(v)} if copy_edata else {})
    new = _carry_frames(g, Graph(torch.cat([g._src, g._dst]), torch.cat([g._dst, g._src]), g._n), edata)
    if not copy_ndata:
        new.ndata = {}
    return new


# batch() keeps the structure (COO, CSR views, plans) of the last BATCH_CACHE_SIZE unions, keyed by WHICH graphs were
# batched in WHICH order: a data loader that shuffles 20 PPI graphs into batches of 2 (scripts/ppi_mle/run.py:14) meets the
# same combination again within a few epochs, and a validation set batched once per epoch always does.  A hit costs the
# frames' concatenation only and inherits the cached union's plans (and, once it has been launched 16 times, its
# XCD-aware order).  The parts are held weakly; 0 switches it off.
# What is kept is a FRAME-LESS structure object: the graph handed to the caller is a copy that shares it (hit or miss), so a
# cached entry never holds a batch's feature tensors.  An entry goes when any of its parts dies (weakref callback), when
# more than BATCH_CACHE_SIZE unions are kept, or when the kept structures exceed BATCH_CACHE_MB of device memory (estimated
# from the union's size: COO, both CSR views, plans and unit orders).
BATCH_CACHE_SIZE = int(os.environ.get("STAG_BATCH_CACHE", "128"))
BATCH_CACHE_MB = float(os.environ.get("STAG_BATCH_CACHE_MB", "1024"))
_batch_cache = collections.OrderedDict()       # key -> (weak refs to the parts, structure, estimated bytes)


def _structure_bytes(g):
    """Device bytes a cached union may come to hold: src, dst, two views of (indptr, indices, eid) + nidx, and per view a
    plan (a 16-byte record per row) with its XCD-aware copies for a few row widths."""
    E, N = g.number_of_edges(), g.number_of_nodes()
    return 4 * (9 * E + 2 * (N + 1)) + 2 * 16 * N * 4


def _view_of(struct):
    """The graph the caller gets: shares the kept structure (views, plans, degree vectors live on `struct`), owns its frames."""
    out = struct._share_structure()
    out.ndata, out.edata, out._origin = {}, {}, struct
    return out


def _batch_cached(graphs):
    if BATCH_CACHE_SIZE <= 0 or len(graphs) > BATCH_CONCAT_MAX_GRAPHS:
        return None, None
    owners = [g._cache_owner() for g in graphs]
    key = tuple(id(o) for o in owners)
    hit = _batch_cache.get(key)
    if hit is not None and all(r() is o for r, o in zip(hit[0], owners)):
        _batch_cache.move_to_end(key)
        return _view_of(hit[1]), key
    return None, key


def _batch_remember(key, graphs, struct):
    if key is None:
        return
    drop = lambda _ref, key=key: _batch_cache.pop(key, None)       # a part died: its ids may be reused, the entry goes now
    _batch_cache[key] = ([weakref.ref(g._cache_owner(), drop) for g in graphs], struct, _structure_bytes(struct))
    cap = BATCH_CACHE_MB * 2 ** 20
    while len(_batch_cache) > 1 and (len(_batch_cache) > BATCH_CACHE_SIZE
                                     or sum(v[2] for v in _batch_cache.values()) > cap):
        _batch_cache.popitem(last=False)


def batch(graphs):
    """Block-diagonal union (`dgl.batch`, scripts/ppi_mle/run.py:12-14)."""
    graphs = list(graphs)
    cached, key = _batch_cached(graphs)
    if cached is not None:
        return _batch_frames(cached, graphs)
    struct = _batch_build(graphs)
    if key is None:
        return _batch_frames(struct, graphs)
    _batch_remember(key, graphs, struct)
    return _batch_frames(_view_of(struct), graphs)


def _batch_build(graphs):
    # one concatenation per array and ONE offset add over all edges (a batch of 4096 molecules is 4096 graphs: an add
    # per graph would be 8192 launches of a few bytes each)
    dev = graphs[0].device
    sizes = [g._n for g in graphs]
    n_edges = [int(g._src.shape[0]) for g in graphs]
    total = int(sum(sizes))
    node_off = np.concatenate([[0], np.cumsum(sizes[:-1])]).astype(np.int64) if graphs else np.zeros(0, np.int64)
    E = int(sum(n_edges))
    if any(g.device != dev for g in graphs):
        # (the one-launch path below hands raw device addresses of every part to a kernel: a part on the host or on
        # another card would be a GPU memory fault there, not the clean error torch.cat used to raise)
        raise ValueError(f"batch(): all graphs must live on one device; got {sorted({str(g.device) for g in graphs})}")
    if dev.type == "cuda" and 1 < len(graphs) <= BATCH_CONCAT_MAX_GRAPHS and E > 0:
        # the union's COO and both CSR views are the parts' arrays laid end to end (the parts keep their views: a data
        # loader hands the same graphs out again every epoch) — every piece of every array in one launch, no sort
        jobs = _ConcatJobs()
        e_off = np.concatenate([[0], np.cumsum(n_edges[:-1])]).astype(np.int64)
        src = torch.empty(E, dtype=torch.int32, device=dev)
        dst = torch.empty(E, dtype=torch.int32, device=dev)
        ptrs = np.stack([g._cache_owner()._array_ptrs() for g in graphs])      # [P, 9] device addresses, kept per part
        jobs.add(ptrs[:, 0], src, e_off, n_edges, add=node_off)
        jobs.add(ptrs[:, 1], dst, e_off, n_edges, add=node_off)
        views = _concat_csr(graphs, node_off, n_edges, None, total, jobs, ptrs)
        jobs.run(dev)
        out = Graph(src, dst, total, batch_num_nodes=torch.tensor(sizes, dtype=torch.int64, device=dev), _trusted=True)
        out._csr, out._csr_t = views
        for v in views:
            v.part_sizes = np.asarray(sizes, np.int64)      # (known on the host: no read-back when the order is built)
        return out
    if dev.type == "cuda":      # on the device, output size given: no host pass over the edges, no read-back
        edge_off = torch.repeat_interleave(torch.from_numpy(node_off).to(dev), torch.tensor(n_edges, dtype=torch.int64, device=dev),
                                           output_size=int(sum(n_edges)))
    else:                       # (numpy: torch's CPU repeat_interleave spins up its thread pool: 80 ms stalls seen)
        edge_off = torch.from_numpy(np.repeat(node_off, n_edges))
    src = torch.cat([g._src for g in graphs]).to(torch.int64) + edge_off
    dst = torch.cat([g._dst for g in graphs]).to(torch.int64) + edge_off
    out = Graph(src, dst, total, batch_num_nodes=torch.tensor(sizes, dtype=torch.int64, device=dev), _trusted=True)
    if 1 < len(graphs) <= BATCH_CONCAT_MAX_GRAPHS and sum(n_edges) > 0:
        # a block-diagonal union's CSRs are its parts' CSRs laid end to end: no sort per batch (the parts keep theirs —
        # a data loader hands the same graphs out again every epoch)
        out._csr, out._csr_t = _concat_csr(graphs, node_off, n_edges, edge_off.to(torch.int32), total)
    return out


def _batch_frames(out, graphs):
    for k in set(graphs[0].ndata):
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], 0)
    for k in set(graphs[0].edata):
        out.edata[k] = torch.cat([g.edata[k] for g in graphs], 0)
    return out


BATCH_CONCAT_MAX_GRAPHS = 256   # (128 molecules: 0.25 against 0.29 ms; a batch of 4096: walking 4096 Python objects for their
                                # sizes and addresses costs 3.5 ms against 2.2 ms for two 4096-way concatenations and two sorts)


class _ConcatJobs:
    """A table of stag_concat_job records (include/stag_hip.h): pieces of device arrays laid end to end with offsets, all
    of them in ONE launch and one small upload — twenty torch concatenations and adds cost a batch 0.3 ms of host time."""

    def __init__(self):
        self.blocks = []

    def add(self, src_ptrs, dst, dst_at, counts, add=0, kind=0):
        """One piece per part: src_ptrs[P] device addresses of int32 arrays (kind 2: none, dst is filled with `add`),
        written into tensor `dst` from element dst_at[P]; counts[P] elements; add: scalar or [P]."""
        counts = np.asarray(counts, np.int64).reshape(-1)
        rows = np.zeros((len(counts), 5), np.int64)
        rows[:, 0] = src_ptrs
        rows[:, 1] = dst.data_ptr() + np.asarray(dst_at, np.int64) * 4
        rows[:, 2], rows[:, 3], rows[:, 4] = counts, add, kind
        self.blocks.append(rows[counts > 0])

    def run(self, dev):
        rows = np.concatenate(self.blocks) if self.blocks else np.zeros((0, 5), np.int64)
        J = len(rows)
        if J == 0:
            return
        t64 = np.zeros((J, 4), np.int64)
        t32 = t64.view(np.int32).reshape(J, 8)
        t64[:, 0:3] = rows[:, 0:3]
        t32[:, 6:8] = rows[:, 3:5]
        start = np.zeros(J + 1, np.int64)
        np.cumsum((rows[:, 2] + 1023) // 1024, out=start[1:])
        buf = torch.from_numpy(np.concatenate([t64.reshape(-1), start])).to(dev)     # the one upload
        with _lib.on_device(dev):
            _lib.check(_lib.lib().stag_concat_jobs(buf.data_ptr(), buf.data_ptr() + J * 32, J, int(start[-1]),
                                                   _lib.stream_of(dev)), "stag_concat_jobs")


def _concat_csr(graphs, node_off, n_edges, node_off_per_edge, total, jobs=None, ptrs=None):
    """(csr, csr_t) of batch(graphs) from the parts' own views: row pointers shifted by the edges before the part, column
    ids by the nodes before it, edge ids and forward positions by the edges before it — array for array what build_csr
    makes of the batch's COO (rows stay in part order, a row's edges in ascending edge id)."""
    dev = graphs[0].device
    E = int(sum(n_edges))
    sizes = [g._n for g in graphs]
    e_off = np.concatenate([[0], np.cumsum(n_edges[:-1])]).astype(np.int64)
    if jobs is not None:         # on the device: pieces of a job table (the caller runs it)
        views = []
        n0, e0 = np.asarray(node_off, np.int64), np.asarray(e_off, np.int64)
        for name, col in (("csr", 2), ("csr_t", 5)):
            indptr = torch.empty(total + 1, dtype=torch.int32, device=dev)
            indices = torch.empty(E, dtype=torch.int32, device=dev)
            eid = torch.empty(E, dtype=torch.int32, device=dev)
            nidx = torch.empty(E, dtype=torch.int32, device=dev) if name == "csr_t" else None
            jobs.add(ptrs[:, col], indptr, n0, sizes, add=e0)
            jobs.add(ptrs[:, col + 1], indices, e0, n_edges, add=n0)
            jobs.add(ptrs[:, col + 2], eid, e0, n_edges, add=e0)
            if nidx is not None:
                jobs.add(ptrs[:, 8], nidx, e0, n_edges, add=e0)
            jobs.add([0], indptr, [total], [1], add=E, kind=2)
            views.append(CsrView(total, total, indptr, indices, eid, nidx))
        return views
    rep = lambda vals, counts, size: torch.from_numpy(np.repeat(vals, counts)).to(torch.int32)
    e_off_per_node = rep(e_off, sizes, total)
    e_off_per_edge = rep(e_off, n_edges, E)
    last = torch.tensor([E], dtype=torch.int32, device=dev)
    views = []
    for name in ("csr", "csr_t"):
        parts = [getattr(g, name) for g in graphs]
        indptr = torch.cat([v.indptr[:-1] for v in parts] + [last])
        indptr[:-1] += e_off_per_node
        indices = torch.cat([v.indices for v in parts]) + node_off_per_edge
        eid = torch.cat([v.eid for v in parts]) + e_off_per_edge
        nidx = (torch.cat([v.nidx for v in parts]) + e_off_per_edge) if name == "csr_t" else None
        views.append(CsrView(total, total, indptr, indices, eid, nidx))
    return views


def _readout(g, name, reduce):
    from . import ops
    x = g.ndata[name]
    sizes = g.batch_num_nodes().to(torch.int64)
    offsets = torch.zeros(sizes.shape[0] + 1, dtype=torch.int32, device=x.device)
    offsets[1:] = torch.cumsum(sizes, 0).to(torch.int32)
    shape = x.shape
    out = ops.segment_reduce(x.reshape(shape[0], -1), offsets, reduce)
    return out.reshape((sizes.shape[0],) + tuple(shape[1:]))


def sum_nodes(g, name):
    return _readout(g, name, "sum")


def mean_nodes(g, name):
    return _readout(g, name, "mean")
