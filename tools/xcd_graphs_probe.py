#!/usr/bin/env python
"""Whole graphs per XCD on block-diagonal batches: the same launches with the XCD-aware order cut
  eighths   at equal eighths of the CSR (round 3: a graph straddles stripes, fine ranges cut graphs),
  graphs    at graph boundaries, bin-packed to the 8 stripes, heavy and light units in their own stripe families,
  merged    the same with ONE family of stripes for the wide shapes (an XCD passes over a fine range once),
interleaved in one process, bit-identity checked.  BASELINE configs[2] (PPI batch) and configs[3] (molecule batch);
`gat` for the cooperative GAT kernels' batches.

    python tools/xcd_graphs_probe.py [gat] [--range-mb 1.5,2.5,3.5]      (GPU box)
"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import _lib, ops, synthetic  # noqa: E402
import bench  # noqa: E402

G = importlib.import_module("stag_amd.graph")


def timeit(fn, steps=200, warm=20):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def build(s, d, n, sizes, mode, dev, views=("csr",)):
    """A graph whose plan carries the order of `mode` (decided when the plan is first asked for)."""
    G.XCD_ORDER = "1"
    G.MERGE_HEAVY_ABOVE = 64          # "merged": wherever a row takes 32 lanes or more
    g = stag_amd.Graph(torch.from_numpy(s), torch.from_numpy(d), n,
                       batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)
    for v in views:
        view = getattr(g, v)
        view.xcd_graphs, view.xcd_merge = mode != "eighths", mode == "merged"      # per view: read at every launch
        view.plan(64, need=True)
    return g


def agg(range_mbs):
    dev = torch.device("cuda:0")
    s3, d3, z3 = synthetic.ppi_like()
    s4, d4, z4 = synthetic.molecules_like()
    for name, s, d, sizes, Ds, red in (("cfg3 PPI batch", s3, d3, z3, (50, 128, 256), "mean"),
                                       ("cfg4 molecules", s4, d4, z4, (128,), "sum")):
        n = int(sizes.sum())
        for mb in range_mbs:
            G.XCD_RANGE_BYTES = int(mb * 1e6)
            for D in Ds:
                x = torch.randn(n, D, device=dev)
                for noise in ("none", "normal"):
                    fns, outs = {}, {}
                    for mode in ("eighths", "graphs", "merged"):
                        g = build(s, d, n, sizes, mode, dev)
                        # the modes are module switches read when an order is built: build it now, under this mode's
                        g.csr.xcd_order(g.csr.plan(64), D)
                        fns[mode] = (lambda i, g=g: ops.aggregate(g, x, bench.make_noise(stag_amd, g, D, noise, i), reduce=red,
                                                                  seg_len=64))
                        outs[mode] = fns[mode](0)
                    same = all(torch.equal(outs["eighths"], o) for o in outs.values())
                    t = {m: [] for m in fns}
                    for r in range(5):
                        for m, f in fns.items():
                            t[m].append(timeit(f))
                    fine = g.csr.xcd_ranges(D)[2]
                    print(f"{name:15s} range {mb:3.1f} MB D={D:4d} {noise:7s} " +
                          "  ".join(f"{m} {np.median(v):6.1f} us" for m, v in t.items()) +
                          f"   (fine {fine}; bit-identical: {same})", flush=True)


def gat(range_mbs):
    dev = torch.device("cuda:0")
    s, d, sizes = synthetic.ppi_like()
    n = int(sizes.sum())
    for mb in range_mbs:
        G.XCD_RANGE_BYTES = int(mb * 1e6)
        for H, F in ((4, 64), (8, 32), (4, 256)):
            el, er = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
            ft, gout = torch.randn(n, H, F, device=dev), torch.randn(n, H, F, device=dev)
            elg, erg, ftg = (t_.clone().requires_grad_(True) for t_ in (el, er, ft))
            mk = lambda g, i: stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i)
            graphs = {}
            for mode in ("eighths", "graphs"):
                g = build(s, d, n, sizes, mode, dev, views=("csr", "csr_t"))
                g.csr.gat_blocks(g.csr.plan(64, need=True), H * F)
                g.csr_t.gat_blocks(g.csr_t.plan(64, need=True), H * F)
                graphs[mode] = g

            def fwd(g):
                def f(i):
                    with torch.no_grad():
                        return ops.gat_aggregate(g, el, er, ft, 0.2, mk(g, i))
                return f

            def train(g):
                def f(i):
                    elg.grad = erg.grad = ftg.grad = None
                    ops.gat_aggregate(g, elg, erg, ftg, 0.2, mk(g, i)).backward(gout)
                    return ftg.grad
                return f
            for what, mkf in (("forward", fwd), ("forward + backward", train)):
                fns = {m: mkf(g) for m, g in graphs.items()}
                same = torch.equal(fns["eighths"](0), fns["graphs"](0))
                t = {m: [] for m in fns}
                for r in range(4):
                    for m, f in fns.items():
                        t[m].append(timeit(f, steps=100))
                print(f"GAT {what:18s} PPI batch range {mb:3.1f} MB H={H} F={F:3d} " +
                      "  ".join(f"{m} {np.median(v):7.1f} us" for m, v in t.items()) + f"   (bit-identical: {same})", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="agg", choices=["agg", "gat"])
    ap.add_argument("--range-mb", default="2.5")
    a = ap.parse_args()
    mbs = [float(v) for v in a.range_mb.split(",")]
    (gat if a.what == "gat" else agg)(mbs)
