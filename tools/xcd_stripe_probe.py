#!/usr/bin/env python
"""What the XCD-aware unit order (stag_plan.xcd_order) is worth: the same launches with and without it, interleaved in
one process.  Workgroups go to the 8 XCDs round-robin and each XCD has its own 4 MB L2: when workgroup b takes its
units from destination-row stripe b mod 8, a graph whose sources are near its destinations (block-diagonal batches:
PPI, molecules) gathers from 1/8 of the table per XCD; a random-source graph (cfg2) has nothing to gain.

    python tools/xcd_stripe_probe.py            (GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import _lib, ops, synthetic  # noqa: E402
import bench  # noqa: E402


def without_xcd_order(view, seg_len):
    p = view.plan(seg_len, need=True)
    p["xcd"], p["xcd_strides"], p["xcd_on"], p["xcd_decided"] = None, (0, 0), False, True
    p.pop("_structs", None)
    p.pop("_ints", None)


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


def main():
    import importlib
    importlib.import_module("stag_amd.graph").XCD_ORDER = "1"     # the comparison wants it on every graph ("auto" decides
    dev = torch.device("cuda:0")                                    # by the locality printed below)
    cases = []
    s, d, sizes = synthetic.ppi_like()
    cases.append(("cfg3 PPI batch", s, d, int(sizes.sum()), (50, 128, 256), "mean"))
    s, d, sizes = synthetic.molecules_like()
    cases.append(("cfg4 molecules", s, d, int(sizes.sum()), (128,), "sum"))
    s, d = synthetic.arxiv_like(seed=1)
    cases.append(("cfg2 arxiv", s, d, synthetic.ARXIV_NODES, (128,), "sum"))
    for name, s, d, n, Ds, red in cases:
        for D in Ds:
            ga = stag_amd.Graph(torch.from_numpy(s), torch.from_numpy(d), n, device=dev)
            gb = stag_amd.Graph(torch.from_numpy(s), torch.from_numpy(d), n, device=dev)
            without_xcd_order(ga.csr, 64)
            x = torch.randn(n, D, device=dev)
            for noise in ("none", "normal"):
                fa = lambda i: ops.aggregate(ga, x, bench.make_noise(stag_amd, ga, D, noise, i), reduce=red, seg_len=64)
                fb = lambda i: ops.aggregate(gb, x, bench.make_noise(stag_amd, gb, D, noise, i), reduce=red, seg_len=64)
                same = torch.equal(fa(0), fb(0))
                ta, tb = [], []
                for r in range(5):
                    ta.append(timeit(fa))
                    tb.append(timeit(fb))
                print(f"{name:16s} D={D:4d} {noise:7s} plan order {np.median(ta):7.1f} us   XCD-aware {np.median(tb):7.1f} us   "
                      f"(bit-identical: {same}; stripe locality {gb.csr.stripe_locality():.3f})", flush=True)


def planless():
    """Launches without a plan (a freshly batched minibatch of short-row graphs never gets one): this build against
    tools/_bin/libstag_noxcdpl.so (python tools/ab_bench.py build noxcdpl="-DSTAG_XCD_PLANLESS=0")."""
    dev = torch.device("cuda:0")
    alt = os.path.join(ROOT, "tools", "_bin", "libstag_noxcdpl.so")
    if not os.path.exists(alt):
        return
    base = _lib.lib()
    other = _lib.bind(alt)
    s, d, sizes = synthetic.molecules_like()
    n = int(sizes.sum())
    g = stag_amd.Graph(torch.from_numpy(s), torch.from_numpy(d), n, device=dev)
    for D in (9, 128):
        x = torch.randn(n, D, device=dev)
        for noise in ("none", "normal"):
            f = lambda i: ops.aggregate(g, x, bench.make_noise(stag_amd, g, D, noise, i), seg_len=None)
            t = {"linear": [], "striped by rows": []}
            outs = {}
            for r in range(5):
                for name, l in (("linear", other), ("striped by rows", base)):
                    _lib._lib = l
                    _lib.lib = lambda l=l: l
                    outs[name] = f(0)
                    t[name].append(timeit(f))
            _lib._lib = base
            _lib.lib = lambda: base
            print(f"cfg4 molecules, no plan D={D:4d} {noise:7s} linear {np.median(t['linear']):7.1f} us   striped by rows "
                  f"{np.median(t['striped by rows']):7.1f} us   (bit-identical: {torch.equal(outs['linear'], outs['striped by rows'])})",
                  flush=True)




def gat():
    """The cooperative GAT kernels with the XCD-aware unit batches (stag_plan_blocks_xcd) and in plan order: forward, and
    forward + backward, on the PPI-sized batch (scripts/ppi_mle/gat/run.py: 4 heads x 256) and on cfg5's arxiv graph."""
    import importlib
    G = importlib.import_module("stag_amd.graph")
    dev = torch.device("cuda:0")
    s, d, sizes = synthetic.ppi_like()
    s2, d2 = synthetic.arxiv_like(seed=1)
    for gname, src, dst, n, shapes in (("PPI batch", s, d, int(sizes.sum()), ((4, 64), (8, 32), (4, 256))),
                                       ("cfg5 arxiv", s2, d2, synthetic.ARXIV_NODES, ((8, 32),))):
        graphs = {}
        for mode in ("0", "1"):
            G.XCD_ORDER = mode
            g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
            for view in (g.csr, g.csr_t):
                view.plan(64, need=True)["xcd_decided"] = True       # (the policy is read at every request for a plan)
            graphs[mode] = g
        for H, F in shapes:
            el, er = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
            ft = torch.randn(n, H, F, device=dev)
            gout = torch.randn(n, H, F, device=dev)
            elg, erg, ftg = (t_.clone().requires_grad_(True) for t_ in (el, er, ft))
            mk = lambda g, i: stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i)

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
                fa, fb = mkf(graphs["0"]), mkf(graphs["1"])
                same = torch.equal(fa(0), fb(0))
                ta, tb = [], []
                for r in range(4):
                    ta.append(timeit(fa, steps=100))
                    tb.append(timeit(fb, steps=100))
                print(f"GAT {what:18s} {gname:10s} H={H} F={F:3d} plan order {np.median(ta):7.1f} us   XCD-aware batches "
                      f"{np.median(tb):7.1f} us   (bit-identical: {same}; stripe locality {graphs['1'].csr.stripe_locality():.3f})",
                      flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gat":
        gat()
    else:
        planless()
        main()
