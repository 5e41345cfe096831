#!/usr/bin/env python
"""Device time of every entry point and layer mode at BASELINE sizes (cfg2 graph unless noted) — a
survey to catch paths that fell off the fast road.  `python tools/op_survey.py` on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import _lib, ops, synthetic  # noqa: E402


def ev(fn, k=20, warm=3):
    for i in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


def main():
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n, E, D = synthetic.ARXIV_NODES, len(src), 128
    t0 = time.perf_counter()
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    g.csr
    torch.cuda.synchronize()
    print(f"Graph + device CSR build (first call)      : {(time.perf_counter() - t0) * 1e3:9.1f} ms")
    t0 = time.perf_counter(); g.csr.plan(64); print(f"launch plan (first call)                    : {(time.perf_counter() - t0) * 1e3:9.1f} ms")
    t0 = time.perf_counter(); g.csr_t; torch.cuda.synchronize(); print(f"source-major twin                           : {(time.perf_counter() - t0) * 1e3:9.1f} ms")
    g.csr_t.plan(64)
    x = torch.randn(n, D, device=dev)
    gout = torch.randn(n, D, device=dev)
    N = torch.distributions.Normal

    def layer_step(layer, xin):
        xin = xin.detach().requires_grad_(True)

        def f():
            for p in layer.parameters():
                p.grad = None
            layer(g, xin).backward(gout[:, :layer.base_layer._out_feats] if hasattr(layer.base_layer, "_out_feats") else gout)
        return f

    print("--- layer training steps (forward + backward), cfg2 graph, 128 -> 128")
    modes = {
        "GCN  fixed Normal(1,.5)                  ": dict(q_a=N(1.0, 0.5)),
        "GCN  Bernoulli(.5) + norm (arxiv_mle)    ": dict(q_a=torch.distributions.Bernoulli(0.5), norm=True),
        "GCN  vi Normal, relu   (r1)              ": dict(q_a=N(1.0, 0.5), vi=True, relu=True),
        "GCN  vi Normal per-channel (rc)          ": dict(q_a=N(torch.ones(D), 0.5 * torch.ones(D)), vi=True),
        "GCN  vi Normal + norm                    ": dict(q_a=N(1.0, 0.5), vi=True, norm=True),
    }
    for name, kw in modes.items():
        layer = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(D, D), **kw).to(dev)
        print(f"{name}: {ev(layer_step(layer, x), 10):9.1f} us")
    for of, tag in ((1, "re "), (D, "rec")):
        q = stag_amd.distributions.AmortizedDistribution(D, of, init_like=N(1.0, 0.3))
        layer = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(D, D), q_a=q, vi=True).to(dev)
        print(f"GCN  amortised [E,{of:3d}] parameters ({tag})    : {ev(layer_step(layer, x), 5):9.1f} us")
    layer = stag_amd.layers.StagLayer(stag_amd.zoo.GraphSAGE(D, D), q_a=N(1.0, 0.5)).to(dev)
    print(f"SAGE fixed Normal                        : {ev(layer_step(layer, x), 10):9.1f} us")
    layer = stag_amd.layers.StagLayer(stag_amd.zoo.GIN(D, D), q_a=N(1.0, 0.5)).to(dev)
    print(f"GIN  fixed Normal                        : {ev(layer_step(layer, x), 10):9.1f} us")
    gat = stag_amd.layers.StagLayer(stag_amd.zoo.GAT(D, 32, num_heads=8), q_a=N(1.0, 0.5)).to(dev)
    xin = x.detach().requires_grad_(True)

    def gat_step():
        for p in gat.parameters():
            p.grad = None
        out = gat(g, xin)
        out.backward(torch.ones_like(out))
    print(f"GAT  8 heads x 32, noise [E,8]           : {ev(gat_step, 5):9.1f} us")
    with torch.no_grad():
        out, attn = gat.base_layer(g, x, get_attention=True)
        print(f"GAT  forward with get_attention=True     : {ev(lambda: gat.base_layer(g, x, get_attention=True), 5):9.1f} us")
    print("--- single entry points")
    nz = stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=0)
    with torch.no_grad():
        print(f"aggregate (headline)                     : {ev(lambda: ops.aggregate(g, x, nz)):9.1f} us")
        print(f"materialise [E,128]                      : {ev(lambda: nz.materialize()):9.1f} us")
        w = nz.materialize()
        print(f"aggregate, explicit [E,128] weights      : {ev(lambda: ops.aggregate(g, x, w)):9.1f} us")
        print(f"agg_bwd_w explicit dw                    : {ev(lambda: ops._bwd_w_raw(g.csr, x, gout, D, None)):9.1f} us")
        print(f"agg_bwd (dx + 2 derivative aggregates)   : {ev(lambda: ops._agg_bwd_raw(g.csr_t, gout, D, nz.spec(), None, None, 64, True)):9.1f} us")
        print(f"coldot (2 outputs)                       : {ev(lambda: ops.coldot(x, gout, gout)):9.1f} us")
        offs = torch.arange(0, n + 1, 41, dtype=torch.int32, device=dev)
        print(f"segment_reduce mean, {len(offs) - 1} graphs of 41 rows  : {ev(lambda: ops.segment_reduce(x[:int(offs[-1])], offs, 'mean')):9.1f} us")
        s32, d32 = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
        from stag_amd.graph import build_csr
        print(f"stag_csr_build (E = {E})             : {ev(lambda: build_csr(s32, d32, n, n), 5):9.1f} us")
        spec = nz.spec()
        spec = spec if not isinstance(spec, tuple) else ops._targs_to_ctypes(spec)
        print(f"agg_bwd_dp (dx + finished dp0, dp1)      : {ev(lambda: ops._agg_bwd_dp_raw(g.csr_t, gout, x, D, spec, None, None, 64)):9.1f} us")
        loc, ls = torch.rand(E, 1, device=dev) + 0.5, torch.rand(E, 1, device=dev) - 1.5
        nz1 = stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, loc, ls, seed=1, offset=0, p1_log=True)
        sp1 = nz1.spec()
        sp1 = sp1 if not isinstance(sp1, tuple) else ops._targs_to_ctypes(sp1)
        print(f"aggregate, [E,1] parameters              : {ev(lambda: ops.aggregate(g, x, nz1)):9.1f} us")
        print(f"agg_bwd_edge (dx + [E,1] gradients)      : {ev(lambda: ops._agg_bwd_edge_raw(g.csr_t, gout, x, D, sp1, None, None, 64)):9.1f} us")
        w2, b2 = torch.randn(D, 2, device=dev), torch.randn(2, device=dev)
        print(f"node_project [N,128] -> [N,2]            : {ev(lambda: ops.node_project(x, w2, b2)):9.1f} us")
        P = ops.node_project(x, w2, b2)
        wh, bh = torch.randn(1, 2, device=dev), torch.randn(2, device=dev)
        print(f"edge_mlp (hidden 1, 2 parameters)        : {ev(lambda: ops.edge_mlp(g, P, wh, bh)):9.1f} us")
        pl, ps = torch.tensor(1.0, device=dev), torch.tensor(0.5, device=dev)
        print(f"normal_kl_mean over [E,1]                : {ev(lambda: ops.normal_kl_mean(loc, ls, pl, ps)):9.1f} us")
        ft = torch.randn(n, 8, 32, device=dev)
        al, ar = torch.randn(1, 8, 32, device=dev), torch.randn(1, 8, 32, device=dev)
        print(f"head_dot [N,8,32] -> el, er              : {ev(lambda: ops.head_dot(ft, al, ar)):9.1f} us")
        print(f"column_sum [N,121]                       : {ev(lambda: ops.column_sum(ft.reshape(n, 256)[:, :121].contiguous())):9.1f} us (incl. the slice copy)")
        g.csr._plans.clear(); g.csr_t._plans.clear()
        import time as _t
        torch.cuda.synchronize(); t0 = _t.perf_counter(); g.csr.plan(64); torch.cuda.synchronize()
        print(f"stag_plan_device (arxiv CSR)             : {(_t.perf_counter() - t0) * 1e6:9.1f} us")


if __name__ == "__main__":
    main()
