#!/usr/bin/env python
"""Repeatability soak at BASELINE size: the same training step (same seed, same offsets) run `--steps` times, every
`--every`-th result compared bit for bit with the first.  Covers the paths whose long rows go through partial sums,
arrival counters or multi-stage reductions: the aggregation forward / backward (cfg2), the one-pass [E,1] backward
and the narrow amortised heads with the KL term, GAT forward and its one-gather backward (cfg5).

    python tools/soak.py [--steps 5000] [--every 50]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stag_amd  # noqa: E402
from stag_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5000)
ap.add_argument("--every", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
src, dst = synthetic.arxiv_like(seed=1)
n = synthetic.ARXIV_NODES
g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
D = 128
torch.manual_seed(0)
x0 = torch.randn(n, D, device=dev)
gout = torch.randn(n, 256, device=dev)
N = torch.distributions.Normal
layers = {
    "gcn vi relu (stag_agg_fwd, stag_agg_bwd_dp, normal_kl)":
        stag_amd.layers.StagLayer(stag_amd.zoo.GCN(D, D), q_a=N(1.0, 0.5), vi=True, relu=True),
    "gcn amortised [E,1] + KL (stag_agg_bwd_edge, node_project, edge_mlp, normal_kl)":
        stag_amd.layers.StagLayer(stag_amd.zoo.GCN(D, D), q_a=stag_amd.distributions.AmortizedDistribution(
            D, 1, init_like=N(1.0, 0.3)), vi=True),
    "gat 8x32 (stag_head_dot, stag_gat_fwd, stag_gat_bwd)":
        stag_amd.layers.StagLayer(stag_amd.zoo.GAT(D, 32, num_heads=8), q_a=N(1.0, 0.5)),
}


def step(layer):
    stag_amd.manual_seed(7)
    layer.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_(True)
    y = layer(g, x)
    kl = layer.kl_divergence()
    outs, grads = [y], [gout[:, :y.shape[1]]]
    if torch.is_tensor(kl):
        outs.append(kl)
        grads.append(torch.ones((), device=dev))
    torch.autograd.backward(outs, grads)
    return [y.detach(), x.grad] + [o.detach() for o in outs[1:]] + [p.grad for p in layer.parameters() if p.grad is not None]


ap_only = os.environ.get("STAG_SOAK_ONLY", "")       # "r04": only the round-4 paths below

# ---- round 4: the partitioned steps on a shard of one (the overlapped forward, the staged GAT backward with its
#      fixed-order combine), and the whole-graphs-per-XCD orders on the PPI-sized batch (two blocks of rows in flight, the
#      XCD-local GAT batches with two rows per round) -----------------------------------------------------------------------
def r04_cases():
    import importlib
    from stag_amd import _lib, ops
    from stag_amd.partition import GraphShard
    G = importlib.import_module("stag_amd.graph")
    sh = GraphShard(src, dst, n, 0, 1, device=dev)
    H, F = 8, 32
    el0, er0, ft0 = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, F, device=dev)
    gG = torch.randn(n, H, F, device=dev)

    def shard_gat():
        el, er, ft = (t.clone().requires_grad_(True) for t in (el0, er0, ft0))
        nz = stag_amd.EdgeNoise(sh, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=3, offset=4)
        out = sh.gat_aggregate(el, er, ft, 0.2, nz, attn_drop=(0.6, 12, 5))
        assert "ShardGat" in type(out.grad_fn).__name__
        out.backward(gG)
        return [out.detach(), ft.grad, el.grad, er.grad]

    def shard_agg():
        x = x0.clone().requires_grad_(True)
        nz = stag_amd.EdgeNoise(sh, D, _lib.NOISE_BERNOULLI, 0.7, None, seed=3, offset=4, in_norm=True)
        out = sh.aggregate(x, nz)
        out.backward(gout[:, :D])
        return [out.detach(), x.grad]

    G.XCD_ORDER = "1"
    s3, d3, z3 = synthetic.ppi_like()
    n3 = int(z3.sum())
    g3 = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), n3, batch_num_nodes=torch.from_numpy(z3).to(dev), device=dev)
    x3 = torch.randn(n3, 256, device=dev)
    g3o = torch.randn(n3, 256, device=dev)
    e3, r3, f3 = torch.randn(n3, 4, device=dev), torch.randn(n3, 4, device=dev), torch.randn(n3, 4, 64, device=dev)
    G3 = torch.randn(n3, 4, 64, device=dev)

    def ppi_agg():
        res = []
        for nz in (None, stag_amd.EdgeNoise(g3, 256, _lib.NOISE_NORMAL, 1.0, 0.5, seed=3, offset=4)):
            x = x3.clone().requires_grad_(True)
            out = ops.aggregate(g3, x, nz, reduce="mean")
            out.backward(g3o)
            res += [out.detach(), x.grad]
        return res

    def ppi_gat():
        el, er, ft = (t.clone().requires_grad_(True) for t in (e3, r3, f3))
        out = ops.gat_aggregate(g3, el, er, ft, 0.2, stag_amd.EdgeNoise(g3, 4, _lib.NOISE_NORMAL, 1.0, 0.5, seed=3, offset=4))
        out.backward(G3)
        return [out.detach(), ft.grad, el.grad, er.grad]

    return {"shard of one: GAT step (_ShardGat, stag_gat_bwd_stages, in-kernel dropout)": shard_gat,
            "shard of one: Bernoulli + in-norm aggregation step (_ShardAggregate)": shard_agg,
            "PPI batch D = 256, whole graphs per XCD: no draw and Normal, forward + dx": ppi_agg,
            "PPI batch GAT 4 x 64 on XCD-local batches, forward + one-gather backward": ppi_gat}


for name, fn in r04_cases().items():
    first = [t.clone() for t in fn()]
    bad, checked, t0 = 0, 0, time.perf_counter()
    for i in range(1, args.steps):
        res = fn()
        if i % args.every == 0:
            checked += 1
            if not all(torch.equal(a, b) for a, b in zip(first, res)):
                bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {args.steps} steps in {time.perf_counter() - t0:.1f} s, {checked} compared, {bad} differing", flush=True)
    if bad:
        raise SystemExit(1)
if ap_only == "r04":
    print("soak ok (round-4 paths)")
    raise SystemExit(0)

for name, layer in layers.items():
    layer = layer.to(dev)
    first = [t.clone() for t in step(layer)]
    bad, checked, t0 = 0, 0, time.perf_counter()
    for i in range(1, args.steps):
        res = step(layer)
        if i % args.every == 0:
            checked += 1
            if not all(torch.equal(a, b) for a, b in zip(first, res)):
                bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {args.steps} steps in {time.perf_counter() - t0:.1f} s, {checked} compared, {bad} differing", flush=True)
    if bad:
        raise SystemExit(1)
print("soak ok")
