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
