#!/usr/bin/env python
"""One training step of BASELINE configs[2] as the reference runs it (scripts/ppi_mle/run.py:70-77): a freshly
batched graph every step (24 PPI-sized graphs), GraphSAGE(mean) 50 -> 256 -> 256 -> 121 with StagLayer noise,
BCE loss, Adam — the graph's construction (batch, both CSRs, plans) is part of the step.

    python tools/ppi_step.py [--steps 30] [--static]      # --static: one graph reused (construction outside the loop)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--static", action="store_true")
ap.add_argument("--batch-cache", action="store_true",
                help="leave graph.batch's structure cache on: the same 24 graphs in the same order every step is then a hit "
                     "(what a validation set, or a loader that has seen the combination before, gets)")
ap.add_argument("--model", default="sage", choices=["sage", "gat"],
                help="gat: scripts/ppi_mle/gat/run.py:21-58 — GAT(50, 256, 4 heads) -> GAT(1024, 256, 4) -> GAT(1024, 121, 4, last)")
args = ap.parse_args()
torch.distributions.Distribution.set_default_validate_args(False)
if not args.batch_cache:
    import importlib
    importlib.import_module("stag_amd.graph").BATCH_CACHE_SIZE = 0      # "fresh" means built, not looked up
dev = torch.device("cuda:0")
s, d, sizes = synthetic.ppi_like()
off = np.concatenate([[0], np.cumsum(sizes)])
gid = np.searchsorted(off, s, side="right") - 1
order = np.argsort(gid, kind="stable")
s, d, gid = s[order], d[order], gid[order]
cuts = np.searchsorted(gid, np.arange(len(sizes) + 1))
parts = [stag_amd.Graph(torch.from_numpy(s[cuts[i]:cuts[i + 1]] - off[i]).to(dev),
                        torch.from_numpy(d[cuts[i]:cuts[i + 1]] - off[i]).to(dev), int(sizes[i]), device=dev)
         for i in range(len(sizes))]
n = int(sizes.sum())
x = torch.randn(n, 50, device=dev)
y = (torch.rand(n, 121, device=dev) < 0.3).float()
N = torch.distributions.Normal
SL, FO, Z = stag_amd.layers.StagLayer, stag_amd.layers.FeatOnlyLayer, stag_amd.zoo
if args.model == "gat":
    elu = torch.nn.functional.elu
    layers = torch.nn.ModuleList([
        SL(Z.GAT(50, 256, num_heads=4, activation=elu), q_a=N(1.0, 0.3)),
        SL(Z.GAT(1024, 256, num_heads=4, activation=elu), q_a=N(1.0, 0.3)),
        SL(Z.GAT(1024, 121, num_heads=4, last=True), q_a=N(1.0, 0.3))]).to(dev)
else:
    layers = torch.nn.ModuleList([
        SL(Z.GraphSAGE(50, 256, aggregator_type="mean", activation=torch.relu), q_a=N(1.0, 0.3)),
        SL(Z.GraphSAGE(256, 256, aggregator_type="mean", activation=torch.relu), q_a=N(1.0, 0.3)),
        SL(Z.GraphSAGE(256, 121, aggregator_type="mean"), q_a=N(1.0, 0.3))]).to(dev)
opt = torch.optim.Adam(layers.parameters(), 1e-3)
static_graph = stag_amd.batch(parts)


def step():
    g = static_graph if args.static else stag_amd.batch(parts)
    opt.zero_grad()
    h = x
    for layer in layers:
        h = layer(g, h)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(h, y)
    loss.backward()
    opt.step()
    return loss


for _ in range(12 if args.static else 5):     # (a static graph's views get their XCD-aware order after 16 launches: untimed)
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
print(f"{'static graph' if args.static else 'batch per step, structure cache hit' if args.batch_cache else 'fresh batch per step'}: N={n} E={len(s)}: "
      f"{(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms per step, loss {loss.item():.4f}")
