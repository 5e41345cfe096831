#!/usr/bin/env python
"""One training epoch (= one full-graph step) of the reference's arxiv GAT model, scripts/arxiv_mle/gat/run.py:29-64:
StagLayer(GAT(128, 8, heads 8, feat_drop 0.6, attn_drop 0.6, elu)) -> StagLayer(GAT(64, 40, heads 8, last=True,
feat_drop 0.6, attn_drop 0.6, softmax)), Normal(1, std) edge noise of width 8, on the arxiv-shaped synthetic graph
with self loops; Adam.

    python tools/arxiv_gat_epoch.py [--composed]      # --composed: the paths this model took before attention dropout
                                                      # and odd head widths (F = 40) were fused (A/B)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--composed", action="store_true")
ap.add_argument("--epochs", type=int, default=20)
args = ap.parse_args()
if args.composed:
    stag_amd.zoo.GAT._padded_width = staticmethod(lambda H, F: F)
    ops.attn_drop_fusable = lambda *a, **k: False
torch.distributions.Distribution.set_default_validate_args(False)
dev = torch.device("cuda:0")
src, dst = synthetic.arxiv_like(seed=1)
n = synthetic.ARXIV_NODES
g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
g = stag_amd.add_self_loop(stag_amd.remove_self_loop(g))
x = torch.randn(n, 128, device=dev)
y = torch.randint(0, 40, (n,), device=dev)
mask = torch.rand(n, device=dev) < 0.54
N = torch.distributions.Normal
SL, Z = stag_amd.layers.StagLayer, stag_amd.zoo
layers = torch.nn.ModuleList([
    SL(Z.GAT(128, 8, num_heads=8, feat_drop=0.6, attn_drop=0.6, activation=torch.nn.functional.elu), q_a=N(1.0, 0.3)),
    SL(Z.GAT(64, 40, num_heads=8, last=True, feat_drop=0.6, attn_drop=0.6,
             activation=lambda t: torch.nn.functional.softmax(t, dim=-1)), q_a=N(1.0, 0.3))])
model = stag_amd.models.StagModel(layers=layers).to(dev)
opt = torch.optim.Adam(model.parameters(), 5e-3)


def epoch():
    model.train()
    opt.zero_grad()
    loss = model.loss(g, x, y, mask=mask)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    epoch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(args.epochs):
    loss = epoch()
e1.record()
torch.cuda.synchronize()
print(f"{'composed paths' if args.composed else 'fused'}: E = {g.number_of_edges()}, epoch "
      f"{(time.perf_counter() - t0) / args.epochs * 1e3:.2f} ms wall, {e0.elapsed_time(e1) / args.epochs:.2f} ms device, "
      f"loss {loss.item():.4f}")
