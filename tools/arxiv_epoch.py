#!/usr/bin/env python
"""One training epoch (= one full-graph step) of BASELINE configs[1]: the 3-layer GCN of
scripts/arxiv_mle/gcn/run.py (hidden 128, 40 classes, BatchNorm + ReLU + Dropout between the layers,
softmax head, Adam) on the arxiv-shaped synthetic graph after the script's preprocessing (self loops
removed and re-added, reverse edges added: E = 2,671,172).

    python tools/arxiv_epoch.py [--distribution Bernoulli|Normal|Uniform] [--n-samples-training S] [--model GCN|GraphSAGE]

With S > 1 (the sweeps run `--n_samples_training 4`: scripts/arxiv_mle/graph_sage/meta_run.sh:29) the epoch is timed
twice: the Monte-Carlo loop with the first layer's S samples batched (stag_agg_fwd_mc) and the sequential loop.
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import synthetic  # noqa: E402


def build(dev, distribution, std=0.3, hidden=128, depth=3, model="GCN", vi=False):
    if distribution == "Normal":
        q_a, norm = torch.distributions.Normal(1.0, std, validate_args=False), False
    elif distribution == "Uniform":
        hr = std * math.sqrt(3.0)
        q_a, norm = torch.distributions.Uniform(1.0 - hr, 1.0 + hr, validate_args=False), False
    else:
        q_a, norm = torch.distributions.Bernoulli(probs=0.5 * (1.0 + math.sqrt(1 - 4.0 * std ** 2))), True
    SL, FO, Z = stag_amd.layers.StagLayer, stag_amd.layers.FeatOnlyLayer, stag_amd.zoo
    mid = lambda: FO(torch.nn.Sequential(torch.nn.BatchNorm1d(hidden), torch.nn.ReLU(), torch.nn.Dropout(0.5)))
    conv = Z.GCN if model == "GCN" else Z.GraphSAGE       # scripts/arxiv_mle/gcn/run.py:60-66: --model
    if vi:                                                # a learned q(A) on every layer (scripts/citation_r1/gcn/run.py:31-46)
        SL = lambda base, q_a, norm, _SL=SL: _SL(base, q_a=q_a, norm=norm, vi=True, relu=True)
    layers = torch.nn.ModuleList([SL(conv(128, hidden), q_a=q_a, norm=norm), mid()])
    for _ in range(depth - 2):
        layers += [SL(conv(hidden, hidden), q_a=q_a, norm=norm), mid()]
    layers.append(SL(conv(hidden, 40, activation=lambda t: torch.nn.functional.softmax(t, dim=-1)), q_a=q_a, norm=norm))
    return stag_amd.models.StagModel(layers=layers).to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--distribution", default="Bernoulli")
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--n-samples-training", type=int, default=1)
    ap.add_argument("--model", default="GCN", choices=["GCN", "GraphSAGE"])
    ap.add_argument("--vi", action="store_true", help="learn q(A) (Normal/Uniform): the KL term joins the loss")
    args = ap.parse_args()
    torch.distributions.Distribution.set_default_validate_args(False)
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    g = stag_amd.add_reverse_edges(stag_amd.add_self_loop(stag_amd.remove_self_loop(g)))
    x = torch.randn(n, 128, device=dev)
    y = torch.randint(0, 40, (n,), device=dev)
    mask = torch.rand(n, device=dev) < 0.54
    model = build(dev, args.distribution, model=args.model, vi=args.vi)
    opt = torch.optim.Adam(model.parameters(), 1e-2)

    S = args.n_samples_training

    def epoch():
        model.train()
        opt.zero_grad()
        loss = model.loss(g, x, y, mask=mask, n_samples=S)
        loss.backward()
        opt.step()
        return loss

    def timed(tag):
        for _ in range(5):
            epoch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.epochs):
            loss = epoch()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.epochs
        print(f"{args.model} {args.distribution}{' vi' if args.vi else ''} n_samples_training={S} [{tag}]: E = {g.number_of_edges()}, epoch {wall * 1e3:.2f} ms wall, "
              f"{e0.elapsed_time(e1) / args.epochs:.2f} ms device, loss {loss.item():.4f}; "
              f"{3 * S * g.number_of_edges() / wall / 1e9:.2f} G edge-aggregations/s forward (3 layers x {S} samples)")

    if S > 1:
        model._mc_batching_off = True
        timed("sequential loop")
        model._mc_batching_off = False
        timed("first layer batched")
    else:
        timed("one sample")


if __name__ == "__main__":
    main()
