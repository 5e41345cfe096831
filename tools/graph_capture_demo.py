#!/usr/bin/env python
"""A whole training step (forward, NLL + KL, backward, Adam) of a 2-layer vi=True GCN StagModel
captured in ONE hipGraph and replayed, on a Cora-sized graph (BASELINE configs[0]: N=2708,
E=10556, 1433 -> 16 -> 7) where the step is launch-bound.  The Philox offsets of the captured
kernels are frozen; the generator's device epoch (advanced inside the graph) gives every replay
fresh noise.  Prints eager vs replay time per step and checks that the loss keeps falling.

    python tools/graph_capture_demo.py            (on the GPU box)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd.random import NoiseGenerator  # noqa: E402


def build(dev, gen):
    N = torch.distributions.Normal
    l1 = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(1433, 16, activation=torch.relu), q_a=N(1.0, 0.4),
                                   vi=True, generator=gen)
    l2 = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(16, 7, activation=lambda t: t.softmax(-1)), q_a=N(1.0, 0.4),
                                   vi=True, generator=gen)
    model = stag_amd.models.StagModel([l1, l2], kl_scaling=1e-3)
    params = [p for l in (l1, l2) for p in l.parameters()]
    for l in (l1, l2):
        l.to(dev)
    return model, params


def main():
    torch.distributions.Distribution.set_default_validate_args(False)   # argument checks read back from the device
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n, E = 2708, 10556
    src, dst = rng.integers(0, n, E), rng.integers(0, n, E)
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    x = (torch.rand(n, 1433, device=dev) < 0.01).float()
    y = torch.randint(0, 7, (n,), device=dev)

    def timed(fn, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e6

    # ---- eager
    gen = NoiseGenerator(seed=1)
    model, params = build(dev, gen)
    opt = torch.optim.Adam(params, lr=1e-2, capturable=True)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        loss = model.loss(g, x, y)
        loss.backward()
        opt.step()
        return loss
    for _ in range(5):
        eager_step()
    t_eager = timed(eager_step, 50)

    # ---- captured
    gen2 = NoiseGenerator(seed=1)
    model2, params2 = build(dev, gen2)
    opt2 = torch.optim.Adam(params2, lr=1e-2, capturable=True)
    gen2.enable_device_epoch(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                      # warm-up on a side stream (torch's capture recipe)
        for _ in range(3):
            opt2.zero_grad(set_to_none=True)
            model2.loss(g, x, y).backward()
            opt2.step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    opt2.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        mark = gen2.offset
        loss = model2.loss(g, x, y)
        loss.backward()
        opt2.step()
        gen2.advance_epoch(gen2.offset - mark)
    losses = []
    for _ in range(60):
        graph.replay()
        losses.append(loss.item())
    t_graph = timed(graph.replay, 200)
    print(f"eager step  : {t_eager:8.1f} us")
    print(f"graph replay: {t_graph:8.1f} us   ({t_eager / t_graph:.1f}x)")
    print(f"loss over 60 replays: {losses[0]:.4f} -> {losses[-1]:.4f}; distinct values: {len(set(losses))}")
    assert losses[-1] < losses[0] and len(set(losses)) > 50


if __name__ == "__main__":
    main()
