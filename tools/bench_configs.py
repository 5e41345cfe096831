#!/usr/bin/env python
"""Timings of the other BASELINE configs' hot ops on one GPU (not the bench line).
  cfg3  PPI-like GraphSAGE: mean aggregation, D=256
  cfg4  molhiv-like GIN: 4096 small graphs batched, D=128, + mean readout
  cfg5  arxiv GAT: H=8, F=32, noise [E,8]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import _lib, ops, synthetic  # noqa: E402


def timeit(fn, steps=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n, E = synthetic.ARXIV_NODES, len(src)
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    g.csr.plan(64)
    # cfg5 GAT
    H, F = 8, 32
    el, er = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    ft = torch.randn(n, H, F, device=dev)
    mk = lambda i: stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i)
    t = timeit(lambda i=0: ops.gat_aggregate(g, el, er, ft, 0.2, mk(i)))
    b_alg = 4 * (n + 1) + 4 * E + 8 * n * H + 2 * 4 * n * H * F
    print(f"cfg5 GAT H=8 F=32 noise[E,8]: {t:8.1f} us  {E / t / 1e3:6.2f} Gedges/s  alg {b_alg / t / 1e3:7.1f} GB/s ({b_alg / t / 1e3 / 80:.1f} % of 8 TB/s)")
    t = timeit(lambda i=0: ops.gat_aggregate(g, el, er, ft, 0.2, None))
    print(f"cfg5 GAT no noise           : {t:8.1f} us")
    # cfg3 SAGE mean D=256 on a PPI-sized graph
    n3, E3 = 56944, 818716
    s3, d3 = synthetic.arxiv_like(n_nodes=n3, n_edges=E3, max_in_degree=700, n_hubs=50, sigma=0.9, seed=3)
    g3 = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), n3, device=dev)
    x3 = torch.randn(n3, 256, device=dev)
    mk3 = lambda i: stag_amd.EdgeNoise(g3, 256, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i)
    t = timeit(lambda i=0: ops.aggregate(g3, x3, mk3(i), reduce="mean"))
    b = 4 * (n3 + 1) + 4 * E3 + 8 * n3 * 256
    print(f"cfg3 SAGE mean D=256        : {t:8.1f} us  {E3 / t / 1e3:6.2f} Gedges/s  alg {b / t / 1e3:7.1f} GB/s ({b / t / 1e3 / 80:.1f} %)")
    x50 = torch.randn(n3, 50, device=dev)          # PPI's input width: the first layer aggregates at D=50
    mk50 = lambda i: stag_amd.EdgeNoise(g3, 50, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i)
    t = timeit(lambda i=0: ops.aggregate(g3, x50, mk50(i), reduce="mean"))
    b = 4 * (n3 + 1) + 4 * E3 + 8 * n3 * 50
    print(f"cfg3 SAGE mean D=50 (layer 1): {t:8.1f} us  {E3 / t / 1e3:6.2f} Gedges/s  alg {b / t / 1e3:7.1f} GB/s ({b / t / 1e3 / 80:.1f} %)")
    # cfg4 molecules D=128 + readout
    s4, d4, sizes = synthetic.molecules_like(4096)
    n4, E4 = int(sizes.sum()), len(s4)
    g4 = stag_amd.Graph(torch.from_numpy(s4), torch.from_numpy(d4), n4,
                        batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)
    x4 = torch.randn(n4, 128, device=dev)
    mk4 = lambda i: stag_amd.EdgeNoise(g4, 128, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i)
    t = timeit(lambda i=0: ops.aggregate(g4, x4, mk4(i)))
    b = 4 * (n4 + 1) + 4 * E4 + 8 * n4 * 128
    print(f"cfg4 GIN sum D=128 (N={n4}, E={E4}): {t:8.1f} us  {E4 / t / 1e3:6.2f} Gedges/s  alg {b / t / 1e3:7.1f} GB/s ({b / t / 1e3 / 80:.1f} %)")
    offs = torch.zeros(len(sizes) + 1, dtype=torch.int32, device=dev)
    offs[1:] = torch.cumsum(torch.from_numpy(sizes).to(dev), 0).to(torch.int32)
    t = timeit(lambda i=0: ops.segment_reduce(x4, offs, "mean"))
    print(f"cfg4 mean readout [4096,128]: {t:8.1f} us")


if __name__ == "__main__":
    main()
