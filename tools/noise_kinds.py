#!/usr/bin/env python
"""The headline launch (arxiv-shaped CSR, D = 128) under every noise kind, interleaved rounds in ONE process:
what the draw costs on top of the gather, and what a cheaper draw could buy at most.

    normal     Philox4x32-10 block + 4 Box-Muller normals (8 transcendentals)
    uniform    the same Philox block + 4 FMAs, NO transcendental: the floor of any table- or polynomial-based
               normal transform (an inverse-CDF table in LDS cannot be cheaper than no transform at all)
    bernoulli  the same block + 4 compares (+ the in-norm weight sums)
    none       no draw: the gather alone

    python tools/noise_kinds.py [--feat 128] [--rounds 7] [--steps 40]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    import stag_amd
    from stag_amd import ops, synthetic
    import bench
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    x = torch.randn(n, args.feat, device=dev)
    kinds = ["normal", "uniform", "bernoulli", "none"]
    times = {k: [] for k in kinds}
    for r in range(args.rounds + 1):
        for k in kinds:
            for i in range(5):
                ops.aggregate(g, x, bench.make_noise(stag_amd, g, args.feat, k, i))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.steps):
                ops.aggregate(g, x, bench.make_noise(stag_amd, g, args.feat, k, i))
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[k].append(e0.elapsed_time(e1) / args.steps * 1e3)
    for k in kinds:
        print(f"{k:10s} median {np.median(times[k]):8.2f} us   min {np.min(times[k]):8.2f} us   (D={args.feat}, {args.rounds} rounds x {args.steps} steps)")


if __name__ == "__main__":
    main()
