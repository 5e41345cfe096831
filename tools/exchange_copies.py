#!/usr/bin/env python
"""Device time of the copies AROUND the collective of one partitioned layer step, before and after round 3, on
one GPU: shard `--rank` of an 8-way node-range partition of the arxiv-shaped graph (the collective itself needs
8 GPUs; what is timed here is what this rank's GPU does besides it and the kernels).

  before: buf = empty; buf[:n].copy_(x); send = x.index_select(0, send_idx)          (+ GAT: cat([ft | el]) before,
          two column slices made contiguous after)
  after : the local rows already live in the persistent buffer; send rows by stag_gather_rows into a persistent
          send buffer; ft and el travel as two tables (no cat, no slices)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import synthetic  # noqa: E402
from stag_amd.partition import GraphShard  # noqa: E402


def timeit(fn, steps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    sh = GraphShard(src, dst, n, args.rank, args.world, device=dev, exchange="halo")
    nr, ns, nb = sh.n_rows, int(sh.send_idx.shape[0]), sh.n_buf
    print(f"shard {args.rank}/{args.world}: {nr} rows, sends {ns} rows, buffer {nb} rows")
    D, H, F = 128, 8, 32
    x = torch.randn(nr, D, device=dev)
    ft, el = torch.randn(nr, H * F, device=dev), torch.randn(nr, H, device=dev)

    def before_agg():
        buf = torch.empty((nb, D), device=dev)
        buf[:nr].copy_(x)
        return buf, x.index_select(0, sh.send_idx)

    xin = sh.local_rows(D)
    xin.copy_(x)
    sendbuf = sh._send_buffer((D,), torch.float32, dev)

    def after_agg():
        return sh._fill_send(xin, sendbuf)

    def before_gat():
        packed = torch.cat([ft, el], 1)
        buf = torch.empty((nb, H * F + H), device=dev)
        buf[:nr].copy_(packed)
        send = packed.index_select(0, sh.send_idx)
        return buf[:, :H * F].contiguous(), buf[:, H * F:].contiguous(), send

    fin, ein = sh.local_rows(H * F), sh.local_rows(H)
    fin.copy_(ft)
    ein.copy_(el)
    sf, se = sh._send_buffer((H * F,), torch.float32, dev), sh._send_buffer((H,), torch.float32, dev)

    def after_gat():
        sh._fill_send(fin, sf)
        return sh._fill_send(ein, se)

    print(f"aggregation step, D={D}:  copies before {timeit(before_agg):7.1f} us   after {timeit(after_agg):7.1f} us")
    print(f"GAT step, H={H} F={F}:      copies before {timeit(before_gat):7.1f} us   after {timeit(after_gat):7.1f} us")


if __name__ == "__main__":
    main()
