#!/usr/bin/env python
"""Where does a launch's time go?  Per-unit timestamps from a -DSTAG_TRACE build of the
aggregation kernel (tools/_bin/libstag_trace.so; build: python tools/ab_bench.py build
trace="-DSTAG_TRACE"), summarised by unit kind and length.

  python tools/trace_units.py [--feat 16] [--noise none]            (on the GPU box)
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--feat", type=int, default=16)
    ap.add_argument("--noise", default="none")
    ap.add_argument("--seg-len", type=int, default=64)
    args = ap.parse_args()
    import stag_amd
    import bench
    from stag_amd import _lib, ops, synthetic
    base = _lib.lib()
    l = C.CDLL(os.path.join(ROOT, "tools", "_bin", "libstag_trace.so"))
    for fn in ("stag_agg_fwd", "stag_plan_workspace_bytes"):
        getattr(l, fn).argtypes = getattr(base, fn).argtypes
        getattr(l, fn).restype = getattr(base, fn).restype
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    D = max(args.feat, 8)          # the trace needs 32 B per unit in an [n, D] fp32 buffer
    x = torch.randn(n, D, device=dev)
    plan = g.csr.plan(args.seg_len)
    units = plan["units"].cpu().numpy().reshape(-1, 4)
    ops._agg_raw(g.csr, x, D, _lib.NoiseSpec(), 0, None, None, args.seg_len)   # plan + counters via the stock library
    _lib._lib = l
    _lib.lib = lambda: l
    nz = bench.make_noise(stag_amd, g, D, args.noise, 0)
    spec = nz.spec() if nz is not None else _lib.NoiseSpec()
    for _ in range(5):
        out, ns = ops._agg_raw(g.csr, x, D, spec, 0, None, None, args.seg_len, want_norm_scale=True)
    torch.cuda.synchronize()
    nu = len(units)
    assert ns.numel() * 4 >= nu * 32, "buffer too small for the trace"
    t = ns.view(torch.int64).reshape(-1)[:nu * 4].cpu().numpy().reshape(nu, 4).astype(np.float64)
    t0 = t[:, 0].min()
    us = lambda a: (a - t0) / 100.0       # 100 MHz -> microseconds
    start, loop_end = us(t[:, 0]), us(t[:, 1])
    end = us(np.where(t[:, 3] > 0, t[:, 3], np.where(t[:, 2] > 0, t[:, 2], t[:, 1])))
    ln, slot = units[:, 2], units[:, 3]
    print(f"D={D} noise={args.noise}: {nu} units, launch span {end.max():.1f} us (first start -> last end)")
    print(f"  last unit START at {start.max():.1f} us; median start {np.median(start):.1f} us")
    seg = slot >= 0
    for name, m in (("segments", seg), ("rows 33..64", ~seg & (ln > 32)), ("rows 9..32", ~seg & (ln > 8) & (ln <= 32)),
                    ("rows 1..8", ~seg & (ln <= 8) & (ln > 0)), ("rows 0", ~seg & (ln == 0))):
        if m.sum() == 0:
            continue
        d = (loop_end - start)[m]
        print(f"  {name:12s} n={m.sum():7d}  start [{start[m].min():6.1f}, {np.median(start[m]):6.1f}, {start[m].max():6.1f}] us"
              f"   loop time med {np.median(d):6.2f} max {d.max():6.2f} us   end max {end[m].max():6.1f} us")
    if seg.any():
        tick = us(t[:, 2])[seg]
        print(f"  segments: loop end max {loop_end[seg].max():.1f} us, ticket taken max {tick.max():.1f} us, "
              f"combine end max {end[seg].max():.1f} us")
        last = seg & (t[:, 3] > 0)
        if last.any():
            d = (us(t[:, 3]) - us(t[:, 2]))[last]
            print(f"  combines: n={last.sum()}  duration med {np.median(d):.2f} max {d.max():.2f} us")
    # edge throughput over time: a unit's edges spread evenly over its loop time
    nb = int(end.max() // 5) + 1
    thr = np.zeros(nb)
    for b in range(nb):
        lo, hi = 5.0 * b, 5.0 * (b + 1)
        ov = np.clip(np.minimum(loop_end, hi) - np.maximum(start, lo), 0, None)
        thr += 0  # (kept for clarity)
        thr[b] = (ln * ov / np.maximum(loop_end - start, 1e-3)).sum()
    print("  edges processed per 5 us bin (k):", " ".join(f"{t / 1e3:.0f}" for t in thr))
    # timeline: units finished per 5 us
    hist, edges = np.histogram(end, bins=np.arange(0, end.max() + 5, 5))
    print("  units finishing per 5 us bin:", " ".join(str(h) for h in hist))
    hist, _ = np.histogram(start, bins=np.arange(0, end.max() + 5, 5))
    print("  units starting  per 5 us bin:", " ".join(str(h) for h in hist))


if __name__ == "__main__":
    main()
