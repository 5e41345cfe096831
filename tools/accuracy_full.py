#!/usr/bin/env python
"""Worst scaled error |got - ref| / (1 + |ref|) of the fused aggregation against the CPU oracle at the
BASELINE cfg2 size, by row length, for the built library and every variant under tools/_bin/
(tools/ab_bench.py build ...):   python tools/accuracy_full.py [--noise normal] [--offsets 3]"""
import argparse
import ctypes as C
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--noise", default="normal")
    ap.add_argument("--offsets", type=int, default=3)
    ap.add_argument("--split", action="store_true", help="separate the draws' share from the arithmetic's")
    args = ap.parse_args()
    import stag_amd
    from stag_amd import _lib, ops, synthetic
    from oracle import oracle as O
    from util import oracle_graph
    import bench
    O.build()
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n, D = synthetic.ARXIV_NODES, 128
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    x = torch.randn(n, D, generator=torch.Generator().manual_seed(0))
    xd = x.to(dev)
    og = oracle_graph(O, g)
    deg = g.in_degrees().cpu().numpy()
    base = _lib.lib()
    handles = {"current": base}
    for path in sorted(glob.glob(os.path.join(ROOT, "tools", "_bin", "libstag_*.so"))):
        l = C.CDLL(path)
        for fn in ("stag_agg_fwd", "stag_plan_workspace_bytes"):
            getattr(l, fn).argtypes = getattr(base, fn).argtypes
            getattr(l, fn).restype = getattr(base, fn).restype
        handles[os.path.basename(path)[len("libstag_"):-3]] = l
    bands = [(0, 64), (65, 256), (257, 1024), (1025, 1 << 30)]
    if args.split:
        # where the difference comes from: (a) the draws (device transcendentals vs fp64 rounded once),
        # (b) the arithmetic alone: the device's own weights handed to the oracle as explicit weights
        nz = bench.make_noise(stag_amd, g, D, "normal", 0)
        w_dev = nz.materialize().cpu().numpy()
        spec = O.make_spec("normal", 1.0, 0.5, seed=0x5747A6, offset=0, Dn=D, n_edges=len(src))
        w_or = O.noise_materialize(og, spec, D)
        dw = np.abs(w_dev.astype(np.float64) - w_or)
        print(f"draws: max |w_dev - w_oracle| {dw.max():.3e} rms {np.sqrt((dw ** 2).mean()):.3e} "
              f"identical {float((dw == 0).mean()) * 100:.1f} %", flush=True)
        got = ops.aggregate(g, xd, nz).cpu().numpy().astype(np.float64)
        ref_w = O.agg_fwd(og, x.numpy(), O.make_spec("explicit", w_dev))
        ref = O.agg_fwd(og, x.numpy(), spec)
        for tag, r in (("fused vs oracle(own draws)", ref), ("fused vs oracle(device draws): arithmetic only", ref_w)):
            err = np.abs(got - r) / (1.0 + np.abs(r))
            print(tag, f"worst {err.max():.3e}", " | ".join(
                f"deg {lo}-{hi if hi < 1 << 29 else 'max'}: max {err[(deg >= lo) & (deg <= hi)].max():.2e}" for lo, hi in bands), flush=True)
        return
    for off in range(args.offsets):
        kind = args.noise
        spec = {"normal": O.make_spec("normal", 1.0, 0.5, seed=0x5747A6, offset=off, Dn=D, n_edges=len(src)),
                "uniform": O.make_spec("uniform", 1.0 - 0.5 * 3 ** 0.5, 1.0 + 0.5 * 3 ** 0.5, seed=0x5747A6, offset=off, Dn=D, n_edges=len(src)),
                "bernoulli": O.make_spec("bernoulli", 0.5, in_norm=True, seed=0x5747A6, offset=off, Dn=D, n_edges=len(src)),
                "none": O.make_spec("none")}[kind]
        ref = O.agg_fwd(og, x.numpy(), spec)
        for name, l in handles.items():
            _lib._lib = l
            _lib.lib = lambda l=l: l
            got = ops.aggregate(g, xd, bench.make_noise(stag_amd, g, D, kind, off)).cpu().numpy().astype(np.float64)
            err = np.abs(got - ref) / (1.0 + np.abs(ref))
            parts = []
            for lo, hi in bands:
                m = (deg >= lo) & (deg <= hi)
                parts.append(f"deg {lo}-{hi if hi < 1 << 29 else 'max'} ({int(m.sum())} rows): max {err[m].max():.2e} rms {np.sqrt((err[m] ** 2).mean()):.2e}")
            print(f"offset {off} {name:10s} worst {err.max():.3e} | " + " | ".join(parts), flush=True)


if __name__ == "__main__":
    main()
