#!/usr/bin/env python
"""A/B timing of libstag_hip.so build variants in ONE process, interleaved rounds
(cdna_hip_programming.md §5.4 rule 24).

  python tools/ab_bench.py build  name1="-DSTAG_BLK_RNG=4" name2="-DSTAG_PIPELINE=0" ...
  python tools/ab_bench.py run [--noise normal] [--rounds 7] [--steps 30]     (on the GPU box)

Variants are built into tools/_bin/libstag_<name>.so (they travel with gpurun)."""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "_bin")
sys.path.insert(0, ROOT)


def build(specs):
    os.makedirs(BIN, exist_ok=True)
    for spec in specs:
        name, _, flags = spec.partition("=")
        out = os.path.join(BIN, f"libstag_{name}.so")
        subprocess.run(["make", "-C", os.path.join(ROOT, "stag_amd", "csrc"), "-j", "8",
                        f"EXTRA={flags}", f"OBJDIR=_obj_{name}", f"OUT={out}", out], check=True,   # the library only
                       stdout=subprocess.DEVNULL)
        print("built", out, flags)


def run(argv):
    import argparse
    import numpy as np
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--noise", default="normal")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--seg-len", type=int, default=64)
    ap.add_argument("--mc", type=int, default=0, help="time ops.aggregate_mc with this many Monte-Carlo samples per call")
    ap.add_argument("--per-edge", action="store_true", help="Normal noise with [E, 1] parameters (an AmortizedDistribution's heads)")
    ap.add_argument("--graph", default="arxiv", choices=["arxiv", "ppi"],
                    help="ppi: the 24-graph PPI-sized batch (BASELINE configs[2]) with the XCD-aware order on, mean reducer")
    args = ap.parse_args(argv)
    import stag_amd
    from stag_amd import _lib, ops, synthetic
    import bench
    dev = torch.device("cuda:0")
    red = "sum"
    if args.graph == "ppi":
        import importlib
        importlib.import_module("stag_amd.graph").XCD_ORDER = "1"
        src, dst, sizes = synthetic.ppi_like()
        n, red = int(sizes.sum()), "mean"
        g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)
    else:
        src, dst = synthetic.arxiv_like(seed=1)
        n = synthetic.ARXIV_NODES
        g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
    x = torch.randn(n, args.feat, device=dev)
    libs = sorted(glob.glob(os.path.join(BIN, "libstag_*.so")))
    base = _lib.lib()
    handles = {"current": base}
    for path in libs:
        l = _lib.bind(path)          # every prototype declared (and the ABI version checked)
        handles[os.path.basename(path)[len("libstag_"):-3]] = l
    times = {k: [] for k in handles}
    ref = None
    if args.per_edge:
        E = g.number_of_edges()
        loc = torch.rand(E, 1, device=dev) + 0.5
        scale = torch.rand(E, 1, device=dev) * 0.5 + 0.1
        one = lambda i: ops.aggregate(g, x, stag_amd.EdgeNoise(g, args.feat, _lib.NOISE_NORMAL, loc, scale, seed=5, offset=i), seg_len=args.seg_len)
    elif args.mc:
        one = lambda i: ops.aggregate_mc(g, x, bench.make_noise(stag_amd, g, args.feat, args.noise, i), args.mc, seg_len=args.seg_len)
    else:
        one = lambda i: ops.aggregate(g, x, bench.make_noise(stag_amd, g, args.feat, args.noise, i), reduce=red, seg_len=args.seg_len)
    for r in range(args.rounds + 1):
        for name, l in handles.items():
            _lib._lib = l
            _lib.lib = lambda l=l: l
            for i in range(3):
                out = one(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.steps):
                out = one(i)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1) / args.steps * 1e3)
            if ref is None:
                ref = out.clone()
            elif not torch.allclose(out, ref, rtol=1e-4, atol=1e-4):
                print(f"WARNING: variant {name} output differs from the first variant")
            elif r == 0 and not torch.equal(out, ref):
                print(f"note: variant {name} is not bit-identical to the first variant "
                      f"(max abs diff {(out - ref).abs().max().item():.3g})")
    for name, t in sorted(times.items(), key=lambda kv: np.median(kv[1])):
        print(f"{name:28s} median {np.median(t):8.2f} us   min {np.min(t):8.2f} us   ({args.noise}, {len(t)} rounds x {args.steps} steps)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])
