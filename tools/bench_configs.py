#!/usr/bin/env python
"""Timings of the other BASELINE configs' hot ops on one GPU (not the bench line).
  cfg3     PPI-like GraphSAGE (24 graphs batched): mean aggregation, D=256
  cfg3_l1  the same batch at D=50 (PPI's input width: the first layer)
  cfg3_gat (not in the default list) GAT forward, 4 heads x 256, on the same batch (scripts/ppi_mle/gat)
  cfg4     molhiv-like GIN: 4096 small graphs batched, sum, D=128, + mean readout
  cfg4_l1  the same batch at D=9 (molhiv's atom features: the first layer)
  cfg5     arxiv GAT: H=8, F=32, noise [E,8], forward
  cfg5_train   the same, forward + backward (gat_bwd_edge_kernel and the d ft / d el / d er passes)

    python tools/bench_configs.py [--only cfg5,cfg5_train] [--steps 50] [--json out.json]

One JSON object per config on stdout (and collected in --json): device time per step (HIP events),
algorithmic bytes and the resulting fraction of the 8 TB/s HBM roofline.
tools/profile_configs.py runs this script under rocprofv3 for profiles/<round>/.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import _lib, ops, synthetic  # noqa: E402

ALL = ["cfg5", "cfg5_train", "cfg3", "cfg3_l1", "cfg4", "cfg4_l1"]


SETTLE_MS = 300.0     # untimed launches before the warm-up, as bench.py does (--settle-ms): a 50-step run started cold
                      # reads a VALU-bound kernel 15-20 % slow (cfg3 with noise: 147 against 123 us) — clocks, not code


def timeit(fn, steps, warmup):
    if SETTLE_MS > 0:
        import time
        t0, i = time.perf_counter(), 0
        while (time.perf_counter() - t0) * 1e3 < SETTLE_MS:
            for _ in range(20):
                fn(i)
                i += 1
            torch.cuda.synchronize()
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(warmup + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3   # us


def report(name, what, t_us, E, b_alg, extra=None):
    line = {"config": name, "what": what, "us_per_step": round(t_us, 2), "edges": E,
            "edges_per_s": E / t_us * 1e6, "algorithmic_bytes": b_alg,
            "achieved_GBs": b_alg / t_us / 1e3, "frac_of_8TBs": b_alg / t_us / 1e3 / 8000.0}
    if extra:
        line.update(extra)
    print(json.dumps(line), flush=True)
    return line


def main():
    global SETTLE_MS
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(ALL))
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--noise", default="normal", choices=["normal", "none"])
    ap.add_argument("--json", default=None)
    ap.add_argument("--settle-ms", type=float, default=SETTLE_MS, help="untimed launches before each timed loop (0: start cold)")
    ap.add_argument("--lib", default=None, help="A/B: a build variant tools/_bin/libstag_<name>.so (tools/ab_bench.py build)")
    ap.add_argument("--blk", default=None, help="A/B: batch budget of the cooperative GAT kernels, 'edges,units' (at most the library's STAG_BLOCK_EDGES, STAG_BLOCK_UNITS)")
    ap.add_argument("--gat-old-bwd", action="store_true",
                    help="cfg5_train: the composed backward (stag_gat_bwd_edge + three stag_agg_fwd calls) for A/B")
    ap.add_argument("--gat-two-pass", action="store_true",
                    help="cfg5_train: stag_gat_bwd_two_pass (edge pass + source pass: two gathers) for A/B")
    args = ap.parse_args()
    only = [s for s in args.only.split(",") if s]
    SETTLE_MS = args.settle_ms
    if args.gat_two_pass:
        ops._GAT_BWD_ONE_GATHER = False
    if args.blk:
        _lib.BLOCK_EDGES, _lib.BLOCK_UNITS = (int(v) for v in args.blk.split(","))
    if args.gat_old_bwd:
        ops._GAT_BWD_FUSED = False
    if args.lib:
        _lib._SO = os.path.join(ROOT, "tools", "_bin", f"libstag_{args.lib}.so")
    dev = torch.device("cuda:0")
    out = []

    def mk(g, dn):
        if args.noise == "none":
            return lambda i: None
        return lambda i: stag_amd.EdgeNoise(g, dn, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i)

    if any(c.startswith("cfg5") for c in only):
        src, dst = synthetic.arxiv_like(seed=1)
        n, E = synthetic.ARXIV_NODES, len(src)
        g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
        g.csr.plan(64)
        H, F = 8, 32
        el, er = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
        ft = torch.randn(n, H, F, device=dev)
        b_alg = 4 * (n + 1) + 4 * E + 8 * n * H + 2 * 4 * n * H * F      # 363.0 MB
        noise = mk(g, H)
        if "cfg5" in only:
            with torch.no_grad():
                t = timeit(lambda i: ops.gat_aggregate(g, el, er, ft, 0.2, noise(i)), args.steps, args.warmup)
            out.append(report("cfg5", f"GAT forward H=8 F=32 noise[E,8]={args.noise}", t, E, b_alg))
        if "cfg5_train" in only:
            g.csr_t.plan(64)
            elg, erg, ftg = (t_.clone().requires_grad_(True) for t_ in (el, er, ft))
            gout = torch.randn(n, H, F, device=dev)

            def step(i):
                elg.grad = erg.grad = ftg.grad = None
                ops.gat_aggregate(g, elg, erg, ftg, 0.2, noise(i)).backward(gout)
            t = timeit(step, args.steps, args.warmup)
            # forward bytes + backward: read g, ft, out once, write d ft once, d el / d er, de[E,H] written + read
            out.append(report("cfg5_train", f"GAT forward+backward (ops.gat_aggregate) noise={args.noise}",
                              t, E, b_alg, {"note": "frac is forward-only algorithmic bytes over fwd+bwd time"}))
        del g

    if any(c.startswith("cfg3") for c in only):
        s3, d3, sizes3 = synthetic.ppi_like()
        n3, E3 = int(sizes3.sum()), len(s3)
        g3 = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), n3,
                            batch_num_nodes=torch.from_numpy(sizes3).to(dev), device=dev)
        g3.csr.plan(64)
        for name, D in (("cfg3", 256), ("cfg3_l1", 50)):
            if name not in only:
                continue
            x3 = torch.randn(n3, D, device=dev)
            noise = mk(g3, D)
            with torch.no_grad():
                t = timeit(lambda i: ops.aggregate(g3, x3, noise(i), reduce="mean"), args.steps, args.warmup)
            b = 4 * (n3 + 1) + 4 * E3 + 8 * n3 * D
            out.append(report(name, f"SAGE mean D={D}, 24 PPI-like graphs batched (N={n3}, E={E3}) noise={args.noise}",
                              t, E3, b))
        if "cfg3_gat" in only:      # scripts/ppi_mle/gat/run.py:21-58: 4 heads x 256 on the same batch (not a BASELINE config)
            H, F = 4, 256
            g3.csr.plan(64, need=True)
            el, er = torch.randn(n3, H, device=dev), torch.randn(n3, H, device=dev)
            ft = torch.randn(n3, H, F, device=dev)
            noise = mk(g3, H)
            with torch.no_grad():
                t = timeit(lambda i: ops.gat_aggregate(g3, el, er, ft, 0.2, noise(i)), args.steps, args.warmup)
            b = 4 * (n3 + 1) + 4 * E3 + 8 * n3 * H + 2 * 4 * n3 * H * F
            out.append(report("cfg3_gat", f"GAT forward H={H} F={F}, 24 PPI-like graphs batched (N={n3}, E={E3}) noise[E,{H}]={args.noise}",
                              t, E3, b))
        del g3

    if any(c.startswith("cfg4") for c in only):
        s4, d4, sizes = synthetic.molecules_like(4096)
        n4, E4 = int(sizes.sum()), len(s4)
        g4 = stag_amd.Graph(torch.from_numpy(s4), torch.from_numpy(d4), n4,
                            batch_num_nodes=torch.from_numpy(sizes).to(dev), device=dev)
        g4.csr.plan(64)
        offs = torch.zeros(len(sizes) + 1, dtype=torch.int32, device=dev)
        offs[1:] = torch.cumsum(torch.from_numpy(sizes).to(dev), 0).to(torch.int32)
        for name, D in (("cfg4", 128), ("cfg4_l1", 9)):
            if name not in only:
                continue
            x4 = torch.randn(n4, D, device=dev)
            noise = mk(g4, D)
            with torch.no_grad():
                t = timeit(lambda i: ops.aggregate(g4, x4, noise(i)), args.steps, args.warmup)
                tr = timeit(lambda i: ops.segment_reduce(x4, offs, "mean"), args.steps, args.warmup)
            b = 4 * (n4 + 1) + 4 * E4 + 8 * n4 * D
            out.append(report(name, f"GIN sum D={D}, 4096 molecules batched (N={n4}, E={E4}) noise={args.noise}",
                              t, E4, b, {"mean_readout_us": round(tr, 2)}))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
