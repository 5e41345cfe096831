#!/usr/bin/env python
"""The kernels of the partitioned aggregation step (`partition._ShardAggregate`, BASELINE configs[1] cut the way
north_star names) shard by shard on ONE GPU: every shard of a `--world`-way node-range partition of the arxiv-shaped graph,
its exchange buffer filled by indexing (the collective itself needs the GPUs).  tools/gat_shard_stages.py is the same for
the GAT step.

  forward : the whole shard in one launch | the rows with only local sources (the window the exchange hides in) + the rest
  backward: the transposed aggregation in one launch | the remote buffer rows first (their gradient has to travel) +
            this rank's own rows (the window of the transposed exchange), then the one combine launch

Device microseconds per launch (HIP events, 300 launches), with the Normal draw | without a draw."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import _lib, ops, synthetic  # noqa: E402
from stag_amd.partition import GraphShard  # noqa: E402


def timeit(fn, steps=300):
    for _ in range(30):
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
    ap.add_argument("--ranks", default="all")
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--seg-len", type=int, default=64)
    ap.add_argument("--lib", default=None, help="A/B: a build variant tools/_bin/libstag_<name>.so (tools/ab_bench.py build)")
    args = ap.parse_args()
    if args.lib:
        lib = _lib.bind(os.path.join(ROOT, "tools", "_bin", f"libstag_{args.lib}.so"))
        _lib._lib = lib
        _lib.lib = lambda: lib
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    D, seg = args.feat, args.seg_len
    ranks = range(args.world) if args.ranks == "all" else [int(r) for r in args.ranks.split(",")]
    none = ops._targs_or_c(ops._none_spec())
    print(f"arxiv aggregation D = {D}, node-range partition x{args.world}; us per launch on one MI355X, Normal draw | no draw")
    print("rank  rows   edges  halo rows | fwd one        local (window)   remote      | bwd one        remote rows    own rows (window)  combine")
    for r in ranks:
        sh = GraphShard(src, dst, n, r, args.world, device=dev)
        csrv, csrt = sh.csr, sh.csr_t
        nb, nr, ns = sh.n_buf, sh.n_rows, int(sh.send_idx.shape[0])
        buf = torch.randn(nb, D, device=dev)
        g = torch.randn(nr, D, device=dev)
        out = torch.empty(nr, D, device=dev)
        T = torch.zeros(nb + ns, D, device=dev)
        whole, whole_t = csrv.plan(seg, need=True), csrt.plan(seg, need=True)
        p_loc, p_rem = sh.plan_split(seg)
        p_first, p_second = sh.plan_split_t(seg)
        comb = sh._combined_csr()
        cells = []
        for drawn in (True, False):
            if drawn:
                nz = stag_amd.EdgeNoise(sh, D, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=3)
                nz.pos_base = sh.pos_base
                spec = ops._targs_or_c(ops._noise_spec(nz))
                spec_t = ops._targs_or_c(ops._noise_spec(nz, in_norm=0))
            else:
                spec = spec_t = none
            fwd = lambda plan: ops._agg_raw(csrv, buf, D, spec, _lib.REDUCE_SUM, None, None, seg, out=out, plan_t=plan)
            bwd = lambda plan: ops._agg_raw(csrt, g, D, spec_t, _lib.REDUCE_SUM, None, None, seg, out=T[:nb], plan_t=plan)
            cells.append((timeit(lambda: fwd(whole)), timeit(lambda: fwd(p_loc)) if p_loc["n_units"] else 0.0,
                          timeit(lambda: fwd(p_rem)) if p_rem["n_units"] else 0.0, timeit(lambda: bwd(whole_t)),
                          timeit(lambda: bwd(p_first)) if p_first["n_units"] else 0.0,
                          timeit(lambda: bwd(p_second)) if p_second["n_units"] else 0.0))
        t_c = timeit(lambda: ops._agg_raw(comb, T, D, none, _lib.REDUCE_SUM, None, None, seg))
        a, b = cells
        print(f"{r:4d} {nr:6d} {sh.number_of_edges():7d} {nb - nr:7d} | {a[0]:5.1f} | {b[0]:5.1f}  {a[1]:5.1f} | {b[1]:5.1f}  {a[2]:5.1f} | {b[2]:5.1f} | "
              f"{a[3]:5.1f} | {b[3]:5.1f}  {a[4]:5.1f} | {b[4]:5.1f}  {a[5]:5.1f} | {b[5]:5.1f}    {t_c:5.1f}", flush=True)
        del sh, csrv, csrt


if __name__ == "__main__":
    main()
