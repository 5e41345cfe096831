#!/usr/bin/env python
"""What the overlapped form of the partitioned GAT step (`partition._ShardGat`, BASELINE configs[4]) costs and what it
buys, as far as ONE GPU can show it: every shard of an 8-way node-range partition of the arxiv-shaped graph, its exchange
buffers filled by indexing (the collective itself needs 8 GPUs).

  before (round 3): wait for the exchange, then ONE stag_gat_fwd; backward ONE stag_gat_bwd, then the transposed exchange,
                    then the segmented sums of what came back.
  after  (round 4): stag_gat_fwd over the rows whose sources are all local WHILE the exchange is in flight, then over the
                    rest; backward by stages (stag_gat_bwd_stages): row dots, the source pass over the REMOTE buffer rows,
                    [transposed exchange starts], the source pass over this rank's own rows + d er [in flight], one
                    combine launch per table.

Printed per shard, device microseconds (HIP events, 200 launches): the kernels of both forms, and the window the new form
gives the collective to hide in (forward: the local launch; backward: the second source pass + d er)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import _lib, ops, synthetic  # noqa: E402
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
    ap.add_argument("--ranks", default="all")
    ap.add_argument("--seg-len", type=int, default=64)
    ap.add_argument("--lib", default=None, help="A/B: a build variant tools/_bin/libstag_<name>.so (tools/ab_bench.py build)")
    ap.add_argument("--fwd-only", action="store_true", help="the forward launches only")
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--feat", type=int, default=32)
    args = ap.parse_args()
    if args.lib:
        lib = _lib.bind(os.path.join(ROOT, "tools", "_bin", f"libstag_{args.lib}.so"))
        _lib._lib = lib
        _lib.lib = lambda: lib
    dev = torch.device("cuda:0")
    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    H, F, seg = args.heads, args.feat, args.seg_len
    HF = H * F
    ranks = range(args.world) if args.ranks == "all" else [int(r) for r in args.ranks.split(",")]
    gen = torch.Generator().manual_seed(3)
    none = ops._targs_or_c(ops._none_spec())
    print(f"arxiv GAT {H} x {F}, noise [E, {H}] = Normal, node-range partition x{args.world}; us per launch on one MI355X")
    print("rank  rows   edges  halo rows | fwd one  fwd local + remote (window) | bwd one  rowdot  src remote  src own  d er  "
          "(window)  combine ft + el | fwd +%  bwd +%")
    tot = {"f1": 0.0, "f2": 0.0, "b1": 0.0, "b2": 0.0}
    for r in ranks:
        sh = GraphShard(src, dst, n, r, args.world, device=dev, exchange="halo")
        csrv, csrt = sh.csr, sh.csr_t
        nb, nr, ns = sh.n_buf, sh.n_rows, int(sh.send_idx.shape[0])
        el = torch.randn(nb, H, generator=gen).to(dev)
        er = torch.randn(nr, H, generator=gen).to(dev)
        ft = torch.randn(nb, H, F, generator=gen).to(dev)
        G = torch.randn(nr, H, F, generator=gen).to(dev)
        noise = stag_amd.EdgeNoise(sh, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=0x5747A6, offset=11)
        noise.pos_base = sh.pos_base
        spec = noise.spec()
        out = torch.empty(nr, H, F, device=dev)
        stats = torch.empty(nr, 2 * H, device=dev)
        whole = csrv.plan(seg, need=True)
        p_loc, p_rem = sh.plan_split(seg)
        drop0 = ops._gat_drop_struct(None)
        fwd = lambda plan: ops._gat_fwd_into(csrv, plan, el, er, ft, H, F, 0.2, spec, None, drop0, out, stats, dev)
        f_one = timeit(lambda: fwd(whole))
        f_loc = timeit(lambda: fwd(p_loc)) if p_loc["n_units"] else 0.0
        f_rem = timeit(lambda: fwd(p_rem)) if p_rem["n_units"] else 0.0
        fwd(whole)
        if args.fwd_only:
            print(f"{r:4d} {nr:6d} {sh.number_of_edges():7d} {nb - nr:7d} | fwd one {f_one:7.1f}  local {f_loc:6.1f} + remote {f_rem:6.1f}; batches {whole['n_blocks']}", flush=True)
            continue
        b_one = timeit(lambda: ops._gat_bwd_fused(csrv, csrt, el, er, ft, stats, G, out, H, F, 0.2, spec, None, False, seg, dev, None))
        T_ft = torch.zeros((nb + ns, HF), device=dev)
        T_el = torch.zeros((nb + ns, H), device=dev)
        d_er = torch.empty((nr, H), device=dev)
        st = ops._GatBwdStages(csrv, csrt, el, er, ft, stats, G, out, H, F, 0.2, spec, None, None, seg, T_el, d_er, T_ft, dev)
        first, second = sh.plan_split_t(seg)
        t_rd = timeit(st.rowdot)
        t_s1 = timeit(lambda: st.source(first)) if first["n_units"] else 0.0
        t_s2 = timeit(lambda: st.source(second)) if second["n_units"] else 0.0
        t_de = timeit(st.der)
        comb = sh._combined_csr()
        t_cf = timeit(lambda: ops._agg_raw(comb, T_ft, HF, none, _lib.REDUCE_SUM, None, None, seg))
        t_ce = timeit(lambda: ops._agg_raw(comb, T_el, H, none, _lib.REDUCE_SUM, None, None, seg))
        # before: the received rows were summed per table by _segsum_back and added to the own rows
        back_ft, back_el = torch.randn(ns, HF, device=dev), torch.randn(ns, H, device=dev)
        own_ft, own_el = torch.randn(nr, HF, device=dev), torch.randn(nr, H, device=dev)
        t_old = timeit(lambda: (own_ft + sh._segsum_back(back_ft), own_el + sh._segsum_back(back_el))) if ns else 0.0
        f2, b1, b2 = f_loc + f_rem, b_one + t_old, t_rd + t_s1 + t_s2 + t_de + t_cf + t_ce
        tot["f1"] += f_one; tot["f2"] += f2; tot["b1"] += b1; tot["b2"] += b2
        print(f"{r:4d} {nr:6d} {sh.number_of_edges():7d} {nb - nr:7d} | {f_one:7.1f}  {f_loc:6.1f} + {f_rem:6.1f} ({f_loc:6.1f}) | "
              f"{b_one:7.1f} (+ {t_old:5.1f} sums) {t_rd:6.1f} {t_s1:8.1f} {t_s2:8.1f} {t_de:6.1f} ({t_s2 + t_de:6.1f}) "
              f"{t_cf:6.1f} + {t_ce:4.1f} | {100 * (f2 / f_one - 1):+5.1f} {100 * (b2 / b1 - 1):+5.1f}")
        del sh, csrv, csrt, st
    k = len(list(ranks))
    if args.fwd_only:
        return
    print(f"mean over {k} shards: forward {tot['f1'] / k:.1f} -> {tot['f2'] / k:.1f} us of kernels, backward {tot['b1'] / k:.1f} -> "
          f"{tot['b2'] / k:.1f} us; the collective of a step has the local launch (forward) and the own-rows source pass + d er "
          f"(backward) to hide behind")


if __name__ == "__main__":
    main()
