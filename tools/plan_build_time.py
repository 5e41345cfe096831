#!/usr/bin/env python
"""What a freshly batched minibatch graph pays before its first launch: Graph + both CSR views + both launch plans
(scripts/ppi_mle and scripts/molhiv_mle build one per training step), with the XCD-aware unit order (its locality
reduction, the stripe sort and fill) built at once and without — the reason "auto" waits until a view has been launched
XCD_AFTER_LAUNCHES times.

    python tools/plan_build_time.py            (GPU box)
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd  # noqa: E402
from stag_amd import synthetic  # noqa: E402

G = importlib.import_module("stag_amd.graph")


def build(src, dst, n, need):
    g = stag_amd.Graph(src, dst, n)
    for view in (g.csr, g.csr_t):
        view.plan(64, need=need)
    return g


def main():
    dev = torch.device("cuda:0")
    cases = []
    s, d, sizes = synthetic.ppi_like()
    cases.append(("PPI batch (24 graphs, E = 818,716)", s, d, int(sizes.sum())))
    s, d, sizes = synthetic.ppi_like(n_graphs=2, n_nodes=4800, n_edges=68000, seed=9)
    cases.append(("PPI minibatch (2 graphs, E = 68,000)", s, d, int(sizes.sum())))
    s, d, sizes = synthetic.molecules_like()
    cases.append(("molecule batch (4096 graphs, E = 218 k)", s, d, int(sizes.sum())))
    for name, s, d, n in cases:
        src, dst = torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)
        for need in (False, True):
            res = {}
            for mode in ("0", "1"):
                G.XCD_ORDER = mode
                ts = []
                for r in range(12):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    g = build(src, dst, n, need)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                res[mode] = float(np.median(ts[2:]))
                has = g.csr._plans.get(64) is not None and g.csr._plans[64].get("xcd") is not None
            t0 = time.perf_counter()
            loc = g.csr.stripe_locality()
            t_loc = (time.perf_counter() - t0) * 1e3
            print(f"{name}: Graph + 2 CSR views + plans (need={need}): {res['0']:.2f} ms without, {res['1']:.2f} ms with the "
                  f"XCD-aware order built at once (STAG_XCD_ORDER=1; built: {has}); the locality reduction 'auto' decides by "
                  f"{t_loc:.2f} ms per view (= {loc:.2f}); 'auto' pays both only after {G.XCD_AFTER_LAUNCHES} launches", flush=True)


if __name__ == "__main__":
    main()
