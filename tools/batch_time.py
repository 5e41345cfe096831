#!/usr/bin/env python
"""What `graph.batch` costs a training step that builds a fresh batch (scripts/ppi_mle/run.py:12-14, 70-77): the union of
24 PPI-sized graphs, then its two CSR views, then its two launch plans — with the parts' arrays laid end to end in one
launch (stag_concat_jobs) and with the union's COO sorted afresh (BATCH_CONCAT_MAX_GRAPHS = 0).

    python tools/batch_time.py            (GPU box)
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
G.BATCH_CACHE_SIZE = 0        # building is what is timed here, not looking up


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    s, d, sizes = synthetic.ppi_like()
    off = np.concatenate([[0], np.cumsum(sizes)])
    gid = np.searchsorted(off, s, side="right") - 1
    parts = [stag_amd.Graph(torch.from_numpy(s[gid == i] - off[i]).to(dev), torch.from_numpy(d[gid == i] - off[i]).to(dev),
                            int(sizes[i]), device=dev) for i in range(len(sizes))]
    for limit, what in ((64, "the parts' arrays laid end to end"), (0, "the union's COO sorted afresh")):
        G.BATCH_CONCAT_MAX_GRAPHS = limit
        tb = timed(lambda: stag_amd.batch(parts))

        def views():
            g = stag_amd.batch(parts)
            g.csr, g.csr_t

        def plans():
            g = stag_amd.batch(parts)
            g.csr.plan(64), g.csr_t.plan(64)
        print(f"{what:36s}: batch() {tb:.2f} ms; with both CSR views {timed(views):.2f} ms; with both plans {timed(plans):.2f} ms",
              flush=True)


if __name__ == "__main__":
    main()
