#!/usr/bin/env python
"""Host time of one ops.aggregate call (launch-bound graphs: Cora-sized), through the dispatcher ops and through
ctypes:   python tools/host_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stag_amd
from stag_amd import _lib, ops

dev = torch.device("cuda:0")
g = stag_amd.rand_graph(2708, 13264, device=dev)
g.csr.plan(64)
x = torch.randn(2708, 16, device=dev)
for tag, env in (("torch.ops.stag.agg_fwd", "1"), ("ctypes", None)):
    if env: os.environ["STAG_TORCH_OPS"] = env
    else: os.environ.pop("STAG_TORCH_OPS", None)
    with torch.no_grad():
        for i in range(200):
            ops.aggregate(g, x, stag_amd.EdgeNoise(g, 16, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5000
        for i in range(n):
            ops.aggregate(g, x, stag_amd.EdgeNoise(g, 16, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=i))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{tag:26s} host {1e6 * (t1 - t0) / n:6.2f} us per call (submit), {1e6 * (t2 - t0) / n:6.2f} us incl. drain")
