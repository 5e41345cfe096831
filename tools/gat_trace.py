#!/usr/bin/env python
"""Per-stage timeline of the workgroup-cooperative GAT forward at cfg5: a build with
-DSTAG_GAT_DBG=16 writes wall-clock stamps at every stage boundary of every batch into the stats buffer.

    python tools/ab_bench.py build trace="-DSTAG_GAT_DBG=16"      # here (no GPU needed)
    python tools/gat_trace.py                                      # on the GPU box
Prints the mean time of each stage for segment batches and row batches, the span of the launch and how many
workgroups were alive on average."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stag_amd import _lib, ops, synthetic
import stag_amd
_lib._SO = os.path.join(ROOT, "tools", "_bin", "libstag_trace.so")
dev = torch.device("cuda:0")
src, dst = synthetic.arxiv_like(seed=1); n = synthetic.ARXIV_NODES
g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
H, F = 8, 32
el, er, ft = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, F, device=dev)
plan = g.csr.plan(64)
nb = plan["n_blocks"]
# call the raw entry with a big stats buffer used as trace storage
csrv = g.csr
out = torch.empty(n, H, F, device=dev)
trace = torch.zeros(max(nb * 8, n * 2 * H // 2 + 8), dtype=torch.int64, device=dev)
nbytes = _lib.lib().stag_gat_workspace_bytes(plan["n_seg"], H, F)
plan_c, keep = ops._plan_struct(csrv, 64, 1, nbytes, dev)
nz = stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=1, offset=0)
spec = nz.spec(); cs = csrv.struct()
for it in range(3):
    rc = _lib.lib().stag_gat_fwd(C.byref(cs), C.byref(plan_c), _lib.ptr(el), _lib.ptr(er), _lib.ptr(ft), H, F, 0.2,
                                 C.byref(spec), None, None, _lib.ptr(out), _lib.ptr(trace), _lib.stream_of(dev))
    assert rc == 0
torch.cuda.synchronize()
t = trace[:nb * 8].cpu().numpy().reshape(nb, 8).astype(np.float64) * 0.01   # us (100 MHz)
t0 = t[:, 0].min()
d = np.diff(t[:, :7], axis=1)
names = ["units+scan+bar1", "phase1", "bar2", "phase1b", "bar3", "phase2"]
bp = plan["block_ptr"].cpu().numpy()
units = plan["units"].cpu().numpy()
nseg_blocks = int(np.searchsorted(bp, plan["n_seg"]))
for tag, sl in (("segment blocks", slice(0, nseg_blocks)), ("row blocks", slice(nseg_blocks, nb))):
    print(tag, d[sl].shape[0], "blocks; mean us per stage:", {k: round(float(v), 2) for k, v in zip(names, d[sl].mean(0))},
          "total", round(float((t[sl, 6] - t[sl, 0]).mean()), 2))
print("kernel span us", round(float(t[:, 6].max() - t0), 1), "blocks", nb)
# concurrency: average number of blocks alive
ev = np.concatenate([np.stack([t[:, 0], np.ones(nb)], 1), np.stack([t[:, 6], -np.ones(nb)], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0])
print("avg blocks alive", round(float((alive[:-1] * dt).sum() / dt.sum()), 1), "= per CU", round(float((alive[:-1] * dt).sum() / dt.sum() / 256), 2))
