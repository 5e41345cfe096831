#!/usr/bin/env python
"""30 training steps (forward + backward) of one StagLayer(GCN 128 -> 128, vi=True, relu=True) on the
cfg2 graph — the workload behind profiles/r01/layer_step_kernel_stats.csv:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_layer -o layer --output-format csv -- \
        python3 tools/layer_step.py
"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import stag_amd
from stag_amd import synthetic
dev = torch.device("cuda:0")
src, dst = synthetic.arxiv_like(seed=1); n = synthetic.ARXIV_NODES
g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
D = 128
x = torch.randn(n, D, device=dev); gout = torch.randn(n, D, device=dev)
layer = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(D, D), q_a=torch.distributions.Normal(1.0, 0.5), vi=True, relu=True).to(dev)
xg = x.clone().requires_grad_(True)
for i in range(30):
    layer.zero_grad(set_to_none=True)
    y = layer(g, xg); y.backward(gout)
torch.cuda.synchronize()
