#!/usr/bin/env python
"""Training steps (forward + backward) of one StagLayer(GCN 128 -> 128) on the cfg2 graph, for a kernel trace:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_layer -o layer --output-format csv -- \
        python3 tools/layer_step.py [--mode r1|re|rec|vi_norm|fixed] [--steps 30] [--kl]

  r1       vi=True, relu=True, Normal(1, 0.5) with learned scalars (profiles/r01/layer_step_kernel_stats.csv)
  re       AmortizedDistribution(128, 1): [E, 1] parameters, what scripts/arxiv_rec/gcn/run.py:85 builds
  rec      AmortizedDistribution(128, 128): [E, 128] parameters
  vi_norm  vi=True, norm=True
  fixed    Normal(1, 0.5), not learned
--kl adds the layer's KL term to the loss (stag/layers.py:132-145), as the training scripts do.
The totals of the trace are per `--steps` steps plus 3 warm-up steps.
"""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import stag_amd
from stag_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="r1", choices=["r1", "re", "rec", "vi_norm", "fixed"])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--kl", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda:0")
src, dst = synthetic.arxiv_like(seed=1)
n = synthetic.ARXIV_NODES
g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
D = 128
x = torch.randn(n, D, device=dev)
gout = torch.randn(n, D, device=dev)
N = torch.distributions.Normal
kw = {"r1": dict(q_a=N(1.0, 0.5), vi=True, relu=True),
      "re": dict(q_a=stag_amd.distributions.AmortizedDistribution(D, 1, init_like=N(1.0, 0.3)), vi=True),
      "rec": dict(q_a=stag_amd.distributions.AmortizedDistribution(D, D, init_like=N(1.0, 0.3)), vi=True),
      "vi_norm": dict(q_a=N(1.0, 0.5), vi=True, norm=True),
      "fixed": dict(q_a=N(1.0, 0.5))}[args.mode]
layer = stag_amd.layers.StagLayer(stag_amd.zoo.GCN(D, D), **kw).to(dev)
xg = x.clone().requires_grad_(True)


def step():
    layer.zero_grad(set_to_none=True)
    xg.grad = None
    y = layer(g, xg)
    if args.kl:      # d loss = <gout, dy> + d kl, without an [N, D] product on the way
        torch.autograd.backward([y, layer.kl_divergence()], [gout, torch.ones((), device=dev)])
    else:
        y.backward(gout)


for i in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(args.steps):
    step()
e1.record()
torch.cuda.synchronize()
print(f"mode={args.mode} kl={args.kl}: {e0.elapsed_time(e1) / args.steps * 1e3:.1f} us per step", flush=True)
