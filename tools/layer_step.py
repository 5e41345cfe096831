#!/usr/bin/env python
"""Training steps (forward + backward) of one StagLayer(GCN 128 -> 128) on the cfg2 graph, for a kernel trace:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_layer -o layer --output-format csv -- \
        python3 tools/layer_step.py [--mode r1|re|rec|vi_norm|fixed] [--steps 30] [--kl]

  r1       vi=True, relu=True, Normal(1, 0.5) with learned scalars (profiles/r01/layer_step_kernel_stats.csv)
  re       AmortizedDistribution(128, 1): [E, 1] parameters, what scripts/arxiv_rec/gcn/run.py:85 builds
  rec      AmortizedDistribution(128, 128): [E, 128] parameters
  vi_norm  vi=True, norm=True
  fixed    Normal(1, 0.5), not learned
  gat | sage | gin   the other base layers with the fixed Normal (GAT: 8 heads x 32)
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
ap.add_argument("--mode", default="r1", choices=["r1", "re", "rec", "vi_norm", "fixed", "gat", "sage", "gin"])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--kl", action="store_true")
ap.add_argument("--attn-drop", type=float, default=0.0, help="--mode gat: attention dropout (the reference's GAT scripts: 0.6)")
ap.add_argument("--lib", default=None, help="A/B: a build variant tools/_bin/libstag_<name>.so (tools/ab_bench.py build)")
args = ap.parse_args()
if args.lib:
    from stag_amd import _lib
    _lib._SO = os.path.join(os.getcwd(), "tools", "_bin", f"libstag_{args.lib}.so")

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
      "fixed": dict(q_a=N(1.0, 0.5))}.get(args.mode, dict(q_a=N(1.0, 0.5)))
base = {"gat": lambda: stag_amd.zoo.GAT(D, 32, num_heads=8, attn_drop=args.attn_drop), "sage": lambda: stag_amd.zoo.GraphSAGE(D, D),
        "gin": lambda: stag_amd.zoo.GIN(D, D)}.get(args.mode, lambda: stag_amd.zoo.GCN(D, D))()
layer = stag_amd.layers.StagLayer(base, **kw).to(dev)
xg = x.clone().requires_grad_(True)


def step():
    layer.zero_grad(set_to_none=True)
    xg.grad = None
    y = layer(g, xg)
    if y.dim() == 3:
        y = y.flatten(1)
    if y.shape[1] != gout.shape[1]:
        y = y[:, :gout.shape[1]] if y.shape[1] > gout.shape[1] else torch.nn.functional.pad(y, (0, gout.shape[1] - y.shape[1]))
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
