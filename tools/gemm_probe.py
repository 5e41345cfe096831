#!/usr/bin/env python
"""Library-GEMM variants for the dense transform around the aggregation (N x 128 x 128, fp32), MI355X:
forward x @ w, dx = g @ w^T (as NT, or NN on a transposed copy of w), dw = x^T g (split-K batched forms).
    python tools/gemm_probe.py [--n 169343] [--k 128] [--m 128]
"""
import argparse
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=169343)
ap.add_argument("--k", type=int, default=128)
ap.add_argument("--m", type=int, default=128)
args = ap.parse_args()
dev = torch.device("cuda:0")
n, K, M = args.n, args.k, args.m
x = torch.randn(n, K, device=dev)
g = torch.randn(n, M, device=dev)
w = torch.randn(K, M, device=dev)
b = torch.randn(M, device=dev)


def ev(fn, k=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


print(f"N={n} K={K} M={M}; 2NKM = {2 * n * K * M / 1e9:.2f} GFLOP; x + y = {(n * K + n * M) * 4 / 1e6:.0f} MB")
print(f"fwd  x @ w                      : {ev(lambda: x @ w):7.1f} us")
print(f"fwd  addmm(b, x, w)             : {ev(lambda: torch.addmm(b, x, w)):7.1f} us")
wt = w.t().contiguous()
print(f"dx   g @ w.t()            (NT)  : {ev(lambda: g @ w.t()):7.1f} us")
print(f"dx   g @ w.t().contiguous() (NN): {ev(lambda: g @ w.t().contiguous()):7.1f} us")
print(f"dx   (w @ g.t()).t()            : {ev(lambda: (w @ g.t()).t()):7.1f} us")
for S in (16, 32, 64, 128, 256, 512):
    n1 = (n // S) * S

    def dw():
        return torch.bmm(x[:n1].view(S, n1 // S, K).transpose(1, 2), g[:n1].view(S, n1 // S, M)).sum(0)
    print(f"dw   split-K bmm S={S:4d} + sum    : {ev(dw):7.1f} us")
    def dw2():
        return torch.bmm(g[:n1].view(S, n1 // S, M).transpose(1, 2), x[:n1].view(S, n1 // S, K)).sum(0).t()
    print(f"dw   split-K bmm (g^T x)^T S={S:4d}: {ev(dw2):7.1f} us")
print(f"dw   x.t() @ g                  : {ev(lambda: x.t() @ g, 10, 3):7.1f} us")
xt = x.t().contiguous()
print(f"dw   xt_contig @ g  (NN, + the transpose {ev(lambda: x.t().contiguous(), 10, 3):.1f} us): {ev(lambda: xt @ g, 10, 3):7.1f} us")
