"""Sizes past the 32-bit / 24-bit fast paths of the kernels (288 GB of HBM is there to be used): more than 2^24
edges (edge ids and per-edge rows take the 64-bit address form, `wide` bit 1) and more than 2^24 source rows (the
feature gather takes it, `wide` bit 0).  The oracle is too slow here, so the check is torch's own index_add of the
same products on the device (the explicit-weight kernel is oracle-checked at small sizes) plus fused == explicit."""
import numpy as np
import pytest
import torch

from util import TOL, assert_close

pytestmark = pytest.mark.gpu


def _ref_aggregate(n, src, dst, x, w):
    out = torch.zeros(n, x.shape[1], dtype=torch.float32, device=x.device)
    step = 4_000_000
    for i in range(0, src.shape[0], step):
        s, d = src[i:i + step], dst[i:i + step]
        out.index_add_(0, d, (w[i:i + step] if w is not None else 1.0) * x[s])
    return out


@pytest.mark.parametrize("case", ["edges_over_2^24", "rows_over_2^24", "x_and_w_over_4GB"])
def test_sizes_past_the_narrow_address_forms(dev, case):
    import stag_amd
    from stag_amd import _lib, ops
    gen = torch.Generator(device=dev).manual_seed(5)
    if case == "edges_over_2^24":
        n, E, D = 2_000_000, 20_000_000, 32
    elif case == "rows_over_2^24":
        n, E, D = 17_500_000, 20_000_000, 16
    else:                                              # x [N, D] and w [E, D] of more than 4 GB each: no 32-bit byte offsets
        n, E, D = 17_500_000, 20_000_000, 64
        assert n * D * 4 > (1 << 32) and E * D * 4 > (1 << 32)
    assert E > (1 << 24) and (case == "edges_over_2^24" or n > (1 << 24))
    src = torch.randint(0, n, (E,), device=dev, generator=gen)
    dst = torch.randint(0, n, (E,), device=dev, generator=gen)
    dst[:50_000] = 7                                   # a hub row: segments and the long-row combine at this scale
    g = stag_amd.Graph(src, dst, n, device=dev)
    x = torch.randn(n, D, device=dev, generator=gen)
    # plain sum and explicit [E, D] weights against torch's index_add of the same products
    out = ops.aggregate(g, x, None)
    ref = _ref_aggregate(n, src, dst, x, None)
    rows = torch.randint(8, n, (200_000,), device=dev, generator=gen)      # (not the hub: index_add's plain fp32
    assert_close(out[rows], ref[rows].cpu().numpy(), tol=5 * TOL, what=f"{case}: sum")   # atomics lose digits there)
    hub_in = (dst == 7).nonzero().squeeze(1)
    assert_close(out[7:8], x[src[hub_in]].double().sum(0, keepdim=True).cpu().numpy(), what=f"{case}: hub row, float64 reference")
    w = torch.rand(E, D, device=dev, generator=gen) + 0.5
    out_w = ops.aggregate(g, x, w)
    ref_w = _ref_aggregate(n, src, dst, x, w)
    assert_close(out_w[rows], ref_w[rows].cpu().numpy(), tol=5 * TOL, what=f"{case}: explicit weights")
    assert_close(out_w[7:8], (w[hub_in].double() * x[src[hub_in]].double()).sum(0, keepdim=True).cpu().numpy(),
                 what=f"{case}: hub row with explicit weights")
    del ref, ref_w, out_w
    # fused Normal noise == its own materialised weights through the explicit kernel; [E, 1] parameters too
    nz = stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, 1.0, 0.5, seed=3, offset=1)
    fused = ops.aggregate(g, x, nz)
    mat = ops.aggregate(g, x, nz.materialize())
    assert_close(fused[rows], mat[rows].cpu().numpy(), tol=5 * TOL, what=f"{case}: fused vs materialised noise")
    del mat
    loc = torch.rand(E, 1, device=dev, generator=gen) + 0.5
    sc = torch.rand(E, 1, device=dev, generator=gen) * 0.5 + 0.1
    nz1 = stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, loc, sc, seed=3, offset=2)
    fused1 = ops.aggregate(g, x, nz1)
    mat1 = ops.aggregate(g, x, nz1.materialize())
    assert_close(fused1[rows], mat1[rows].cpu().numpy(), tol=5 * TOL, what=f"{case}: [E,1] parameters")
    # backward: <A x, y> == <x, A^T y> with the same noise (the source-major twin at this scale)
    xr = x.clone().requires_grad_(True)
    y = torch.randn(n, D, device=dev, generator=gen)
    o = ops.aggregate(g, xr, stag_amd.EdgeNoise(g, D, _lib.NOISE_NORMAL, 1.0, 0.5, seed=3, offset=1))
    o.backward(y)
    lhs, rhs = float((o.detach().double() * y.double()).sum()), float((x.double() * xr.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * (abs(lhs) + abs(rhs)), (lhs, rhs)


def test_gat_past_the_narrow_address_forms(dev):
    """The cooperative GAT kernels with ft [N, H, F] of more than 4 GB (no buffer descriptor for the row gather) and
    more than 2^24 edges: forward and the one-gather backward against the composed statement of the same layer
    (torch ops over [E, H] + the aggregation kernels), with attention dropout against its materialised mask."""
    import stag_amd
    from stag_amd import _lib, ops
    gen = torch.Generator(device=dev).manual_seed(9)
    n, E, H, F = 4_400_000, 18_000_000, 8, 32
    assert n * H * F * 4 > (1 << 32) and E > (1 << 24)
    src = torch.randint(0, n, (E,), device=dev, generator=gen)
    dst = torch.randint(0, n, (E,), device=dev, generator=gen)
    dst[:30_000] = 11
    g = stag_amd.Graph(src, dst, n, device=dev)
    el, er = (torch.randn(n, H, device=dev, generator=gen) for _ in range(2))
    ft = torch.randn(n, H, F, device=dev, generator=gen)
    G = torch.randn(n, H, F, device=dev, generator=gen)
    rows = torch.randint(0, n, (100_000,), device=dev, generator=gen)
    rows[0] = 11
    noise = lambda: stag_amd.EdgeNoise(g, H, _lib.NOISE_NORMAL, 1.0, 0.5, seed=4, offset=6)
    t = [a.clone().requires_grad_(True) for a in (el, er, ft)]
    out = ops.gat_aggregate(g, *t, 0.2, noise())
    out.backward(G)
    t2 = [a.clone().requires_grad_(True) for a in (el, er, ft)]
    out2 = ops.gat_aggregate(g, *t2, 0.2, noise(), attn_fn=lambda a_: a_)          # the composed path
    out2.backward(G)
    assert_close(out[rows], out2[rows].detach().cpu().numpy(), tol=5 * TOL, what="forward")
    for a_, b_, nm in zip(t, t2, ("d el", "d er", "d ft")):
        sc = max(1.0, float(b_.grad[rows].abs().max()))
        assert_close(a_.grad[rows] / sc, (b_.grad[rows] / sc).cpu().numpy(), tol=5 * TOL, what=nm)
    del t2, out2
    keep_prob = float(np.float32(1.0 - 0.6))      # as ops._gat_drop_struct hands it over: 1 - p in double, then fp32
    keep = stag_amd.EdgeNoise(g, H, _lib.NOISE_BERNOULLI, keep_prob, seed=21, offset=3).materialize()
    with torch.no_grad():
        fused = ops.gat_aggregate(g, el, er, ft, 0.2, noise(), attn_drop=(0.6, 21, 3))
        comp = ops.gat_aggregate(g, el, er, ft, 0.2, noise(), attn_fn=lambda a_: a_ * keep / keep_prob)
    assert_close(fused[rows], comp[rows].cpu().numpy(), tol=5 * TOL, what="attention dropout")
