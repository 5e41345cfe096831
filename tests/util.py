"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

TOL = 1e-5   # north_star: aggregated features within 1e-5 fp32 => |a-b| <= TOL * (1 + |b|)


def scaled_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.max(np.abs(got - ref) / (1.0 + np.abs(ref)))) if ref.size else 0.0


def assert_close(got, ref, tol=TOL, what=""):
    if torch.is_tensor(got):
        got = got.detach().cpu().numpy()
    err = scaled_err(got, ref)
    if os.environ.get("STAG_PRINT_ERR"):      # how close each comparison runs to its bar (tolerance audits)
        print(f"ERR {what}: {err:.3e} (tol {tol:.1e})")
    assert err <= tol, f"{what}: scaled error {err:.3e} > {tol:.1e}"


def assert_close_cond(got, ref, abs_terms, tol=TOL, what=""):
    """assert_close with the conditioning of each sum taken into account:
        |got - ref| <= tol (1 + |ref|) + 2^-24 sum_e |term_e|
    The kernels — like the reference's fp32 dataflow, stag/zoo/gcn.py:94-96: `w * x'[src]` materialised, then summed —
    round every term w (g s) to fp32 once; the oracle forms it in double.  2^-24 sum |terms| is the most those
    roundings can add up to, and it only matters where a sum cancels: the case that found it is a 3-node graph with
    2243 edges, whose source rows sum TWO distinct destination rows ~380 times each with like signs — +476 and
    -476.5 at one channel, result -0.529, where the one rounding of g[v] s[v] is shared by all 380 terms.  There the
    plain bar asks for 1e-5 of a number 900 times smaller than what was summed (observed 2.9e-5; 4.6e-8 of
    sum |terms|), whatever the order of summation."""
    if torch.is_tensor(got):
        got = got.detach().cpu().numpy()
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if not ref.size:
        return
    bar = tol * (1.0 + np.abs(ref)) + 2.0 ** -24 * np.asarray(abs_terms, np.float64)
    over = np.abs(got - ref) / bar
    if os.environ.get("STAG_PRINT_ERR"):
        print(f"ERR {what}: {scaled_err(got, ref):.3e} plain, {over.max():.3f} of the conditioned bar")
    assert over.max() <= 1.0, (f"{what}: |got - ref| = {np.abs(got - ref).flat[over.argmax()]:.3e} is "
                               f"{over.max():.2f} x the bar {bar.flat[over.argmax()]:.3e} (scaled error {scaled_err(got, ref):.3e})")


def oracle_graph(O, g, transposed=False):
    c = g.csr_t if transposed else g.csr
    return O.CsrGraph(c.indptr.cpu().numpy(), c.indices.cpu().numpy(), c.eid.cpu().numpy(),
                      nidx=None if c.nidx is None else c.nidx.cpu().numpy(), n_src=c.n_src)


def random_graph(n, e, seed, hub=None, device=None):
    """Random multigraph with an optional hub destination carrying `hub` extra edges and
    node n-1 left without in-edges."""
    import stag_amd
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = rng.integers(0, max(n - 1, 1), e)
    if hub:
        src = np.concatenate([src, rng.integers(0, n, hub)])
        dst = np.concatenate([dst, np.full(hub, min(3, n - 1))])
    return stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=device)


_HW_TABLES = {}


class hw_normals:
    """`with hw_normals(oracle, dev):` — inside, the oracle draws its normals from the DEVICE's tables of
    the three hardware functions (stag_normal_tables): the same weights, bit for bit, as the kernels draw.
    What a comparison then shows is arithmetic on identical weights, so the 1e-5 bar needs no allowance for
    the length of a row.  The tables themselves are pinned against libm, exhaustively, by
    test_gpu_parity.py::test_normal_tables_exhaustive."""

    def __init__(self, oracle, dev):
        self.oracle, self.dev = oracle, dev

    def __enter__(self):
        if "t" not in _HW_TABLES:
            from stag_amd import ops
            _HW_TABLES["t"] = ops.normal_tables(self.dev).cpu().numpy()
        self.oracle.set_normal_tables(_HW_TABLES["t"])
        return _HW_TABLES["t"]

    def __exit__(self, *exc):
        self.oracle.set_normal_tables(None)
        return False


def assert_gat_grads_vs_oracle(oracle, og, el, er, ft, G, spec, got, keep=None, keep_prob=1.0, got_dw=None,
                               what="", dev=None, tol=TOL):
    """The GAT backward (stag_gat_bwd / stag_gat_bwd_two_pass / the composed calls) against its CPU twin
    oracle.gat_bwd (stag_gat_bwd_cpu: float64, what autograd returns for stag/zoo/gat.py:109-126).
    got = (d el, d er, d ft) device tensors; got_dw: d w [E, H] for explicit weights.  Gradients are compared
    relative to their own scale; d el / d er — sums of d s[e,h] = a' <G, ft[u]> - a <G, out[v]>, two dot products
    that can cancel exactly (a row with one kept in-edge) — also relative to the largest term a' <G, ft[u]>."""
    el, er, ft, G = (np.asarray(a, np.float32) for a in (el, er, ft, G))
    ctx = hw_normals(oracle, dev) if dev is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        d_el, d_er, d_ft, dw = oracle.gat_bwd(og, el, er, ft, G, 0.2, spec, keep=keep, keep_prob=keep_prob,
                                              want_dw=got_dw is not None)
        _, attn = oracle.gat_fwd(og, el, er, ft, 0.2, spec, want_attn=True, keep=keep, keep_prob=keep_prob)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    term = 0.0
    if og.n_edges:
        eid = og.eid if og.eid is not None else np.arange(og.n_edges)
        u, v = og.indices, og.dst_of_pos
        dots = np.einsum("phf,phf->ph", G[v].astype(np.float64), ft[u].astype(np.float64))
        term = float(np.abs(attn[eid].astype(np.float64) * dots).max())
    for g_, r_, nm in zip(got, (d_el, d_er, d_ft), ("d el", "d er", "d ft")):
        sc = max(1.0, float(np.abs(r_).max()) if r_.size else 0.0, term if nm != "d ft" else 0.0)
        assert_close(g_.detach().cpu().numpy() / sc, r_.astype(np.float64) / sc, tol=tol, what=f"{what} {nm} vs oracle")
    if got_dw is not None:
        sc = max(1.0, float(np.abs(dw).max()) if dw.size else 0.0)
        assert_close(got_dw.detach().cpu().numpy() / sc, dw.astype(np.float64) / sc, tol=tol, what=f"{what} d w vs oracle")
