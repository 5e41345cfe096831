"""Distribution evidence at SCALE: what `q_a.expand([E, Dn]).sample()` promises (stag/layers.py:117-127)
— independent draws from the stated law — checked on ONE FULL FIELD of BASELINE configs[1]: E = 1,166,243 edges x 128
channels = 149,279,104 draws per launch, the stream the fused kernels consume (`ops.materialize_noise` = the same
counters, keys and transforms; bit-equal to the in-kernel draw: tests/test_gpu_parity.py).  Tier (iii) of the parity
contract (hard part 2) was so far checked on 320 k oracle draws; here, per kind:
  * moments to 4th order against the law's, each within 5 standard errors of its estimator at n = 1.49e8;
  * Normal: tail counts beyond 3, 4 and 5 sigma against 2 n (1 - Phi(k)) within 5 sqrt(expected) — the 23-bit radius
    caps |z| at 5.65 sigma, so the 5-sigma count is the one that would show a truncated tail;
  * Uniform: chi-square of a 256-bin histogram; Bernoulli: chi-square of the per-channel keep counts;
  * lag-1 correlations across edges (same channel), across channels (inside a Philox block and across its boundary)
    and across offsets (the same position one generator step later), each |r| <= 5 / sqrt(n).
Thresholds are 5-sigma bounds of the estimators: a correct stream passes with probability > 0.9999 per statistic."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = 128
SEED = 0x5747A6


@pytest.fixture(scope="module")
def arxiv(dev):
    import stag_amd
    from stag_amd import synthetic
    src, dst = synthetic.arxiv_like(seed=1)
    g = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), synthetic.ARXIV_NODES, device=dev)
    g.csr.plan(64)
    return g


def _field(g, kind, p0, p1, offset):
    import stag_amd
    from stag_amd import ops
    w = ops.materialize_noise(g, stag_amd.EdgeNoise(g, D, kind, p0, p1, seed=SEED, offset=offset))
    assert w.shape == (g.number_of_edges(), D)
    return w


def _moments(z):
    """(mean, E z^2, E z^3, E z^4) in float64, chunked (no [E, D] float64 temporary)."""
    s = torch.zeros(4, dtype=torch.float64, device=z.device)
    for part in z.split(1 << 16):
        p = part.double()
        p2 = p * p
        s += torch.stack([p.sum(), p2.sum(), (p2 * p).sum(), (p2 * p2).sum()])
    return (s / z.numel()).tolist()


def _corr(a, b):
    """Pearson correlation of two equally shaped fields, float64, chunked."""
    n = a.numel()
    s = torch.zeros(5, dtype=torch.float64, device=a.device)
    for pa, pb in zip(a.split(1 << 16), b.split(1 << 16)):
        x, y = pa.double(), pb.double()
        s += torch.stack([x.sum(), y.sum(), (x * x).sum(), (y * y).sum(), (x * y).sum()])
    sx, sy, sxx, syy, sxy = s.tolist()
    cov = sxy / n - (sx / n) * (sy / n)
    return cov / math.sqrt((sxx / n - (sx / n) ** 2) * (syy / n - (sy / n) ** 2)), n


def _independence(w0, w1, what):
    """lag-1 correlations of one field (edges, channels inside / across Philox blocks) and against the next offset."""
    checks = {
        "edges (e, e+1), same channel": _corr(w0[:-1], w0[1:]),
        "channels (k, k+1) inside a Philox block": _corr(w0[:, 0::4], w0[:, 1::4]),
        "channels (k, k+1) across a block boundary": _corr(w0[:, 3:-1:4], w0[:, 4::4]),
        "channels (k, k+2): the two Box-Muller pairs of a block": _corr(w0[:, 0::4], w0[:, 2::4]),
        "offsets (o, o+1), same position": _corr(w0, w1),
    }
    for name, (r, n) in checks.items():
        assert abs(r) <= 5.0 / math.sqrt(n), f"{what}: lag-1 correlation over {name}: r = {r:.3e} over n = {n}"


def test_normal_field_at_cfg2_scale(arxiv):
    from stag_amd import _lib
    w0 = _field(arxiv, _lib.NOISE_NORMAL, 1.0, 0.5, 0)
    n = w0.numel()
    assert n == 149_279_104
    z = (w0 - 1.0) / 0.5
    m1, m2, m3, m4 = _moments(z)
    se = lambda var: 5.0 * math.sqrt(var / n)                 # 5 standard errors of a sample mean of that variance
    assert abs(m1) <= se(1.0), f"mean {m1:.3e}"
    assert abs(m2 - 1.0) <= se(2.0), f"E z^2 = {m2:.6f}"       # Var(z^2) = 2
    assert abs(m3) <= se(15.0), f"E z^3 = {m3:.3e}"            # Var(z^3) = 15
    assert abs(m4 - 3.0) <= se(96.0), f"E z^4 = {m4:.6f}"      # Var(z^4) = 96
    az = z.abs()
    for k in (3.0, 4.0, 5.0):
        expect = n * math.erfc(k / math.sqrt(2.0))
        got = int((az > k).sum())
        assert abs(got - expect) <= 5.0 * math.sqrt(expect), f"|z| > {k}: {got} draws, expected {expect:.1f}"
    assert float(az.max()) <= 5.66, "the 23-bit radius caps |z| at sqrt(2 * 23 * ln 2) = 5.65"
    del z, az
    w1 = _field(arxiv, _lib.NOISE_NORMAL, 1.0, 0.5, 1)
    _independence(w0, w1, "Normal")


def test_uniform_field_at_cfg2_scale(arxiv):
    from stag_amd import _lib
    lo, hi = 1.0 - 0.5 * math.sqrt(3.0), 1.0 + 0.5 * math.sqrt(3.0)
    w0 = _field(arxiv, _lib.NOISE_UNIFORM, lo, hi, 0)
    n = w0.numel()
    u = (w0 - lo) / (hi - lo)
    assert float(u.min()) >= 0.0 and float(u.max()) <= 1.0
    m1, m2, m3, m4 = _moments(u - 0.5)
    se = lambda var: 5.0 * math.sqrt(var / n)
    assert abs(m1) <= se(1 / 12), f"mean - 1/2 = {m1:.3e}"
    assert abs(m2 - 1 / 12) <= se(1 / 80 - 1 / 144), f"variance {m2:.7f}"         # Var((u-1/2)^2) = 1/80 - 1/144
    assert abs(m3) <= se(1 / 448), f"third central moment {m3:.3e}"              # E (u-1/2)^6 = 1/448
    assert abs(m4 - 1 / 80) <= se(1 / 2304 - 1 / 6400), f"fourth central moment {m4:.7f}"
    hist = torch.histc(u, bins=256, min=0.0, max=1.0).double()
    chi2 = float(((hist - n / 256) ** 2 / (n / 256)).sum())
    assert abs(chi2 - 255.0) <= 5.0 * math.sqrt(2 * 255.0), f"chi-square of 256 bins: {chi2:.1f} (255 d.o.f.)"
    del u
    w1 = _field(arxiv, _lib.NOISE_UNIFORM, lo, hi, 1)
    _independence(w0, w1, "Uniform")


@pytest.mark.parametrize("p", [0.5, 0.9])
def test_bernoulli_field_at_cfg2_scale(arxiv, p):
    from stag_amd import _lib
    w0 = _field(arxiv, _lib.NOISE_BERNOULLI, p, None, 0)
    E = w0.shape[0]
    n = w0.numel()
    assert bool(((w0 == 0) | (w0 == 1)).all())
    keep = w0.double().sum(0)                                   # per channel
    # p as the kernels compare it: fp32(p) against a 23-bit uniform
    p32 = float(np.float32(p))
    chi2 = float((((keep - E * p32) ** 2) / (E * p32 * (1 - p32))).sum())
    assert abs(chi2 - D) <= 5.0 * math.sqrt(2.0 * D), f"chi-square of the per-channel keep counts: {chi2:.1f} ({D} d.o.f.)"
    tot = float(keep.sum())
    assert abs(tot - n * p32) <= 5.0 * math.sqrt(n * p32 * (1 - p32)), f"kept {tot:.0f} of {n}"
    w1 = _field(arxiv, _lib.NOISE_BERNOULLI, p, None, 1)
    _independence(w0, w1, f"Bernoulli({p})")
