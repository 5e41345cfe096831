"""ctypes front-end of oracle/_build/libstag_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY — see oracle/stag_oracle.c for what is restated and the
reference lines (yuanqing-wang/stag) each function follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("STAG_ORACLE_SO") or os.path.join(_HERE, "_build", "libstag_oracle.so")

NOISE_NONE, NOISE_EXPLICIT, NOISE_NORMAL, NOISE_UNIFORM, NOISE_BERNOULLI = range(5)
PARAM_SCALAR, PARAM_PER_CHANNEL, PARAM_PER_EDGE1, PARAM_PER_EDGE = range(4)
REDUCE_SUM, REDUCE_MEAN = 0, 1
KIND = {"none": 0, "explicit": 1, "normal": 2, "uniform": 3, "bernoulli": 4}

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


class Csr(C.Structure):
    _fields_ = [("n_dst", C.c_int32), ("n_src", C.c_int32), ("n_edges", C.c_int64),
                ("indptr", _i32p), ("indices", _i32p), ("eid", _i32p), ("nidx", _i32p)]


class NoiseSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("param_mode", C.c_int32),
                ("p0", _f32p), ("p1", _f32p),
                ("p0_scalar", C.c_float), ("p1_scalar", C.c_float),
                ("relu", C.c_int32), ("in_norm", C.c_int32), ("deriv", C.c_int32),
                ("group", C.c_int32), ("seed", C.c_uint64), ("offset", C.c_uint64), ("pos_base", C.c_int64),
                ("chunk_base", C.c_int32), ("p1_log", C.c_int32), ("epoch", C.c_void_p)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("stag_oracle.c", "stag_oracle.h")):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.stag_philox_raw_cpu.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64,
                                             C.c_int32, _u32p]
        _lib.stag_noise_materialize_cpu.argtypes = [C.POINTER(Csr), C.POINTER(NoiseSpec),
                                                    C.c_int32, _f32p, C.c_int64]
        _lib.stag_agg_fwd_cpu.argtypes = [C.POINTER(Csr), _f32p, C.c_int64, C.c_int32,
                                          C.POINTER(NoiseSpec), C.c_int32, _f32p, _f32p,
                                          _f32p, C.c_int64, _f32p]
        _lib.stag_agg_ref_dataflow_cpu.argtypes = [C.POINTER(Csr), _i32p, _i32p, _f32p,
                                                   C.c_int64, C.c_int32, C.POINTER(NoiseSpec),
                                                   _f32p, _f32p, _f32p, C.c_int64]
        _lib.stag_agg_bwd_w_cpu.argtypes = [C.POINTER(Csr), _f32p, C.c_int64, _f32p, C.c_int64,
                                            C.c_int32, _f32p, C.POINTER(NoiseSpec), C.c_int32,
                                            _f32p, C.c_int64]
        _lib.stag_csr_build_cpu.argtypes = [_i32p, _i32p, C.c_int32, C.c_int32, C.c_int64,
                                            _i32p, _i32p, _i32p, _i32p, _i32p]
        _lib.stag_segment_reduce_cpu.argtypes = [_f32p, C.c_int64, C.c_int32, _i32p, C.c_int32,
                                                 C.c_int32, _f32p, C.c_int64]
        _lib.stag_gat_fwd_cpu.argtypes = [C.POINTER(Csr), _f32p, _f32p, _f32p, C.c_int32,
                                          C.c_int32, C.c_float, C.POINTER(NoiseSpec), _f32p, _f32p]
        _lib.stag_gat_fwd_drop_cpu.argtypes = [C.POINTER(Csr), _f32p, _f32p, _f32p, C.c_int32, C.c_int32, C.c_float,
                                               C.POINTER(NoiseSpec), _f32p, C.c_float, _f32p, _f32p]
        _lib.stag_philox4x32_10_cpu.argtypes = [_u32p, _u32p, _u32p]
        _lib.stag_gat_bwd_cpu.argtypes = [C.POINTER(Csr), _f32p, _f32p, _f32p, _f32p, C.c_int32, C.c_int32, C.c_float,
                                          C.POINTER(NoiseSpec), _f32p, C.c_float, _f32p, _f32p, _f32p, _f32p]
    return _lib


def set_threads(n):
    """Number of OpenMP threads the oracle uses (libgomp's omp_set_num_threads)."""
    gomp = C.CDLL("libgomp.so.1")
    gomp.omp_set_num_threads(int(n))


_normal_tables = None


def set_normal_tables(tables):
    """tables: None (default: normals are the fp64 expression rounded once), or a [3, 2^23] float32 array
    (rad, cos, sin) produced by the DEVICE (stag_normal_tables): the oracle then redraws the device's
    normals bit for bit, z = rad[m_a] * cos|sin[m_b].  The array is kept alive here."""
    global _normal_tables
    if tables is None:
        _normal_tables = None
        _check(lib().stag_set_normal_tables_cpu(None, None, None), "set_normal_tables")
        return
    t = np.ascontiguousarray(tables, dtype=np.float32)
    if t.shape != (3, 1 << 23):
        raise ValueError("normal tables must be [3, 2^23]")
    _normal_tables = t
    _check(lib().stag_set_normal_tables_cpu(_p(t[0], _f32p), _p(t[1], _f32p), _p(t[2], _f32p)), "set_normal_tables")


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed: rc={rc}")


class CsrGraph:
    """Host CSR bundle; keeps the numpy arrays alive behind the ctypes struct."""

    def __init__(self, indptr, indices, eid=None, nidx=None, n_src=None):
        self.indptr = _i32(indptr)
        self.indices = _i32(indices)
        self.eid = _i32(eid)
        self.nidx = _i32(nidx)
        self.n_dst = len(self.indptr) - 1
        self.n_src = int(n_src if n_src is not None else
                         (self.indices.max() + 1 if len(self.indices) else 0))
        self.n_edges = len(self.indices)
        self.c = Csr(self.n_dst, self.n_src, self.n_edges, _p(self.indptr, _i32p),
                     _p(self.indices, _i32p), _p(self.eid, _i32p), _p(self.nidx, _i32p))

    @property
    def dst_of_pos(self):
        return np.repeat(np.arange(self.n_dst, dtype=np.int32), np.diff(self.indptr))

    def transpose(self):
        """src-major CSR of the same edges (stable in original edge id, like csr_build on the
        swapped COO) with nidx = forward CSR position of each edge (for the backward pass)."""
        E = self.n_edges
        eid = self.eid if self.eid is not None else np.arange(E, dtype=np.int32)
        pos_of_eid = np.empty(E, dtype=np.int32)
        pos_of_eid[eid] = np.arange(E, dtype=np.int32)
        src_coo = np.empty(E, dtype=np.int32)
        dst_coo = np.empty(E, dtype=np.int32)
        src_coo[eid] = self.indices
        dst_coo[eid] = self.dst_of_pos
        indptr_t, indices_t, eid_t, _, _ = csr_build(dst_coo, src_coo, self.n_dst, self.n_src)
        return CsrGraph(indptr_t, indices_t, eid=eid_t, nidx=pos_of_eid[eid_t], n_src=self.n_dst)


def csr_build(src, dst, n_src, n_dst):
    src, dst = _i32(src), _i32(dst)
    E = len(src)
    indptr = np.zeros(n_dst + 1, np.int32)
    indices = np.zeros(E, np.int32)
    eid = np.zeros(E, np.int32)
    in_deg = np.zeros(n_dst, np.int32)
    out_deg = np.zeros(n_src, np.int32)
    _check(lib().stag_csr_build_cpu(_p(src, _i32p), _p(dst, _i32p), n_src, n_dst, E,
                                    _p(indptr, _i32p), _p(indices, _i32p), _p(eid, _i32p),
                                    _p(in_deg, _i32p), _p(out_deg, _i32p)), "csr_build")
    return indptr, indices, eid, in_deg, out_deg


def make_spec(kind="none", p0=None, p1=None, relu=False, in_norm=False, seed=0, offset=0,
              pos_base=0, Dn=None, n_edges=None, deriv=0, chunk_base=0, p1_log=False):
    """Build a NoiseSpec; p0/p1 may be python floats or arrays ([Dn], [E,1], [E,Dn])."""
    k = KIND[kind] if isinstance(kind, str) else int(kind)
    keep = []
    s = NoiseSpec()
    s.kind, s.relu, s.in_norm, s.deriv = k, int(relu), int(in_norm), int(deriv)
    s.seed, s.offset, s.pos_base, s.chunk_base = int(seed), int(offset), int(pos_base), int(chunk_base)
    s.p1_log = int(p1_log)
    s.param_mode = PARAM_SCALAR

    def classify(a):
        a = np.asarray(a, dtype=np.float32)
        if a.ndim == 0:
            return PARAM_SCALAR, a
        if a.ndim == 1:
            return PARAM_PER_CHANNEL, a
        if a.shape[-1] == 1:
            return PARAM_PER_EDGE1, a
        return PARAM_PER_EDGE, a

    if k == NOISE_EXPLICIT:
        a = _f32(p0)
        keep.append(a)
        s.p0 = _p(a, _f32p)
    elif k >= NOISE_NORMAL:
        m0, a0 = classify(p0)
        mode = m0
        a1 = None
        if k != NOISE_BERNOULLI:
            m1, a1 = classify(p1)
            mode = max(m0, m1)
        s.param_mode = mode
        if mode == PARAM_SCALAR:
            s.p0_scalar = float(a0)
            if a1 is not None:
                s.p1_scalar = float(a1)
        else:
            shape = {PARAM_PER_CHANNEL: (Dn,), PARAM_PER_EDGE1: (n_edges, 1),
                     PARAM_PER_EDGE: (n_edges, Dn)}[mode]
            a0 = _f32(np.broadcast_to(a0, shape))
            keep.append(a0)
            s.p0 = _p(a0, _f32p)
            if a1 is not None:
                a1 = _f32(np.broadcast_to(a1, shape))
                keep.append(a1)
                s.p1 = _p(a1, _f32p)
    s._keep = keep
    return s


def philox_raw(seed, offset, pos0, n_pos, n_chunk):
    out = np.zeros((n_pos, n_chunk, 4), np.uint32)
    _check(lib().stag_philox_raw_cpu(seed, offset, pos0, n_pos, n_chunk, _p(out, _u32p)),
           "philox_raw")
    return out


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, np.uint32).copy()
    k = np.asarray(key, np.uint32).copy()
    o = np.zeros(4, np.uint32)
    lib().stag_philox4x32_10_cpu(_p(c, _u32p), _p(k, _u32p), _p(o, _u32p))
    return o


def noise_materialize(g, spec, Dn):
    w = np.zeros((g.n_edges, Dn), np.float32)
    _check(lib().stag_noise_materialize_cpu(C.byref(g.c), C.byref(spec), Dn, _p(w, _f32p), Dn),
           "noise_materialize")
    return w


def agg_fwd(g, x, spec, reduce=REDUCE_SUM, src_scale=None, dst_scale=None, want_norm_scale=False):
    x = _f32(x)
    D = x.shape[1]
    out = np.zeros((g.n_dst, D), np.float32)
    ss, ds = _f32(src_scale), _f32(dst_scale)
    ns = np.zeros((g.n_dst, D), np.float32) if want_norm_scale else None
    _check(lib().stag_agg_fwd_cpu(C.byref(g.c), _p(x, _f32p), x.strides[0] // 4, D, C.byref(spec),
                                  reduce, _p(ss, _f32p), _p(ds, _f32p), _p(out, _f32p), D,
                                  _p(ns, _f32p)), "agg_fwd")
    return (out, ns) if want_norm_scale else out


def agg_ref_dataflow(g, coo_src, coo_dst, x, spec, bufs=None):
    x = _f32(x)
    D = x.shape[1]
    coo_src, coo_dst = _i32(coo_src), _i32(coo_dst)
    if bufs is None:
        bufs = (np.empty((g.n_edges, D), np.float32), np.empty((g.n_edges, D), np.float32))
    out = np.zeros((g.n_dst, D), np.float32)
    _check(lib().stag_agg_ref_dataflow_cpu(C.byref(g.c), _p(coo_src, _i32p), _p(coo_dst, _i32p),
                                           _p(x, _f32p), x.strides[0] // 4, D, C.byref(spec),
                                           _p(bufs[0], _f32p), _p(bufs[1], _f32p),
                                           _p(out, _f32p), D), "agg_ref_dataflow")
    return out


def agg_bwd_w(g, x, grad, src_scale=None, spec=None, reduce_k=False):
    x, grad = _f32(x), _f32(grad)
    D = x.shape[1]
    ss = _f32(src_scale)
    dw = np.zeros((g.n_edges, 1 if reduce_k else D), np.float32)
    _check(lib().stag_agg_bwd_w_cpu(C.byref(g.c), _p(x, _f32p), D, _p(grad, _f32p), D, D,
                                    _p(ss, _f32p), C.byref(spec) if spec is not None else None,
                                    int(reduce_k), _p(dw, _f32p), dw.shape[1]), "agg_bwd_w")
    return dw


def segment_reduce(x, offsets, reduce=REDUCE_SUM):
    x, offsets = _f32(x), _i32(offsets)
    B, D = len(offsets) - 1, x.shape[1]
    out = np.zeros((B, D), np.float32)
    _check(lib().stag_segment_reduce_cpu(_p(x, _f32p), D, D, _p(offsets, _i32p), B, reduce,
                                         _p(out, _f32p), D), "segment_reduce")
    return out


def gat_fwd(g, el, er, ft, neg_slope, spec, want_attn=False, keep=None, keep_prob=1.0):
    """keep [E, H] by edge id (0/1) + keep_prob: attention dropout between the softmax and the sum
    (stag/zoo/gat.py:122); the returned attention is then the dropped one, as the reference's edata is."""
    el, er, ft = _f32(el), _f32(er), _f32(ft)
    H = el.shape[1]
    F = ft.shape[-1] if ft.ndim == 3 else ft.shape[1] // H
    out = np.zeros((g.n_dst, H, F), np.float32)
    attn = np.zeros((g.n_edges, H), np.float32) if want_attn else None
    keep = None if keep is None else _f32(keep)
    _check(lib().stag_gat_fwd_drop_cpu(C.byref(g.c), _p(el, _f32p), _p(er, _f32p), _p(ft, _f32p), H, F,
                                       float(neg_slope), C.byref(spec), _p(keep, _f32p), float(keep_prob),
                                       _p(out, _f32p), _p(attn, _f32p)), "gat_fwd")
    return (out, attn) if want_attn else out


def gat_bwd(g, el, er, ft, gout, neg_slope, spec, keep=None, keep_prob=1.0, want_dw=False):
    """CPU twin of stag_gat_bwd: (d_el [n_src, H], d_er [n_dst, H], d_ft [n_src, H, F], dw [E, H] | None) — the
    gradients autograd returns for stag/zoo/gat.py:109-126 given d out = gout, in float64 rounded once; keep /
    keep_prob: the attention-dropout mask of the forward (by edge id)."""
    el, er, ft, gout = _f32(el), _f32(er), _f32(ft), _f32(gout)
    H = el.shape[1]
    F = ft.shape[-1] if ft.ndim == 3 else ft.shape[1] // H
    d_el = np.zeros((g.n_src, H), np.float32)
    d_er = np.zeros((g.n_dst, H), np.float32)
    d_ft = np.zeros((g.n_src, H, F), np.float32)
    dw = np.zeros((g.n_edges, H), np.float32) if want_dw else None
    keep = None if keep is None else _f32(keep)
    _check(lib().stag_gat_bwd_cpu(C.byref(g.c), _p(el, _f32p), _p(er, _f32p), _p(ft, _f32p), _p(gout, _f32p), H, F,
                                  float(neg_slope), C.byref(spec), _p(keep, _f32p), float(keep_prob),
                                  _p(d_el, _f32p), _p(d_er, _f32p), _p(d_ft, _f32p), _p(dw, _f32p)), "gat_bwd")
    return d_el, d_er, d_ft, dw


def coldot(x, t0, t1=None):
    """CPU twin of stag_coldot: out_i[k] = sum_n x[n,k] * t_i[n,k], accumulated in fp64."""
    x64 = np.asarray(x, dtype=np.float64)
    o0 = (x64 * np.asarray(t0, dtype=np.float64)).sum(0)
    return o0, (None if t1 is None else (x64 * np.asarray(t1, dtype=np.float64)).sum(0))


def agg_bwd(g_t, gout, spec, g_scale=None, row_scale=None, want_dp=True):
    """CPU twin of stag_agg_bwd on the source-major CSR `g_t` (its nidx = forward positions):
    (dx, dp0_rows, dp1_rows) = the aggregation with spec.deriv = 0, 1, 2."""
    outs = []
    for deriv in ((0, 1, 2) if want_dp else (0,)):
        s = NoiseSpec.from_buffer_copy(spec)
        s.deriv = deriv
        outs.append(agg_fwd(g_t, gout, s, src_scale=g_scale, dst_scale=row_scale))
    return tuple(outs) if want_dp else (outs[0], None, None)


def agg_fwd_mc(g, x, spec, n_samples, offset_stride=1, **kw):
    """CPU twin of stag_agg_fwd_mc: [n_samples, N, D], sample s drawn at offset + s * offset_stride."""
    outs = []
    for s_i in range(n_samples):
        s = NoiseSpec.from_buffer_copy(spec)
        s.offset = spec.offset + s_i * offset_stride
        outs.append(agg_fwd(g, x, s, **kw))
    return np.stack(outs, 0)


# ---- amortised per-edge parameters and their KL term (float64 numpy; no C twin: dense arithmetic) -----------
def amortized_parameters(src, dst, feat, w_embed, b_embed, heads):
    """AmortizedDistribution.condition (stag/distributions.py:221-233) restated:
        h_e   = SiLU(W_e [feat[src_e] || feat[dst_e]] + b_e)       :225-227 (embedding_mlp = Linear + SiLU, :178-183)
        par_c = W_c h_e + b_c  for every head c                    :229-231 (parameters_mlp, :186-191)
    heads: {name: (weight [out, hidden], bias [out])} -> {name: [E, out]} in float64."""
    f = np.asarray(feat, np.float64)
    cat = np.concatenate([f[np.asarray(src)], f[np.asarray(dst)]], 1)
    pre = cat @ np.asarray(w_embed, np.float64).T + (0.0 if b_embed is None else np.asarray(b_embed, np.float64))
    h = pre / (1.0 + np.exp(-pre))
    return {k: h @ np.asarray(w, np.float64).T + (0.0 if b is None else np.asarray(b, np.float64))
            for k, (w, b) in heads.items()}


def normal_kl_mean(loc, log_scale, p_loc, p_scale):
    """StagLayer.kl_divergence for a Normal pair (stag/layers.py:132-139): torch's closed form
    (torch/distributions/kl.py _kl_normal_normal: 0.5 (vr + t1 - 1 - log vr), vr = (s_q / s_p)^2,
    t1 = ((m_q - m_p) / s_p)^2), averaged over every element."""
    m, ls = np.asarray(loc, np.float64), np.asarray(log_scale, np.float64)
    vr = (np.exp(ls) / float(p_scale)) ** 2
    t1 = ((m - float(p_loc)) / float(p_scale)) ** 2
    return float((0.5 * (vr + t1 - 1.0 - np.log(vr))).mean())
