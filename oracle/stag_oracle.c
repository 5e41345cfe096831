/*
 * stag_oracle.c — CPU restatement of the stochastic-aggregation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under stag_amd/ may import, link or call
 * this file; it is the checker for tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py.
 *
 * What it restates (reference = yuanqing-wang/stag, read-only at /root/reference):
 *   - StagLayer.forward / rsample_noise / relu / _in_norm   stag/layers.py:8-36, 84-129
 *   - ParametrizedDistribution broadcast rules              stag/distributions.py:93-144
 *   - GCN / GraphSAGE aggregation lines                      stag/zoo/gcn.py:59-75, 94-108
 *                                                            stag/zoo/graph_sage.py:53-57, 70-75
 *   - GAT noisy-logit softmax aggregation                    stag/zoo/gat.py:109-126
 *   - readout                                                stag/layers.py:156-178
 *   - DGL's u_mul_e / copy_e / sum / mean / edge_softmax semantics, which the
 *     reference delegates to the un-vendored, unpinned `dgl` package (absent
 *     from /root/reference and from this image): restated from DGL's documented
 *     behaviour.  For that arithmetic the reference's own tests assert shapes
 *     only (stag/tests/test_layers.py:21,32,43,54) => parity for it is pinned by
 *     tests/golden/ fixtures produced by running the reference's Python source
 *     here against a stand-in graph object (tests/golden/make_golden.py).
 *
 * Pinning status
 *   - Philox4x32-10: pinned by the Random123 known-answer vectors
 *     (tests/test_oracle_philox.py).
 *   - explicit-weight aggregation, relu, _in_norm, GCN/SAGE/GAT glue: pinned by
 *     tests/golden fixtures (reference source executed in the build container).
 *   - the fused Philox noise stream itself has no counterpart in the reference
 *     (it draws from torch's global generator): defined in include/stag_hip.h,
 *     checked distributionally against torch.distributions.
 *
 * Arithmetic: sums are accumulated in double and rounded once to fp32, so the
 * oracle is independent of any summation order the GPU kernel chooses; the GPU
 * result must agree within |a-b| <= 1e-5 * (1 + |b|).
 */
#include "stag_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as    */
/* easy as 1, 2, 3", SC'11; constants as in Random123 philox.h)              */
/* ------------------------------------------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void stag_philox4x32_10_cpu(const uint32_t ctr[4], const uint32_t key[2],
                            uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void noise_words(uint64_t seed, uint64_t offset, int64_t gpos,
                               uint32_t chunk, uint32_t r[4]) {
  uint32_t ctr[4], key[2];
  ctr[0] = (uint32_t)((uint64_t)gpos & 0xFFFFFFFFu);
  ctr[1] = chunk | ((uint32_t)((uint64_t)gpos >> 32) << 20);
  ctr[2] = (uint32_t)(offset & 0xFFFFFFFFu);
  ctr[3] = (uint32_t)(offset >> 32);
  key[0] = (uint32_t)(seed & 0xFFFFFFFFu);
  key[1] = (uint32_t)(seed >> 32);
  stag_philox4x32_10_cpu(ctr, key, r);
}

int stag_philox_raw_cpu(uint64_t seed, uint64_t offset, int64_t pos0,
                        int64_t n_pos, int32_t n_chunk, uint32_t* out) {
  if (!out || n_pos < 0 || n_chunk < 0) return STAG_EINVAL;
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < n_pos; ++p)
    for (int32_t c = 0; c < n_chunk; ++c)
      noise_words(seed, offset, pos0 + p, (uint32_t)c, out + (p * n_chunk + c) * 4);
  return STAG_OK;
}

/* standard (parameter-free) draws for 4 channels: uniform u in [0,1) or normal z.
 * f12(r) = the float in [1,2) whose mantissa is the low 23 bits of r (include/stag_hip.h). */
static inline float f12(uint32_t r) {
  union { uint32_t u; float f; } v;
  v.u = (r & 0x007FFFFFu) | 0x3F800000u;
  return v.f;
}

static inline void std_uniform4(const uint32_t r[4], float u[4]) {
  for (int j = 0; j < 4; ++j) u[j] = f12(r[j]) - 1.0f;
}

/* Optional: the device's own evaluation of the three functions a normal draw is made of, for all
 * 2^23 mantissas (include/stag_hip.h: stag_normal_tables).  With the tables loaded the oracle redraws
 * the device's normals BIT FOR BIT (z = rad[m_a] * cos[m_b], one fp32 multiply, as in the kernel), so
 * what is left between the two is arithmetic on identical weights.  Without them (the default; what
 * the CPU tests and the golden fixtures use) the draw is the fp64 expression rounded once. */
static const float* g_tab_rad = NULL;
static const float* g_tab_cos = NULL;
static const float* g_tab_sin = NULL;

int stag_set_normal_tables_cpu(const float* rad, const float* cosv, const float* sinv) {
  if ((rad == NULL) != (cosv == NULL) || (rad == NULL) != (sinv == NULL)) return STAG_EINVAL;
  g_tab_rad = rad; g_tab_cos = cosv; g_tab_sin = sinv;
  return STAG_OK;
}

static inline void std_normal4(const uint32_t r[4], float z[4]) {
  if (g_tab_rad) {
    for (int h = 0; h < 2; ++h) {
      const float rad = g_tab_rad[r[2 * h] & 0x007FFFFFu];
      z[2 * h] = rad * g_tab_cos[r[2 * h + 1] & 0x007FFFFFu];
      z[2 * h + 1] = rad * g_tab_sin[r[2 * h + 1] & 0x007FFFFFu];
    }
    return;
  }
  for (int h = 0; h < 2; ++h) {
    double u1 = 2.0 - (double)f12(r[2 * h]);           /* (0, 1] */
    double u2 = (double)f12(r[2 * h + 1]) - 1.0;       /* [0, 1) */
    double rad = sqrt(-2.0 * log(u1));
    double ang = 6.283185307179586476925286766559 * u2;
    z[2 * h] = (float)(rad * cos(ang));
    z[2 * h + 1] = (float)(rad * sin(ang));
  }
}

/* parameter lookup following `q_a.expand([E, Dn])` broadcasting */
static inline float param_at(const float* p, float scalar, int mode, int64_t eid,
                             int32_t k, int32_t Dn) {
  switch (mode) {
    case STAG_PARAM_SCALAR: return scalar;
    case STAG_PARAM_PER_CHANNEL: return p[k];
    case STAG_PARAM_PER_EDGE1: return p[eid];
    default: return p[eid * (int64_t)Dn + k];
  }
}

/* w for channels [4c, 4c+4) of the edge at CSR position p — before in-norm.
 * stag/layers.py:117-127 (sample) and :98-99 (relu). */
static void edge_weights4(const stag_csr* csr, const stag_noise_spec* s,
                          int64_t p, int32_t c, int32_t Dn, float w[4]) {
  int64_t eid = csr->eid ? csr->eid[p] : p;
  int32_t k0 = 4 * c;
  if (s->kind == STAG_NOISE_NONE) {
    for (int j = 0; j < 4; ++j) w[j] = 1.0f;
  } else if (s->kind == STAG_NOISE_EXPLICIT) {
    int32_t grp = s->group > 1 ? s->group : 1;   /* one weight per `grp` channels */
    for (int j = 0; j < 4; ++j)
      w[j] = (k0 + j < Dn) ? s->p0[eid * (int64_t)(Dn / grp) + (k0 + j) / grp] : 0.0f;
  } else {
    int64_t gpos = s->pos_base + (csr->nidx ? (int64_t)csr->nidx[p] : (int64_t)p);
    uint32_t r[4];
    noise_words(s->seed, s->offset + (s->epoch ? *s->epoch : 0) /* host pointer here */, gpos, (uint32_t)(c + s->chunk_base), r);
    float t[4];
    if (s->kind == STAG_NOISE_NORMAL) std_normal4(r, t); else std_uniform4(r, t);
    for (int j = 0; j < 4; ++j) {
      int32_t k = k0 + j;
      if (k >= Dn) { w[j] = 0.0f; continue; }
      float a = param_at(s->p0, s->p0_scalar, s->param_mode, eid, k, Dn);
      float d1 = 1.0f, d2 = t[j];                     /* dw/dp0, dw/dp1 */
      if (s->kind == STAG_NOISE_NORMAL) {
        float b = param_at(s->p1, s->p1_scalar, s->param_mode, eid, k, Dn);
        if (s->p1_log) {                              /* p1 = log(scale): d/dlog_scale = z * scale */
          b = expf(b);
          d2 = t[j] * b;
        }
        w[j] = fmaf(b, t[j], a);                      /* loc + scale * z */
      } else if (s->kind == STAG_NOISE_UNIFORM) {
        float b = param_at(s->p1, s->p1_scalar, s->param_mode, eid, k, Dn);
        w[j] = fmaf(b - a, t[j], a);                  /* low + (high-low) * u */
        d1 = 1.0f - t[j];
      } else {
        w[j] = t[j] < a ? 1.0f : 0.0f;                /* Bernoulli(probs) */
      }
      if (s->deriv) {                                 /* backward of rsample, stag/layers.py:123-124 */
        float mask = (s->relu && !(w[j] > 0.0f)) ? 0.0f : 1.0f;
        w[j] = (s->deriv == 1 ? d1 : d2) * mask;
      }
    }
    if (s->deriv) return;
  }
  if (s->relu)
    for (int j = 0; j < 4; ++j) w[j] = w[j] > 0.0f ? w[j] : 0.0f;
}

static int check_spec(const stag_noise_spec* s) {
  if (!s) return STAG_EINVAL;
  if (s->kind < STAG_NOISE_NONE || s->kind > STAG_NOISE_BERNOULLI) return STAG_EINVAL;
  if (s->kind == STAG_NOISE_EXPLICIT && !s->p0) return STAG_EINVAL;
  if (s->deriv < 0 || s->deriv > 2) return STAG_EINVAL;
  if (s->deriv != 0 && (s->in_norm || (s->kind != STAG_NOISE_NORMAL && s->kind != STAG_NOISE_UNIFORM)))
    return STAG_EINVAL;
  if (s->kind >= STAG_NOISE_NORMAL) {
    if (s->param_mode < STAG_PARAM_SCALAR || s->param_mode > STAG_PARAM_PER_EDGE)
      return STAG_EINVAL;
    if (s->param_mode != STAG_PARAM_SCALAR) {
      if (!s->p0) return STAG_EINVAL;
      if (s->kind != STAG_NOISE_BERNOULLI && !s->p1) return STAG_EINVAL;
    }
  }
  return STAG_OK;
}

/* in-norm factor of one destination row, per channel: stag/layers.py:17-28
 *   cur = sum_in w ; s = cur != 0 ? indeg / cur : 1                           */
static void row_norm_scale(const stag_csr* csr, const stag_noise_spec* s,
                           int32_t v, int32_t Dn, float* scale /*[Dn]*/) {
  int32_t b = csr->indptr[v], e = csr->indptr[v + 1];
  int32_t nchunk = (Dn + 3) / 4;
  for (int32_t c = 0; c < nchunk; ++c) {
    double cur[4] = {0, 0, 0, 0};
    for (int32_t p = b; p < e; ++p) {
      float w[4];
      edge_weights4(csr, s, p, c, Dn, w);
      for (int j = 0; j < 4; ++j) cur[j] += (double)w[j];
    }
    for (int j = 0; j < 4; ++j) {
      int32_t k = 4 * c + j;
      if (k >= Dn) break;
      float curf = (float)cur[j];
      scale[k] = (curf != 0.0f) ? (float)((double)(e - b) / cur[j]) : 1.0f;
    }
  }
}

/* ------------------------------------------------------------------------- */
int stag_noise_materialize_cpu(const stag_csr* csr, const stag_noise_spec* spec,
                               int32_t Dn, float* w_out, int64_t ldw) {
  int rc = check_spec(spec);
  if (rc) return rc;
  if (!csr || !w_out || Dn <= 0 || ldw < Dn) return STAG_EINVAL;
  int32_t nchunk = (Dn + 3) / 4;
#pragma omp parallel
  {
    float* scale = (float*)malloc(sizeof(float) * (size_t)Dn);
#pragma omp for schedule(dynamic, 64)
    for (int32_t v = 0; v < csr->n_dst; ++v) {
      if (spec->in_norm) row_norm_scale(csr, spec, v, Dn, scale);
      for (int32_t p = csr->indptr[v]; p < csr->indptr[v + 1]; ++p) {
        int64_t eid = csr->eid ? csr->eid[p] : p;
        for (int32_t c = 0; c < nchunk; ++c) {
          float w[4];
          edge_weights4(csr, spec, p, c, Dn, w);
          for (int j = 0; j < 4; ++j) {
            int32_t k = 4 * c + j;
            if (k >= Dn) break;
            w_out[eid * ldw + k] = spec->in_norm ? w[j] * scale[k] : w[j];
          }
        }
      }
    }
    free(scale);
  }
  return STAG_OK;
}

/* ------------------------------------------------------------------------- */
/* Fused twin: noise generated on the fly, destination-major segmented sum.   */
int stag_agg_fwd_cpu(const stag_csr* csr, const float* x, int64_t ldx, int32_t D,
                     const stag_noise_spec* spec, int32_t reduce,
                     const float* src_scale, const float* dst_scale, float* out,
                     int64_t ldo, float* norm_scale_out) {
  int rc = check_spec(spec);
  if (rc) return rc;
  if (!csr || !x || !out || D <= 0 || ldx < D || ldo < D) return STAG_EINVAL;
  if (reduce != STAG_REDUCE_SUM && reduce != STAG_REDUCE_MEAN) return STAG_EINVAL;
  int32_t nchunk = (D + 3) / 4;
#pragma omp parallel
  {
    double* acc = (double*)malloc(sizeof(double) * (size_t)D);
    float* scale = (float*)malloc(sizeof(float) * (size_t)D);
#pragma omp for schedule(dynamic, 64)
    for (int32_t v = 0; v < csr->n_dst; ++v) {
      int32_t b = csr->indptr[v], e = csr->indptr[v + 1];
      for (int32_t k = 0; k < D; ++k) { acc[k] = 0.0; scale[k] = 1.0f; }
      if (spec->in_norm) row_norm_scale(csr, spec, v, D, scale);
      for (int32_t p = b; p < e; ++p) {
        int32_t u = csr->indices[p];
        double su = src_scale ? (double)src_scale[u] : 1.0;
        const float* xr = x + (int64_t)u * ldx;
        for (int32_t c = 0; c < nchunk; ++c) {
          float w[4];
          edge_weights4(csr, spec, p, c, D, w);
          for (int j = 0; j < 4; ++j) {
            int32_t k = 4 * c + j;
            if (k >= D) break;
            acc[k] += (double)w[j] * ((double)xr[k] * su);
          }
        }
      }
      double dv = dst_scale ? (double)dst_scale[v] : 1.0;
      if (reduce == STAG_REDUCE_MEAN) dv /= (double)((e - b) > 1 ? (e - b) : 1);
      for (int32_t k = 0; k < D; ++k) {
        out[(int64_t)v * ldo + k] = (float)(acc[k] * (double)scale[k] * dv);
        if (norm_scale_out) norm_scale_out[(int64_t)v * D + k] = scale[k];
      }
    }
    free(acc);
    free(scale);
  }
  return STAG_OK;
}

/* ------------------------------------------------------------------------- */
/* Reference dataflow twin (what the reference's CPU path does, in its order):
 *   1. materialise w[E, D]                       stag/layers.py:117-127
 *   2. m[e, :] = x[src_e, :] * w[e, :]           dgl u_mul_e   (zoo/gcn.py:63)
 *   3. out[dst_e, :] += m[e, :]                  dgl sum       (zoo/gcn.py:95)
 * `w_buf` and `m_buf` are caller-provided [E, D] scratch so the timing covers
 * the traffic the unfused path pays.  fp32 throughout, like the reference.     */
int stag_agg_ref_dataflow_cpu(const stag_csr* csr, const int32_t* coo_src,
                              const int32_t* coo_dst, const float* x, int64_t ldx,
                              int32_t D, const stag_noise_spec* spec, float* w_buf,
                              float* m_buf, float* out, int64_t ldo) {
  if (!csr || !coo_src || !coo_dst || !x || !w_buf || !m_buf || !out) return STAG_EINVAL;
  int rc = stag_noise_materialize_cpu(csr, spec, D, w_buf, D);
  if (rc) return rc;
  int64_t E = csr->n_edges;
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < E; ++e) {
    const float* xr = x + (int64_t)coo_src[e] * ldx;
    const float* wr = w_buf + e * D;
    float* mr = m_buf + e * D;
    for (int32_t k = 0; k < D; ++k) mr[k] = xr[k] * wr[k];
  }
  /* segmented sum: row-parallel over the dst-major order (DGL's CPU SpMM is
   * row-parallel over the CSR too) */
#pragma omp parallel for schedule(dynamic, 64)
  for (int32_t v = 0; v < csr->n_dst; ++v) {
    float* o = out + (int64_t)v * ldo;
    for (int32_t k = 0; k < D; ++k) o[k] = 0.0f;
    for (int32_t p = csr->indptr[v]; p < csr->indptr[v + 1]; ++p) {
      int64_t eid = csr->eid ? csr->eid[p] : p;
      const float* mr = m_buf + eid * D;
      for (int32_t k = 0; k < D; ++k) o[k] += mr[k];
    }
  }
  return STAG_OK;
}

/* ------------------------------------------------------------------------- */
int stag_agg_bwd_w_cpu(const stag_csr* csr, const float* x, int64_t ldx,
                       const float* g, int64_t ldg, int32_t D,
                       const float* src_scale, const stag_noise_spec* spec, int32_t reduce_k,
                       float* dw, int64_t ldw) {
  if (!csr || !x || !g || !dw || D <= 0) return STAG_EINVAL;
  const int use_d = spec && spec->kind >= STAG_NOISE_NORMAL && spec->deriv != 0;
  int32_t nchunk = (D + 3) / 4;
#pragma omp parallel for schedule(dynamic, 64)
  for (int32_t v = 0; v < csr->n_dst; ++v) {
    for (int32_t p = csr->indptr[v]; p < csr->indptr[v + 1]; ++p) {
      int64_t eid = csr->eid ? csr->eid[p] : p;
      int32_t u = csr->indices[p];
      double su = src_scale ? (double)src_scale[u] : 1.0;
      double tot = 0.0;
      for (int32_t c = 0; c < nchunk; ++c) {
        float d[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (use_d) edge_weights4(csr, spec, p, c, D, d);
        for (int j = 0; j < 4; ++j) {
          int32_t k = 4 * c + j;
          if (k >= D) break;
          double val = (double)d[j] * (double)x[(int64_t)u * ldx + k] * su * (double)g[(int64_t)v * ldg + k];
          if (reduce_k) tot += val; else dw[eid * ldw + k] = (float)val;
        }
      }
      if (reduce_k) dw[eid * ldw] = (float)tot;
    }
  }
  return STAG_OK;
}

/* ------------------------------------------------------------------------- */
/* Stable destination-major CSR from COO (counting sort): position order inside
 * a row = ascending original edge id.                                          */
int stag_csr_build_cpu(const int32_t* src, const int32_t* dst, int32_t n_src,
                       int32_t n_dst, int64_t E, int32_t* indptr, int32_t* indices,
                       int32_t* eid, int32_t* in_deg, int32_t* out_deg) {
  if (E < 0 || n_src < 0 || n_dst < 0 || !indptr) return STAG_EINVAL;
  if (E > 0 && (!src || !dst || !indices || !eid)) return STAG_EINVAL;
  for (int32_t v = 0; v <= n_dst; ++v) indptr[v] = 0;
  if (out_deg) for (int32_t u = 0; u < n_src; ++u) out_deg[u] = 0;
  for (int64_t e = 0; e < E; ++e) {
    if (dst[e] < 0 || dst[e] >= n_dst || src[e] < 0 || src[e] >= n_src) return STAG_EINVAL;
    indptr[dst[e] + 1]++;
    if (out_deg) out_deg[src[e]]++;
  }
  if (in_deg) for (int32_t v = 0; v < n_dst; ++v) in_deg[v] = indptr[v + 1];
  for (int32_t v = 0; v < n_dst; ++v) indptr[v + 1] += indptr[v];
  int32_t* cursor = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_dst > 0 ? n_dst : 1));
  if (!cursor) return STAG_ENOMEM;
  memcpy(cursor, indptr, sizeof(int32_t) * (size_t)n_dst);
  for (int64_t e = 0; e < E; ++e) {
    int32_t p = cursor[dst[e]]++;
    indices[p] = src[e];
    eid[p] = (int32_t)e;
  }
  free(cursor);
  return STAG_OK;
}

/* ------------------------------------------------------------------------- */
int stag_segment_reduce_cpu(const float* x, int64_t ldx, int32_t D,
                            const int32_t* offsets, int32_t n_seg, int32_t reduce,
                            float* out, int64_t ldo) {
  if (!x || !offsets || !out || D <= 0 || n_seg < 0) return STAG_EINVAL;
#pragma omp parallel for schedule(dynamic, 16)
  for (int32_t b = 0; b < n_seg; ++b) {
    int32_t lo = offsets[b], hi = offsets[b + 1];
    for (int32_t k = 0; k < D; ++k) {
      double a = 0.0;
      for (int32_t i = lo; i < hi; ++i) a += (double)x[(int64_t)i * ldx + k];
      /* dgl.mean_nodes of an empty graph is 0/0 in DGL; we define it as 0 */
      if (reduce == STAG_REDUCE_MEAN) a = (hi > lo) ? a / (double)(hi - lo) : 0.0;
      out[(int64_t)b * ldo + k] = (float)a;
    }
  }
  return STAG_OK;
}

/* ------------------------------------------------------------------------- */
/* GAT: stag/zoo/gat.py:114-126.  Noise width is H (`sample_dimension`,
 * stag/zoo/gat.py:11); the weight multiplies the leaky-relu'd logit BEFORE the
 * softmax (:117-119).  edge_softmax = per-destination, per-head softmax with
 * the usual max subtraction [DGL].                                            */
int stag_gat_fwd_cpu(const stag_csr* csr, const float* el, const float* er,
                     const float* ft, int32_t H, int32_t F, float neg_slope,
                     const stag_noise_spec* spec, float* out, float* attn_out) {
  return stag_gat_fwd_drop_cpu(csr, el, er, ft, H, F, neg_slope, spec, NULL, 1.0f, out, attn_out);
}

/* ... with attention dropout between the softmax and the sum (stag/zoo/gat.py:122:
 * `graph.edata["a"] = self.attn_drop(edge_softmax(graph, e))`): keep [E, H] by edge id is the 0/1 mask
 * (the caller draws it: a Bernoulli(keep_prob) stream of its own), a -> a * keep / keep_prob; attn_out
 * receives the dropped attention, as the reference's edata does. */
int stag_gat_fwd_drop_cpu(const stag_csr* csr, const float* el, const float* er,
                          const float* ft, int32_t H, int32_t F, float neg_slope,
                          const stag_noise_spec* spec, const float* keep, float keep_prob,
                          float* out, float* attn_out) {
  int rc = check_spec(spec);
  if (rc) return rc;
  if (!csr || !el || !er || !ft || !out || H <= 0 || F <= 0) return STAG_EINVAL;
  int32_t nchunk = (H + 3) / 4;
#pragma omp parallel
  {
    float* scale = (float*)malloc(sizeof(float) * (size_t)H);
    double* acc = (double*)malloc(sizeof(double) * (size_t)F);
#pragma omp for schedule(dynamic, 64)
    for (int32_t v = 0; v < csr->n_dst; ++v) {
      int32_t b = csr->indptr[v], e = csr->indptr[v + 1];
      for (int32_t h = 0; h < H; ++h) scale[h] = 1.0f;
      if (spec->in_norm) row_norm_scale(csr, spec, v, H, scale);
      float* logit = (float*)malloc(sizeof(float) * (size_t)((e - b) > 0 ? (e - b) : 1) * H);
      for (int32_t p = b; p < e; ++p) {
        int32_t u = csr->indices[p];
        for (int32_t c = 0; c < nchunk; ++c) {
          float w[4];
          edge_weights4(csr, spec, p, c, H, w);
          for (int j = 0; j < 4; ++j) {
            int32_t h = 4 * c + j;
            if (h >= H) break;
            float s = el[(int64_t)u * H + h] + er[(int64_t)v * H + h];
            float lr = s > 0.0f ? s : neg_slope * s;
            logit[(int64_t)(p - b) * H + h] = (w[j] * scale[h]) * lr;
          }
        }
      }
      for (int32_t h = 0; h < H; ++h) {
        double mx = -INFINITY, den = 0.0;
        for (int32_t p = b; p < e; ++p) {
          double l = logit[(int64_t)(p - b) * H + h];
          if (l > mx) mx = l;
        }
        for (int32_t p = b; p < e; ++p)
          den += exp((double)logit[(int64_t)(p - b) * H + h] - mx);
        for (int32_t f = 0; f < F; ++f) acc[f] = 0.0;
        for (int32_t p = b; p < e; ++p) {
          int32_t u = csr->indices[p];
          double a = exp((double)logit[(int64_t)(p - b) * H + h] - mx) / den;
          int64_t eid = csr->eid ? csr->eid[p] : p;
          if (keep) a = a * (double)keep[eid * H + h] / (double)keep_prob;
          if (attn_out) attn_out[eid * H + h] = (float)a;
          const float* fr = ft + ((int64_t)u * H + h) * F;
          for (int32_t f = 0; f < F; ++f) acc[f] += a * (double)fr[f];
        }
        for (int32_t f = 0; f < F; ++f)
          out[((int64_t)v * H + h) * F + f] = (float)acc[f];
      }
      free(logit);
    }
    free(scale);
    free(acc);
  }
  return STAG_OK;
}

/* ------------------------------------------------------------------------- */
/* GAT backward: what autograd returns for stag/zoo/gat.py:109-126 given d out = G.
 *   s = el[u] + er[v];  lr = leaky_relu(s);  e = wt * lr  (wt = relu?(w) * in-norm factor, a constant of the
 *   backward unless the weights are EXPLICIT and dw is asked for);  a = edge_softmax(e);  a' = a * keep / q
 *   (attn_drop, :122);  out[v,h,:] = sum_e a'_e ft[u_e,h,:]  (:125-126).
 *   d ft[u,h,:] += a'_e G[v,h,:];   dot_e = <G[v,h,:], ft[u_e,h,:]>;   d a_e = dot_e keep_e / q;
 *   d e_e = a_e (d a_e - sum_e' a_e' d a_e')   [softmax];   d s_e = d e_e wt_e lrelu'(s_e);
 *   d el[u_e,h] += d s_e;  d er[v,h] += d s_e;   d w_e = d e_e lr_e (1[w_e > 0] under relu; in-norm must be off).
 * Everything in double, rounded once; edge data by edge id.  d_el [n_src, H], d_ft [n_src, H, F], d_er [n_dst, H],
 * dw [E, H] (EXPLICIT weights only) — each may be NULL. */
int stag_gat_bwd_cpu(const stag_csr* csr, const float* el, const float* er, const float* ft,
                     const float* gout, int32_t H, int32_t F, float neg_slope,
                     const stag_noise_spec* spec, const float* keep, float keep_prob,
                     float* d_el, float* d_er, float* d_ft, float* dw) {
  int rc = check_spec(spec);
  if (rc) return rc;
  if (!csr || !el || !er || !ft || !gout || H <= 0 || F <= 0) return STAG_EINVAL;
  if (dw && (spec->kind != STAG_NOISE_EXPLICIT || spec->in_norm)) return STAG_EINVAL;
  const int64_t E = csr->n_edges;
  const int32_t nchunk = (H + 3) / 4;
  /* pass 1 (rows in parallel): a' and d s of every (position, head) */
  double* ap = (double*)malloc(sizeof(double) * (size_t)(E > 0 ? E : 1) * H);
  double* ds = (double*)malloc(sizeof(double) * (size_t)(E > 0 ? E : 1) * H);
  if (!ap || !ds) { free(ap); free(ds); return STAG_EINVAL; }
#pragma omp parallel
  {
    float* scale = (float*)malloc(sizeof(float) * (size_t)H);
#pragma omp for schedule(dynamic, 64)
    for (int32_t v = 0; v < csr->n_dst; ++v) {
      const int32_t b = csr->indptr[v], e = csr->indptr[v + 1];
      const int32_t deg = e - b;
      if (deg == 0) { if (d_er) for (int32_t h = 0; h < H; ++h) d_er[(int64_t)v * H + h] = 0.0f; continue; }
      for (int32_t h = 0; h < H; ++h) scale[h] = 1.0f;
      if (spec->in_norm) row_norm_scale(csr, spec, v, H, scale);
      double* logit = (double*)malloc(sizeof(double) * (size_t)deg * H);
      double* wt = (double*)malloc(sizeof(double) * (size_t)deg * H);
      double* lr = (double*)malloc(sizeof(double) * (size_t)deg * H);
      double* sl = (double*)malloc(sizeof(double) * (size_t)deg * H);   /* lrelu'(s) */
      for (int32_t p = b; p < e; ++p) {
        const int32_t u = csr->indices[p];
        for (int32_t c = 0; c < nchunk; ++c) {
          float w[4];
          edge_weights4(csr, spec, p, c, H, w);
          for (int j = 0; j < 4; ++j) {
            const int32_t h = 4 * c + j;
            if (h >= H) break;
            const float s = el[(int64_t)u * H + h] + er[(int64_t)v * H + h];
            const int64_t i = (int64_t)(p - b) * H + h;
            lr[i] = s > 0.0f ? (double)s : (double)neg_slope * (double)s;
            sl[i] = s > 0.0f ? 1.0 : (double)neg_slope;
            wt[i] = (double)(w[j] * scale[h]);
            /* the forward's logit is the fp32 product (stag_gat_fwd_cpu); the softmax is taken of that */
            logit[i] = (double)((w[j] * scale[h]) * (s > 0.0f ? s : neg_slope * s));
          }
        }
      }
      for (int32_t h = 0; h < H; ++h) {
        double mx = -INFINITY, den = 0.0, corr = 0.0;
        for (int32_t q = 0; q < deg; ++q) if (logit[(int64_t)q * H + h] > mx) mx = logit[(int64_t)q * H + h];
        for (int32_t q = 0; q < deg; ++q) den += exp(logit[(int64_t)q * H + h] - mx);
        const float* gr = gout + ((int64_t)v * H + h) * F;
        /* first sweep: a, d a, the softmax correction sum_e a_e d a_e */
        for (int32_t q = 0; q < deg; ++q) {
          const int64_t p = b + q, i = (int64_t)q * H + h;
          const int64_t eid = csr->eid ? csr->eid[p] : p;
          const double a = exp(logit[i] - mx) / den;
          const double kq = keep ? (double)keep[eid * H + h] / (double)keep_prob : 1.0;
          const float* fr = ft + ((int64_t)csr->indices[p] * H + h) * F;
          double dot = 0.0;
          for (int32_t f = 0; f < F; ++f) dot += (double)gr[f] * (double)fr[f];
          ap[p * H + h] = a * kq;
          ds[p * H + h] = dot * kq;          /* d a for now */
          logit[i] = a;                      /* reuse: a */
          corr += a * dot * kq;
        }
        double der = 0.0;
        for (int32_t q = 0; q < deg; ++q) {
          const int64_t p = b + q, i = (int64_t)q * H + h;
          const double de = logit[i] * (ds[p * H + h] - corr);
          if (dw) {
            const int64_t eid = csr->eid ? csr->eid[p] : p;
            double m = 1.0;
            if (spec->relu && !(spec->p0[eid * H + h] > 0.0f)) m = 0.0;
            dw[eid * H + h] = (float)(de * lr[i] * m);
          }
          ds[p * H + h] = de * wt[i] * sl[i];
          der += ds[p * H + h];
        }
        if (d_er) d_er[(int64_t)v * H + h] = (float)der;
      }
      free(logit); free(wt); free(lr); free(sl);
    }
    free(scale);
  }
  /* pass 2: scatter by source, in position order, double accumulators */
  if (d_el || d_ft) {
    const int64_t ns = csr->n_src;
    double* ael = d_el ? (double*)calloc((size_t)(ns > 0 ? ns : 1) * H, sizeof(double)) : NULL;
    double* aft = d_ft ? (double*)calloc((size_t)(ns > 0 ? ns : 1) * H * F, sizeof(double)) : NULL;
    for (int32_t v = 0; v < csr->n_dst; ++v)
      for (int32_t p = csr->indptr[v]; p < csr->indptr[v + 1]; ++p) {
        const int64_t u = csr->indices[p];
        for (int32_t h = 0; h < H; ++h) {
          if (ael) ael[u * H + h] += ds[(int64_t)p * H + h];
          if (aft) {
            const double a = ap[(int64_t)p * H + h];
            const float* gr = gout + ((int64_t)v * H + h) * F;
            double* o = aft + (u * H + h) * F;
            for (int32_t f = 0; f < F; ++f) o[f] += a * (double)gr[f];
          }
        }
      }
    if (ael) { for (int64_t i = 0; i < ns * H; ++i) d_el[i] = (float)ael[i]; free(ael); }
    if (aft) { for (int64_t i = 0; i < ns * H * F; ++i) d_ft[i] = (float)aft[i]; free(aft); }
  }
  free(ap); free(ds);
  return STAG_OK;
}
