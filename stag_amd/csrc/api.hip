// api.hip — the C ABI of include/stag_hip.h: argument checks, host-side launch
// planning, and the auxiliary kernels (noise materialisation, weight gradient,
// readout, raw Philox test hook).  The hot kernel lives in agg_kernel.hpp.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/stag_hip.h"
#include "agg_kernel.hpp"

using namespace stag;

namespace stag {
template <int KIND>
hipError_t agg_launch(const AggArgs& a, bool vec, hipStream_t stream);
}

namespace {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// n_edges: per-edge arrays (explicit weights, [E, 1 | Dn] parameters) of a graph without edges have no address
int check_spec(const stag_noise_spec* s, int64_t n_edges = 1) {
  if (!s) return STAG_EINVAL;
  if (s->kind < STAG_NOISE_NONE || s->kind > STAG_NOISE_BERNOULLI) return STAG_EINVAL;
  if (s->kind == STAG_NOISE_EXPLICIT && !s->p0 && n_edges > 0) return STAG_EINVAL;
  if (s->deriv < 0 || s->deriv > 2 || s->chunk_base < 0 || s->chunk_base >= (1 << 20)) return STAG_EINVAL;
  if (s->deriv != 0 && (s->in_norm || (s->kind != STAG_NOISE_NORMAL && s->kind != STAG_NOISE_UNIFORM)))
    return STAG_EINVAL;   // only reparameterised draws have a derivative; in-norm is not differentiated here
  if (s->p1_log != 0 && (s->p1_log != 1 || s->kind != STAG_NOISE_NORMAL)) return STAG_EINVAL;   // a log-scale is a Normal's
  if (s->p1_log && s->param_mode == STAG_PARAM_PER_CHANNEL) return STAG_ENOSYS;   // exponentiate a [Dn] row yourself
  if (s->kind >= STAG_NOISE_NORMAL) {
    if (s->param_mode < STAG_PARAM_SCALAR || s->param_mode > STAG_PARAM_PER_EDGE) return STAG_EINVAL;
    const bool per_edge = s->param_mode == STAG_PARAM_PER_EDGE1 || s->param_mode == STAG_PARAM_PER_EDGE;
    if (s->param_mode != STAG_PARAM_SCALAR && !(per_edge && n_edges == 0)) {
      if (!s->p0) return STAG_EINVAL;
      if (s->kind != STAG_NOISE_BERNOULLI && !s->p1) return STAG_EINVAL;
    }
  }
  return STAG_OK;
}

int check_csr(const stag_csr* g) {
  if (!g || g->n_dst < 0 || g->n_src < 0 || g->n_edges < 0) return STAG_EINVAL;
  if (g->n_edges > 0x7FFFFFFFll) return STAG_EINVAL;   // int32 CSR positions
  if (!g->indptr) return STAG_EINVAL;
  if (g->n_edges > 0 && !g->indices) return STAG_EINVAL;
  return STAG_OK;
}

PhiloxKey make_key(const stag_noise_spec* s) {
  PhiloxKey k;
  k.k0 = (uint32_t)(s->seed & 0xFFFFFFFFull);
  k.k1 = (uint32_t)(s->seed >> 32);
  k.o0 = (uint32_t)(s->offset & 0xFFFFFFFFull);
  k.o1 = (uint32_t)(s->offset >> 32);
  k.epoch = s->epoch;
  return k;
}

// ------------------------------------------------------------------------- //
__global__ void philox_raw_kernel(PhiloxKey key, int64_t pos0, int64_t n_pos, int n_chunk,
                                  uint32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pos * n_chunk) return;
  uint32_t r[4];
  philox_at(pos0 + i / n_chunk, (uint32_t)(i % n_chunk), key, r);
  *reinterpret_cast<uint4*>(out + i * 4) = make_uint4(r[0], r[1], r[2], r[3]);
}

// the hardware functions of a normal draw, tabulated over all 2^23 mantissas (stag_normal_tables)
__global__ void normal_tables_kernel(float* rad, float* cosv, float* sinv) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= (1u << 23)) return;
  rad[m] = bm_radius(m);
  cosv[m] = bm_cos(m);
  sinv[m] = bm_sin(m);
}

// w of the 4 channels [k0, k0+4) of the edge at position p, before in-norm
struct NoiseArgs {
  const int32_t* indptr;
  const int32_t* eid;
  const int32_t* nidx;
  int32_t n_rows;
  int32_t Dn;
  int32_t kind;
  const float* p0;
  const float* p1;
  float p0s, p1s;
  int32_t pmode, nflags, in_norm;   // nflags: relu | deriv << 1 (noise.hpp)
  PhiloxKey key;
  int64_t pos_base;
  uint32_t chunk_base;
  float* w;
  int64_t ldw;
  // launch plan (nullable): units bound the edges one team walks, so a hub row is many teams
  const stag_unit* units;
  const int32_t* long_rows;
  int32_t n_units;
  const float* norm_scale;   // [n_rows, Dn] in-norm factor (materialise), or null
};

// the unit a team owns: destination row, first position, number of edges
__device__ __forceinline__ bool unit_of(const NoiseArgs& a, int unit, int& row, int& b, int& len) {
  if (unit >= a.n_units) return false;
  if (a.units) {
    const int4 q = *reinterpret_cast<const int4*>(a.units + unit);
    row = q.w >= 0 ? a.long_rows[q.x] : q.x; b = q.y; len = q.z;
  } else {
    row = unit; b = a.indptr[unit]; len = a.indptr[unit + 1] - b;
  }
  return true;
}

__device__ __forceinline__ void edge_params4(const NoiseArgs& a, int64_t ed, int k0, float (&pa)[4], float (&pb)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    const bool in = k < a.Dn;
    float q0 = a.p0s, q1 = a.p1s;
    if (a.pmode == 1) { q0 = in ? a.p0[k] : 0.f; q1 = (in && a.p1) ? a.p1[k] : 0.f; }
    else if (a.pmode == 2) { q0 = a.p0[ed]; q1 = a.p1 ? a.p1[ed] : 0.f; }
    else if (a.pmode == 3) { q0 = in ? a.p0[ed * a.Dn + k] : 0.f; q1 = (in && a.p1) ? a.p1[ed * a.Dn + k] : 0.f; }
    if (a.pmode != 0 && (a.nflags & kFlagLogScale)) q1 = exp_scale(q1);     // scalar: exponentiated on the host
    pa[j] = q0; pb[j] = q1;
  }
}

// w (after relu) and both parameter derivatives of the edge at position p (Normal | Uniform)
__device__ __forceinline__ void edge_w4_grad(const NoiseArgs& a, const PhiloxKey& key, int p, int64_t ed,
                                             uint32_t chunk, float (&w)[4], float (&d0)[4], float (&d1)[4]) {
  float pa[4], pb[4];
  edge_params4(a, ed, (int)chunk * 4, pa, pb);
  const int64_t gpos = a.pos_base + (a.nidx ? (int64_t)a.nidx[p] : (int64_t)p);
  chunk += a.chunk_base;
  if (a.kind == kNormal) draw4_grad<kNormal>((uint32_t)gpos, ctr1_of(gpos, chunk), key, pa, pb, a.nflags, w, d0, d1);
  else draw4_grad<kUniform>((uint32_t)gpos, ctr1_of(gpos, chunk), key, pa, pb, a.nflags, w, d0, d1);
}

// as above with the edge's parameters already loaded (per-edge [E, 1] pairs travel with the rows)
__device__ __forceinline__ void edge_w4_grad_p(const NoiseArgs& a, const PhiloxKey& key, int p, uint32_t chunk,
                                               const float (&pa)[4], const float (&pb)[4], float (&w)[4],
                                               float (&d0)[4], float (&d1)[4]) {
  const int64_t gpos = a.pos_base + (a.nidx ? (int64_t)a.nidx[p] : (int64_t)p);
  chunk += a.chunk_base;
  if (a.kind == kNormal) draw4_grad<kNormal>((uint32_t)gpos, ctr1_of(gpos, chunk), key, pa, pb, a.nflags, w, d0, d1);
  else draw4_grad<kUniform>((uint32_t)gpos, ctr1_of(gpos, chunk), key, pa, pb, a.nflags, w, d0, d1);
}

__device__ __forceinline__ void edge_w4(const NoiseArgs& a, int p, int64_t ed, uint32_t chunk,
                                        float (&w)[4]) {
  const int k0 = (int)chunk * 4;
  float pa[4], pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    const bool in = k < a.Dn;
    float q0 = a.p0s, q1 = a.p1s;
    if (a.pmode == 1) { q0 = in ? a.p0[k] : 0.f; q1 = (in && a.p1) ? a.p1[k] : 0.f; }
    else if (a.pmode == 2) { q0 = a.p0[ed]; q1 = a.p1 ? a.p1[ed] : 0.f; }
    else if (a.pmode == 3) { q0 = in ? a.p0[ed * a.Dn + k] : 0.f; q1 = (in && a.p1) ? a.p1[ed * a.Dn + k] : 0.f; }
    if (a.pmode != 0 && (a.nflags & kFlagLogScale)) q1 = exp_scale(q1);     // scalar: exponentiated on the host
    pa[j] = q0; pb[j] = q1;
  }
  const int64_t gpos = a.pos_base + (a.nidx ? (int64_t)a.nidx[p] : (int64_t)p);
  chunk += a.chunk_base;   // global channel group (channel shards)
  const PhiloxKey key = resolve_epoch(a.key);
  switch (a.kind) {
    case kNormal: draw4<kNormal>((uint32_t)gpos, ctr1_of(gpos, chunk), key, pa, pb, a.nflags, w); break;
    case kUniform: draw4<kUniform>((uint32_t)gpos, ctr1_of(gpos, chunk), key, pa, pb, a.nflags, w); break;
    case kBernoulli: draw4<kBernoulli>((uint32_t)gpos, ctr1_of(gpos, chunk), key, pa, pb, a.nflags, w); break;
    case kExplicit:
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = (k0 + j < a.Dn) ? a.p0[ed * a.Dn + k0 + j] : 0.f;
        w[j] = (a.nflags & kFlagRelu) ? fmaxf(t, 0.f) : t;
      }
      break;
    default: w[0] = w[1] = w[2] = w[3] = 1.0f;
  }
}

// A team of LPE lanes per unit of the plan (a row, or a <= seg_len piece of a long row — a hub is
// many teams, not one wave walking 13k edges), 4 channels per lane, two edges in flight.
// Writes what StagLayer keeps in `_edge_weight_sample` (stag/layers.py:107); the in-norm factor
// of the row comes in as `norm_scale` (computed by stag_agg_fwd on a broadcast row of ones).
template <int LPE, bool VEC>
__global__ __launch_bounds__(256) void noise_materialize_kernel(const NoiseArgs a) {
  const int c = threadIdx.x % LPE;
  const int unit = blockIdx.x * (256 / LPE) + threadIdx.x / LPE;
  const uint32_t chunk = blockIdx.y * LPE + c;
  const int k0 = (int)chunk * 4;
  int row, b, len;
  if (!unit_of(a, unit, row, b, len) || k0 >= a.Dn) return;
  float s[4] = {1.f, 1.f, 1.f, 1.f};
  if (a.norm_scale) load4(a.norm_scale + (int64_t)row * a.Dn, k0, a.Dn, VEC, s);
  for (int p = b; p < b + len; p += 2) {
    float w[2][4];
    int64_t ed[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (p + j < b + len) {
        ed[j] = a.eid ? a.eid[p + j] : p + j;
        edge_w4(a, p + j, ed[j], chunk, w[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (p + j < b + len) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w[j][q] *= s[q];
        store4(a.w + ed[j] * a.ldw, k0, a.Dn, VEC, w[j]);
      }
    }
  }
}

// dw[eid, k] = D[p,k] * sscale[u] * x[u,k] * g[v,k]   (stag_agg_bwd_w)
// A team of LPE lanes per unit of the plan, 4 channels per lane, two edges in flight; the row
// of g is the unit's own row.  D = 1 (explicit weights), one regenerated derivative
// (spec.deriv), or BOTH derivatives from one Philox block (w1 non-null).  With reduce_k the
// channel tiles are walked inside the team so that the sum over k is one fixed-order sum.
struct BwdWArgs {
  NoiseArgs n;          // n.kind < kNormal or (n.nflags >> 1 == 0 and !w1)  =>  D = 1
  const int32_t* indices;
  const float* x;
  int64_t ldx;
  const float* g;
  int64_t ldg;
  const float* src_scale;
  int32_t reduce_k;
  float* w1;            // second output (d / d p1) when both derivatives are asked for
};

// MODE 0: D = 1 (no draw); 1: one derivative selected by the flags (any kind / parameter mode);
// 2: both derivatives.  Separate instantiations: together they cost 140 VGPRs (3 waves per SIMD).
template <int LPE, bool VEC, int MODE>
__device__ __forceinline__ void agg_bwd_w_body(const BwdWArgs& b) {
  const NoiseArgs& a = b.n;
  const int c = threadIdx.x % LPE;
  const int unit = blockIdx.x * (256 / LPE) + threadIdx.x / LPE;
  int v, rb, len;
  if (!unit_of(a, unit, v, rb, len)) return;
  const int D = a.Dn;
  const int ntile = ((D + 3) / 4 + LPE - 1) / LPE;
  constexpr bool both = MODE == 2;
  constexpr bool use_d = MODE >= 1;
  const PhiloxKey key = MODE == 2 ? resolve_epoch(a.key) : a.key;
  float gv[4] = {0.f, 0.f, 0.f, 0.f};
  if (ntile == 1 && c * 4 < D) load4(b.g + (int64_t)v * b.ldg, c * 4, D, VEC, gv);
  for (int p0 = rb; p0 < rb + len; p0 += 2) {
    int u[2];
    int64_t ed[2];
    float ss[2], tot0[2] = {0.f, 0.f}, tot1[2] = {0.f, 0.f}, q0[2] = {0.f, 0.f}, q1[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool live = p0 + j < rb + len;
      u[j] = live ? b.indices[p0 + j] : 0;
      ed[j] = live ? (a.eid ? a.eid[p0 + j] : p0 + j) : 0;
      ss[j] = (live && b.src_scale) ? b.src_scale[u[j]] : 1.0f;
      if (both && a.pmode == 2 && live) {     // [E, 1] parameters: in flight with the rows
        q0[j] = a.p0[ed[j]];
        q1[j] = a.p1 ? a.p1[ed[j]] : 0.f;
      }
    }
    for (int tile = 0; tile < ntile; ++tile) {
      const uint32_t chunk = tile * LPE + c;
      const int k0 = (int)chunk * 4;
      if (k0 >= D) continue;
      if (ntile > 1) load4(b.g + (int64_t)v * b.ldg, k0, D, VEC, gv);
      float xv[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (p0 + j < rb + len) load4(b.x + (int64_t)u[j] * b.ldx, k0, D, VEC, xv[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (p0 + j < rb + len) {
          float val[4], o0[4], o1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) val[q] = (k0 + q < D) ? (ss[j] * xv[j][q]) * gv[q] : 0.f;
          if constexpr (both) {
            float w[4], d0[4], d1[4];
            if (a.pmode == 2) {
              const float s1 = (a.nflags & kFlagLogScale) ? exp_scale(q1[j]) : q1[j];
              const float pa[4] = {q0[j], q0[j], q0[j], q0[j]}, pb[4] = {s1, s1, s1, s1};
              edge_w4_grad_p(a, key, p0 + j, chunk, pa, pb, w, d0, d1);
            } else {
              edge_w4_grad(a, key, p0 + j, ed[j], chunk, w, d0, d1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { o0[q] = d0[q] * val[q]; o1[q] = d1[q] * val[q]; }
          } else if constexpr (use_d) {
            float dwt[4];
            edge_w4(a, p0 + j, ed[j], chunk, dwt);
#pragma unroll
            for (int q = 0; q < 4; ++q) o0[q] = dwt[q] * val[q];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) o0[q] = val[q];
          }
          if (!b.reduce_k) {
            store4(a.w + ed[j] * a.ldw, k0, D, VEC, o0);
            if (both) store4(b.w1 + ed[j] * a.ldw, k0, D, VEC, o1);
          } else {
            tot0[j] += (o0[0] + o0[1]) + (o0[2] + o0[3]);
            tot1[j] += (o1[0] + o1[1]) + (o1[2] + o1[3]);
          }
        }
      }
    }
    if (b.reduce_k) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        tot0[j] = team_sum<LPE>(tot0[j]);
        if (both) tot1[j] = team_sum<LPE>(tot1[j]);
        if (p0 + j < rb + len && c == 0) {
          a.w[ed[j] * a.ldw] = tot0[j];
          if (both) b.w1[ed[j] * a.ldw] = tot1[j];
        }
      }
    }
  }
}

template <int LPE, bool VEC>
__global__ __launch_bounds__(256) void agg_bwd_w_kernel0(const BwdWArgs b) { agg_bwd_w_body<LPE, VEC, 0>(b); }
template <int LPE, bool VEC>
__global__ __launch_bounds__(256) void agg_bwd_w_kernel1(const BwdWArgs b) { agg_bwd_w_body<LPE, VEC, 1>(b); }
template <int LPE, bool VEC>
__global__ __launch_bounds__(256) void agg_bwd_w_kernel2(const BwdWArgs b) { agg_bwd_w_body<LPE, VEC, 2>(b); }

// out[b, :] = sum | mean of x[offsets[b]:offsets[b+1], :]   (dgl.sum_nodes / mean_nodes)
// A team of LPE lanes per graph of the batch, 4 channels (one dwordx4) per lane, 4 rows in
// flight; graphs are small (molhiv: ~26 nodes), so several teams share a wave.
template <int LPE, bool VEC>
__global__ __launch_bounds__(256) void segment_reduce_kernel(const float* x, int64_t ldx, int D,
                                                             const int32_t* offsets, int n_seg,
                                                             int mean, float* out, int64_t ldo) {
  const int c = threadIdx.x % LPE;
  const int s = blockIdx.x * (256 / LPE) + threadIdx.x / LPE;
  const int k0 = (blockIdx.y * LPE + c) * 4;
  if (s >= n_seg || k0 >= D) return;
  const int lo = offsets[s], hi = offsets[s + 1];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = lo; i < hi; i += 4) {
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i + j < hi) load4(x + (int64_t)(i + j) * ldx, k0, D, VEC, t[j]);
      else t[j][0] = t[j][1] = t[j][2] = t[j][3] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += t[j][q];
  }
  if (mean) {
    const float inv = (hi > lo) ? 1.0f / (float)(hi - lo) : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] *= inv;
  }
  store4(out + (int64_t)s * ldo, k0, D, VEC, acc);
}


// ---- column dots: out_i[k] = sum_n x[n,k] * t_i[n,k]  (the last step of a per-channel
//      parameter gradient: dp_i[k] = sum_u x[u,k] * dp_i_rows[u,k]; stag_agg_bwd) ------------
// Stage 1: block b walks rows b*R, b*R+1, ... (R rows side by side), CW lanes across the
// columns; the R row-lanes of a column are added through LDS; one partial row per block.
// Stage 2: the partials of a column are added in a fixed two-level order.  Deterministic.
constexpr int kColdotBlocks = 1024;
template <int W>   // W = 4: dwordx4 columns, W = 1: scalar
__global__ __launch_bounds__(256) void coldot_partial_kernel(const float* x, int64_t ldx, const float* t0,
                                                             const float* t1, int64_t ldt, int64_t n_rows,
                                                             int D, int cw, float* part) {
  __shared__ float red[2][256][W];
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, R = 256 / cw;
  const int col = (blockIdx.y * cw + tx) * W;
  float a0[W], a1[W];
#pragma unroll
  for (int q = 0; q < W; ++q) a0[q] = a1[q] = 0.f;
  if (col < D) {
    for (int64_t n = (int64_t)blockIdx.x * R + ty; n < n_rows; n += (int64_t)gridDim.x * R) {
      float xv[W], u0[W], u1[W];
      if constexpr (W == 4) {
        const float4 xx = *reinterpret_cast<const float4*>(x + n * ldx + col);
        const float4 aa = *reinterpret_cast<const float4*>(t0 + n * ldt + col);
        xv[0] = xx.x; xv[1] = xx.y; xv[2] = xx.z; xv[3] = xx.w;
        u0[0] = aa.x; u0[1] = aa.y; u0[2] = aa.z; u0[3] = aa.w;
        if (t1) {
          const float4 bb = *reinterpret_cast<const float4*>(t1 + n * ldt + col);
          u1[0] = bb.x; u1[1] = bb.y; u1[2] = bb.z; u1[3] = bb.w;
        }
      } else {
        xv[0] = x[n * ldx + col]; u0[0] = t0[n * ldt + col];
        if (t1) u1[0] = t1[n * ldt + col];
      }
#pragma unroll
      for (int q = 0; q < W; ++q) {
        a0[q] = __builtin_fmaf(xv[q], u0[q], a0[q]);
        if (t1) a1[q] = __builtin_fmaf(xv[q], u1[q], a1[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < W; ++q) { red[0][threadIdx.x][q] = a0[q]; red[1][threadIdx.x][q] = a1[q]; }
  __syncthreads();
  if (ty == 0 && col < D) {
    for (int o = 0; o < (t1 ? 2 : 1); ++o)
#pragma unroll
      for (int q = 0; q < W; ++q) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += red[o][r * cw + tx][q];
        if (col + q < D) part[((int64_t)blockIdx.x * 2 + o) * D + col + q] = s;
      }
  }
}

// 16 columns x 16 groups per block: group y adds partials y*G .. y*G+G-1 (Kahan, loads batched 8
// at a time), then thread (column, 0) adds the 16 group sums in group order.
__global__ __launch_bounds__(256) void coldot_final_kernel(const float* part, int n_blocks, int D,
                                                           float* out0, float* out1) {
  __shared__ float red[2][16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + tx;
  const int G = (n_blocks + 15) / 16;
  const int b0 = ty * G, b1 = min(b0 + G, n_blocks);
  float s0 = 0.f, c0 = 0.f, s1 = 0.f, c1 = 0.f;
  if (k < D) {
    for (int b = b0; b < b1; b += 8) {
      float v0[8], v1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool in = b + i < b1;
        v0[i] = in ? part[((int64_t)(b + i) * 2) * D + k] : 0.f;
        v1[i] = (in && out1) ? part[((int64_t)(b + i) * 2 + 1) * D + k] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float y0 = v0[i] - c0; const float n0 = s0 + y0; c0 = (n0 - s0) - y0; s0 = n0;
        const float y1 = v1[i] - c1; const float n1 = s1 + y1; c1 = (n1 - s1) - y1; s1 = n1;
      }
    }
  }
  red[0][ty][tx] = s0; red[1][ty][tx] = s1;
  __syncthreads();
  if (ty == 0 && k < D) {
    float a = 0.f, b = 0.f;
    for (int y = 0; y < 16; ++y) { a += red[0][y][tx]; b += red[1][y][tx]; }
    out0[k] = a;
    if (out1) out1[k] = b;
  }
}

}  // namespace

// ------------------------------------------------------------------------- //
// out[i, :] = x[idx[i], :]: a thread per 4 floats of a row (dwordx4 when the rows allow it)
template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const int32_t* __restrict__ idx, int64_t n, int32_t width,
                                                          int32_t nchunk, float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = t / nchunk;
  const int k0 = (int)(t - i * nchunk) * 4;
  if (i >= n) return;
  const float* src = x + (int64_t)idx[i] * ldx;
  float* dst = out + i * ldo;
  if (VEC) {
    *reinterpret_cast<float4*>(dst + k0) = *reinterpret_cast<const float4*>(src + k0);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < width) dst[k0 + j] = src[k0 + j];
  }
}

extern "C" {

int stag_abi_version(void) { return STAG_ABI_VERSION; }

const char* stag_strerror(int code) {
  switch (code) {
    case STAG_OK: return "ok";
    case STAG_EINVAL: return "invalid argument";
    case STAG_ENOMEM: return "workspace too small";
    case STAG_EIO: return "HIP runtime error at launch";
    case STAG_ENOSYS: return "not implemented";
    default: return "unknown error";
  }
}

int stag_plan_count(const int32_t* indptr_host, int32_t n_dst, int32_t seg_len,
                    int32_t* n_units_out, int32_t* n_long_out, int32_t* n_seg_out,
                    int32_t* n_heavy_out) {
  if (!indptr_host || n_dst < 0 || seg_len <= 0 || seg_len > (1 << 20) || !n_units_out || !n_long_out || !n_seg_out)
    return STAG_EINVAL;
  int64_t nl = 0, ns = 0, nh = 0;
  for (int32_t v = 0; v < n_dst; ++v) {
    const int32_t deg = indptr_host[v + 1] - indptr_host[v];
    if (deg < 0) return STAG_EINVAL;
    if (deg > seg_len) { ++nl; ns += (deg + seg_len - 1) / seg_len; }
    else if (deg > STAG_HEAVY_LEN) ++nh;
  }
  if (n_heavy_out) *n_heavy_out = (int32_t)std::min<int64_t>(ns + nh, 0x7FFFFFFFll);
  const int64_t nu = (int64_t)n_dst - nl + ns;
  if (nu > 0x7FFFFFFFll) return STAG_EINVAL;
  *n_units_out = (int32_t)nu;
  *n_long_out = (int32_t)nl;
  *n_seg_out = (int32_t)ns;
  return STAG_OK;
}

int stag_plan_fill(const int32_t* indptr_host, int32_t n_dst, int32_t seg_len,
                   stag_unit* units_host, int32_t* long_rows_host, int32_t* long_seg_ptr_host) {
  if (!indptr_host || n_dst < 0 || seg_len <= 0 || seg_len > (1 << 20) || !long_seg_ptr_host)
    return STAG_EINVAL;
  if (n_dst > 0 && !units_host) return STAG_EINVAL;
  std::vector<int32_t> longs;
  std::vector<int64_t> bucket((size_t)seg_len + 2, 0);   // whole rows by length
  for (int32_t v = 0; v < n_dst; ++v) {
    const int32_t deg = indptr_host[v + 1] - indptr_host[v];
    if (deg > seg_len) longs.push_back(v); else bucket[deg] += 1;
  }
  if (!longs.empty() && !long_rows_host) return STAG_EINVAL;
  // Segments first — the row with the most segments gets the earliest blocks, so the segment
  // that arrives last (and adds the partials) is done long before the launch ends — then the
  // whole rows, longest first (counting sort; every length is in [0, seg_len]).
  std::stable_sort(longs.begin(), longs.end(), [&](int32_t x, int32_t y) {
    return indptr_host[x + 1] - indptr_host[x] > indptr_host[y + 1] - indptr_host[y];
  });
  int32_t s = 0;
  long_seg_ptr_host[0] = 0;
  for (size_t r = 0; r < longs.size(); ++r) {
    const int32_t v = longs[r];
    const int32_t b = indptr_host[v], deg = indptr_host[v + 1] - b;
    const int32_t nseg = (deg + seg_len - 1) / seg_len;
    const int32_t base = deg / nseg, rem = deg % nseg;    // balanced: lengths differ by at most 1
    long_rows_host[r] = v;
    int32_t p = b;
    for (int32_t i = 0; i < nseg; ++i) {
      const int32_t l = base + (i < rem ? 1 : 0);
      units_host[s] = stag_unit{(int32_t)r, p, l, s};
      p += l;
      ++s;
    }
    long_seg_ptr_host[r + 1] = s;
  }
  std::vector<int64_t> cursor((size_t)seg_len + 1, 0);
  int64_t run = s;
  for (int32_t l = seg_len; l >= 0; --l) { cursor[l] = run; run += bucket[l]; }
  for (int32_t v = 0; v < n_dst; ++v) {
    const int32_t b = indptr_host[v], e = indptr_host[v + 1];
    if (e - b <= seg_len) units_host[cursor[e - b]++] = stag_unit{v, b, e - b, -1};
  }
  return STAG_OK;
}

size_t stag_plan_xcd_ints(int32_t stride_heavy, int32_t stride_light) {
  if (stride_heavy < 0 || stride_light < 0) return 0;
  return (size_t)STAG_XCD_HEADER + 4u * (size_t)STAG_XCD_STRIPES * ((size_t)stride_heavy + (size_t)stride_light);
}

int32_t stag_plan_xcd_fine(int32_t n_dst) {
  // fine stripes per XCD stripe: row ranges of about STAG_XCD_FINE_ROWS rows, walked one after the other
  if (n_dst <= 0) return 1;
  const int64_t m = ((int64_t)n_dst + (int64_t)STAG_XCD_STRIPES * STAG_XCD_FINE_ROWS - 1) / ((int64_t)STAG_XCD_STRIPES * STAG_XCD_FINE_ROWS);
  return (int32_t)(m < 1 ? 1 : m > STAG_XCD_FINE_MAX ? STAG_XCD_FINE_MAX : m);
}

extern "C++" {
namespace {
// key of a unit in [0, 8 * fine): which XCD stripe, and which of the stripe's `fine` ranges, its rows belong to.
//   by position: where its first edge lies in the CSR, in 8 * fine equal parts (contiguous destination-row ranges with the
//                same number of edges each);
//   by range table: cuts[R + 1] ascending CSR positions, keys[R] — the key of every unit whose first edge lies in
//                [cuts[r], cuts[r + 1]) (a block-diagonal batch: whole graphs, bin-packed to the stripes).
struct XcdKeyByPosition {
  int64_t E; int S;
  int operator()(int32_t start) const {
    const int64_t k = (int64_t)start * S / E;
    return (int)(k < 0 ? 0 : k >= S ? S - 1 : k);
  }
};
struct XcdKeyByRange {
  const int64_t* cuts; const int32_t* keys; int32_t R;
  int operator()(int32_t start) const {
    int32_t lo = 0, hi = R;                       // last r with cuts[r] <= start (r = 0 below the table)
    while (hi - lo > 1) {
      const int32_t mid = lo + (hi - lo) / 2;
      if (cuts[mid] <= (int64_t)start) lo = mid; else hi = mid;
    }
    return keys[lo];
  }
};

template <class KeyFn>
int plan_xcd_impl(const stag_unit* units_host, int32_t n_units, int32_t n_heavy, int32_t fine, const KeyFn& keyfn,
                  int32_t* xcd_host, int32_t* strides_out) {
  // A stable partition of the heavy prefix and of the rest by key: inside one key the plan's order (segments, then
  // longest first) stands; the `fine` ranges of one XCD stripe lie one after the other and the XCD walks them in turn
  // (the rows one of them gathers should fit its L2).
  const int S = STAG_XCD_STRIPES * fine;
  auto key = [&](int32_t i) { return (int)(i >= n_heavy ? S : 0) + keyfn(units_host[i].start); };
  std::vector<int32_t> count((size_t)2 * S, 0);
  for (int32_t i = 0; i < n_units; ++i) count[key(i)] += 1;
  int32_t per[2 * STAG_XCD_STRIPES] = {0};           // units per XCD stripe: heavy [0, 8), the others [8, 16)
  for (int k = 0; k < 2 * S; ++k) per[k / fine] += count[k];
  int32_t sh = 0, sl = 0;
  for (int k = 0; k < STAG_XCD_STRIPES; ++k) {
    sh = std::max(sh, per[k]);
    sl = std::max(sl, per[STAG_XCD_STRIPES + k]);
  }
  strides_out[0] = sh; strides_out[1] = sl;
  if (!xcd_host) return STAG_OK;
  for (int k = 0; k < STAG_XCD_HEADER; ++k) xcd_host[k] = 0;
  for (int k = 0; k < 2 * STAG_XCD_STRIPES; ++k) xcd_host[k] = per[k];
  xcd_host[2 * STAG_XCD_STRIPES] = sh;
  xcd_host[2 * STAG_XCD_STRIPES + 1] = sl;
  xcd_host[2 * STAG_XCD_STRIPES + 2] = fine;
  stag_unit* rec = reinterpret_cast<stag_unit*>(xcd_host + STAG_XCD_HEADER);
  const int64_t n_rec = (int64_t)STAG_XCD_STRIPES * ((int64_t)sh + sl);
  for (int64_t i = 0; i < n_rec; ++i) rec[i] = stag_unit{-1, 0, 0, -1};
  std::vector<int64_t> cursor((size_t)2 * S);
  for (int x = 0; x < 2 * STAG_XCD_STRIPES; ++x) {   // fine ranges of one XCD stripe lie one after the other
    int64_t at = x < STAG_XCD_STRIPES ? (int64_t)x * sh : (int64_t)STAG_XCD_STRIPES * sh + (int64_t)(x - STAG_XCD_STRIPES) * sl;
    for (int f = 0; f < fine; ++f) { cursor[(size_t)x * fine + f] = at; at += count[(size_t)x * fine + f]; }
  }
  for (int32_t i = 0; i < n_units; ++i) rec[cursor[key(i)]++] = units_host[i];
  return STAG_OK;
}

bool xcd_ranges_ok(const int64_t* cuts, const int32_t* keys, int32_t R, int32_t fine) {
  if (R < 1 || !cuts || !keys) return false;
  for (int32_t r = 0; r < R; ++r)
    if (cuts[r + 1] < cuts[r] || keys[r] < 0 || keys[r] >= STAG_XCD_STRIPES * fine) return false;
  return true;
}
}  // namespace
}  // extern "C++"

int stag_plan_xcd(const stag_unit* units_host, int32_t n_units, int32_t n_heavy, int64_t n_edges, int32_t fine,
                  int32_t* xcd_host, int32_t* strides_out) {
  if (n_units < 0 || n_heavy < 0 || n_heavy > n_units || n_edges < 0 || !strides_out || (n_units > 0 && !units_host) ||
      fine < 1 || fine > STAG_XCD_FINE_MAX) return STAG_EINVAL;
  return plan_xcd_impl(units_host, n_units, n_heavy, fine, XcdKeyByPosition{n_edges > 0 ? n_edges : 1, STAG_XCD_STRIPES * fine},
                       xcd_host, strides_out);
}

int stag_plan_xcd_ranges(const stag_unit* units_host, int32_t n_units, int32_t n_heavy, const int64_t* cuts_host,
                         const int32_t* keys_host, int32_t n_ranges, int32_t fine, int32_t* xcd_host, int32_t* strides_out) {
  if (n_units < 0 || n_heavy < 0 || n_heavy > n_units || !strides_out || (n_units > 0 && !units_host) || fine < 1 ||
      fine > STAG_XCD_FINE_MAX || !xcd_ranges_ok(cuts_host, keys_host, n_ranges, fine)) return STAG_EINVAL;
  return plan_xcd_impl(units_host, n_units, n_heavy, fine, XcdKeyByRange{cuts_host, keys_host, n_ranges}, xcd_host, strides_out);
}

int stag_plan_blocks(const stag_unit* units_host, int32_t n_units, int32_t max_edges, int32_t max_units,
                     int32_t* block_ptr_host, int32_t* n_blocks_out) {
  if (n_units < 0 || max_edges <= 0 || max_units <= 0 || !n_blocks_out || (n_units > 0 && !units_host))
    return STAG_EINVAL;
  int32_t nb = 0, edges = 0, units = 0;
  if (block_ptr_host) block_ptr_host[0] = 0;
  for (int32_t i = 0; i < n_units; ++i) {
    const int32_t len = units_host[i].len;
    if (len < 0) return STAG_EINVAL;
    if (units > 0 && (edges + len > max_edges || units == max_units)) {
      ++nb;
      if (block_ptr_host) block_ptr_host[nb] = i;
      edges = 0; units = 0;
    }
    edges += len; ++units;
  }
  if (units > 0) {
    ++nb;
    if (block_ptr_host) block_ptr_host[nb] = n_units;
  }
  *n_blocks_out = nb;
  return STAG_OK;
}

extern "C++" {
namespace {
template <class KeyFn>
int plan_blocks_xcd_impl(const stag_unit* units_host, int32_t n_units, int32_t fine, const KeyFn& keyfn, int32_t max_edges,
                         int32_t max_units, stag_unit* units_out_host, int32_t* block_ptr_host, int32_t* n_blocks_out) {
  // units by key (stable: the plan's order inside one), batched greedily inside each fine range, then the batches of the
  // 8 XCD stripes dealt out in turn: batch b belongs to stripe b mod 8, a stripe that has run out of batches gets empty ones
  const int S = STAG_XCD_STRIPES * fine;
  std::vector<int32_t> key((size_t)n_units), start((size_t)S + 1, 0), order((size_t)n_units);
  for (int32_t i = 0; i < n_units; ++i) {
    if (units_host[i].len < 0) return STAG_EINVAL;
    key[i] = (int32_t)keyfn(units_host[i].start);
    start[(size_t)key[i] + 1] += 1;
  }
  for (int k = 0; k < S; ++k) start[(size_t)k + 1] += start[k];
  {
    std::vector<int32_t> cur(start.begin(), start.end() - 1);
    for (int32_t i = 0; i < n_units; ++i) order[(size_t)cur[key[i]]++] = i;
  }
  std::vector<std::vector<std::pair<int32_t, int32_t>>> batches(STAG_XCD_STRIPES);   // (first position in `order`, units)
  for (int x = 0; x < STAG_XCD_STRIPES; ++x)
    for (int f = 0; f < fine; ++f) {
      const int32_t lo = start[(size_t)x * fine + f], hi = start[(size_t)x * fine + f + 1];
      int32_t first = lo, edges = 0, units = 0;
      for (int32_t p = lo; p < hi; ++p) {
        const int32_t len = units_host[order[p]].len;
        if (units > 0 && (edges + len > max_edges || units == max_units)) {
          batches[x].emplace_back(first, units);
          first = p; edges = 0; units = 0;
        }
        edges += len; ++units;
      }
      if (units > 0) batches[x].emplace_back(first, units);
    }
  size_t per = 0;
  for (auto& b : batches) per = std::max(per, b.size());
  const int64_t nb = (int64_t)per * STAG_XCD_STRIPES;
  if (nb > 0x7FFFFFFFll) return STAG_EINVAL;
  *n_blocks_out = (int32_t)nb;
  if (!units_out_host) return STAG_OK;
  int32_t at = 0;
  block_ptr_host[0] = 0;
  for (size_t j = 0; j < per; ++j)
    for (int x = 0; x < STAG_XCD_STRIPES; ++x) {
      if (j < batches[x].size())
        for (int32_t p = 0; p < batches[x][j].second; ++p) units_out_host[at++] = units_host[order[(size_t)batches[x][j].first + p]];
      block_ptr_host[j * STAG_XCD_STRIPES + x + 1] = at;
    }
  return STAG_OK;
}
}  // namespace
}  // extern "C++"

int stag_plan_blocks_xcd(const stag_unit* units_host, int32_t n_units, int64_t n_edges, int32_t fine, int32_t max_edges,
                         int32_t max_units, stag_unit* units_out_host, int32_t* block_ptr_host, int32_t* n_blocks_out) {
  if (n_units < 0 || n_edges < 0 || fine < 1 || fine > STAG_XCD_FINE_MAX || max_edges <= 0 || max_units <= 0 ||
      !n_blocks_out || (n_units > 0 && !units_host) || ((units_out_host == nullptr) != (block_ptr_host == nullptr)))
    return STAG_EINVAL;
  return plan_blocks_xcd_impl(units_host, n_units, fine, XcdKeyByPosition{n_edges > 0 ? n_edges : 1, STAG_XCD_STRIPES * fine},
                              max_edges, max_units, units_out_host, block_ptr_host, n_blocks_out);
}

int stag_plan_blocks_xcd_ranges(const stag_unit* units_host, int32_t n_units, const int64_t* cuts_host,
                                const int32_t* keys_host, int32_t n_ranges, int32_t fine, int32_t max_edges, int32_t max_units,
                                stag_unit* units_out_host, int32_t* block_ptr_host, int32_t* n_blocks_out) {
  if (n_units < 0 || fine < 1 || fine > STAG_XCD_FINE_MAX || max_edges <= 0 || max_units <= 0 || !n_blocks_out ||
      (n_units > 0 && !units_host) || ((units_out_host == nullptr) != (block_ptr_host == nullptr)) ||
      !xcd_ranges_ok(cuts_host, keys_host, n_ranges, fine)) return STAG_EINVAL;
  return plan_blocks_xcd_impl(units_host, n_units, fine, XcdKeyByRange{cuts_host, keys_host, n_ranges}, max_edges, max_units,
                              units_out_host, block_ptr_host, n_blocks_out);
}

size_t stag_plan_workspace_bytes(int32_t n_seg, int32_t D, int32_t in_norm) {
  if (n_seg <= 0 || D <= 0) return 0;
  return (size_t)n_seg * (size_t)D * (in_norm ? 2u : 1u) * sizeof(float);
}

int stag_philox_raw(uint64_t seed, uint64_t offset, int64_t pos0, int64_t n_pos, int32_t n_chunk,
                    uint32_t* out, void* stream) {
  if (!out || n_pos < 0 || n_chunk <= 0 || !aligned16(out)) return STAG_EINVAL;
  const int64_t n = n_pos * n_chunk;
  if (n == 0) return STAG_OK;
  stag_noise_spec s{};
  s.seed = seed; s.offset = offset;
  hipLaunchKernelGGL(philox_raw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, make_key(&s), pos0, n_pos, n_chunk, out);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_normal_tables(float* rad, float* cosv, float* sinv, void* stream) {
  if (!rad || !cosv || !sinv) return STAG_EINVAL;
  hipLaunchKernelGGL(normal_tables_kernel, dim3((1u << 23) / 256), dim3(256), 0, (hipStream_t)stream, rad, cosv, sinv);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

// one aggregation launch; nout > 1: extra outputs ride along — the two parameter-derivative
// aggregates (nout = 3, mc = 0) or Monte-Carlo samples 1.. (nout = 2 | 4, mc = 1)
struct EdgeGradOut {      // stag_agg_bwd_edge / stag_agg_bwd_dp: parameter gradients out of the same pass
  const float* xown;      // [n_dst, ldxo]: the rows the units own (stag_agg_bwd_dp: may be null = ones)
  int64_t ldxo;
  float* eg0;             // per edge (stag_agg_bwd_edge)
  float* eg1;
  float* dp_part;         // block partials (stag_agg_bwd_dp)
};
static int agg_common(const stag_csr* csr, const stag_plan* plan, const float* x, int64_t ldx,
                      int32_t D, const stag_noise_spec* spec, int32_t reduce, const float* src_scale,
                      const float* dst_scale, float* out, int64_t ldo, float* norm_scale_out,
                      int nout, float* const* extra, int mc, uint64_t mc_stride, void* stream,
                      const EdgeGradOut* eg = nullptr) {
  int rc = check_csr(csr);
  if (rc) return rc;
  rc = check_spec(spec, csr->n_edges);
  if (rc) return rc;
  if (D <= 0 || (ldx != 0 && ldx < D) || ldo < D) return STAG_EINVAL;
  if (reduce != STAG_REDUCE_SUM && reduce != STAG_REDUCE_MEAN) return STAG_EINVAL;
  if (csr->n_dst == 0) return STAG_OK;        // no row, nothing to write (an output of no rows has no address: a shard
                                              // whose cut left it without rows still makes the call)
  // ldx == 0: one broadcast row; out may be NULL when only the in-norm factor (or the edge gradients) is wanted
  if (!out && !norm_scale_out && !eg) return STAG_EINVAL;
  if (csr->n_edges > 0 && !x) return STAG_EINVAL;

  AggArgs a{};
  a.indptr = csr->indptr; a.indices = csr->indices; a.eid = csr->eid; a.nidx = csr->nidx;
  a.n_rows = csr->n_dst;
  a.x = x; a.ldx = ldx; a.D = D;
  a.ldxb = (uint32_t)(ldx * 4); a.ldwb = (uint32_t)D * 4u;
  {
    // 32-bit byte offsets + 24-bit multiplies when everything fits, else 64-bit addressing
    const uint64_t xbytes = (uint64_t)csr->n_src * (uint64_t)ldx * 4u;
    const uint64_t wbytes = (uint64_t)csr->n_edges * (uint64_t)D * 4u;   // (upper bound when grouped)
    const bool x_narrow = xbytes < (1ull << 32) && csr->n_src < (1 << 24) && (uint64_t)ldx * 4u < (1u << 24);
    const bool w_narrow = wbytes < (1ull << 32) && csr->n_edges < (1 << 24) && (uint64_t)D * 4u < (1u << 24);
    a.wide = (x_narrow ? 0 : 1) | (w_narrow ? 0 : 2);
    a.x_bytes = x_narrow ? (uint32_t)(ldx == 0 ? (uint64_t)D * 4u : xbytes) : 0u;
  }
  const bool logs = spec->kind == STAG_NOISE_NORMAL && spec->p1_log;
  a.p0 = spec->p0; a.p1 = spec->p1; a.p0s = spec->p0_scalar; a.p1s = logs ? expf(spec->p1_scalar) : spec->p1_scalar;
  a.pmode = spec->kind >= STAG_NOISE_NORMAL ? spec->param_mode : 0;
  a.relu = (spec->relu ? kFlagRelu : 0) | (spec->deriv << kDerivShift) | (logs ? kFlagLogScale : 0);
  a.in_norm = spec->in_norm;
  a.wgroup = (spec->kind == STAG_NOISE_EXPLICIT && spec->group > 1) ? spec->group : 1;
  if (a.wgroup > 1 && (D % a.wgroup != 0 || spec->in_norm)) return STAG_EINVAL;
  a.key = make_key(spec);
  a.pos_lo = (uint32_t)((uint64_t)spec->pos_base & 0xFFFFFFFFull);
  a.pos_hi = (uint32_t)((uint64_t)spec->pos_base >> 32);
  a.chunk_base = (uint32_t)spec->chunk_base;
  // one launch must not straddle a 2^32 boundary of the global position space (the kernel keeps
  // hi32 in a scalar and adds pos_lo to the local index in 32 bits, with or without nidx): shards
  // are < 2^31 edges, so split the call at the boundary
  if (spec->kind >= STAG_NOISE_NORMAL && (uint64_t)a.pos_lo + (uint64_t)csr->n_edges > (1ull << 32))
    return STAG_ENOSYS;
  a.src_scale = src_scale; a.dst_scale = dst_scale; a.mean = (reduce == STAG_REDUCE_MEAN);
  a.out = out; a.ldo = ldo; a.norm_scale_out = norm_scale_out;
  for (int o = 0; o + 1 < nout; ++o) a.outx[o] = extra[o];
  a.mc = mc; a.mc_stride = mc_stride;
  if (eg) {
    a.xown = eg->xown; a.ldxo = eg->ldxo; a.own_scale = dst_scale; a.eg0 = eg->eg0; a.eg1 = eg->eg1;
    a.dp_part = eg->dp_part;
  }

  const bool use_plan = plan && plan->n_units > 0;
  const bool has_segs = use_plan && plan->n_seg > 0;
  a.n_units = csr->n_dst;
  if (use_plan) {
    if (!plan->units || !aligned16(plan->units)) return STAG_EINVAL;
    a.units = a.units_plan = static_cast<const stag_unit*>(plan->units);
    a.n_units = plan->n_units;
    if (plan->n_heavy < 0 || plan->n_heavy > plan->n_units) return STAG_EINVAL;
    a.n_heavy = plan->n_heavy;
    // the XCD-aware order of the same records (stag_plan_xcd).  Not for the block partials of stag_agg_bwd_dp: they
    // are added in block order, which stays the plan's own
    if (plan->xcd_order && !(eg && eg->dp_part)) {
      const int64_t sh = plan->xcd_stride_heavy, sl = plan->xcd_stride_light;
      if (!aligned16(plan->xcd_order) || sh < 0 || sl < 0 || sh > plan->n_heavy || sl > plan->n_units ||     /* (sl may count heavy units: stag_plan_xcd_ranges with n_heavy = 0) */
          STAG_XCD_STRIPES * (sh + sl) < plan->n_units || STAG_XCD_STRIPES * (sh + sl) > 0x7FFFFFFFll) return STAG_EINVAL;
      a.xcd = plan->xcd_order;
      a.units = reinterpret_cast<const stag_unit*>(plan->xcd_order + STAG_XCD_HEADER);
      a.walk.sh = (int32_t)sh;     // agg_launch_shape completes the walk for its block size
      a.walk.sl = (int32_t)sl;
    }
  }
  if (has_segs) {
    if (!plan->long_rows || !plan->long_seg_ptr || !plan->workspace || !plan->seg_counters)
      return STAG_EINVAL;
    // partial rows: [D sums | D weight sums if in-norm], or [nout x D] with extra outputs
    const size_t need = stag_plan_workspace_bytes(plan->n_seg, nout * D, spec->in_norm);
    if (plan->workspace_bytes < need) return STAG_ENOMEM;
    a.long_rows = plan->long_rows; a.long_seg_ptr = plan->long_seg_ptr;
    if (need >= (1ull << 32)) return STAG_ENOSYS;   // partials go through a 32-bit buffer descriptor
    a.ws = plan->workspace; a.ws_stride = D * nout * (spec->in_norm ? 2 : 1); a.ws_bytes = (uint32_t)need;
    a.n_long = plan->n_long; a.seg_counters = plan->seg_counters; a.n_seg = plan->n_seg;
  }

  // dwordx4 path needs 16-B aligned rows everywhere a float4 is formed
  bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(x) && (!out || aligned16(out));
  if (norm_scale_out) vec = vec && aligned16(norm_scale_out);
  for (int o = 0; o + 1 < nout; ++o) vec = vec && aligned16(extra[o]);
  if (spec->kind == STAG_NOISE_EXPLICIT) vec = vec && aligned16(spec->p0);
  if (spec->kind >= STAG_NOISE_NORMAL && spec->param_mode != STAG_PARAM_SCALAR &&
      spec->param_mode != STAG_PARAM_PER_EDGE1)
    vec = vec && aligned16(spec->p0) && (!spec->p1 || aligned16(spec->p1));
  if (has_segs) vec = vec && aligned16(plan->workspace);
  if (eg && eg->xown) vec = vec && aligned16(eg->xown) && (eg->ldxo % 4 == 0);

  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](const AggArgs& args) -> hipError_t {
    switch (spec->kind) {
      case STAG_NOISE_NONE: return agg_launch<kNone>(args, vec, s);
      case STAG_NOISE_EXPLICIT: return agg_launch<kExplicit>(args, vec, s);
      case STAG_NOISE_NORMAL: return agg_launch<kNormal>(args, vec, s);
      case STAG_NOISE_UNIFORM: return agg_launch<kUniform>(args, vec, s);
      default: return agg_launch<kBernoulli>(args, vec, s);
    }
  };
  if (launch(a) != hipSuccess) return STAG_EIO;
  return STAG_OK;
}

int stag_agg_fwd(const stag_csr* csr, const stag_plan* plan, const float* x, int64_t ldx,
                 int32_t D, const stag_noise_spec* spec, int32_t reduce, const float* src_scale,
                 const float* dst_scale, float* out, int64_t ldo, float* norm_scale_out,
                 void* stream) {
  return agg_common(csr, plan, x, ldx, D, spec, reduce, src_scale, dst_scale, out, ldo, norm_scale_out,
                    1, nullptr, 0, 0, stream);
}

int stag_agg_fwd_mc(const stag_csr* csr, const stag_plan* plan, const float* x, int64_t ldx,
                    int32_t D, const stag_noise_spec* spec, int32_t n_samples, int64_t offset_stride,
                    int32_t reduce, const float* src_scale, const float* dst_scale, float* out,
                    int64_t ldo, int64_t sample_stride, void* stream) {
  if (!spec || n_samples < 1 || offset_stride < 0 || !out) return STAG_EINVAL;
  if (spec->kind < STAG_NOISE_NORMAL || spec->deriv) return STAG_EINVAL;
  if (spec->param_mode != STAG_PARAM_SCALAR && spec->param_mode != STAG_PARAM_PER_CHANNEL) return STAG_EINVAL;
  if (n_samples > 1 && sample_stride < (int64_t)(csr ? csr->n_dst : 0) * ldo) return STAG_EINVAL;
  // 4 (then 2, then 1) samples per pass over the gathered rows; every launch starts its own
  // samples at the right offset, so the result is that of n_samples separate stag_agg_fwd calls.
  // With in-norm (stag/layers.py:8-36) every sample carries its own weight sums: 2 per pass (registers).
  stag_noise_spec sp = *spec;
  const int kmax = spec->in_norm ? 2 : 4;
  for (int32_t s0 = 0; s0 < n_samples;) {
    const int k = (n_samples - s0 >= 4 && kmax >= 4) ? 4 : (n_samples - s0 >= 2) ? 2 : 1;
    sp.offset = spec->offset + (uint64_t)s0 * (uint64_t)offset_stride;
    float* extra[3] = {nullptr, nullptr, nullptr};
    for (int o = 1; o < k; ++o) extra[o - 1] = out + (int64_t)(s0 + o) * sample_stride;
    const int rc = agg_common(csr, plan, x, ldx, D, &sp, reduce, src_scale, dst_scale,
                              out + (int64_t)s0 * sample_stride, ldo, nullptr, k, extra, k > 1 ? 1 : 0,
                              (uint64_t)offset_stride, stream);
    if (rc) return rc;
    s0 += k;
  }
  return STAG_OK;
}

int stag_agg_bwd(const stag_csr* csr_t, const stag_plan* plan_t, const float* g, int64_t ldg,
                 int32_t D, const stag_noise_spec* spec, const float* g_scale, const float* row_scale,
                 float* dx, float* dp0_rows, float* dp1_rows, int64_t ldo, void* stream) {
  if (!spec) return STAG_EINVAL;
  if ((dp0_rows == nullptr) != (dp1_rows == nullptr)) return STAG_EINVAL;
  if (spec->in_norm || spec->deriv) return STAG_EINVAL;      // in-norm is not differentiated here
  if (dp0_rows) {
    // parameter derivatives exist for the reparameterised kinds with scalar / per-channel
    // parameters; per-edge (amortised) parameters take their gradients from stag_agg_bwd_w
    if (spec->kind != STAG_NOISE_NORMAL && spec->kind != STAG_NOISE_UNIFORM) return STAG_EINVAL;
    if (spec->param_mode != STAG_PARAM_SCALAR && spec->param_mode != STAG_PARAM_PER_CHANNEL) return STAG_EINVAL;
    if (csr_t && csr_t->n_edges > 0 && !csr_t->nidx) return STAG_EINVAL;   // must redraw the FORWARD's noise
  }
  float* extra[2] = {dp0_rows, dp1_rows};
  return agg_common(csr_t, plan_t, g, ldg, D, spec, STAG_REDUCE_SUM, g_scale, row_scale, dx, ldo, nullptr,
                    dp0_rows ? 3 : 1, extra, 0, 0, stream);
}

int stag_agg_bwd_edge(const stag_csr* csr_t, const stag_plan* plan_t, const float* g, int64_t ldg,
                      int32_t D, const stag_noise_spec* spec, const float* g_scale,
                      const float* row_scale, const float* x, int64_t ldx, float* dx, int64_t ldo,
                      float* dp0_edge, float* dp1_edge, void* stream) {
  if (!spec || !csr_t || ldx < D) return STAG_EINVAL;
  if (csr_t->n_edges > 0 && (!x || !dp0_edge)) return STAG_EINVAL;     // (no edge: nothing per edge to write)
  if (spec->kind != STAG_NOISE_NORMAL && spec->kind != STAG_NOISE_UNIFORM) return STAG_EINVAL;
  if (spec->param_mode != STAG_PARAM_PER_EDGE1 || spec->in_norm || spec->deriv) return STAG_EINVAL;
  // the parameters and their gradients live at edge ids, the noise at forward positions
  if (csr_t->n_edges > 0 && (!csr_t->nidx || !csr_t->eid)) return STAG_EINVAL;
  if (D > 256) return STAG_ENOSYS;      // the channel sum of an edge is one team sum: one channel tile
  const EdgeGradOut eg{x, ldx, dp0_edge, dp1_edge, nullptr};
  return agg_common(csr_t, plan_t, g, ldg, D, spec, STAG_REDUCE_SUM, g_scale, row_scale, dx, ldo, nullptr,
                    1, nullptr, 0, 0, stream, &eg);
}

// ---- stag_agg_bwd_dp: the block partials of agg_dp_kernel -> dp0 [D], dp1 [D], two fixed-order stages ----------
constexpr int kDpSlabs = 256;
// stage 1: block (channel group of 64, derivative i, slab s) adds its slab of the gx block partials: 4 slices of
// the slab side by side, then the slices in order -> part2[s][i][k]
__global__ __launch_bounds__(256) void dp_stage1_kernel(const float* part, int gx, int lpe4, int D, float* part2) {
  __shared__ float red[4][64];
  const int kx = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kx, i = blockIdx.y, slab = blockIdx.z;
  const int per = (gx + kDpSlabs - 1) / kDpSlabs;
  const int b0 = slab * per, b1 = min(gx, b0 + per);
  float sum = 0.f;
  if (k < D) {
    const int tile = k / lpe4, kk = k - tile * lpe4;
    const float* base = part + ((int64_t)tile * gx) * (2 * lpe4) + (int64_t)i * lpe4 + kk;
    for (int b = b0 + slice; b < b1; b += 4) sum += base[(int64_t)b * (2 * lpe4)];
  }
  red[slice][kx] = sum;
  __syncthreads();
  if (slice == 0 && k < D) part2[((int64_t)slab * 2 + i) * D + k] = (red[0][kx] + red[1][kx]) + (red[2][kx] + red[3][kx]);
}
// stage 2: a wave per channel: lane s adds slabs s, s + 64, ... in order, then the wave's butterfly
__global__ __launch_bounds__(256) void dp_stage2_kernel(const float* part2, int D, float* dp0, float* dp1) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, i = blockIdx.y;
  float sum = 0.f;
  if (k < D)
    for (int s = lane; s < kDpSlabs; s += 64) sum += part2[((int64_t)s * 2 + i) * D + k];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
  if (lane == 0 && k < D) (i == 0 ? dp0 : dp1)[k] = sum;
}

}  // extern "C" (reopened below)
namespace stag {
// per-block partials [gx][2][D] (one channel tile) -> dp0 [D], dp1 [D]; part2: [kDpSlabs][2][D] scratch.  Shared with
// the GAT backward (gat.hip: stag_gat_bwd_dp), whose batches leave [2][H] partials.
int dp_reduce_partials(const float* part, int64_t gx, int32_t D, float* part2, float* dp0, float* dp1, hipStream_t s) {
  hipLaunchKernelGGL(dp_stage1_kernel, dim3((D + 63) / 64, 2, kDpSlabs), dim3(256), 0, s, part, (int)gx, D, D, part2);
  hipLaunchKernelGGL(dp_stage2_kernel, dim3((D + 3) / 4, 2), dim3(256), 0, s, part2, D, dp0, dp1);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}
}  // namespace stag
extern "C" {

static void dp_shape(int32_t D, int64_t n_units, int& lpe, int& tiles, int64_t& gx) {
  const int nchunk = (D + 3) / 4;
  lpe = 1;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  tiles = (nchunk + lpe - 1) / lpe;
  const int tpb = STAG_BLOCK_THREADS / lpe;
  gx = (n_units + tpb - 1) / tpb;
}

size_t stag_agg_bwd_dp_workspace_bytes(int64_t n_units, int32_t D) {
  if (n_units <= 0 || D <= 0) return 0;
  int lpe, tiles;
  int64_t gx;
  dp_shape(D, n_units, lpe, tiles, gx);
  return ((size_t)tiles * (size_t)gx * 2u * (size_t)lpe * 4u + (size_t)kDpSlabs * 2u * (size_t)D) * sizeof(float);
}

int stag_agg_bwd_dp(const stag_csr* csr_t, const stag_plan* plan_t, const float* g, int64_t ldg,
                    int32_t D, const stag_noise_spec* spec, const float* g_scale, const float* row_scale,
                    const float* x, int64_t ldx, float* dx, int64_t ldo, float* dp0, float* dp1,
                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!spec || !csr_t || !dp0 || !dp1 || D <= 0) return STAG_EINVAL;
  if (x && ldx < D) return STAG_EINVAL;
  if (spec->kind != STAG_NOISE_NORMAL && spec->kind != STAG_NOISE_UNIFORM) return STAG_EINVAL;
  if (spec->param_mode != STAG_PARAM_SCALAR && spec->param_mode != STAG_PARAM_PER_CHANNEL) return STAG_EINVAL;
  if (spec->in_norm || spec->deriv) return STAG_EINVAL;
  if (csr_t->n_edges > 0 && !csr_t->nidx) return STAG_EINVAL;      // must redraw the FORWARD's noise
  hipStream_t s = (hipStream_t)stream;
  if (csr_t->n_dst == 0 || csr_t->n_edges == 0) {
    if (hipMemsetAsync(dp0, 0, sizeof(float) * D, s) != hipSuccess) return STAG_EIO;
    if (hipMemsetAsync(dp1, 0, sizeof(float) * D, s) != hipSuccess) return STAG_EIO;
    if (dx && csr_t->n_dst > 0 && hipMemset2DAsync(dx, sizeof(float) * ldo, 0, sizeof(float) * D, csr_t->n_dst, s) != hipSuccess)
      return STAG_EIO;
    return STAG_OK;
  }
  const int64_t n_units = (plan_t && plan_t->n_units > 0) ? plan_t->n_units : csr_t->n_dst;
  int lpe, tiles;
  int64_t gx;
  dp_shape(D, n_units, lpe, tiles, gx);
  if (!workspace || workspace_bytes < stag_agg_bwd_dp_workspace_bytes(n_units, D)) return STAG_ENOMEM;
  float* part = static_cast<float*>(workspace);
  float* part2 = part + (size_t)tiles * (size_t)gx * 2u * (size_t)lpe * 4u;
  const EdgeGradOut eg{x, ldx, nullptr, nullptr, part};
  const int rc = agg_common(csr_t, plan_t, g, ldg, D, spec, STAG_REDUCE_SUM, g_scale, row_scale, dx, ldo, nullptr,
                            1, nullptr, 0, 0, stream, &eg);
  if (rc) return rc;
  hipLaunchKernelGGL(dp_stage1_kernel, dim3((D + 63) / 64, 2, kDpSlabs), dim3(256), 0, s, part, (int)gx, lpe * 4, D, part2);
  hipLaunchKernelGGL(dp_stage2_kernel, dim3((D + 3) / 4, 2), dim3(256), 0, s, part2, D, dp0, dp1);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

// plan fields the unit-walking auxiliary kernels need
static int set_units(NoiseArgs& a, const stag_csr* csr, const stag_plan* plan) {
  a.n_units = csr->n_dst;
  if (plan && plan->n_units > 0) {
    if (!plan->units || !aligned16(plan->units)) return STAG_EINVAL;
    if (plan->n_seg > 0 && !plan->long_rows) return STAG_EINVAL;
    a.units = plan->units; a.long_rows = plan->long_rows; a.n_units = plan->n_units;
  }
  return STAG_OK;
}

#define STAG_LPE_DISPATCH(KERNEL, lpe, vec, grid, s, args)                                      \
  do {                                                                                          \
    switch (lpe) {                                                                              \
      case 64: if (vec) hipLaunchKernelGGL((KERNEL<64, true>), grid, dim3(256), 0, s, args);    \
               else     hipLaunchKernelGGL((KERNEL<64, false>), grid, dim3(256), 0, s, args); break; \
      case 32: if (vec) hipLaunchKernelGGL((KERNEL<32, true>), grid, dim3(256), 0, s, args);    \
               else     hipLaunchKernelGGL((KERNEL<32, false>), grid, dim3(256), 0, s, args); break; \
      case 16: if (vec) hipLaunchKernelGGL((KERNEL<16, true>), grid, dim3(256), 0, s, args);    \
               else     hipLaunchKernelGGL((KERNEL<16, false>), grid, dim3(256), 0, s, args); break; \
      case 8:  if (vec) hipLaunchKernelGGL((KERNEL<8, true>), grid, dim3(256), 0, s, args);     \
               else     hipLaunchKernelGGL((KERNEL<8, false>), grid, dim3(256), 0, s, args); break;  \
      case 4:  if (vec) hipLaunchKernelGGL((KERNEL<4, true>), grid, dim3(256), 0, s, args);     \
               else     hipLaunchKernelGGL((KERNEL<4, false>), grid, dim3(256), 0, s, args); break;  \
      case 2:  if (vec) hipLaunchKernelGGL((KERNEL<2, true>), grid, dim3(256), 0, s, args);     \
               else     hipLaunchKernelGGL((KERNEL<2, false>), grid, dim3(256), 0, s, args); break;  \
      default: if (vec) hipLaunchKernelGGL((KERNEL<1, true>), grid, dim3(256), 0, s, args);     \
               else     hipLaunchKernelGGL((KERNEL<1, false>), grid, dim3(256), 0, s, args); break;  \
    }                                                                                           \
  } while (0)

int stag_noise_materialize(const stag_csr* csr, const stag_plan* plan, const stag_noise_spec* spec,
                           int32_t Dn, float* w, int64_t ldw, float* norm_scale, void* stream) {
  int rc = check_csr(csr);
  if (rc) return rc;
  rc = check_spec(spec, csr->n_edges);
  if (rc) return rc;
  if (!w || Dn <= 0 || ldw < Dn) return STAG_EINVAL;
  if (spec->in_norm && !norm_scale) return STAG_EINVAL;   // [n_dst, Dn] scratch for the row factors
  if (csr->n_dst == 0 || csr->n_edges == 0) return STAG_OK;
  if (spec->in_norm) {
    // the in-norm factor indeg / sum_in w of every (row, channel): the aggregation kernel on one
    // broadcast row of ones, which draws the same weights (stag/layers.py:12-28)
    stag_noise_spec sp = *spec;
    sp.deriv = 0;
    // (x = one broadcast row, whose values do not enter the factor: the first Dn floats of w;
    //  out = NULL: only norm_scale_out is written)
    rc = agg_common(csr, plan, w, 0, Dn, &sp, STAG_REDUCE_SUM, nullptr, nullptr, nullptr, Dn, norm_scale,
                    1, nullptr, 0, 0, stream);
    if (rc) return rc;
  }
  NoiseArgs a{};
  a.indptr = csr->indptr; a.eid = csr->eid; a.nidx = csr->nidx; a.n_rows = csr->n_dst;
  a.Dn = Dn; a.kind = spec->kind; a.p0 = spec->p0; a.p1 = spec->p1;
  const bool logs = spec->kind == STAG_NOISE_NORMAL && spec->p1_log;
  a.p0s = spec->p0_scalar; a.p1s = logs ? expf(spec->p1_scalar) : spec->p1_scalar;
  a.pmode = spec->kind >= STAG_NOISE_NORMAL ? spec->param_mode : 0;
  a.nflags = (spec->relu ? kFlagRelu : 0) | (spec->deriv << kDerivShift) | (logs ? kFlagLogScale : 0);
  a.in_norm = spec->in_norm;
  a.key = make_key(spec); a.pos_base = spec->pos_base; a.chunk_base = (uint32_t)spec->chunk_base;
  a.w = w; a.ldw = ldw; a.norm_scale = spec->in_norm ? norm_scale : nullptr;
  rc = set_units(a, csr, plan);
  if (rc) return rc;
  const int nchunk = (Dn + 3) / 4;
  int lpe = 1;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  const bool vec = (Dn % 4 == 0) && (ldw % 4 == 0) && aligned16(w) && (!a.norm_scale || aligned16(norm_scale));
  const dim3 grid((a.n_units + 256 / lpe - 1) / (256 / lpe), (nchunk + lpe - 1) / lpe);
  hipStream_t s = (hipStream_t)stream;
  STAG_LPE_DISPATCH(noise_materialize_kernel, lpe, vec, grid, s, a);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_agg_bwd_w(const stag_csr* csr, const stag_plan* plan, const float* x, int64_t ldx,
                   const float* g, int64_t ldg, int32_t D, const float* src_scale,
                   const stag_noise_spec* spec, int32_t reduce_k, float* dw, float* dw1, int64_t ldw,
                   void* stream) {
  int rc = check_csr(csr);
  if (rc) return rc;
  if (spec) { rc = check_spec(spec, csr->n_edges); if (rc) return rc; }
  if (!x || !g || !dw || D <= 0 || (ldx != 0 && ldx < D) || ldg < D) return STAG_EINVAL;
  if (ldw < (reduce_k ? 1 : D)) return STAG_EINVAL;
  if (dw1 && !(spec && (spec->kind == STAG_NOISE_NORMAL || spec->kind == STAG_NOISE_UNIFORM))) return STAG_EINVAL;
  if (csr->n_dst == 0 || csr->n_edges == 0) return STAG_OK;
  BwdWArgs b{};
  NoiseArgs& a = b.n;
  a.indptr = csr->indptr; a.eid = csr->eid; a.nidx = csr->nidx; a.n_rows = csr->n_dst; a.Dn = D;
  if (spec && spec->kind >= STAG_NOISE_NORMAL && (spec->deriv != 0 || dw1)) {
    a.kind = spec->kind; a.p0 = spec->p0; a.p1 = spec->p1;
    const bool logs = spec->kind == STAG_NOISE_NORMAL && spec->p1_log;
    a.p0s = spec->p0_scalar; a.p1s = logs ? expf(spec->p1_scalar) : spec->p1_scalar; a.pmode = spec->param_mode;
    a.nflags = (spec->relu ? kFlagRelu : 0) | ((dw1 ? 0 : spec->deriv) << kDerivShift) | (logs ? kFlagLogScale : 0);
    a.key = make_key(spec); a.pos_base = spec->pos_base; a.chunk_base = (uint32_t)spec->chunk_base;
  }
  a.w = dw; a.ldw = ldw;
  rc = set_units(a, csr, plan);
  if (rc) return rc;
  b.indices = csr->indices; b.x = x; b.ldx = ldx; b.g = g; b.ldg = ldg; b.src_scale = src_scale;
  b.reduce_k = reduce_k ? 1 : 0; b.w1 = dw1;
  const int nchunk = (D + 3) / 4;
  int lpe = 1;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (ldg % 4 == 0) && aligned16(x) && aligned16(g);
  if (!reduce_k) vec = vec && (ldw % 4 == 0) && aligned16(dw) && (!dw1 || aligned16(dw1));
  if (a.kind >= kNormal && a.pmode == STAG_PARAM_PER_CHANNEL) vec = vec && aligned16(a.p0) && (!a.p1 || aligned16(a.p1));
  const dim3 grid((a.n_units + 256 / lpe - 1) / (256 / lpe));
  hipStream_t s = (hipStream_t)stream;
  if (dw1) STAG_LPE_DISPATCH(agg_bwd_w_kernel2, lpe, vec, grid, s, b);
  else if (a.kind >= kNormal) STAG_LPE_DISPATCH(agg_bwd_w_kernel1, lpe, vec, grid, s, b);
  else STAG_LPE_DISPATCH(agg_bwd_w_kernel0, lpe, vec, grid, s, b);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_segment_reduce(const float* x, int64_t ldx, int32_t D, const int32_t* offsets,
                        int32_t n_seg, int32_t reduce, float* out, int64_t ldo, void* stream) {
  if (!offsets || !out || D <= 0 || n_seg < 0 || ldx < D || ldo < D) return STAG_EINVAL;
  if (reduce != STAG_REDUCE_SUM && reduce != STAG_REDUCE_MEAN) return STAG_EINVAL;
  if (n_seg == 0) return STAG_OK;
  if (!x) return STAG_EINVAL;
  const int nchunk = (D + 3) / 4;
  int lpe = 1;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(x) && aligned16(out);
  const dim3 grid((n_seg + 256 / lpe - 1) / (256 / lpe), (nchunk + lpe - 1) / lpe);
  const int m = reduce == STAG_REDUCE_MEAN ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
#define STAG_SEG_LAUNCH(L)                                                                          \
  do {                                                                                              \
    if (vec) hipLaunchKernelGGL((segment_reduce_kernel<L, true>), grid, dim3(256), 0, s, x, ldx, D, \
                                offsets, n_seg, m, out, ldo);                                       \
    else     hipLaunchKernelGGL((segment_reduce_kernel<L, false>), grid, dim3(256), 0, s, x, ldx, D,\
                                offsets, n_seg, m, out, ldo);                                       \
  } while (0)
  switch (lpe) {
    case 64: STAG_SEG_LAUNCH(64); break;
    case 32: STAG_SEG_LAUNCH(32); break;
    case 16: STAG_SEG_LAUNCH(16); break;
    case 8: STAG_SEG_LAUNCH(8); break;
    case 4: STAG_SEG_LAUNCH(4); break;
    case 2: STAG_SEG_LAUNCH(2); break;
    default: STAG_SEG_LAUNCH(1); break;
  }
#undef STAG_SEG_LAUNCH
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_gather_rows(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t width,
                     float* out, int64_t ldo, void* stream) {
  if (n < 0 || width <= 0 || ldx < width || ldo < width) return STAG_EINVAL;
  if (n == 0) return STAG_OK;
  if (!x || !idx || !out) return STAG_EINVAL;
  const int nchunk = (width + 3) / 4;
  const int64_t threads = n * nchunk;
  if ((threads + 255) / 256 >= (1ll << 31)) return STAG_ENOSYS;
  const bool vec = (width % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(x) && aligned16(out);
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (vec) hipLaunchKernelGGL((gather_rows_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, idx, n, width, nchunk, out, ldo);
  else     hipLaunchKernelGGL((gather_rows_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, idx, n, width, nchunk, out, ldo);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

size_t stag_coldot_workspace_bytes(int32_t D) {
  return D > 0 ? (size_t)kColdotBlocks * 2u * (size_t)D * sizeof(float) : 0;
}

int stag_coldot(const float* x, int64_t ldx, const float* t0, const float* t1, int64_t ldt,
                int64_t n_rows, int32_t D, float* out0, float* out1, void* workspace,
                size_t workspace_bytes, void* stream) {
  if (D <= 0 || n_rows < 0 || ldx < D || ldt < D || !out0 || (t1 != nullptr) != (out1 != nullptr)) return STAG_EINVAL;
  if (n_rows > 0 && (!x || !t0)) return STAG_EINVAL;
  if (!workspace || workspace_bytes < stag_coldot_workspace_bytes(D)) return STAG_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (ldt % 4 == 0) && aligned16(x) && aligned16(t0) &&
                   (!t1 || aligned16(t1));
  const int elems = vec ? D / 4 : D;
  int cw = 1;
  while (cw < elems && cw < 256) cw <<= 1;
  const int R = 256 / cw;
  const int nb = (int)std::min<int64_t>(kColdotBlocks, std::max<int64_t>(1, (n_rows + R - 1) / R));
  const dim3 grid(nb, (elems + cw - 1) / cw);
  float* part = static_cast<float*>(workspace);
  if (vec) hipLaunchKernelGGL((coldot_partial_kernel<4>), grid, dim3(256), 0, s, x, ldx, t0, t1, ldt, n_rows, D, cw, part);
  else     hipLaunchKernelGGL((coldot_partial_kernel<1>), grid, dim3(256), 0, s, x, ldx, t0, t1, ldt, n_rows, D, cw, part);
  hipLaunchKernelGGL(coldot_final_kernel, dim3((D + 15) / 16), dim3(256), 0, s, part, nb, D, out0, out1);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

}  // extern "C"
