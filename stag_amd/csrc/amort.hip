// Amortised per-edge parameters with NARROW heads (include/stag_hip.h: stag_node_project_* / stag_edge_mlp_* /
// stag_normal_kl_*).
//
// What the reference computes (stag/distributions.py:178-191, 225-242) for AmortizedDistribution(in, 1) — the
// form every scripts/*_rec/run.py builds; hidden_features defaults to out_features = 1:
//     h_e   = SiLU(W_e [feat[src] || feat[dst]] + b_e)            [E, hidden]
//     par_c = W_c h_e + b_c            (loc, log_scale, ...)      [E, 1] each
// and then KL(N(loc, exp(log_scale)) || prior).mean() over the edges (stag/layers.py:132-145).
// As dense torch that is a chain of degenerate GEMMs (N x 128 x 1, E x 1 x 2) and ~40 elementwise launches over
// [E, 1] tensors: 1.2 ms of a 2.0 ms layer step on MI355X.  Here:
//   node_project : P = feat [N, K] . W [K, C] + b, C = 2 hidden <= 16 columns, ONE pass over feat (both halves of
//                  W_e at once: the concatenation is a sum of two projections), and its backward (dx, dW, db) in
//                  one pass over feat as well;
//   edge_mlp     : a thread per edge gathers the two projected rows, applies SiLU and the heads;
//                  backward: d pre [E, hidden] (the caller segment-sums it by source and by destination with the
//                  aggregation kernel) and the head gradients by a fixed-order two-stage reduction;
//   normal_kl    : the KL mean and its gradients, one pass each.
// Every reduction has a fixed order (bit-identical from run to run).
#include <hip/hip_runtime.h>

#include "../../include/stag_hip.h"
#include "agg_kernel.hpp"

namespace {
using stag::load4;
using stag::team_sum;

constexpr int kRedBlocks = 512;     // first-stage blocks of every reduction here
constexpr int kMaxC = 16, kMaxHidden = 8, kMaxPar = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// ---- y[n, c] = sum_k x[n, k] w[k, c] + b[c] ---------------------------------------------------------------
// A team of LPE lanes per kFwdRows consecutive rows (that many row loads in flight per lane), 4 columns of x per
// lane and tile; the lane's 4 x C slice of w sits in registers.
constexpr int kFwdRows = 4, kBwdRows = 4;
template <int LPE, int CT>
__global__ __launch_bounds__(256) void node_project_fwd_kernel(const float* x, int64_t ldx, int n, int K,
                                                               const float* w, const float* b, int C, bool vec,
                                                               float* y) {
  constexpr int R = kFwdRows;
  const int c = threadIdx.x % LPE;
  const int row0 = (blockIdx.x * (256 / LPE) + threadIdx.x / LPE) * R;     // every lane stays for the team sums
  float acc[R][CT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) acc[r][cc] = 0.f;
  for (int kt = 0; kt < K; kt += LPE * 4) {
    const int k0 = kt + c * 4;
    float wv[4][CT], xv[R][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) wv[q][cc] = (k0 + q < K && cc < C) ? w[(k0 + q) * C + cc] : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      xv[r][0] = xv[r][1] = xv[r][2] = xv[r][3] = 0.f;
      if (row0 + r < n && k0 < K) load4(x + (int64_t)(row0 + r) * ldx, k0, K, vec, xv[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) acc[r][cc] = __builtin_fmaf(xv[r][q], wv[q][cc], acc[r][cc]);
  }
  const float bias = (b && c < C) ? b[c] : 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float mine = 0.f;
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      const float s = team_sum<LPE>(acc[r][cc]);
      if (cc == c) mine = s;
    }
    if (row0 + r < n && c < C) y[(int64_t)(row0 + r) * C + c] = mine + bias;
  }
}

// ---- backward: dx[n, k] = sum_c gy[n, c] w[k, c];  dw[k, c] = sum_n x[n, k] gy[n, c];  db[c] = sum_n gy[n, c]
// Team t of T walks rows t, t + T, ...; a lane keeps the 4 x C partial sums of its columns; the teams of a block
// are added through LDS in team order; one partial [(K + 1) x C] per block; stage 2 adds the blocks in order.
template <int LPE, int CT>
__global__ __launch_bounds__(256) void node_project_bwd_kernel(const float* x, int64_t ldx, int n, int K,
                                                               const float* w, int C, const float* gy, bool vec,
                                                               float* dx, int64_t lddx, float* part) {
  constexpr int TEAMS = 256 / LPE;
  __shared__ float s_red[TEAMS][LPE * 4 * CT];
  __shared__ float s_db[TEAMS][CT];
  const int c = threadIdx.x % LPE, tm = threadIdx.x / LPE;
  const int T = gridDim.x * TEAMS;
  const int t0 = blockIdx.x * TEAMS + tm;
  float* mypart = part + (int64_t)blockIdx.x * (K + 1) * C;
  float dbacc[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) dbacc[cc] = 0.f;
  for (int kt = 0; kt < K; kt += LPE * 4) {
    const int k0 = kt + c * 4;
    float wv[4][CT], dwacc[4][CT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        wv[q][cc] = (k0 + q < K && cc < C) ? w[(k0 + q) * C + cc] : 0.f;
        dwacc[q][cc] = 0.f;
      }
    // kBwdRows of the team's rows per trip (that many row loads in flight per lane), folded in row order
    constexpr int RB = CT <= 4 ? kBwdRows : 1;
    for (int row0 = t0; row0 < n; row0 += RB * T) {
      float g[RB][CT], xv[RB][4];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = row0 + r * T;
        xv[r][0] = xv[r][1] = xv[r][2] = xv[r][3] = 0.f;
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) g[r][cc] = (cc < C && row < n) ? gy[(int64_t)row * C + cc] : 0.f;
        if (k0 < K && row < n) load4(x + (int64_t)row * ldx, k0, K, vec, xv[r]);
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = row0 + r * T;
        if (row >= n) break;
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int cc = 0; cc < CT; ++cc) {
            d[q] = __builtin_fmaf(g[r][cc], wv[q][cc], d[q]);
            dwacc[q][cc] = __builtin_fmaf(xv[r][q], g[r][cc], dwacc[q][cc]);
          }
        if (dx && k0 < K) stag::store4(dx + (int64_t)row * lddx, k0, K, vec, d);
        if (kt == 0 && c == 0) {
#pragma unroll
          for (int cc = 0; cc < CT; ++cc) dbacc[cc] += g[r][cc];
        }
      }
    }
    // the block's teams, in team order
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) s_red[tm][(c * 4 + q) * CT + cc] = dwacc[q][cc];
    __syncthreads();
    for (int i = threadIdx.x; i < LPE * 4 * CT; i += 256) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < TEAMS; ++j) s += s_red[j][i];
      const int k = kt + i / CT, cc = i % CT;
      if (k < K && cc < C) mypart[k * C + cc] = s;
    }
    __syncthreads();
  }
  if (c == 0) {
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) s_db[tm][cc] = dbacc[cc];
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < TEAMS; ++j) s += s_db[j][threadIdx.x];
    mypart[K * C + threadIdx.x] = s;
  }
}

// sum_b part[b * nv + i] over the nb first-stage partials by one block of 256 threads: thread t adds partials
// t, t + 256, ... in order, then the wave butterflies and the four waves in order.  Fixed for a given nb.
__device__ __forceinline__ float block_sum_partials(const float* part, int nb, int nv, int i) {
  __shared__ float s_w[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256) s += part[(int64_t)b * nv + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  return (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// out[i] = scale * sum_b part[b][i]: block i of the grid owns value i
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* part, int nb, int nv, float* out0, int n0,
                                                           float* out1, const float* scale_dev, float scale) {
  const int i = blockIdx.x;
  float s = block_sum_partials(part, nb, nv, i);
  if (threadIdx.x != 0) return;
  s *= scale * (scale_dev ? scale_dev[0] : 1.0f);
  if (i < n0) { if (out0) out0[i] = s; }
  else if (out1) out1[i - n0] = s;
}

// ---- per-head dots: y[c][n][g] = sum_f x[n, g F + f] w[c][g F + f]  (GAT's el / er, stag/zoo/gat.py:109-110) ------
// The block-diagonal special case of node_project: a lane's 4 columns belong to ONE head, so a row costs CG dots
// per lane and a sum over the F / 4 lanes of the head.  Same team / row walk as above.
__device__ __forceinline__ float lanes_sum(float v, int nl) {     // aligned groups of nl = 2^i lanes, all alive
  if (nl > 1) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  if (nl > 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  if (nl > 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  if (nl > 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  if (nl > 16) v += __shfl_xor(v, 16);
  if (nl > 32) v += __shfl_xor(v, 32);
  return v;
}

// lane L of a team -> (head, chunk of 4 channels): a head takes gl = F / 4 rounded up to a power of two lanes (the
// head sums are butterflies); lanes past a head's F channels idle (F = 40: 10 of 16).  F / 4 a power of two: k0 = 4 L.
__device__ __forceinline__ void head_lane(int L, int G, int F, int gl, int& k0, bool& in, int& head) {
  head = L / gl;
  const int j = L - head * gl;
  in = head < G && 4 * j < F;
  k0 = in ? head * F + 4 * j : 0;
}

template <int LPE, int CG>
__global__ __launch_bounds__(256) void head_dot_fwd_kernel(const float* x, int64_t ldx, int n, int K, int F, int gl,
                                                           const float* w, float* y) {
  constexpr int R = kFwdRows;
  const int c = threadIdx.x % LPE;
  const int row0 = (blockIdx.x * (256 / LPE) + threadIdx.x / LPE) * R;
  const int G = K / F;                              // heads; gl lanes per head
  for (int L0 = 0; L0 < G * gl; L0 += LPE) {
    int k0, hd;
    bool in;
    head_lane(L0 + c, G, F, gl, k0, in, hd);
    float wv[CG][4], xv[R][4];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) {
      wv[cc][0] = wv[cc][1] = wv[cc][2] = wv[cc][3] = 0.f;
      if (in) load4(w + (int64_t)cc * K, k0, K, true, wv[cc]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      xv[r][0] = xv[r][1] = xv[r][2] = xv[r][3] = 0.f;
      if (row0 + r < n && in) load4(x + (int64_t)(row0 + r) * ldx, k0, K, true, xv[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) {
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) d = __builtin_fmaf(xv[r][q], wv[cc][q], d);
        d = lanes_sum(d, gl);
        if (in && row0 + r < n && k0 == hd * F) y[((int64_t)cc * n + (row0 + r)) * G + hd] = d;
      }
  }
}

// dx[n, k] = sum_c gy[c][n][g(k)] w[c][k];  dw[c][k] = sum_n gy[c][n][g(k)] x[n, k]: one pass over x
template <int LPE, int CG>
__global__ __launch_bounds__(256) void head_dot_bwd_kernel(const float* x, int64_t ldx, int n, int K, int F, int gl,
                                                           const float* w, const float* gy, float* dx, int64_t lddx,
                                                           float* part) {
  constexpr int TEAMS = 256 / LPE;
  __shared__ float s_red[TEAMS][LPE * 4 * CG];
  const int c = threadIdx.x % LPE, tm = threadIdx.x / LPE;
  const int T = gridDim.x * TEAMS;
  const int t0 = blockIdx.x * TEAMS + tm;
  const int G = K / F;
  float* mypart = part + (int64_t)blockIdx.x * CG * K;
  for (int L0 = 0; L0 < G * gl; L0 += LPE) {
    int k0, g;
    bool in;
    head_lane(L0 + c, G, F, gl, k0, in, g);
    float wv[CG][4], dwacc[CG][4];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) {
      wv[cc][0] = wv[cc][1] = wv[cc][2] = wv[cc][3] = 0.f;
      if (in) load4(w + (int64_t)cc * K, k0, K, true, wv[cc]);
      dwacc[cc][0] = dwacc[cc][1] = dwacc[cc][2] = dwacc[cc][3] = 0.f;
    }
    if (in) {
      constexpr int RB = kBwdRows;       // rows of the team per trip, folded in row order
      for (int row0 = t0; row0 < n; row0 += RB * T) {
        float gv[RB][CG], xv[RB][4];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int row = row0 + r * T;
          if (row < n) {
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) gv[r][cc] = gy[((int64_t)cc * n + row) * G + g];
            load4(x + (int64_t)row * ldx, k0, K, true, xv[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int row = row0 + r * T;
          if (row >= n) break;
          float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < CG; ++cc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              d[q] = __builtin_fmaf(gv[r][cc], wv[cc][q], d[q]);
              dwacc[cc][q] = __builtin_fmaf(gv[r][cc], xv[r][q], dwacc[cc][q]);
            }
          if (dx) stag::store4(dx + (int64_t)row * lddx, k0, K, true, d);
        }
      }
    }
#pragma unroll
    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
      for (int q = 0; q < 4; ++q) s_red[tm][(cc * LPE + c) * 4 + q] = dwacc[cc][q];
    __syncthreads();
    for (int i = threadIdx.x; i < LPE * 4 * CG; i += 256) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < TEAMS; ++j) s += s_red[j][i];
      const int cc = i / (LPE * 4), lane = (i % (LPE * 4)) >> 2, q = i & 3;
      int kk, hh;
      bool kin;
      head_lane(L0 + lane, G, F, gl, kk, kin, hh);
      if (kin) mypart[(int64_t)cc * K + kk + q] = s;
    }
    __syncthreads();
  }
}

// ---- per-edge heads -----------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int HT>
__global__ __launch_bounds__(256) void edge_mlp_fwd_kernel(const int32_t* src, const int32_t* dst, int64_t E,
                                                           const float* ps, const float* pd, int ldp, int hidden,
                                                           const float* wh, const float* bh, int P, float* par) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int u = src[e], v = dst[e];
  float out[kMaxPar];
#pragma unroll
  for (int p = 0; p < kMaxPar; ++p) out[p] = (p < P && bh) ? bh[p] : 0.f;
#pragma unroll
  for (int j = 0; j < HT; ++j) {
    if (j < hidden) {
      const float pre = ps[(int64_t)u * ldp + j] + pd[(int64_t)v * ldp + j];
      const float h = pre * sigmoidf(pre);
#pragma unroll
      for (int p = 0; p < kMaxPar; ++p)
        if (p < P) out[p] = __builtin_fmaf(h, wh[j * P + p], out[p]);
    }
  }
#pragma unroll
  for (int p = 0; p < kMaxPar; ++p)
    if (p < P) par[(int64_t)p * E + e] = out[p];
}

// d pre[e, j] and the block partials of d wh [hidden x P] | d bh [P]; thread i of the grid walks edges
// i, i + total, ... (fixed), waves and blocks are added in order
template <int HT>
__global__ __launch_bounds__(256) void edge_mlp_bwd_kernel(const int32_t* src, const int32_t* dst, int64_t E,
                                                           const float* ps, const float* pd, int ldp, int hidden,
                                                           const float* wh, int P, const float* gpar, float* dpre,
                                                           float* part) {
  constexpr int NV = HT * kMaxPar + kMaxPar;
  __shared__ float s_w[4][NV];
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  const int64_t total = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += total) {
    const int u = src[e], v = dst[e];
    float g[kMaxPar];
#pragma unroll
    for (int p = 0; p < kMaxPar; ++p) {
      g[p] = p < P ? gpar[(int64_t)p * E + e] : 0.f;
      acc[HT * kMaxPar + p] += g[p];
    }
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      if (j < hidden) {
        const float pre = ps[(int64_t)u * ldp + j] + pd[(int64_t)v * ldp + j];
        const float sg = sigmoidf(pre);
        const float h = pre * sg;
        float dh = 0.f;
#pragma unroll
        for (int p = 0; p < kMaxPar; ++p) {
          if (p < P) dh = __builtin_fmaf(g[p], wh[j * P + p], dh);
          acc[j * kMaxPar + p] = __builtin_fmaf(h, g[p], acc[j * kMaxPar + p]);
        }
        dpre[e * hidden + j] = dh * (sg * (1.0f + pre * (1.0f - sg)));     // SiLU'(pre)
      }
    }
  }
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) s_w[wv][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const float s = (s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + (s_w[2][threadIdx.x] + s_w[3][threadIdx.x]);
    part[(int64_t)blockIdx.x * NV + threadIdx.x] = s;
  }
}

// partials [nb][HT * kMaxPar + kMaxPar] -> dwh [hidden x P], dbh [P]; block i owns value i
__global__ __launch_bounds__(256) void edge_mlp_final_kernel(const float* part, int nb, int HT, int hidden, int P,
                                                             float* dwh, float* dbh) {
  const int i = blockIdx.x;
  const int NV = HT * kMaxPar + kMaxPar;
  const int j = i / kMaxPar, p = i % kMaxPar;
  if (p >= P || (j < HT && j >= hidden)) return;          // block-uniform
  const float s = block_sum_partials(part, nb, NV, i);
  if (threadIdx.x != 0) return;
  if (j < HT) { if (dwh) dwh[j * P + p] = s; }
  else if (dbh) dbh[p] = s;
}

// ---- KL(N(m1, exp(l1)) || N(m2, s2)) per element, m2 / s2 one-element device tensors -----------------------
//   kl = 0.5 (vr + t1 - 1 - log vr),  vr = (s1 / s2)^2,  t1 = ((m1 - m2) / s2)^2       (torch's _kl_normal_normal)
__global__ __launch_bounds__(256) void normal_kl_fwd_kernel(const float* loc, const float* ls, int64_t n,
                                                            const float* p_loc, const float* p_scale, float* part) {
  __shared__ float s_w[4];
  const float m2 = p_loc[0], s2 = p_scale[0];
  const float inv = 1.0f / s2, lg2 = logf(s2);
  float acc = 0.f;
  const int64_t total = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += total) {
    const float l1 = ls[i];
    const float r = expf(l1) * inv, d = (loc[i] - m2) * inv;
    acc += 0.5f * (r * r + d * d - 1.0f - 2.0f * (l1 - lg2));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// d kl_mean: d loc[i] = c (m1 - m2) / s2^2, d ls[i] = c (vr - 1), c = g / n; partials of sum_i (m1 - m2) / s2^2
// and sum_i (1 - vr - t1) / s2 for the prior
__global__ __launch_bounds__(256) void normal_kl_bwd_kernel(const float* loc, const float* ls, int64_t n,
                                                            const float* p_loc, const float* p_scale,
                                                            const float* g, float* dloc, float* dls, float* part) {
  __shared__ float s_w[4][2];
  const float m2 = p_loc[0], s2 = p_scale[0];
  const float inv = 1.0f / s2, cg = g[0] / (float)n;
  float a0 = 0.f, a1 = 0.f;
  const int64_t total = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += total) {
    const float r = expf(ls[i]) * inv, d = (loc[i] - m2) * inv;
    const float vr = r * r, t1 = d * d;
    if (dloc) dloc[i] = cg * (d * inv);
    if (dls) dls[i] = cg * (vr - 1.0f);
    a0 += d * inv;
    a1 += (1.0f - vr - t1) * inv;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1);
  if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6][0] = a0; s_w[threadIdx.x >> 6][1] = a1; }
  __syncthreads();
  if (threadIdx.x < 2)
    part[(int64_t)blockIdx.x * 2 + threadIdx.x] =
        (s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + (s_w[2][threadIdx.x] + s_w[3][threadIdx.x]);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int lpe_for(int K, int CT) {
  int lpe = 8;
  while (lpe * 4 < K && lpe < 64) lpe <<= 1;
  while (lpe < CT) lpe <<= 1;
  return lpe;
}
inline int ct_for(int C) { return C <= 2 ? 2 : C <= 4 ? 4 : C <= 8 ? 8 : 16; }
inline int ht_for(int h) { return h <= 1 ? 1 : h <= 2 ? 2 : h <= 4 ? 4 : 8; }

#define STAG_NP_DISPATCH(KERNEL, lpe, ct, grid, s, ...)                                                       \
  do {                                                                                                        \
    switch (lpe * 100 + ct) {                                                                                 \
      case 802:  hipLaunchKernelGGL((KERNEL<8, 2>), grid, dim3(256), 0, s, __VA_ARGS__); break;               \
      case 804:  hipLaunchKernelGGL((KERNEL<8, 4>), grid, dim3(256), 0, s, __VA_ARGS__); break;               \
      case 808:  hipLaunchKernelGGL((KERNEL<8, 8>), grid, dim3(256), 0, s, __VA_ARGS__); break;               \
      case 1602: hipLaunchKernelGGL((KERNEL<16, 2>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 1604: hipLaunchKernelGGL((KERNEL<16, 4>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 1608: hipLaunchKernelGGL((KERNEL<16, 8>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 1616: hipLaunchKernelGGL((KERNEL<16, 16>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 3202: hipLaunchKernelGGL((KERNEL<32, 2>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 3204: hipLaunchKernelGGL((KERNEL<32, 4>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 3208: hipLaunchKernelGGL((KERNEL<32, 8>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 3216: hipLaunchKernelGGL((KERNEL<32, 16>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 6402: hipLaunchKernelGGL((KERNEL<64, 2>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 6404: hipLaunchKernelGGL((KERNEL<64, 4>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      case 6408: hipLaunchKernelGGL((KERNEL<64, 8>), grid, dim3(256), 0, s, __VA_ARGS__); break;              \
      default:   hipLaunchKernelGGL((KERNEL<64, 16>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
    }                                                                                                         \
  } while (0)

#define STAG_HT_DISPATCH(KERNEL, ht, grid, s, ...)                                                            \
  do {                                                                                                        \
    switch (ht) {                                                                                             \
      case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, s, __VA_ARGS__); break;                     \
      case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, s, __VA_ARGS__); break;                     \
      case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), 0, s, __VA_ARGS__); break;                     \
      default: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), 0, s, __VA_ARGS__); break;                    \
    }                                                                                                         \
  } while (0)

}  // namespace

extern "C" {

size_t stag_amort_workspace_bytes(int32_t n_values) {
  return n_values > 0 ? (size_t)kRedBlocks * (size_t)n_values * sizeof(float) : 0;
}

int stag_node_project_fwd(const float* x, int64_t ldx, int64_t n_rows, int32_t K, const float* w, const float* b,
                          int32_t C, float* y, void* stream) {
  if (n_rows < 0 || K <= 0 || C <= 0 || C > kMaxC || n_rows >= (1ll << 31)) return STAG_EINVAL;
  if (n_rows == 0) return STAG_OK;
  if (!x || !w || !y || ldx < K) return STAG_EINVAL;
  const int ct = ct_for(C), lpe = lpe_for(K, ct);
  const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
  const int64_t rows_per_block = (int64_t)(256 / lpe) * kFwdRows;
  const dim3 grid((unsigned)((n_rows + rows_per_block - 1) / rows_per_block));
  hipStream_t s = (hipStream_t)stream;
  STAG_NP_DISPATCH(node_project_fwd_kernel, lpe, ct, grid, s, x, ldx, (int)n_rows, K, w, b, C, vec, y);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_node_project_bwd(const float* x, int64_t ldx, int64_t n_rows, int32_t K, const float* w, int32_t C,
                          const float* gy, float* dx, int64_t lddx, float* dw, float* db, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (n_rows < 0 || K <= 0 || C <= 0 || C > kMaxC || n_rows >= (1ll << 31)) return STAG_EINVAL;
  if (n_rows > 0 && (ldx < K || (dx && lddx < K))) return STAG_EINVAL;
  if (!dw && !db && !dx) return STAG_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nv = (K + 1) * C;
  if (n_rows == 0) {
    if (dw && hipMemsetAsync(dw, 0, sizeof(float) * K * C, s) != hipSuccess) return STAG_EIO;
    if (db && hipMemsetAsync(db, 0, sizeof(float) * C, s) != hipSuccess) return STAG_EIO;
    return STAG_OK;
  }
  if (!x || !w || !gy) return STAG_EINVAL;
  const int ct = ct_for(C), lpe = lpe_for(K, ct);
  // fewer first-stage blocks when a partial is large (K = 1433, C = 16: 92 KB each)
  int nb = kRedBlocks;
  while (nb > 16 && (size_t)nb * nv * sizeof(float) > (32u << 20)) nb >>= 1;
  while (nb > 1 && (int64_t)(nb / 2) * (256 / lpe) >= n_rows) nb >>= 1;    // no more teams than rows
  if (!workspace || workspace_bytes < (size_t)nb * nv * sizeof(float)) return STAG_ENOMEM;
  const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(x) && (!dx || (aligned16(dx) && lddx % 4 == 0));
  float* part = static_cast<float*>(workspace);
  STAG_NP_DISPATCH(node_project_bwd_kernel, lpe, ct, dim3(nb), s, x, ldx, (int)n_rows, K, w, C, gy, vec, dx, lddx, part);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(nv), dim3(256), 0, s, part, nb, nv, dw, K * C, db,
                     (const float*)nullptr, 1.0f);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_head_dot_fwd(const float* x, int64_t ldx, int64_t n_rows, int32_t G, int32_t F, const float* w, int32_t C,
                      float* y, void* stream) {
  if (n_rows < 0 || G <= 0 || F < 4 || F > 256 || F % 4 != 0 || C < 1 || C > 2 || n_rows >= (1ll << 31))
    return STAG_EINVAL;
  if (n_rows == 0) return STAG_OK;
  const int K = G * F;
  if (!x || !w || !y || ldx < K || ldx % 4 != 0 || !aligned16(x) || !aligned16(w)) return STAG_EINVAL;
  int gl = 1;                                     // lanes per head: F / 4 rounded up to a power of two
  while (gl * 4 < F) gl <<= 1;
  const int lpe = lpe_for(G * gl * 4, 1) < gl ? gl : lpe_for(G * gl * 4, 1);
  const int64_t rows_per_block = (int64_t)(256 / lpe) * kFwdRows;
  const dim3 grid((unsigned)((n_rows + rows_per_block - 1) / rows_per_block));
  hipStream_t s = (hipStream_t)stream;
#define STAG_HD_FWD(L)                                                                                          \
  do {                                                                                                          \
    if (C == 1) hipLaunchKernelGGL((head_dot_fwd_kernel<L, 1>), grid, dim3(256), 0, s, x, ldx, (int)n_rows, K, F, gl, w, y); \
    else        hipLaunchKernelGGL((head_dot_fwd_kernel<L, 2>), grid, dim3(256), 0, s, x, ldx, (int)n_rows, K, F, gl, w, y); \
  } while (0)
  switch (lpe) {
    case 8: STAG_HD_FWD(8); break;
    case 16: STAG_HD_FWD(16); break;
    case 32: STAG_HD_FWD(32); break;
    default: STAG_HD_FWD(64); break;
  }
#undef STAG_HD_FWD
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_head_dot_bwd(const float* x, int64_t ldx, int64_t n_rows, int32_t G, int32_t F, const float* w, int32_t C,
                      const float* gy, float* dx, int64_t lddx, float* dw, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (n_rows < 0 || G <= 0 || F < 4 || F > 256 || F % 4 != 0 || C < 1 || C > 2 || n_rows >= (1ll << 31))
    return STAG_EINVAL;
  if (!dx && !dw) return STAG_EINVAL;
  const int K = G * F, nv = C * K;
  hipStream_t s = (hipStream_t)stream;
  if (n_rows == 0) {
    if (dw && hipMemsetAsync(dw, 0, sizeof(float) * nv, s) != hipSuccess) return STAG_EIO;
    return STAG_OK;
  }
  if (!x || !w || !gy || ldx < K || ldx % 4 != 0 || !aligned16(x) || !aligned16(w)) return STAG_EINVAL;
  if (dx && (lddx < K || lddx % 4 != 0 || !aligned16(dx))) return STAG_EINVAL;
  int gl = 1;
  while (gl * 4 < F) gl <<= 1;
  const int lpe = lpe_for(G * gl * 4, 1) < gl ? gl : lpe_for(G * gl * 4, 1);
  int nb = kRedBlocks;
  while (nb > 16 && (size_t)nb * nv * sizeof(float) > (32u << 20)) nb >>= 1;
  while (nb > 1 && (int64_t)(nb / 2) * (256 / lpe) >= n_rows) nb >>= 1;
  if (!workspace || workspace_bytes < (size_t)nb * nv * sizeof(float)) return STAG_ENOMEM;
  float* part = static_cast<float*>(workspace);
#define STAG_HD_BWD(L)                                                                                          \
  do {                                                                                                          \
    if (C == 1) hipLaunchKernelGGL((head_dot_bwd_kernel<L, 1>), dim3(nb), dim3(256), 0, s, x, ldx, (int)n_rows, K, F, gl, w, gy, dx, lddx, part); \
    else        hipLaunchKernelGGL((head_dot_bwd_kernel<L, 2>), dim3(nb), dim3(256), 0, s, x, ldx, (int)n_rows, K, F, gl, w, gy, dx, lddx, part); \
  } while (0)
  switch (lpe) {
    case 8: STAG_HD_BWD(8); break;
    case 16: STAG_HD_BWD(16); break;
    case 32: STAG_HD_BWD(32); break;
    default: STAG_HD_BWD(64); break;
  }
#undef STAG_HD_BWD
  if (dw)
    hipLaunchKernelGGL(reduce_final_kernel, dim3(nv), dim3(256), 0, s, part, nb, nv, dw, nv, (float*)nullptr,
                       (const float*)nullptr, 1.0f);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_edge_mlp_fwd(const int32_t* src, const int32_t* dst, int64_t n_edges, const float* ps, const float* pd,
                      int64_t ldp, int32_t hidden, const float* wh, const float* bh, int32_t n_par, float* par,
                      void* stream) {
  if (n_edges < 0 || hidden <= 0 || hidden > kMaxHidden || n_par <= 0 || n_par > kMaxPar || ldp < hidden)
    return STAG_EINVAL;
  if (n_edges == 0) return STAG_OK;
  if (!src || !dst || !ps || !pd || !wh || !par || ldp >= (1 << 30)) return STAG_EINVAL;
  const dim3 grid((unsigned)((n_edges + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  STAG_HT_DISPATCH(edge_mlp_fwd_kernel, ht_for(hidden), grid, s, src, dst, n_edges, ps, pd, (int)ldp, hidden, wh, bh,
                   n_par, par);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_edge_mlp_bwd(const int32_t* src, const int32_t* dst, int64_t n_edges, const float* ps, const float* pd,
                      int64_t ldp, int32_t hidden, const float* wh, int32_t n_par, const float* gpar, float* dpre,
                      float* dwh, float* dbh, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_edges < 0 || hidden <= 0 || hidden > kMaxHidden || n_par <= 0 || n_par > kMaxPar || ldp < hidden)
    return STAG_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n_edges == 0) {
    if (dwh && hipMemsetAsync(dwh, 0, sizeof(float) * hidden * n_par, s) != hipSuccess) return STAG_EIO;
    if (dbh && hipMemsetAsync(dbh, 0, sizeof(float) * n_par, s) != hipSuccess) return STAG_EIO;
    return STAG_OK;
  }
  if (!src || !dst || !ps || !pd || !wh || !gpar || !dpre || ldp >= (1 << 30)) return STAG_EINVAL;
  const int ht = ht_for(hidden), nvp = ht * kMaxPar + kMaxPar;
  if (!workspace || workspace_bytes < stag_amort_workspace_bytes(nvp)) return STAG_ENOMEM;
  float* part = static_cast<float*>(workspace);
  STAG_HT_DISPATCH(edge_mlp_bwd_kernel, ht, dim3(kRedBlocks), s, src, dst, n_edges, ps, pd, (int)ldp, hidden, wh,
                   n_par, gpar, dpre, part);
  hipLaunchKernelGGL(edge_mlp_final_kernel, dim3(nvp), dim3(256), 0, s, part, kRedBlocks, ht, hidden, n_par, dwh, dbh);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_normal_kl_fwd(const float* loc, const float* log_scale, int64_t n, const float* p_loc, const float* p_scale,
                       float* kl_mean, void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || !loc || !log_scale || !p_loc || !p_scale || !kl_mean) return STAG_EINVAL;
  if (!workspace || workspace_bytes < stag_amort_workspace_bytes(1)) return STAG_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(normal_kl_fwd_kernel, dim3(kRedBlocks), dim3(256), 0, s, loc, log_scale, n, p_loc, p_scale, part);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, s, part, kRedBlocks, 1, kl_mean, 1, (float*)nullptr,
                     (const float*)nullptr, 1.0f / (float)n);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

int stag_normal_kl_bwd(const float* loc, const float* log_scale, int64_t n, const float* p_loc, const float* p_scale,
                       const float* g, float* dloc, float* dlog_scale, float* dp_loc, float* dp_scale,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || !loc || !log_scale || !p_loc || !p_scale || !g) return STAG_EINVAL;
  if (!workspace || workspace_bytes < stag_amort_workspace_bytes(2)) return STAG_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(normal_kl_bwd_kernel, dim3(kRedBlocks), dim3(256), 0, s, loc, log_scale, n, p_loc, p_scale, g,
                     dloc, dlog_scale, part);
  // d m2 = -c sum_i (m1 - m2) / s2^2,  d s2 = c sum_i (1 - vr - t1) / s2,  c = g / n
  if (dp_loc)
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, s, part, kRedBlocks, 2, dp_loc, 1, (float*)nullptr,
                       g, -1.0f / (float)n);          // block 0: value 0
  if (dp_scale)
    hipLaunchKernelGGL(reduce_final_kernel, dim3(2), dim3(256), 0, s, part, kRedBlocks, 2, (float*)nullptr, 1, dp_scale,
                       g, 1.0f / (float)n);           // block 1: value 1 (block 0 has no output)
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

}  // extern "C"
