// agg_kernel.hpp — fused noise x CSR gather -> weighted segmented sum (gfx950).
//
// Replaces, in one pass and without an [E, D] tensor:
//   StagLayer.rsample_noise / relu / _in_norm        stag/layers.py:84-129, 8-36
//   update_all(u_mul_e('h','_edge_weight'), sum)      stag/zoo/gcn.py:94-96
//   the degree scalings around it                     stag/zoo/gcn.py:67-75, 100-108
//
// Work decomposition (wave = 64 lanes):
//   A "team" of LPE*EPT lanes owns one destination row (or one segment of a long
//   row).  LPE lanes span the channel tile, 4 channels (one dwordx4, one Philox
//   block) per lane; EPT edge slots run side by side.  For D = 128 a team is a
//   whole wave: 32 lanes x float4 = one 512-B row per half-wave, 2 edges at once.
//   Rows longer than plan.seg_len are cut into segments (long mode) whose partial
//   sums go to a workspace and are added in segment order by agg_combine_kernel,
//   so hub rows neither serialise a wave nor make the result order-dependent.
#pragma once
#include "noise.hpp"

namespace stag {

struct AggArgs {
  // graph
  const int32_t* indptr;
  const int32_t* indices;
  const int32_t* eid;    // may be null (identity)
  const int32_t* nidx;   // may be null (pos_base + position)
  int32_t n_rows;
  // gathered matrix
  const float* x;
  int64_t ldx;
  int32_t D;
  // noise
  const float* p0;
  const float* p1;
  float p0s, p1s;
  int32_t pmode;   // STAG_PARAM_*
  int32_t relu, in_norm;
  PhiloxKey key;
  int64_t pos_base;
  // scaling / reduce
  const float* src_scale;
  const float* dst_scale;
  int32_t mean;
  // output
  float* out;
  int64_t ldo;
  float* norm_scale_out;   // [n_rows, D] or null
  // plan
  int32_t long_mode;   // 0: one team per row (rows > seg_len skipped); 1: one team per segment
  int32_t seg_len;     // <= 0: no splitting
  int32_t n_units;     // rows (short) or segments (long)
  const int32_t* long_rows;
  const int32_t* long_seg_ptr;
  const int32_t* seg_row;
  const int32_t* seg_start;
  float* ws;           // [n_seg][ws_stride]: D partial sums, then D weight sums if in_norm
  int32_t ws_stride;
  int32_t n_long;
};

__device__ __forceinline__ void load4(const float* p, int k0, int D, bool vec, float (&v)[4]) {
  if (vec) {
    const float4 t = *reinterpret_cast<const float4*>(p + k0);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (k0 + j < D) ? p[k0 + j] : 0.0f;
  }
}

__device__ __forceinline__ void store4(float* p, int k0, int D, bool vec, const float (&v)[4]) {
  if (vec) {
    *reinterpret_cast<float4*>(p + k0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < D) p[k0 + j] = v[j];
  }
}

// epilogue shared by the short path and the combine kernel
__device__ __forceinline__ void agg_epilogue(const AggArgs& a, int v, int deg, int k0, bool vec,
                                             float (&acc)[4], const float (&wsum)[4]) {
  float dv = a.dst_scale ? a.dst_scale[v] : 1.0f;
  if (a.mean) dv /= (float)(deg > 1 ? deg : 1);
  float s[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  if (a.in_norm) {
    // stag/layers.py:24-28: indeg / cur where cur != 0, else 1
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (wsum[j] != 0.0f) ? (float)deg / wsum[j] : 1.0f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = acc[j] * s[j] * dv;
  store4(a.out + (int64_t)v * a.ldo, k0, a.D, vec, acc);
  if (a.norm_scale_out) store4(a.norm_scale_out + (int64_t)v * a.D, k0, a.D, vec, s);
}

template <int KIND, int LPE, int EPT, bool VEC, bool PEDGE>
__global__ __launch_bounds__(256) void agg_kernel(const AggArgs a) {
  constexpr int TEAM = LPE * EPT;
  constexpr int TEAMS_PER_BLOCK = 256 / TEAM;
  static_assert(TEAM <= 64 && (64 % TEAM) == 0, "a team must not straddle waves");
  constexpr int UNROLL = 4;

  const int t = threadIdx.x % TEAM;
  const int c = t % LPE;    // chunk lane inside the channel tile
  const int ep = t / LPE;   // edge slot
  const int unit = blockIdx.x * TEAMS_PER_BLOCK + threadIdx.x / TEAM;
  const uint32_t chunk = blockIdx.y * LPE + c;
  const int k0 = (int)chunk * 4;
  const bool active = (unit < a.n_units) && (k0 < a.D);

  int v = 0, b = 0, e = 0, row_deg = 0;
  if (unit < a.n_units) {
    if (a.long_mode) {
      v = a.long_rows[a.seg_row[unit]];
      b = a.seg_start[unit];
      const int row_end = a.indptr[v + 1];
      e = min(b + a.seg_len, row_end);
    } else {
      v = unit;
      b = a.indptr[v];
      e = a.indptr[v + 1];
      row_deg = e - b;
      if (a.seg_len > 0 && row_deg > a.seg_len) e = b;   // long row: long mode owns it
    }
  }
  const bool skip_store = (!a.long_mode) && (a.seg_len > 0) && (row_deg > a.seg_len);

  // distribution parameters of this lane's 4 channels
  float pa[4] = {a.p0s, a.p0s, a.p0s, a.p0s};
  float pb[4] = {a.p1s, a.p1s, a.p1s, a.p1s};
  if constexpr (KIND >= kNormal) {
    if (a.pmode == 1 && active) {   // per-channel
      load4(a.p0, k0, a.D, VEC, pa);
      if (a.p1) load4(a.p1, k0, a.D, VEC, pb);
    }
  }

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float wsum[4] = {0.f, 0.f, 0.f, 0.f};

  if (active) {
    for (int p0 = b + ep; p0 < e; p0 += UNROLL * EPT) {
      int u[UNROLL];
      float xs[UNROLL];
      float xv[UNROLL][4];
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) {
        const int p = p0 + i * EPT;
        u[i] = (p < e) ? a.indices[p] : -1;
      }
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) {
        if (u[i] >= 0) {
          load4(a.x + (int64_t)u[i] * a.ldx, k0, a.D, VEC, xv[i]);
          xs[i] = a.src_scale ? a.src_scale[u[i]] : 1.0f;
        }
      }
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) {
        if (u[i] >= 0) {
          const int p = p0 + i * EPT;
          float w[4];
          if constexpr (KIND == kNone) {
            w[0] = w[1] = w[2] = w[3] = 1.0f;
          } else if constexpr (KIND == kExplicit) {
            const int64_t ed = a.eid ? a.eid[p] : p;
            load4(a.p0 + ed * (int64_t)a.D, k0, a.D, VEC, w);
            if (a.relu) {
#pragma unroll
              for (int j = 0; j < 4; ++j) w[j] = fmaxf(w[j], 0.0f);
            }
          } else {
            if constexpr (PEDGE) {
              const int64_t ed = a.eid ? a.eid[p] : p;
              if (a.pmode == 2) {
                const float q0 = a.p0[ed];
                const float q1 = a.p1 ? a.p1[ed] : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pa[j] = q0; pb[j] = q1; }
              } else {
                load4(a.p0 + ed * (int64_t)a.D, k0, a.D, VEC, pa);
                if (a.p1) load4(a.p1 + ed * (int64_t)a.D, k0, a.D, VEC, pb);
              }
            }
            const int64_t gpos = a.nidx ? (int64_t)a.nidx[p] : a.pos_base + p;
            draw4<KIND>(gpos, chunk, a.key, pa, pb, a.relu != 0, w);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_fmaf(w[j], xv[i][j] * xs[i], acc[j]);
            wsum[j] += w[j];
          }
        }
      }
    }
  }

  // add the EPT edge slots of the team (fixed order => deterministic)
#pragma unroll
  for (int m = LPE; m < TEAM; m <<= 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] += __shfl_xor(acc[j], m);
      wsum[j] += __shfl_xor(wsum[j], m);
    }
  }

  if (!active || ep != 0) return;
  if (a.long_mode) {
    float* wrow = a.ws + (int64_t)unit * a.ws_stride;
    store4(wrow, k0, a.D, VEC, acc);
    if (a.in_norm) store4(wrow + a.D, k0, a.D, VEC, wsum);
  } else if (!skip_store) {
    agg_epilogue(a, v, row_deg, k0, VEC, acc, wsum);
  }
}

// Launch one (KIND, PEDGE) family; defined per kind in agg_<kind>.hip so the
// instantiations compile in parallel.
template <int KIND>
hipError_t agg_launch(const AggArgs& a, bool vec, hipStream_t stream);

template <int KIND, int LPE, int EPT>
inline void agg_launch_shape(const AggArgs& a, bool vec, bool pedge, dim3 tiles, hipStream_t s) {
  constexpr int TPB = 256 / (LPE * EPT);
  dim3 grid((a.n_units + TPB - 1) / TPB, tiles.y);
  if (grid.x == 0) return;
  if constexpr (KIND >= kNormal) {
    if (pedge) {
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, EPT, true, true>), grid, dim3(256), 0, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, EPT, false, true>), grid, dim3(256), 0, s, a);
      return;
    }
  }
  if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, EPT, true, false>), grid, dim3(256), 0, s, a);
  else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, EPT, false, false>), grid, dim3(256), 0, s, a);
}

template <int KIND>
inline hipError_t agg_launch_impl(const AggArgs& a, bool vec, hipStream_t s) {
  const int nchunk = (a.D + 3) / 4;
  const bool pedge = (KIND >= kNormal) && (a.pmode >= 2);
  // lanes per edge: smallest power of two covering the row, capped at a wave
  int lpe = 1;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  dim3 tiles(1, (nchunk + lpe - 1) / lpe);
  switch (lpe) {
    case 64: agg_launch_shape<KIND, 64, 1>(a, vec, pedge, tiles, s); break;
    case 32: agg_launch_shape<KIND, 32, 2>(a, vec, pedge, tiles, s); break;
    case 16: agg_launch_shape<KIND, 16, 4>(a, vec, pedge, tiles, s); break;
    case 8:  agg_launch_shape<KIND, 8, 4>(a, vec, pedge, tiles, s); break;
    case 4:  agg_launch_shape<KIND, 4, 4>(a, vec, pedge, tiles, s); break;
    case 2:  agg_launch_shape<KIND, 2, 8>(a, vec, pedge, tiles, s); break;
    default: agg_launch_shape<KIND, 1, 16>(a, vec, pedge, tiles, s); break;
  }
  return hipGetLastError();
}

}  // namespace stag
