// agg_kernel.hpp — fused noise x CSR gather -> weighted segmented sum (gfx950).
//
// Replaces, in one pass and without an [E, D] tensor:
//   StagLayer.rsample_noise / relu / _in_norm        stag/layers.py:84-129, 8-36
//   update_all(u_mul_e('h','_edge_weight'), sum)      stag/zoo/gcn.py:94-96
//   the degree scalings around it                     stag/zoo/gcn.py:67-75, 100-108
//
// Work decomposition (wave = 64 lanes)
//   The launch walks a list of UNITS (stag_plan): a unit is a whole destination row,
//   or one segment (<= seg_len edges) of a long row.  Units are sorted by length,
//   longest first, so (a) the teams that share a wave have equal trip counts and
//   (b) the heavy units are dispatched first (no hub-row tail).
//   A TEAM of LPE lanes owns one unit: lane c holds channels [4c, 4c+4) of the
//   channel tile — one dwordx4 of the gathered row, one Philox block of noise, four
//   accumulators.  D = 128 -> LPE = 32, two rows per wave; D = 256 -> one row per
//   wave; D = 16 -> sixteen rows per wave.  Nothing is reduced across lanes.
//   The kernel is VALU(RNG)-bound at D = 128, so the loop is built to add as few
//   vector instructions as possible around draw4(): indices come in with one
//   coalesced load per LPE edges and are handed round with ds_bpermute, rows are
//   fetched four edges at a time, optional work hides behind wave-uniform branches.
//   Segment partials go to a workspace; agg_combine_kernel adds them in segment
//   order, so results do not depend on scheduling.
#pragma once
#include "../../include/stag_hip.h"
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
  const stag_unit* units;  // null: unit i = row i, unsplit
  int32_t n_units;
  const int32_t* long_rows;
  const int32_t* long_seg_ptr;
  int32_t n_long;
  float* ws;               // [n_seg][ws_stride]: D partial sums, then D weight sums if in_norm
  int32_t ws_stride;
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

// epilogue shared by whole-row units and the combine kernel
__device__ __forceinline__ void agg_epilogue(const AggArgs& a, int v, int deg, int k0, bool vec,
                                             float (&acc)[4], const float (&wsum)[4]) {
  float dv = a.dst_scale ? a.dst_scale[v] : 1.0f;
  if (a.mean) dv *= __builtin_amdgcn_rcpf((float)(deg > 1 ? deg : 1));
  if (a.in_norm) {
    // stag/layers.py:24-28: indeg / cur where cur != 0, else 1
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (wsum[j] != 0.0f) ? (float)deg / wsum[j] : 1.0f;
      acc[j] *= s[j];
    }
    if (a.norm_scale_out) store4(a.norm_scale_out + (int64_t)v * a.D, k0, a.D, vec, s);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] *= dv;
  store4(a.out + (int64_t)v * a.ldo, k0, a.D, vec, acc);
}

template <int KIND, int LPE, bool VEC, bool PEDGE>
__global__ __launch_bounds__(256) void agg_kernel(const AggArgs a) {
  constexpr int TEAMS_PER_BLOCK = 256 / LPE;
  constexpr int BLK = 4;   // edges whose rows are in flight together

  const int lane = threadIdx.x & 63;
  const int c = threadIdx.x % LPE;                // chunk lane inside the channel tile
  const int team_lane0 = lane - c;                // first lane of this team inside the wave
  const int unit = blockIdx.x * TEAMS_PER_BLOCK + threadIdx.x / LPE;
  const uint32_t chunk = blockIdx.y * LPE + c;
  const int k0 = (int)chunk * 4;
  const bool has_unit = unit < a.n_units;
  const bool active = has_unit && (k0 < a.D);

  int v = 0, b = 0, len = 0, slot = -1;
  if (has_unit) {
    if (a.units) {
      const int4 q = *reinterpret_cast<const int4*>(a.units + unit);
      v = q.x; b = q.y; len = q.z; slot = q.w;
    } else {
      v = unit;
      b = a.indptr[v];
      len = a.indptr[v + 1] - b;
    }
  }

  // distribution parameters of this lane's 4 channels
  float pa[4] = {a.p0s, a.p0s, a.p0s, a.p0s};
  float pb[4] = {a.p1s, a.p1s, a.p1s, a.p1s};
  if constexpr (KIND >= kNormal) {
    if (a.pmode == STAG_PARAM_PER_CHANNEL && active) {
      load4(a.p0, k0, a.D, VEC, pa);
      if (a.p1) load4(a.p1, k0, a.D, VEC, pb);
    }
  }

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float wsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_ss = a.src_scale != nullptr;
  const bool want_wsum = a.in_norm != 0;

  // Edges are consumed LPE at a time: lane c fetches the column id (and, if asked for,
  // the edge id) of edge i0 + c with one coalesced load; ds_bpermute then hands edge
  // i0 + j to every lane of the team.
  for (int i0 = 0; i0 < len; i0 += LPE) {
    const int nb = min(LPE, len - i0);
    int my_u = 0, my_e = 0;
    if (c < nb) {
      const int p = b + i0 + c;
      my_u = a.indices[p];
      if constexpr (KIND == kExplicit || PEDGE) my_e = a.eid ? a.eid[p] : p;
      else if constexpr (KIND >= kNormal) my_e = a.nidx ? a.nidx[p] : 0;
    }
    for (int j0 = 0; j0 < nb; j0 += BLK) {
      int u[BLK], ee[BLK];
      float xv[BLK][4], xs[BLK];
#pragma unroll
      for (int j = 0; j < BLK; ++j) {
        const int src_lane = (team_lane0 + j0 + j) << 2;
        u[j] = __builtin_amdgcn_ds_bpermute(src_lane, my_u);
        if constexpr (KIND != kNone) ee[j] = __builtin_amdgcn_ds_bpermute(src_lane, my_e);
      }
#pragma unroll
      for (int j = 0; j < BLK; ++j) {
        if (j0 + j < nb && active) {
          load4(a.x + (int64_t)u[j] * a.ldx, k0, a.D, VEC, xv[j]);
          if (want_ss) xs[j] = a.src_scale[u[j]];
        }
      }
#pragma unroll
      for (int j = 0; j < BLK; ++j) {
        if (j0 + j < nb && active) {
          const int p = b + i0 + j0 + j;
          float w[4];
          if constexpr (KIND == kNone) {
            w[0] = w[1] = w[2] = w[3] = 1.0f;
          } else if constexpr (KIND == kExplicit) {
            load4(a.p0 + (int64_t)ee[j] * a.D, k0, a.D, VEC, w);
            if (a.relu) {
#pragma unroll
              for (int q = 0; q < 4; ++q) w[q] = fmaxf(w[q], 0.0f);
            }
          } else {
            if constexpr (PEDGE) {
              if (a.pmode == STAG_PARAM_PER_EDGE1) {
                const float q0 = a.p0[ee[j]];
                const float q1 = a.p1 ? a.p1[ee[j]] : 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { pa[q] = q0; pb[q] = q1; }
              } else {
                load4(a.p0 + (int64_t)ee[j] * a.D, k0, a.D, VEC, pa);
                if (a.p1) load4(a.p1 + (int64_t)ee[j] * a.D, k0, a.D, VEC, pb);
              }
            }
            int64_t gpos;
            if constexpr (PEDGE) gpos = a.nidx ? (int64_t)a.nidx[p] : a.pos_base + p;
            else gpos = a.nidx ? (int64_t)ee[j] : a.pos_base + p;
            draw4<KIND>(gpos, chunk, a.key, pa, pb, a.relu != 0, w);
          }
          if (want_ss) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[j][q] *= xs[j];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_fmaf(w[q], xv[j][q], acc[q]);
          if (want_wsum) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wsum[q] += w[q];
          }
        }
      }
    }
  }

  if (!active) return;
  if (slot >= 0) {
    float* wrow = a.ws + (int64_t)slot * a.ws_stride;
    store4(wrow, k0, a.D, VEC, acc);
    if (want_wsum) store4(wrow + a.D, k0, a.D, VEC, wsum);
  } else {
    agg_epilogue(a, v, len, k0, VEC, acc, wsum);
  }
}

// Launch one (KIND, PEDGE) family; defined per kind in agg_<kind>.hip so the
// instantiations compile in parallel.
template <int KIND>
hipError_t agg_launch(const AggArgs& a, bool vec, hipStream_t stream);

template <int KIND, int LPE>
inline void agg_launch_shape(const AggArgs& a, bool vec, bool pedge, int tiles, hipStream_t s) {
  constexpr int TPB = 256 / LPE;
  dim3 grid((a.n_units + TPB - 1) / TPB, tiles);
  if (grid.x == 0) return;
  if constexpr (KIND >= kNormal) {
    if (pedge) {
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, true>), grid, dim3(256), 0, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, true>), grid, dim3(256), 0, s, a);
      return;
    }
  }
  if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, false>), grid, dim3(256), 0, s, a);
  else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, false>), grid, dim3(256), 0, s, a);
}

template <int KIND>
inline hipError_t agg_launch_impl(const AggArgs& a, bool vec, hipStream_t s) {
  const int nchunk = (a.D + 3) / 4;
  const bool pedge = (KIND >= kNormal) && (a.pmode >= STAG_PARAM_PER_EDGE1);
  // lanes per unit: smallest power of two covering the row, capped at a wave
  int lpe = 1;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  const int tiles = (nchunk + lpe - 1) / lpe;
  switch (lpe) {
    case 64: agg_launch_shape<KIND, 64>(a, vec, pedge, tiles, s); break;
    case 32: agg_launch_shape<KIND, 32>(a, vec, pedge, tiles, s); break;
    case 16: agg_launch_shape<KIND, 16>(a, vec, pedge, tiles, s); break;
    case 8:  agg_launch_shape<KIND, 8>(a, vec, pedge, tiles, s); break;
    case 4:  agg_launch_shape<KIND, 4>(a, vec, pedge, tiles, s); break;
    case 2:  agg_launch_shape<KIND, 2>(a, vec, pedge, tiles, s); break;
    default: agg_launch_shape<KIND, 1>(a, vec, pedge, tiles, s); break;
  }
  return hipGetLastError();
}

}  // namespace stag
