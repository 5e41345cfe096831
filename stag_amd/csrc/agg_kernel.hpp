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
//   longest first, so (a) the units that share a wave have equal trip counts and
//   (b) the heavy units are dispatched first (no hub-row tail).
//   LPE lanes cover a unit's channels: lane c holds channels [4c, 4c+4) of the channel
//   tile — one dwordx4 of the gathered row, one Philox block of noise, four
//   accumulators.  D = 128 -> LPE = 32, two rows per wave; D = 256 -> one row per
//   wave; D = 16 -> sixteen rows per wave.  Nothing is reduced across channel lanes.
//   Summation order is fixed: edges are taken in BLOCKS of 2, a block's products are
//   summed from zero and the block sums are folded into the unit's sum in order
//   (compensated once the unit is longer than 16 edges).  Everything below that changes
//   how much is in flight (blocks fetched together, edge slots) keeps exactly this
//   order, so a result does not depend on D, on the launch shape or on scheduling —
//   which is what lets channel shards reproduce the whole-width result bit for bit.
//   D = 128 is VALU(RNG)- and gather-bound, so the loop adds as few vector
//   instructions as possible around draw4(): column ids are broadcast loads with
//   immediate offsets, row addresses are 32-bit offsets behind a buffer descriptor
//   (64-bit vector adds are quarter rate), optional work hides behind wave-uniform
//   branches.  D <= 64 is latency-bound (tools/trace_units.py): the heavy units (all
//   segments and the whole rows longer than STAG_HEAVY_LEN edges; the first n_heavy of
//   the plan) are spread over 2-4 EDGE SLOTS of LPE lanes each, which draw and gather
//   side by side and exchange block sums by ds_bpermute.
//   Segment partials go to a workspace; the segment of a row that finishes LAST
//   (write-through stores / arrival counter / acquire, cdna_hip_programming.md
//   Guideline 16) adds them — Kahan sums of groups of 16 in segment order, then of the
//   group sums — so there is no second launch.
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
  uint32_t ldxb;   // row stride of x in bytes (0: one broadcast row)
  uint32_t ldwb;   // row stride in bytes of explicit weights / per-edge parameters
  uint32_t x_bytes;   // extent of x when it fits a buffer descriptor
  int32_t wide;    // bit 0: x, bit 1: edge data — byte offsets / ids too big for the 32-bit form
  // noise
  const float* p0;
  const float* p1;
  float p0s, p1s;
  int32_t pmode;   // STAG_PARAM_*
  int32_t relu, in_norm;   // relu: noise flags = relu | deriv << 1 | log-scale << 3 (noise.hpp)
  int32_t wgroup;          // EXPLICIT: channels sharing one weight column (<= 1: one per channel)
  PhiloxKey key;
  uint32_t pos_lo, pos_hi;   // lo32 / hi32 of the shard's global position base
  uint32_t chunk_base;       // global chunk (channel / 4) of this shard's channel 0
  int32_t n_heavy;           // leading units of the plan longer than STAG_HEAVY_LEN edges
  int32_t n_heavy_blocks;    // blocks that serve them (set per launch shape)
  // Which units a workgroup walks — eight scalars the launch shape sets (one s_load): workgroup b serves stripe
  // b & smask with its (b >> sshift)-th block.  XCD-aware (stag_plan.xcd_order; smask 7, sshift 3): workgroups go to
  // the 8 XCDs round-robin and every XCD has its own L2, so each stripe is a range of destination rows (1/8 of the
  // edges) whose units lie together — `units` then holds 8 heavy stripes of `sh` records and 8 light ones of `sl`,
  // padded with null records (row < 0).  Plan order (smask 0, sshift 0): one stripe, sh = n_heavy, sl = the rest.
  // Without a plan (unit i = row i): stripes of `sl` rows.
  struct alignas(32) Walk {
    int32_t smask, sshift;
    int32_t jh_heavy;        // blocks per stripe that serve the heavy units with the slotted loop, or 0
    int32_t jh_light;        // ... with the one-slot loop (shapes without edge slots), or 0: one of the two is 0
    int32_t sh;              // heavy records per stripe
    int32_t lbase;           // first light record
    int32_t sl;              // light records per stripe
    int32_t n_total;         // records in all
  } walk;
  const int32_t* xcd;        // stag_plan.xcd_order or null (host side only: agg_launch_shape reads the strides)
  // scaling / reduce
  const float* src_scale;
  const float* dst_scale;
  int32_t mean;
  // output
  float* out;
  float* outx[3];          // extra outputs: stag_agg_bwd's two derivative aggregates, or the
                           // Monte-Carlo samples 1.. of stag_agg_fwd_mc; outx[0] null: none
  int32_t mc;              // extra outputs are MC samples: sample s draws at offset + s * mc_stride
  uint64_t mc_stride;
  int64_t ldo;
  float* norm_scale_out;   // [n_rows, D] or null
  // per-edge parameter gradients out of the same pass (stag_agg_bwd_edge; PEDGE 3): the unit's OWN row of
  // xown times the gathered row, summed over the channels, times each derivative of the draw
  const float* xown;       // [n_rows, ldxo]
  int64_t ldxo;
  const float* own_scale;  // [n_rows] or null: factor on the own row
  float* eg0;              // [E] by edge id: d / d p0
  float* eg1;              // [E] by edge id: d / d p1 (d / d log p1 under the log-scale flag); may be null
  // gradients of scalar / per-channel parameters out of the same pass (stag_agg_bwd_dp; PEDGE 4): every lane sums
  // dw/dp_i * (gathered row) * (own row of xown; null: ones) over the edges it walks, the block adds its teams
  float* dp_part;          // [gridDim.y][gridDim.x][2][LPE * 4] block partials, or null
  // plan
  const stag_unit* units;  // null: unit i = row i, unsplit
  const stag_unit* units_plan;   // (host side) the plan's own order, for the launch families that do not take the walk
  int32_t n_units;
  const int32_t* long_rows;
  const int32_t* long_seg_ptr;
  int32_t n_long;
  int32_t n_seg;           // units[0, n_seg) are segments, the rest whole rows
  int32_t* seg_counters;   // [n_long] arrival counters, zero between launches
  float* ws;               // [n_seg][ws_stride]: D partial sums, then D weight sums if in_norm
  int32_t ws_stride;
  uint32_t ws_bytes;
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

// write-through (sc1) store of 4 channels through a buffer descriptor: aux = 16
__device__ __forceinline__ void store4_sc1(__amdgpu_buffer_rsrc_t rsrc, uint32_t off, int k0, int D,
                                           bool vec, const float (&v)[4]) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  if (vec) {
    const u32x4 t = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                     __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(t, rsrc, (int)off, 0, 16);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < D)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), rsrc, (int)(off + 4u * j), 0, 16);
  }
}

// sum[q] = sum over segments s0..s1-1 of ws[s][k0+q], in segment order, Kahan-compensated
template <int NF, bool VEC>
__device__ __forceinline__ void kahan_sum_partials(const float* ws, int ws_stride, int s0, int s1,
                                                   int k0, int D, float (&sum)[4], float (&comp)[4]) {
  comp[0] = comp[1] = comp[2] = comp[3] = 0.f;
  sum[0] = sum[1] = sum[2] = sum[3] = 0.f;
#pragma unroll 1
  for (int s = s0; s < s1; s += NF) {
    float t[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (s + i < s1) load4(ws + (int64_t)(s + i) * ws_stride, k0, D, VEC, t[i]);
      else t[i][0] = t[i][1] = t[i][2] = t[i][3] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float y = t[i][q] - comp[q];
        const float n = sum[q] + y;
        comp[q] = (n - sum[q]) - y;
        sum[q] = n;
      }
  }
}

// a derivative aggregate: same row scale as the main output, no in-norm
__device__ __forceinline__ void agg_epilogue_extra(const AggArgs& a, float* out, int v, int deg, int k0,
                                                   bool vec, float (&acc)[4]);

// Two-level form: Kahan sums of groups of kCombineGroup partials, then a Kahan sum of the group sums in
// group order.  SLOTS edge slots of the unit take one group each per round and exchange the group
// sums (ds_bpermute; slot0 = byte address of slot 0's lane with my channels): same arithmetic
// for every SLOTS.
// What a group sum loses to rounding (its Kahan residual) is not dropped: it joins the second level's
// compensation, so the second level does not undo the first: the result is the sum of the partials
// to about one rounding of the result.  (Written to stay inside the hot loop's register budget.)
constexpr int kCombineGroup = 16;
template <int NF, bool VEC, int LPE, int SLOTS>
__device__ __forceinline__ void two_level_sum(const float* ws, int ws_stride, int s0, int s1, int k0,
                                              int D, int sl, int slot0, float (&sum)[4]) {
  float comp[4] = {0.f, 0.f, 0.f, 0.f};
  sum[0] = sum[1] = sum[2] = sum[3] = 0.f;
#pragma unroll 1
  for (int g0 = s0; g0 < s1; g0 += kCombineGroup * SLOTS) {
    const int gs = g0 + sl * kCombineGroup;
    float gsum[4] = {0.f, 0.f, 0.f, 0.f}, gres[4] = {0.f, 0.f, 0.f, 0.f};   // group sum, what it still owes
    if (gs < s1) {
      const int ge = min(gs + kCombineGroup, s1);
      kahan_sum_partials<NF, VEC>(ws, ws_stride, gs, ge, k0, D, gsum, gres);
    }
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      if (g0 + j * kCombineGroup < s1) {            // uniform over the unit's lanes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float gj = SLOTS > 1 ? __int_as_float(__builtin_amdgcn_ds_bpermute(
                                           slot0 + j * (LPE << 2), __float_as_int(gsum[q])))
                                     : gsum[q];
          const float rj = SLOTS > 1 ? __int_as_float(__builtin_amdgcn_ds_bpermute(
                                           slot0 + j * (LPE << 2), __float_as_int(gres[q])))
                                     : gres[q];
          comp[q] += rj;                            // true group sum = gj - rj
          const float y = gj - comp[q];
          const float n = sum[q] + y;
          comp[q] = (n - sum[q]) - y;
          sum[q] = n;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sum[q] -= comp[q];
}

// epilogue shared by whole-row units and the last-arriver combine
__device__ __forceinline__ void agg_epilogue_to(const AggArgs& a, float* out, float* norm_scale_out, int v, int deg,
                                                int k0, bool vec, float (&acc)[4], const float (&wsum)[4]);
__device__ __forceinline__ void agg_epilogue(const AggArgs& a, int v, int deg, int k0, bool vec,
                                             float (&acc)[4], const float (&wsum)[4]) {
  agg_epilogue_to(a, a.out, a.norm_scale_out, v, deg, k0, vec, acc, wsum);
}
__device__ __forceinline__ void agg_epilogue_to(const AggArgs& a, float* out, float* norm_scale_out, int v, int deg,
                                                int k0, bool vec, float (&acc)[4], const float (&wsum)[4]) {
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
    if (norm_scale_out) store4(norm_scale_out + (int64_t)v * a.D, k0, a.D, vec, s);
  }
  if (!out) return;      // only the in-norm factor was asked for
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] *= dv;
  store4(out + (int64_t)v * a.ldo, k0, a.D, vec, acc);
}

__device__ __forceinline__ void agg_epilogue_extra(const AggArgs& a, float* out, int v, int deg, int k0,
                                                   bool vec, float (&acc)[4]) {
  float dv = a.dst_scale ? a.dst_scale[v] : 1.0f;
  if (a.mean) dv *= __builtin_amdgcn_rcpf((float)(deg > 1 ? deg : 1));
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] *= dv;
  store4(out + (int64_t)v * a.ldo, k0, a.D, vec, acc);
}

// row `idx` of a row-major fp32 matrix, `koff` bytes into the row.  Narrow form: one
// v_mul_u32_u24 and a 32-bit offset from a scalar base (global_load ... saddr), instead of
// 64-bit vector address arithmetic, which is quarter rate on gfx950.
__device__ __forceinline__ const float* row_at(const float* base, int idx, uint32_t stride_bytes,
                                               uint32_t koff, bool wide) {
  const char* b = reinterpret_cast<const char*>(base);
  if (wide) return reinterpret_cast<const float*>(b + (uint64_t)(uint32_t)idx * stride_bytes + koff);
  return reinterpret_cast<const float*>(b + (__umul24((uint32_t)idx, stride_bytes) + koff));
}

// Gathered row through a buffer descriptor: ONE v_mad_u32_u24 of address arithmetic and a
// `buffer_load_dwordx4 ... offen` with a 32-bit offset (out-of-range offsets read 0).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bufrow4(__amdgpu_buffer_rsrc_t rsrc, int idx, uint32_t stride_bytes,
                                        uint32_t koff, float (&v)[4]) {
#ifndef STAG_X_AUX
#define STAG_X_AUX 0   // cache-policy bits of the row gather (2 = nt, 16 = sc1, ...): A/B knob
#endif
  const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(
      rsrc, (int)(__umul24((uint32_t)idx, stride_bytes) + koff), 0, STAG_X_AUX);
  v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y);
  v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
}

__device__ __forceinline__ void loadrow4(const float* p, int k0, int D, bool vec, float (&v)[4]) {
  if (vec) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (k0 + j < D) ? p[j] : 0.0f;
  }
}

#ifndef STAG_KAHAN_MIN_LEN
#define STAG_KAHAN_MIN_LEN 16
#endif
constexpr int kKahanMinLen = STAG_KAHAN_MIN_LEN;

// Tuning knobs (tools/ab_bench.py builds variants with -D...): edges per block for the
// RNG-bound and the gather-bound kinds, waves per SIMD asked of the register allocator.
#ifndef STAG_BLK_RNG
#define STAG_BLK_RNG 2
#endif
#ifndef STAG_BLK_MEM
#define STAG_BLK_MEM 2
#endif
#ifndef STAG_WAVES_PER_SIMD
#define STAG_WAVES_PER_SIMD 1
#endif
#ifndef STAG_BLOCK_THREADS
#define STAG_BLOCK_THREADS 256
#endif
#ifndef STAG_LOAD_PRIO
#define STAG_LOAD_PRIO 1
#endif
#ifndef STAG_PRIO_MIN_LEN
#define STAG_PRIO_MIN_LEN 24
#endif
// blocks fetched together by a light unit, by lanes per unit (LPE 4: D <= 16; 8: D <= 32; 16: D <= 64;
// wider); values are for the RNG kinds, mult_of() scales them for the gather-bound kinds
#ifndef STAG_IDX_PREFETCH
#define STAG_IDX_PREFETCH 1
#endif
#ifndef STAG_PIN_IDX
#define STAG_PIN_IDX 1
#endif
#ifndef STAG_DRAW_FIRST
#define STAG_DRAW_FIRST 1
#endif
#ifndef STAG_PREFETCH_MAX_LPE
#define STAG_PREFETCH_MAX_LPE 16
#endif
#ifndef STAG_PREFETCH_MAX_LPE_MEM
#define STAG_PREFETCH_MAX_LPE_MEM 32    // the gather-bound kinds (no draw): D = 128 gains 1.6 % (97.4 -> 95.9 us), D = 256 nothing
#endif
#ifndef STAG_MULT_LPE4
#define STAG_MULT_LPE4 4
#endif
#ifndef STAG_MULT_LPE8
#define STAG_MULT_LPE8 1
#endif
#ifndef STAG_MULT_LPE16
#define STAG_MULT_LPE16 1
#endif
#ifndef STAG_MULT_WIDE
#define STAG_MULT_WIDE 1
#endif

// sum over the LPE lanes of a team, result in every lane: DPP inside a row of 16 lanes (quad
// swaps, half-row and row mirrors: 4 full-rate ops), ds_bpermute only across rows
template <int LPE>
__device__ __forceinline__ float team_sum(float v) {
  if constexpr (LPE >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm 1,0,3,2
  if constexpr (LPE >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm 2,3,0,1
  if constexpr (LPE >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  if constexpr (LPE >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)); // row_mirror
  if constexpr (LPE >= 32) v += __shfl_xor(v, 16);
  if constexpr (LPE >= 64) v += __shfl_xor(v, 32);
  return v;
}

// The same sum when lanes at the END of the team may have left (channel chunks past the row's end):
// a shift-down tree, whose reads only ever go to higher lanes, valid in lane 0 only.  c = this lane's
// index in the team, nlive = the lanes that stayed.  (A butterfly needs every lane's partial sums.)
template <int LPE>
__device__ __forceinline__ float team_sum_lane0(float v, int c, int nlive) {
  // row_shl:n — lane i of a row of 16 reads lane i + n; past the row or from a lane that left: 0 (bound_ctrl)
  if constexpr (LPE >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, true));
  if constexpr (LPE >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x102, 0xF, 0xF, true));
  if constexpr (LPE >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xF, 0xF, true));
  if constexpr (LPE >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x108, 0xF, 0xF, true));
  if constexpr (LPE >= 32) { const float t = __shfl_down(v, 16); v += (c + 16 < nlive) ? t : 0.f; }
  if constexpr (LPE >= 64) { const float t = __shfl_down(v, 32); v += (c + 32 < nlive) ? t : 0.f; }
  return v;
}

// Register image of one block of BLK edges of a unit.
template <int BLK>
struct EdgeIdx {        // what the index fetch brings in
  int u[BLK];           // column id (row of x)
  int ee[BLK];          // original edge id (explicit weights / per-edge parameters)
  uint32_t nn[BLK];     // Philox counter word 0 (noise index)
};
// per-edge distribution parameters travel with the rows (their loads overlap the gathers):
// PEDGE 1 = one (p0, p1) pair per edge ([E, 1] parameters), 2 = a row of 4 + 4 ([E, D])
template <int BLK, int PEDGE>
struct EdgeParams {};
template <int BLK>
struct EdgeParams<BLK, 1> { float q0[BLK], q1[BLK]; };
template <int BLK>
struct EdgeParams<BLK, 2> { float pa[BLK][4], pb[BLK][4]; };
// PEDGE 3 = PEDGE 1 whose pass also returns the gradients of the pair (stag_agg_bwd_edge)
template <int BLK>
struct EdgeParams<BLK, 3> : EdgeParams<BLK, 1> {};
// PEDGE 4 = PEDGE 0 (scalar / per-channel parameters) whose pass also sums their gradients (stag_agg_bwd_dp)
template <bool ON>
struct DpAcc { float v[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}; };
template <>
struct DpAcc<false> {};
template <bool ON>
struct OwnRow { float v[4]; };
template <>
struct OwnRow<false> {};

template <int BLK, int PEDGE = 0>
struct EdgeRows {       // what the row fetch brings in
  float xv[BLK][4];
  float xs[BLK];
  [[no_unique_address]] EdgeParams<BLK, PEDGE> P;
};

// BLK: edges per arithmetic block (fixes the summation order, so the same for every shape);
// MULT: blocks fetched together (BLK * MULT rows in flight per team; no effect on the sums).
// accumulators of the extra outputs; nothing at all when there are none
template <int NX>
struct ExtraAcc {
  float acc[NX][4] = {}, comp[NX][4] = {};
};
template <>
struct ExtraAcc<0> {};
template <int N>
struct ExtraW {           // MC + in-norm: the weight sums of samples 1..N (stag/layers.py:12-15, per sample)
  float w[N][4] = {};
};
template <>
struct ExtraW<0> {};
template <int N>
struct ExtraKeys {        // MC: the Philox keys of samples 1..N (SGPRs)
  PhiloxKey k[N];
};
template <>
struct ExtraKeys<0> {};

// NOUT outputs from one pass over the gathered rows.  MC = false: 1, or 3 = the weight and its two
// parameter derivatives (stag_agg_bwd).  MC = true: NOUT Monte-Carlo samples, sample s drawn at
// offset + s * stride (stag_agg_fwd_mc; the n_samples loop of stag/models.py:45-55 on layer 1).
// WN (MC only): every sample keeps its own in-norm weight sums (registers: two samples per pass, not four)
template <int KIND, int LPE, bool VEC, int PEDGE, int BLK, int MULT = 1, int NOUT = 1, bool MC = false, bool WN = false>
struct AggTeam {
  static_assert(!WN || MC, "per-sample weight sums belong to the Monte-Carlo form");
  static constexpr int NB = BLK * MULT;
  static constexpr int NX = NOUT - 1;              // extra outputs
  static_assert(NOUT == 1 || PEDGE == 0, "extra outputs: scalar / per-channel parameters");
  static_assert(NOUT == 1 || MC || (NOUT == 3 && (KIND == kNormal || KIND == kUniform)), "derivatives: reparameterised draws");
  static_assert(!MC || KIND >= kNormal, "Monte-Carlo samples need sampled noise");
  static constexpr bool NEED_EID = (KIND == kExplicit) || (PEDGE != 0 && PEDGE != 4);
  const AggArgs& a;
  const PhiloxKey key;     // a.key with the device epoch folded in
  const int k0;
  const uint32_t koff, c1;
  int pend;
  const __amdgpu_buffer_rsrc_t rx;
  const bool x_buf;
  float pa[4], pb[4];
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, comp[4] = {0.f, 0.f, 0.f, 0.f};
  float wsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool kahan;
  [[no_unique_address]] ExtraAcc<NX> X;
  [[no_unique_address]] ExtraKeys<MC ? NX : 0> KX;
  [[no_unique_address]] ExtraW<WN ? NX : 0> XW;
  [[no_unique_address]] OwnRow<PEDGE == 3 || PEDGE == 4> XO;     // the unit's own row of a.xown (times own_scale)
  [[no_unique_address]] DpAcc<PEDGE == 4> DP;      // this lane's share of the two parameter gradients
  static constexpr bool P1 = PEDGE == 1 || PEDGE == 3;

  // every lane of the team reads the same BLK column ids: broadcast dword loads with
  // immediate offsets, no per-edge vector arithmetic
  __device__ __forceinline__ void fetch_idx(EdgeIdx<NB>& I, int p0) const {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int p = p0 + j;
      if (p < pend) {
        I.u[j] = a.indices[p];
        if constexpr (NEED_EID) I.ee[j] = a.eid ? a.eid[p] : p;
        if constexpr (KIND >= kNormal) I.nn[j] = a.pos_lo + (a.nidx ? (uint32_t)a.nidx[p] : (uint32_t)p);
      }
    }
  }

  // The edge records are COMPLETE in registers before the row gathers are issued: the column ids have to be (the row
  // addresses are made of them), and pinning the noise indices here as well keeps the draw free of any dependence on
  // a load.  Without it the compiler sinks `pos_lo + nidx[p]` to the top of the Philox block and, unable to count
  // the conditional loads issued since, guards it with `s_waitcnt vmcnt(0)` — which also waits for the ROWS: the whole
  // draw of a block then ran after its rows had arrived instead of while they were in flight.
  __device__ __forceinline__ void pin_idx(EdgeIdx<NB>& I) const {
#if STAG_PIN_IDX
    if constexpr (KIND >= kNormal) {
#pragma unroll
      for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(I.nn[j]));
    }
#endif
    // ... and, for the narrowest shapes (4-8 blocks of rows fetched together), the column / edge ids: every row address
    // is then made of registers and no `s_waitcnt` for an id stands between two row gathers.  Measured per shape, no
    // draw | Normal, us: D = 16 31.6 -> 31.2 | 35.2 -> 35.1, D = 32 37.6 -> 35.0 | 45.7 -> 44.9; at LPE 16 it loses (D = 64 no
    // draw 54.9 -> 57.2), wider shapes and the Monte-Carlo kernels do not move.
    if constexpr (STAG_PIN_IDX && LPE <= 8) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        asm volatile("" : "+v"(I.u[j]));
        if constexpr (NEED_EID) asm volatile("" : "+v"(I.ee[j]));
      }
    }
  }

  __device__ __forceinline__ void fetch_rows(EdgeRows<NB, PEDGE>& R, const EdgeIdx<NB>& I, int p0) const {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (p0 + j < pend) {
#ifdef STAG_EXP_NO_ROWS   // experiment (tools/ab_bench.py): the launch without its row gathers — what the ids, the draw and the adds cost alone
        for (int q = 0; q < 4; ++q) R.xv[j][q] = __int_as_float(0x3f800000 | ((I.u[j] + q) & 0xff));
#else
        if (x_buf) bufrow4(rx, I.u[j], a.ldxb, koff, R.xv[j]);
        else loadrow4(row_at(a.x, I.u[j], a.ldxb, koff, (a.wide & 1) != 0), k0, a.D, VEC, R.xv[j]);
#endif
        if (a.src_scale) R.xs[j] = a.src_scale[I.u[j]];
        if constexpr (KIND >= kNormal && P1) {
          R.P.q0[j] = a.p0[I.ee[j]];
          R.P.q1[j] = a.p1 ? a.p1[I.ee[j]] : 0.0f;
        } else if constexpr (KIND >= kNormal && PEDGE == 2) {
          loadrow4(row_at(a.p0, I.ee[j], a.ldwb, koff, (a.wide & 2) != 0), k0, a.D, VEC, R.P.pa[j]);
          if (a.p1) loadrow4(row_at(a.p1, I.ee[j], a.ldwb, koff, (a.wide & 2) != 0), k0, a.D, VEC, R.P.pb[j]);
        }
      }
    }
  }

  // the blocks fetched at p0: draw, multiply, fold — one block (BLK edges) at a time
  __device__ __forceinline__ void compute(EdgeRows<NB, PEDGE>& R, const EdgeIdx<NB>& I, int p0) {
#pragma unroll
    for (int m = 0; m < MULT; ++m) {
    if (m > 0 && p0 + m * BLK >= pend) break;   // an empty block must not touch the Kahan state
    // block sums go into fresh accumulators (small magnitudes => small rounding)
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] ExtraAcc<NX> TX;              // block sums of the extra outputs (acc only)
#if STAG_DRAW_FIRST
    // The plain fused draw (one output, scalar / per-channel parameters): ALL the block's draws first, then the
    // multiplies — the draws need nothing from memory (pin_idx), so they run while the block's rows are in flight and
    // the wave meets its first `s_waitcnt` with the whole RNG of the block behind it.
    constexpr bool DRAW_FIRST = KIND >= kNormal && NX == 0 && PEDGE == 0;
    [[maybe_unused]] float wb[DRAW_FIRST ? BLK : 1][4];
    if constexpr (DRAW_FIRST) {
#pragma unroll
      for (int j = m * BLK; j < (m + 1) * BLK; ++j)
        if (p0 + j < pend) draw4<KIND>(I.nn[j], c1, key, pa, pb, a.relu, wb[j - m * BLK]);
    }
#else
    constexpr bool DRAW_FIRST = false;
    [[maybe_unused]] float wb[1][4];
#endif
#pragma unroll
    for (int j = m * BLK; j < (m + 1) * BLK; ++j) {
      if (p0 + j < pend) {
        float w[4];
        [[maybe_unused]] ExtraAcc<NX> dd;          // dd.acc[o] = derivative o of this edge's draw
        [[maybe_unused]] float g0[4], g1[4];        // PEDGE 3: the two derivatives of this edge's draw
        if constexpr (PEDGE == 3) {
          const float s1 = (a.relu & kFlagLogScale) ? exp_scale(R.P.q1[j]) : R.P.q1[j];
#pragma unroll
          for (int q = 0; q < 4; ++q) { pa[q] = R.P.q0[j]; pb[q] = s1; }
          draw4_grad<KIND>(I.nn[j], c1, key, pa, pb, a.relu, w, g0, g1);
        } else if constexpr (PEDGE == 4) {
          draw4_grad<KIND>(I.nn[j], c1, key, pa, pb, a.relu, w, g0, g1);
        } else if constexpr (DRAW_FIRST) {
#pragma unroll
          for (int q = 0; q < 4; ++q) w[q] = wb[j - m * BLK][q];
        } else if constexpr (NX == 0) {
          edge_weight(R, I, j, w);
        } else if constexpr (!MC) {
          draw4_grad<KIND>(I.nn[j], c1, key, pa, pb, a.relu, w, dd.acc[0], dd.acc[1]);
        } else {
          draw4<KIND>(I.nn[j], c1, key, pa, pb, a.relu, w);
#pragma unroll
          for (int o = 0; o < NX; ++o) draw4<KIND>(I.nn[j], c1, KX.k[o], pa, pb, a.relu, dd.acc[o]);
          if constexpr (WN) {
#pragma unroll
            for (int o = 0; o < NX; ++o)
#pragma unroll
              for (int q = 0; q < 4; ++q) XW.w[o][q] += dd.acc[o][q];
          }
        }
        if (a.src_scale) {
          asm volatile("" ::: "memory");   // keep this a branch: as selects it costs 6 VALU ops per edge
#pragma unroll
          for (int q = 0; q < 4; ++q) R.xv[j][q] *= R.xs[j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = __builtin_fmaf(w[q], R.xv[j][q], t[q]);
        if constexpr (NX > 0) {
#pragma unroll
          for (int o = 0; o < NX; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) TX.acc[o][q] = __builtin_fmaf(dd.acc[o][q], R.xv[j][q], TX.acc[o][q]);
        }
        if constexpr (PEDGE == 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float gx = R.xv[j][q] * XO.v[q];
            DP.v[0][q] = __builtin_fmaf(g0[q], gx, DP.v[0][q]);
            DP.v[1][q] = __builtin_fmaf(g1[q], gx, DP.v[1][q]);
          }
        }
        if constexpr (PEDGE == 3) {
          // d L / d p_i of this edge = sum_k dw/dp_i[k] * (gathered row)[k] * (own row)[k]: the channel
          // tile is the whole row (one tile: checked on the host), so one team sum finishes it
          float e0 = 0.f, e1 = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float gx = R.xv[j][q] * XO.v[q];
            e0 = __builtin_fmaf(g0[q], gx, e0);
            e1 = __builtin_fmaf(g1[q], gx, e1);
          }
          e0 = team_sum_lane0<LPE>(e0, k0 >> 2, (a.D + 3) >> 2);
          e1 = team_sum_lane0<LPE>(e1, k0 >> 2, (a.D + 3) >> 2);
          if (k0 == 0) {
            a.eg0[I.ee[j]] = e0;
            if (a.eg1) a.eg1[I.ee[j]] = e1;
          }
        }
        if (PEDGE != 3 && PEDGE != 4 && a.in_norm) {     // (stag_agg_bwd_edge / _dp: in_norm 0, checked on the host)
          asm volatile("" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) wsum[q] += w[q];   // edge order; 0/1 draws (Bernoulli + norm): exact
        }
      }
    }
    fold(t);
    if constexpr (NX > 0) {
#pragma unroll
      for (int o = 0; o < NX; ++o) fold_into(X.acc[o], X.comp[o], TX.acc[o]);
    }
    }   // m
  }

  // fold one block into the unit's sum; compensated (Kahan) once a unit is long enough for
  // the running sum to dwarf a block, so a 13k-edge hub row keeps ~1e-6 relative accuracy
  __device__ __forceinline__ void fold(const float (&t)[4]) { fold_into(acc, comp, t); }

  __device__ __forceinline__ void fold_into(float (&s)[4], float (&cmp)[4], const float (&t)[4]) const {
    if (kahan) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float y = t[q] - cmp[q];
        const float sum = s[q] + y;
        cmp[q] = (sum - s[q]) - y;
        s[q] = sum;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += t[q];
    }
  }

  // Block sums only (register staging), for the slotted loop: t[m] of this lane's MULT blocks
  // starting at p0 and the weights we[j] of its NB edges (in-norm sums them in edge order);
  // blocks past the unit's end stay zero and are never folded.
  __device__ __forceinline__ void block_sums(EdgeRows<NB, PEDGE>& R, const EdgeIdx<NB>& I, int p0,
                                             float (&t)[MULT][4], float (&we)[NB][4]) {
    // (all the round's weights first, then the multiplies — what the one-slot loop does since round 3 — was measured
    // here too: Bernoulli at D = 64 68.1 -> 69.9 us, everything else within noise; the narrow shapes keep draw-then-multiply)
#pragma unroll
    for (int m = 0; m < MULT; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) t[m][q] = 0.f;
#pragma unroll
      for (int j = m * BLK; j < (m + 1) * BLK; ++j) {
        if (p0 + j < pend) {
          edge_weight(R, I, j, we[j]);       // (straight into the slot's weight record: no copy — at LPE 16 the copy
                                             //  cost 2 VGPRs and with them the 6th wave per SIMD: D = 64 Normal 72.9 -> 66.9 us)
          if (a.src_scale) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) R.xv[j][q] *= R.xs[j];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) t[m][q] = __builtin_fmaf(we[j][q], R.xv[j][q], t[m][q]);
        }
      }
    }
  }

  // w[0..3]: the multiplicative weight of edge j of the fetched set on this lane's channels
  __device__ __forceinline__ void edge_weight(const EdgeRows<NB, PEDGE>& R, const EdgeIdx<NB>& I, int j,
                                              float (&w)[4]) {
    if constexpr (KIND == kNone) {
      w[0] = w[1] = w[2] = w[3] = 1.0f;
    } else if constexpr (KIND == kExplicit) {
      if (a.wgroup > 1) {   // one weight per `wgroup` channels (GAT heads: a[e,h] over F)
        const float* wr = a.p0 + (int64_t)I.ee[j] * (a.D / a.wgroup);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (k0 + q < a.D) ? wr[(k0 + q) / a.wgroup] : 0.0f;
      } else {
        loadrow4(row_at(a.p0, I.ee[j], a.ldwb, koff, (a.wide & 2) != 0), k0, a.D, VEC, w);
      }
      if (a.relu & kFlagRelu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = fmaxf(w[q], 0.0f);
      }
    } else {
      if constexpr (PEDGE == 1) {
        const float s1 = (a.relu & kFlagLogScale) ? exp_scale(R.P.q1[j]) : R.P.q1[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) { pa[q] = R.P.q0[j]; pb[q] = s1; }
        draw4<KIND>(I.nn[j], c1, key, pa, pb, a.relu, w);
        return;
      } else if constexpr (PEDGE == 2) {
        if (a.relu & kFlagLogScale) {         // [E, D] log-scales exponentiated where they are used: no [E, D] exp pass
          float pbe[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pbe[q] = exp_scale(R.P.pb[j][q]);
          draw4<KIND>(I.nn[j], c1, key, R.P.pa[j], pbe, a.relu, w);
        } else {
          draw4<KIND>(I.nn[j], c1, key, R.P.pa[j], R.P.pb[j], a.relu, w);
        }
        return;
      }
      draw4<KIND>(I.nn[j], c1, key, pa, pb, a.relu, w);
    }
  }
};

// One unit (a whole row or a segment of a long row) on LPE x SLOTS lanes of a wave:
// c = this lane's chunk (4 channels) of the channel tile, sl = its edge slot.
template <int KIND, int LPE, bool VEC, int PEDGE, int SLOTS, int MULT, int NOUT = 1, bool MC = false, bool WN = false,
          bool NULLS = false>
__device__ __forceinline__ void agg_unit(const AggArgs& a, const int unit, const int c, const int sl,
                                         float (*dp_out)[4] = nullptr) {
  static_assert(LPE * SLOTS <= 64 && 64 % (LPE * SLOTS) == 0, "a unit's lanes stay inside one wave");
  static_assert(NOUT == 1 || SLOTS == 1, "the derivative outputs take the one-slot loop");
  // edges per block: the RNG kinds are VALU-bound and register-hungry, the others want
  // more rows in flight
  // with the per-edge gradients one edge per block: 66 VGPRs = 7 waves per SIMD instead of 76 = 6 (the layer step
  // of tools/layer_step.py --mode re --kl: 988 against 1019 us)
#ifndef STAG_BLK_EG
#define STAG_BLK_EG 1
#endif
  // (the three-accumulator backward, NOUT 3, at one edge per block: 84 VGPRs = 5 waves instead of 104 = 4, and
  // no faster: 915-1040 against 869-885 us for the r1 layer step; it keeps the block of its separate passes)
  // (PEDGE 4 at one edge per block as well: agg_dp_kernel 189.5 against 204 us at cfg2)
#ifndef STAG_BLK_DP
#define STAG_BLK_DP 1
#endif
  constexpr int BLK = PEDGE == 3 ? STAG_BLK_EG : PEDGE == 4 ? STAG_BLK_DP : (KIND >= kNormal) ? STAG_BLK_RNG : STAG_BLK_MEM;
  constexpr int NB = BLK * MULT;

  const uint32_t chunk = blockIdx.y * LPE + c;
  const int k0 = (int)chunk * 4;
  // lanes past the row's end (units never talk: no barrier below; the team sums of PEDGE 3 are written
  // for teams whose last lanes have left: team_sum_lane0)
  if (k0 >= a.D) return;

  int v, b, len, slot = -1;
  if (a.units) {
    const int4 q = *reinterpret_cast<const int4*>(a.units + unit);
    v = q.x; b = q.y; len = q.z; slot = q.w;
    if constexpr (NULLS) {
      if (v < 0) return;        // a null record: the padding of an XCD stripe (stag_plan.xcd_order)
    }
  } else {
    v = unit;
    b = a.indptr[v];
    len = a.indptr[v + 1] - b;
  }

  // Long units are the critical path of the launch: a SIMD round-robins its waves, so a
  // 64-edge unit would take 8x its own issue time at 8 waves per SIMD.  They are dispatched
  // first (plan order) and run at raised priority; the short rows fill in behind them.
  if (len > STAG_PRIO_MIN_LEN) __builtin_amdgcn_s_setprio(2);
#ifdef STAG_TRACE   // tools/trace_units.py: per-unit timestamps (100 MHz) into the norm-scale buffer
  uint64_t* trace = (!a.in_norm && a.norm_scale_out && blockIdx.y == 0 && c == 0 && sl == 0)
                        ? reinterpret_cast<uint64_t*>(a.norm_scale_out) + (int64_t)unit * 4 : nullptr;
  if (trace) { trace[0] = wall_clock64(); trace[1] = trace[2] = trace[3] = 0; }
#endif

  AggTeam<KIND, LPE, VEC, PEDGE, BLK, MULT, NOUT, MC, WN> T{
      a, (KIND >= kNormal) ? resolve_epoch(a.key) : a.key, k0, (uint32_t)k0 * 4u,
      (chunk + a.chunk_base) | (a.pos_hi << 20),   // Philox counter word 1: a per-lane constant
      b + len,
      // descriptor of x for the narrow (< 4 GB, ids < 2^24) case; kernel arguments only
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000),
      // rows go through the descriptor as ONE dwordx4 whenever the narrow form applies — also when
      // D is not a multiple of 4: the load is dword-aligned, the tail lanes pick up neighbouring
      // floats (or 0 past the end) that never reach a store
      (a.wide & 1) == 0 && a.x_bytes != 0,
      {a.p0s, a.p0s, a.p0s, a.p0s}, {a.p1s, a.p1s, a.p1s, a.p1s},
      {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f},
      len > kKahanMinLen};
  if constexpr (MC) {
#pragma unroll
    for (int o = 0; o < NOUT - 1; ++o) T.KX.k[o] = key_plus(T.key, (uint64_t)(o + 1) * a.mc_stride);
  }
  if constexpr (PEDGE == 3 || PEDGE == 4) {
    const int row = slot >= 0 ? a.long_rows[v] : v;
    const float os = a.own_scale ? a.own_scale[row] : 1.0f;
    if (PEDGE == 3 || a.xown) {
      load4(a.xown + (int64_t)row * a.ldxo, k0, a.D, VEC, T.XO.v);
    } else {                      // no own row: ones (the in-norm term of the parameter gradients)
#pragma unroll
      for (int q = 0; q < 4; ++q) T.XO.v[q] = (k0 + q < a.D) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) T.XO.v[q] *= os;
  }
  if constexpr (KIND >= kNormal) {
    if (a.pmode == STAG_PARAM_PER_CHANNEL) {   // distribution parameters of this lane's 4 channels
      load4(a.p0, k0, a.D, VEC, T.pa);
      if (a.p1) load4(a.p1, k0, a.D, VEC, T.pb);
    }
  }

  // One block at a time: ids -> rows -> draws.  Measured on MI355X (tools/ab_bench.py, cfg2,
  // Normal noise): this loop at 2 edges per block 113 us/step; 4 edges per block 126 us;
  // an A/B software pipeline that keeps the next block's rows in flight 133 us (the extra
  // registers cost a wave per SIMD, and occupancy hides the gather latency better);
  // persistent teams striding over the unit list (grid = chip) 140-170 us, also without
  // noise (104 vs 98 us): many short-lived waves beat few long-lived ones here.
  const int pend = b + len;
  EdgeIdx<NB> I;
  EdgeRows<NB, PEDGE> R;
  {
    // the next block's edge records are fetched while this block's rows are in flight: one
    // round trip per block on the unit's critical path instead of two.  Narrow shapes only
    // (their launch is latency-bound: -2..3 us); at LPE >= 32 the extra registers would cost
    // the RNG kinds their 8th wave per SIMD for no gain.
    // (with the draw, at LPE >= 32, it loses: 110 against 104 us at D = 128, 221 against 209 at D = 256, round 3)
    constexpr bool PREFETCH = STAG_IDX_PREFETCH && LPE <= ((KIND >= kNormal) ? STAG_PREFETCH_MAX_LPE : STAG_PREFETCH_MAX_LPE_MEM);
    if constexpr (SLOTS > 1) {
      constexpr int RB = SLOTS * NB;               // edges of the unit per round
      const int base = ((int)(threadIdx.x & 63) - sl * LPE) << 2;   // slot 0's lane of my channels (bpermute address)
      for (int r0 = b; r0 < pend; r0 += RB) {
        const int p0 = r0 + sl * NB;
#if STAG_LOAD_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        T.fetch_idx(I, p0);
        T.pin_idx(I);
        T.fetch_rows(R, I, p0);
#if STAG_LOAD_PRIO
        if (len > STAG_PRIO_MIN_LEN) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
        float t[MULT][4], we[NB][4];
        T.block_sums(R, I, p0, t, we);
        // every slot folds all the round's blocks in block order: the same sequence of adds as
        // the one-slot loop, and every slot ends with the unit's sum
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
#pragma unroll
          for (int m = 0; m < MULT; ++m) {
            const int pb = r0 + (j * MULT + m) * BLK;     // first edge of slot j's block m
            if (pb < pend) {                              // uniform over the unit's lanes
              float tj[4];
#pragma unroll
              for (int q = 0; q < 4; ++q)
                tj[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(base + j * (LPE << 2), __float_as_int(t[m][q])));
              T.fold(tj);
              if (a.in_norm) {
#pragma unroll
                for (int e = 0; e < BLK; ++e) {
                  if (pb + e < pend) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                      T.wsum[q] += __int_as_float(__builtin_amdgcn_ds_bpermute(
                          base + j * (LPE << 2), __float_as_int(we[m * BLK + e][q])));
                  }
                }
              }
            }
          }
        }
      }
    } else if constexpr (PREFETCH) {
    EdgeIdx<NB> In;
    T.fetch_idx(I, b);
    for (int p0 = b; p0 < pend; p0 += NB) {
#if STAG_LOAD_PRIO
      __builtin_amdgcn_s_setprio(3);   // get the loads out ahead of other waves' draws
#endif
      T.pin_idx(I);
      T.fetch_rows(R, I, p0);
      if (p0 + NB < pend) T.fetch_idx(In, p0 + NB);
#if STAG_LOAD_PRIO
      if (len > STAG_PRIO_MIN_LEN) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
      T.compute(R, I, p0);
      I = In;
    }
    } else {
    for (int p0 = b; p0 < pend; p0 += NB) {
#if STAG_LOAD_PRIO
      __builtin_amdgcn_s_setprio(3);   // get the loads out ahead of other waves' draws
#endif
      T.fetch_idx(I, p0);
      T.pin_idx(I);
      T.fetch_rows(R, I, p0);
#if STAG_LOAD_PRIO
      if (len > STAG_PRIO_MIN_LEN) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
      T.compute(R, I, p0);
    }
    }
  }

#ifdef STAG_TRACE
  if (trace) trace[1] = wall_clock64();
#endif
  if constexpr (PEDGE == 4) {     // the lane's share of the parameter gradients: complete once the edges are walked
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) dp_out[i][q] = T.DP.v[i][q];
  }
  if (slot < 0) {
    if (sl != 0) return;                // every slot holds the row's sum; slot 0 writes it
    agg_epilogue(a, v, len, k0, VEC, T.acc, T.wsum);
    if constexpr (NOUT > 1) {
#pragma unroll
      for (int o = 0; o < NOUT - 1; ++o) {
        if constexpr (WN) agg_epilogue_to(a, a.outx[o], nullptr, v, len, k0, VEC, T.X.acc[o], T.XW.w[o]);
        else agg_epilogue_extra(a, a.outx[o], v, len, k0, VEC, T.X.acc[o]);
      }
    }
#ifdef STAG_TRACE
    if (trace) trace[3] = wall_clock64();
#endif
    return;
  }
  // ---- segment of a long row: publish the partial, the last arriver sums the row ----------
  const int r = v;                      // for segments the unit names the long row by index
  // Producer side (Guideline 16, R1): the partial is stored WRITE-THROUGH (sc1), so there is
  // no release fence (a buffer_wbl2 per segment made the whole launch 2x slower); every
  // storing wave drains its stores, then one lane per team takes a ticket.
  const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)a.ws_bytes, 0x00020000);
  const uint32_t woff = (uint32_t)slot * ((uint32_t)a.ws_stride * 4u) + (uint32_t)k0 * 4u;
  if (sl == 0) {
    store4_sc1(rws, woff, k0, a.D, VEC, T.acc);
    // a segment's row of the workspace: the partial sums of output o at o * D, its weight sums (in-norm) at (NOUT + o) * D
    if (PEDGE != 3 && PEDGE != 4 && a.in_norm) store4_sc1(rws, woff + (uint32_t)(NOUT * a.D) * 4u, k0, a.D, VEC, T.wsum);
    if constexpr (NOUT > 1) {
#pragma unroll
      for (int o = 0; o < NOUT - 1; ++o) {
        store4_sc1(rws, woff + (uint32_t)(o + 1) * (uint32_t)a.D * 4u, k0, a.D, VEC, T.X.acc[o]);
        if constexpr (WN) store4_sc1(rws, woff + (uint32_t)(NOUT + o + 1) * (uint32_t)a.D * 4u, k0, a.D, VEC, T.XW.w[o]);
      }
    }
  }
  const int s0 = a.long_seg_ptr[r], s1 = a.long_seg_ptr[r + 1];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int ticket = 0;
  int32_t* counter = a.seg_counters + (int64_t)blockIdx.y * a.n_long + r;   // per channel tile
  if (c == 0 && sl == 0)
    ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int lane0 = ((int)(threadIdx.x & 63) - c - sl * LPE) << 2;   // the unit's first lane
  ticket = __builtin_amdgcn_ds_bpermute(lane0, ticket);
#ifdef STAG_TRACE
  if (trace) trace[2] = wall_clock64();
#endif
  if (ticket != (s1 - s0) - 1) return;
  // consumer side: this team drew the last ticket; acquire, then plain loads
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (c == 0 && sl == 0) *counter = 0;  // leave the counter ready for the next call
  const int row = a.long_rows[r];
  const int deg = a.indptr[row + 1] - a.indptr[row];
  // Sum the row's partials with compensated (Kahan) fp32 adds, two levels: groups of 16
  // partials in segment order, then the group sums in group order — a hub row has hundreds of
  // partials and the unit's slots take a group each (the sum is the same for any slot count).
  // NF partials are in flight at a time; the tail is written to stay under the hot loop's
  // register count (fp64 accumulators or a wider NF cost a wave per SIMD).
#ifndef STAG_COMBINE_NF
#define STAG_COMBINE_NF 8
#endif
  constexpr int NF = STAG_COMBINE_NF;
  float facc[4], fws[4] = {0.f, 0.f, 0.f, 0.f};
  const int slot0 = lane0 + (c << 2);   // slot 0's lane of my channels
  two_level_sum<NF, VEC, LPE, SLOTS>(a.ws, a.ws_stride, s0, s1, k0, a.D, sl, slot0, facc);
  if (PEDGE != 3 && PEDGE != 4 && a.in_norm) two_level_sum<NF, VEC, LPE, SLOTS>(a.ws + NOUT * a.D, a.ws_stride, s0, s1, k0, a.D, sl, slot0, fws);
  if (sl != 0) return;
  agg_epilogue(a, row, deg, k0, VEC, facc, fws);
#pragma unroll
  for (int o = 0; o < NOUT - 1; ++o) {
    two_level_sum<NF, VEC, LPE, SLOTS>(a.ws + (o + 1) * a.D, a.ws_stride, s0, s1, k0, a.D, sl, slot0, facc);
    if constexpr (WN) {
      two_level_sum<NF, VEC, LPE, SLOTS>(a.ws + (NOUT + o + 1) * a.D, a.ws_stride, s0, s1, k0, a.D, sl, slot0, fws);
      agg_epilogue_to(a, a.outx[o], nullptr, row, deg, k0, VEC, facc, fws);
    } else {
      agg_epilogue_extra(a, a.outx[o], row, deg, k0, VEC, facc);
    }
  }
#ifdef STAG_TRACE
  if (trace) trace[3] = wall_clock64();
#endif
}

// Lanes per unit.  Light units (the many short rows) take LPE lanes; HEAVY units — the first
// a.n_heavy of the plan, every unit longer than STAG_HEAVY_LEN edges: segments of long rows and
// the longest whole rows — are the launch's critical path when rows are narrow (D <= 64: the
// launch is latency-bound, tools/trace_units.py) and take LPE x heavy_slots lanes.
template <int KIND, int LPE>
constexpr int mult_of() {
  // blocks a LIGHT unit fetches together.  Measured with heavy slots on (tools/ab_bench.py, arxiv
  // CSR, us per launch none | normal): D=16 8 | 4 (42 | 47), D=32 4 | 1 (46 | 51), D=64 2 | 1
  // (59 | 81); deeper costs the RNG kinds their occupancy, shallower the others their overlap.
  constexpr bool RNG = KIND >= kNormal;
  return LPE <= 4 ? (RNG ? STAG_MULT_LPE4 : 2 * STAG_MULT_LPE4)
         : LPE == 8 ? (RNG ? STAG_MULT_LPE8 : 4 * STAG_MULT_LPE8)
         : LPE == 16 ? (RNG ? STAG_MULT_LPE16 : 2 * STAG_MULT_LPE16)
                     : STAG_MULT_WIDE;
}
#ifndef STAG_HSLOTS_LPE4
#define STAG_HSLOTS_LPE4 4
#endif
#ifndef STAG_HSLOTS_LPE8
#define STAG_HSLOTS_LPE8 4
#endif
#ifndef STAG_HSLOTS_LPE16
#define STAG_HSLOTS_LPE16 2
#endif
#ifndef STAG_HSLOTS_LPE32
#define STAG_HSLOTS_LPE32 1
#endif
#ifndef STAG_HMULT
#define STAG_HMULT 1
#endif
// At 16 lanes per row (D in 33..64) Normal and Uniform take no slots (round 4): the slotted loop costs them registers and
// with them a wave per SIMD, and their launch is the draw's, not the hub rows' — tools/ab_bench.py, us, two slots | none:
//   PPI batch (BASELINE configs[2], layer 1) D = 52 Normal 41.8 | 36.8, D = 64 43.1 | 38.5; no draw at D = 52 24.2 | 26.6;
//   arxiv (a 13k-edge hub) D = 64 Normal 67.6 | 68.7, Uniform 69.1 | 69.1, D = 48 Normal 65.8 | 66.7 — and Bernoulli + in-norm
//   68.6 | 87.2: its draw is cheap, its launch IS the hub row's, so Bernoulli (like no draw, explicit weights) keeps them.
#ifndef STAG_HSLOTS_LPE16_RNG
#define STAG_HSLOTS_LPE16_RNG 1
#endif
// At 32 lanes per row (D in 65..128) a launch too small to be bound by the fabric is bound by its longest unit: a 64-edge
// unit walked two edges per round trip is 32 trips = 30 us, whatever else the launch holds (a shard of an 8-way partition
// of the arxiv graph: 146 k edges, 33 us).  Such launches (SMALL: a plan of at most STAG_SMALL_UNITS
// units, plain family, plan order) give their heavy units two edge slots — tools/ab_bench-style A/B of four builds in one
// process, us with | without the draw, one slot | two:  shard of 8: 34.3 | 33.1 -> 31.9 | 25.1; the hub's shard 52.6 | 51.2 ->
// 42.9 | 36.3; shard of 4: 58.3 | 57.9 -> 47.9 | 42.8; shard of 2 (583 k edges): 54.5 | 60.5 -> 60.5 | 49.9; the whole graph
// 103.6 | 95.0 -> 119.3 | 100.7 — so the big launches keep one.  Same bits (the slots fold their blocks in block order).
#ifndef STAG_HSLOTS_LPE32_SMALL
#define STAG_HSLOTS_LPE32_SMALL 2
#endif
#ifndef STAG_SMALL_UNITS
#define STAG_SMALL_UNITS 49152
#endif
template <int KIND, int LPE, bool SMALL = false>
constexpr int heavy_slots_of() {
  return LPE <= 4 ? STAG_HSLOTS_LPE4 : LPE == 8 ? STAG_HSLOTS_LPE8
         : LPE == 16 ? ((KIND == kNormal || KIND == kUniform) ? STAG_HSLOTS_LPE16_RNG : STAG_HSLOTS_LPE16)
         : LPE == 32 ? (SMALL ? STAG_HSLOTS_LPE32_SMALL : STAG_HSLOTS_LPE32) : 1;
}

// The kernel arguments a unit needs before its first gather, fetched TOGETHER at the top of the kernel: left to
// itself the compiler loads each group at its first use, behind the early exits — a chain of 8 scalar loads, each
// waited for on its own, before the unit record can even be asked for.  The empty asm makes every value live here,
// so the loads are issued back to back and waited for once.  Measured at cfg2 (tools/ab_bench.py, one process): Normal
// 102.3 -> 101.2 us, Bernoulli 101.9 -> 101.7; the no-noise kernel 96.3 -> 97.7 (its waves reach the gather sooner
// than the fabric wants them), so the kinds that draw nothing keep the compiler's lazy loads.
#ifndef STAG_HOIST_ARGS
#define STAG_HOIST_ARGS 1
#endif
#ifndef STAG_HOIST_PLAIN
#define STAG_HOIST_PLAIN 0      // the kinds that draw nothing: hoisting measured 1.5 us SLOWER at cfg2 (97.8 against 96.3)
#endif
#define STAG_PIN_S(x) asm volatile("" : "+s"(x))
#ifndef STAG_HOIST_MIN_LPE
#define STAG_HOIST_MIN_LPE 32   // the narrow shapes (two inlined loops, SGPRs spilling already) lose 2 us of 35 to it
#endif
template <int KIND, int LPE, bool WALK>
__device__ __forceinline__ void hoist_args(AggArgs& l) {
#if STAG_HOIST_ARGS
  if constexpr (WALK) {     // the walk (one aligned s_load_dwordx8 instead of three loads and two waits) for every kind
    STAG_PIN_S(l.walk.smask); STAG_PIN_S(l.walk.sshift); STAG_PIN_S(l.walk.jh_heavy); STAG_PIN_S(l.walk.jh_light);
    STAG_PIN_S(l.walk.sh); STAG_PIN_S(l.walk.lbase); STAG_PIN_S(l.walk.sl); STAG_PIN_S(l.walk.n_total);
  }
  if constexpr ((KIND >= kNormal || STAG_HOIST_PLAIN) && LPE >= STAG_HOIST_MIN_LPE) {
    STAG_PIN_S(l.D); STAG_PIN_S(l.units); STAG_PIN_S(l.indptr); STAG_PIN_S(l.indices);
    STAG_PIN_S(l.x); STAG_PIN_S(l.ldxb); STAG_PIN_S(l.x_bytes); STAG_PIN_S(l.wide); STAG_PIN_S(l.src_scale);
  }
  if constexpr (KIND >= kNormal && LPE >= STAG_HOIST_MIN_LPE) {
    STAG_PIN_S(l.eid); STAG_PIN_S(l.nidx);
    STAG_PIN_S(l.p0); STAG_PIN_S(l.p1); STAG_PIN_S(l.p0s); STAG_PIN_S(l.p1s); STAG_PIN_S(l.pmode); STAG_PIN_S(l.relu);
    STAG_PIN_S(l.in_norm);
    STAG_PIN_S(l.key.k0); STAG_PIN_S(l.key.k1); STAG_PIN_S(l.key.o0); STAG_PIN_S(l.key.o1); STAG_PIN_S(l.key.epoch);
    STAG_PIN_S(l.pos_lo); STAG_PIN_S(l.pos_hi); STAG_PIN_S(l.chunk_base);
  }
#endif
}

// WALK = false: block b takes units [b * teams, (b + 1) * teams) of the plan's order (heavy blocks first).  WALK = true
// (the plain family only: one output, scalar / per-channel parameters): the stripes of AggArgs::Walk — its own
// instantiation, so that the plan-order kernels keep the code they were tuned with (as one kernel with a run-time
// walk the narrow shapes lost 2 us of 35 with noise, tools/ab_bench.py).
template <int KIND, int LPE, bool VEC, int PEDGE, int NOUT = 1, bool MC = false, bool WN = false, bool WALK = false,
          bool SMALL = false>
__global__ __launch_bounds__(STAG_BLOCK_THREADS, STAG_WAVES_PER_SIMD) void agg_kernel(const AggArgs a_in) {
  AggArgs a = a_in;
  hoist_args<KIND, LPE, WALK>(a);
  static_assert(!SMALL || (LPE == 32 && !WALK && PEDGE == 0 && NOUT == 1), "SMALL: the plain plan-order launch at 32 lanes per row");
  constexpr int HS = (NOUT == 1 && PEDGE != 3) ? heavy_slots_of<KIND, LPE, SMALL>() : 1;
  const int c = threadIdx.x % LPE;                // chunk lane inside the channel tile
  constexpr int TPB = STAG_BLOCK_THREADS / LPE, TPBH = STAG_BLOCK_THREADS / (LPE * HS);
  if constexpr (!WALK) {
    int first = 0, blk = blockIdx.x;
    if constexpr (HS > 1) {
      if (blk < a.n_heavy_blocks) {                 // block-uniform
        const int unit = blk * TPBH + threadIdx.x / (LPE * HS);
        if (unit >= a.n_heavy) return;              // teams never talk to each other: no barrier below
        agg_unit<KIND, LPE, VEC, PEDGE, HS, STAG_HMULT>(a, unit, c, (threadIdx.x / LPE) % HS);
        return;
      }
      first = a.n_heavy;
      blk -= a.n_heavy_blocks;
    }
    const int unit = first + blk * TPB + threadIdx.x / LPE;
    if (unit >= a.n_units) return;
    agg_unit<KIND, LPE, VEC, PEDGE, 1, NOUT == 1 ? mult_of<KIND, LPE>() : 1, NOUT, MC, WN>(a, unit, c, 0);
  } else {
    // first unit of this block, the end of the range it walks, heavy: the slotted loop (all block-uniform).  An XCD
    // stripe is padded with null records, which agg_unit skips (reading the stripe's own length from the order's
    // header instead is one more dependent load in every block's prologue: the molecule batch with noise 27.0 -> 29.9 us).
    const AggArgs::Walk w = a.walk;
    const int stripe = blockIdx.x & w.smask;
    int j = blockIdx.x >> w.sshift;
    int unit0, end;
    bool heavy = false;
    if (HS > 1 && j < w.jh_heavy) {
      heavy = true;
      unit0 = stripe * w.sh + j * TPBH;
      end = (stripe + 1) * w.sh;
    } else {
      if (HS > 1) j -= w.jh_heavy;
      if (j < w.jh_light) {
        unit0 = stripe * w.sh + j * TPB;
        end = (stripe + 1) * w.sh;
      } else {
        j -= w.jh_light;
        unit0 = w.lbase + stripe * w.sl + j * TPB;
        end = min(w.lbase + (stripe + 1) * w.sl, w.n_total);
      }
    }
    if constexpr (HS > 1) {
      if (heavy) {
        const int unit = unit0 + threadIdx.x / (LPE * HS);
        if (unit >= end) return;
        agg_unit<KIND, LPE, VEC, PEDGE, HS, STAG_HMULT, 1, false, false, true>(a, unit, c, (threadIdx.x / LPE) % HS);
        return;
      }
    }
    const int unit = unit0 + threadIdx.x / LPE;
    if (unit >= end) return;
    // The walk instantiation of the no-draw kernel — what an XCD-aware order or a plan-less row-striped launch runs, i.e. a
    // graph whose rows come out of an XCD's L2 — fetches TWO blocks (four rows) at a time where a row takes a whole wave
    // (round 4): the L2-resident gather is bound by latency, not by the fabric.  Two blocks of two edges, not one
    // block of four: a block's sum is formed from zero and folded into the unit's sum, so the order of additions — and
    // with it every bit of the result — is that of every other launch (one block of four was 2 us faster at D = 256 and
    // is not the same sum).  tools/ab_bench.py, us, 1 | 2 blocks: PPI batch D = 256 71.8 | 68.0, D = 128 36.1 | 35.1;
    // narrower rows already fetch several blocks together (mult_of); on the arxiv graph (rows from the Infinity Cache:
    // plan order, the other instantiation) more rows in flight change nothing (95.3 | 95.6).
#ifndef STAG_MULT_WALK_WIDE
#define STAG_MULT_WALK_WIDE 2
#endif
    // (LPE 64 only: at 32 lanes per row the PPI batch gains 1 us of 36 and the molecule batch — rows of 2-4 edges, for
    // which the second block is empty — loses 0.7 us of 22.4)
    constexpr int MULT_W = NOUT != 1 ? 1 : (KIND == kNone && PEDGE == 0 && LPE >= 64) ? STAG_MULT_WALK_WIDE : mult_of<KIND, LPE>();
    agg_unit<KIND, LPE, VEC, PEDGE, 1, MULT_W, NOUT, MC, WN, true>(a, unit, c, 0);
  }
}

// The dx pass that also sums the gradients of scalar / per-channel parameters (PEDGE 4; stag_agg_bwd_dp).  A unit's
// lanes keep their share of sum_e dw/dp_i * g * x in registers; the block's teams are added through LDS in team
// order and the block leaves ONE partial [2][LPE * 4]; a second launch adds the blocks in order (api.hip).  No
// [N, D] aggregate per derivative, no column-dot pass over them.
template <int KIND, int LPE, bool VEC>
__global__ __launch_bounds__(STAG_BLOCK_THREADS, STAG_WAVES_PER_SIMD) void agg_dp_kernel(const AggArgs a) {
  constexpr int TEAMS = STAG_BLOCK_THREADS / LPE;
  __shared__ float s_dp[TEAMS][2][LPE * 4];
  const int c = threadIdx.x % LPE, tm = threadIdx.x / LPE;
  const int unit = blockIdx.x * TEAMS + tm;
  float dp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (unit < a.n_units) agg_unit<KIND, LPE, VEC, 4, 1, 1>(a, unit, c, 0, dp);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) s_dp[tm][i][c * 4 + q] = dp[i][q];
  __syncthreads();
  float* part = a.dp_part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * LPE * 4);
  for (int i = threadIdx.x; i < 2 * LPE * 4; i += STAG_BLOCK_THREADS) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < TEAMS; ++j) sum += s_dp[j][i / (LPE * 4)][i % (LPE * 4)];
    part[i] = sum;
  }
}

// Launch one (KIND, PEDGE) family; defined per kind in agg_<kind>.hip so the
// instantiations compile in parallel.
template <int KIND>
hipError_t agg_launch(const AggArgs& a, bool vec, hipStream_t stream);

// Dynamic LDS the launch asks for although the kernel uses none: caps the workgroups a CU admits
// (160 KB / bytes).  A/B knob (tools/ab_bench.py): the gather saturates the fabric with fewer waves in
// flight than the register budget allows, and past that point more of them slow it down.
#ifndef STAG_XCD_PLANLESS
#define STAG_XCD_PLANLESS 1     // A/B knob: 0 = launches without a plan keep the linear block -> row mapping
#endif
#ifndef STAG_AGG_LDS_BYTES
#define STAG_AGG_LDS_BYTES 0
#endif
// the SMALL instantiation of the plain plan-order launch at 32 lanes per row (heavy_slots_of)
template <int KIND>
inline void agg_launch_small(const AggArgs& a_in, bool vec, int tiles, hipStream_t s) {
  AggArgs a = a_in;
  constexpr int LPE = 32, TPB = STAG_BLOCK_THREADS / LPE, TPBH = STAG_BLOCK_THREADS / (LPE * heavy_slots_of<KIND, LPE, true>());
  a.n_heavy_blocks = (a.n_heavy + TPBH - 1) / TPBH;
  const dim3 grid(a.n_heavy_blocks + (a.n_units - a.n_heavy + TPB - 1) / TPB, tiles), block(STAG_BLOCK_THREADS);
  if (grid.x == 0) return;
  if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 0, 1, false, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
  else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 0, 1, false, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
}

template <int KIND, int LPE>
inline void agg_launch_shape(const AggArgs& a_in, bool vec, int pedge, int tiles, hipStream_t s) {
  if constexpr (LPE == 32 && heavy_slots_of<KIND, 32, true>() != heavy_slots_of<KIND, 32, false>()) {
    if (!a_in.outx[0] && !a_in.dp_part && pedge == 0 && !a_in.xcd && a_in.units && a_in.n_heavy > 0 &&
        a_in.n_units <= STAG_SMALL_UNITS) {
      agg_launch_small<KIND>(a_in, vec, tiles, s);
      return;
    }
  }
  AggArgs a = a_in;
  constexpr int TPB = STAG_BLOCK_THREADS / LPE;
  constexpr int HS = heavy_slots_of<KIND, LPE>();
  constexpr int TPBH = STAG_BLOCK_THREADS / (LPE * HS);
  const bool slotted = !(HS == 1 || a.outx[0] || pedge == 3 || a.dp_part);
  if (!slotted) a.n_heavy = 0;
  // Two sets of instantiations.  The plain family (one output, scalar / per-channel parameters: the headline) keeps a
  // kernel WITHOUT the walk for plan-order launches — as one kernel with a run-time walk its narrow shapes lost 2 us of 35
  // (tools/ab_bench.py).  Every other family that walks units (Monte-Carlo samples, the derivative outputs, per-edge
  // parameters and their gradients) exists ONLY with the walk (round 4): plan order is the one-stripe walk
  // (smask 0: sh = the heavy prefix, sl = the rest), so the XCD-aware order and the row stripes of plan-less launches
  // reach them without a second set of kernels.  (stag_agg_bwd_dp's kernel adds block partials in block order: plan order.)
  const bool plain = !a.outx[0] && !a.dp_part && pedge == 0;
  if (a.xcd && a.dp_part) { a.xcd = nullptr; a.units = a.units_plan; }
  a.n_heavy_blocks = (a.n_heavy + TPBH - 1) / TPBH;
  dim3 grid(a.n_heavy_blocks + (a.n_units - a.n_heavy + TPB - 1) / TPB, tiles);
  bool walk = false;
  AggArgs::Walk& w = a.walk;
  if (a.xcd) {                     // a.walk.sh / sl arrive holding the plan's stripe lengths
    walk = true;
    w.smask = STAG_XCD_STRIPES - 1; w.sshift = 3;
    w.jh_heavy = slotted ? (w.sh + TPBH - 1) / TPBH : 0;
    w.jh_light = slotted ? 0 : (w.sh + TPB - 1) / TPB;
    w.lbase = STAG_XCD_STRIPES * w.sh;
    w.n_total = STAG_XCD_STRIPES * (w.sh + w.sl);
    grid.x = STAG_XCD_STRIPES * (w.jh_heavy + w.jh_light + (w.sl + TPB - 1) / TPB);
  } else if (STAG_XCD_PLANLESS && !a.dp_part && !a.units && a.n_units >= STAG_XCD_STRIPES * TPB && a.n_units < (1 << 30)) {
    // a graph that runs without a plan (short rows only: a freshly batched minibatch of molecules) is striped by rows
    walk = true;
    w = AggArgs::Walk{STAG_XCD_STRIPES - 1, 3, 0, 0, 0, 0, 0, a.n_units};
    w.sl = ((a.n_units + STAG_XCD_STRIPES - 1) / STAG_XCD_STRIPES + TPB - 1) / TPB * TPB;
    grid.x = STAG_XCD_STRIPES * (w.sl / TPB);
  } else if (!plain && !a.dp_part) {
    // plan order (or row order without a plan) as the one-stripe walk
    walk = true;
    w = AggArgs::Walk{0, 0, 0, 0, a.n_heavy, a.n_heavy, a.n_units - a.n_heavy, a.n_units};
    w.jh_heavy = slotted ? a.n_heavy_blocks : 0;
    w.jh_light = slotted ? 0 : (w.sh + TPB - 1) / TPB;
    grid.x = w.jh_heavy + w.jh_light + (w.sl + TPB - 1) / TPB;
  }
  if (grid.x == 0) return;
  const dim3 block(STAG_BLOCK_THREADS);
  if constexpr (KIND >= kNormal) {
    if (a.mc && a.outx[2]) {   // four Monte-Carlo samples per gathered row (validated on the host: !pedge)
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 0, 4, true, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 0, 4, true, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
    if (a.mc && a.outx[0] && a.in_norm) {   // two, each with its own in-norm weight sums
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 0, 2, true, true, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 0, 2, true, true, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
    if (a.mc && a.outx[0]) {   // two
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 0, 2, true, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 0, 2, true, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
  }
  if constexpr (KIND == kNormal || KIND == kUniform) {
    if (a.outx[0]) {        // weight + both parameter derivatives in one pass (validated on the host: !pedge)
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 0, 3, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 0, 3, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
  }
  if constexpr (KIND == kNormal || KIND == kUniform) {
    if (a.dp_part) {        // scalar / per-channel parameters and their gradients (stag_agg_bwd_dp): no heavy blocks
      if (vec) hipLaunchKernelGGL((agg_dp_kernel<KIND, LPE, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_dp_kernel<KIND, LPE, false>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
    if (pedge == 3) {       // [E, 1] parameters and their gradients (stag_agg_bwd_edge)
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 3, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 3, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
  }
  if constexpr (KIND >= kNormal) {
    if (pedge == 1) {       // [E, 1] parameters: one pair per edge
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 1, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 1, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
    if (pedge == 2) {       // [E, D] parameters: a row per edge
      if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 2, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 2, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
      return;
    }
  }
  if (walk) {
    if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 0, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
    else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 0, 1, false, false, true>), grid, block, STAG_AGG_LDS_BYTES, s, a);
    return;
  }
  if (vec) hipLaunchKernelGGL((agg_kernel<KIND, LPE, true, 0>), grid, block, STAG_AGG_LDS_BYTES, s, a);
  else     hipLaunchKernelGGL((agg_kernel<KIND, LPE, false, 0>), grid, block, STAG_AGG_LDS_BYTES, s, a);
}

template <int KIND>
inline hipError_t agg_launch_impl(const AggArgs& a, bool vec, hipStream_t s) {
  const int nchunk = (a.D + 3) / 4;
  const int pedge = (KIND < kNormal) ? 0 : a.pmode == STAG_PARAM_PER_EDGE1 ? (a.eg0 ? 3 : 1)
                    : a.pmode == STAG_PARAM_PER_EDGE ? 2 : 0;
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
