// gat.hip — GAT edge attention with noisy logits, softmax and aggregation fused
// (stag/zoo/gat.py:114-126):
//     e[p,h]     = w[p,h] * leaky_relu(el[u_p,h] + er[v,h])
//     a[p,h]     = softmax_p(e[.,h])           (DGL edge_softmax: per dst, per head)
//     out[v,h,:] = sum_p a[p,h] * ft[u_p,h,:]
// Noise width is H (`sample_dimension`, stag/zoo/gat.py:11): one Philox block per
// (edge, 4 heads), so this path is gather-bound (an H*F*4-byte row per edge), not RNG-bound.
//
// Same decomposition as the aggregation kernel (agg_kernel.hpp): the launch walks the
// plan's UNITS (whole rows, or <= seg_len-edge segments of long rows, longest first); a
// TEAM of LPE = H*F/4 lanes owns one unit.  Per batch of LPE edges of the unit:
//   phase 1 (edge-parallel)    lane i draws the H weights of edge i, forms the H logits and
//                              parks them in LDS — no RNG work is repeated across lanes;
//   phase 2 (channel-parallel) lane c owns channels [4c, 4c+4) of head h_c: batch max, then
//                              p = exp(logit - max), running sum and weighted row sum.
// A unit ends with a softmax state (m, l, acc) per lane.  Whole rows normalise and store;
// segments publish their state (sc1 stores + ticket, as in agg_kernel) and the last arriver
// merges the states in segment order: m = max m_i, l = sum l_i e^(m_i-m), acc likewise.
#include "../../include/stag_hip.h"
#include "agg_kernel.hpp"

using namespace stag;

namespace {

#ifndef STAG_GAT_MERGE_NF
#define STAG_GAT_MERGE_NF 12
#endif
constexpr int kGatMergeNF = STAG_GAT_MERGE_NF;   // segment states of a long row fetched per round trip by the row's merge

struct GatArgs {
  const int32_t* indptr;
  const int32_t* indices;
  const int32_t* eid;
  const int32_t* nidx;
  int32_t n_rows;
  const float* el;
  const float* er;
  const float* ft;
  const float* nscale;   // [n_rows, H] in-norm factor (null: 1)
  int32_t H, F, HF;
  float neg_slope;
  int32_t kind;
  const float* p0;
  const float* p1;
  float p0s, p1s;
  int32_t pmode, relu;
  PhiloxKey key;
  uint32_t pos_lo, pos_hi;
  float* out;
  float* attn;           // [E, H] by edge id, or null (gat_attn_kernel / backward by-product)
  float* stats;          // [n_rows, 2H] softmax statistics per row: m[H] then l[H], or null
  uint32_t ft_bytes;     // extent of ft when it fits a buffer descriptor, else 0
  // plan
  const stag_unit* units;
  int32_t n_units;
  const int32_t* long_rows;
  const int32_t* long_seg_ptr;
  int32_t n_long;
  int32_t* seg_counters;
  float* ws;             // [n_seg][HF + 2*H]: acc, then m[H], l[H]
  int32_t n_seg;         // units [0, n_seg) are the segments of long rows
  int32_t ws_stride;
  uint32_t ws_bytes;
  const int32_t* block_ptr;   // [n_blocks+1] unit batches of the workgroup-cooperative kernels
  int32_t hvec;               // el / er / nscale are 16-byte aligned (rows of H % 4 == 0 heads load as dwordx4)
  int32_t lphp;               // lanes per head in the cooperative kernels: F / 4 rounded up to a power of two
  // attention dropout (stag/zoo/gat.py:122): a[e,h] -> a[e,h] keep[e,h] / keep_prob, keep from its own Philox stream
  float drop_keep;            // keep probability; 0: no dropout
  float drop_scale;           // 1 / keep probability
  PhiloxKey drop_key;
};

// bit j = head 4c + j of the edge with noise index n survives the attention dropout
__device__ __forceinline__ uint32_t drop_keep4(const GatArgs& a, const PhiloxKey& dkey, uint32_t n, uint32_t chunk) {
  const float q[4] = {a.drop_keep, a.drop_keep, a.drop_keep, a.drop_keep};
  float k[4];
  draw4<kBernoulli>(n, chunk | (a.pos_hi << 20), dkey, q, q, 0, k);
  return (k[0] != 0.f ? 1u : 0u) | (k[1] != 0.f ? 2u : 0u) | (k[2] != 0.f ? 4u : 0u) | (k[3] != 0.f ? 8u : 0u);
}

// the 4 weights of heads [4c, 4c+4) of the edge with noise index n and edge id ed
__device__ __forceinline__ void head_w4(const GatArgs& a, const PhiloxKey& key, uint32_t n, int64_t ed,
                                        uint32_t chunk, float (&w)[4]) {
  const int h0 = (int)chunk * 4;
  float pa[4], pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int h = h0 + j;
    const bool in = h < a.H;
    float q0 = a.p0s, q1 = a.p1s;
    if (a.pmode == STAG_PARAM_PER_CHANNEL) { q0 = in ? a.p0[h] : 0.f; q1 = (in && a.p1) ? a.p1[h] : 0.f; }
    else if (a.pmode == STAG_PARAM_PER_EDGE1) { q0 = a.p0[ed]; q1 = a.p1 ? a.p1[ed] : 0.f; }
    else if (a.pmode == STAG_PARAM_PER_EDGE) {
      q0 = in ? a.p0[ed * a.H + h] : 0.f;
      q1 = (in && a.p1) ? a.p1[ed * a.H + h] : 0.f;
    }
    if (a.pmode != STAG_PARAM_SCALAR && (a.relu & kFlagLogScale)) q1 = exp_scale(q1);
    pa[j] = q0; pb[j] = q1;
  }
  const uint32_t c1 = chunk | (a.pos_hi << 20);
  switch (a.kind) {
    case kNormal: draw4<kNormal>(n, c1, key, pa, pb, a.relu, w); break;
    case kUniform: draw4<kUniform>(n, c1, key, pa, pb, a.relu, w); break;
    case kBernoulli: draw4<kBernoulli>(n, c1, key, pa, pb, a.relu, w); break;
    case kExplicit:
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = (h0 + j < a.H) ? a.p0[ed * a.H + h0 + j] : 0.f;
        w[j] = (a.relu & kFlagRelu) ? fmaxf(t, 0.f) : t;
      }
      break;
    default: w[0] = w[1] = w[2] = w[3] = 1.0f;
  }
}

// Lane L of a team -> its 4 channels.  A head takes lphp lanes (F / 4 rounded up to a power of two: the head sums
// are DPP butterflies); lanes past a head's F channels idle, so a width like F = 40 (lphp = 16) costs idle lanes but
// no padded bytes.  F / 4 a power of two: k0 = 4 L, the plain contiguous mapping.
__device__ __forceinline__ void lane_chunk(int L, int H, int F, int lphp, int& k0, bool& kin, int& hl) {
  const int head = L / lphp, j = L - head * lphp;
  kin = head < H && 4 * j < F;
  k0 = kin ? head * F + 4 * j : 0;
  hl = kin ? head : 0;
}

// The row output is written once and not read again by this launch: a non-temporal store keeps it from
// displacing ft in the 256 MB Infinity Cache (cfg5: ft 173 MB + out 173 MB do not fit together; measured
// 298 -> 289 us on the one-unit-per-team kernel).
__device__ __forceinline__ void store4_out(float* p, int k0, int D, bool vec, const float (&v)[4]) {
  if (vec) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p + k0));
    return;
  }
  store4(p, k0, D, vec, v);
}

// LDS hand-off between lanes of ONE wave: the LDS unit serves a wave's requests in
// order, so only the compiler has to be kept from reordering around the hand-off.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// LPE lanes per unit (H*F/4 rounded up to a power of two, 4..64); VEC: dwordx4 rows
template <int LPE, bool VEC>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const GatArgs a) {
  extern __shared__ __align__(16) float lds[];
  const PhiloxKey key = resolve_epoch(a.key);
  constexpr int TEAMS_PER_BLOCK = 256 / LPE;
  const int H = a.H, F = a.F, HF = a.HF;
  const int team = threadIdx.x / LPE;
  const int c = threadIdx.x % LPE;
  const int team_lane0 = (int)(threadIdx.x & 63) - c;
  float* logit = lds + (size_t)team * (LPE * H);      // [LPE edges][H]
  const int unit = blockIdx.x * TEAMS_PER_BLOCK + team;
  if (unit >= a.n_units) return;   // teams are independent: no block-level barrier below

  int v, b, len, slot = -1;
  if (a.units) {
    const int4 q = *reinterpret_cast<const int4*>(a.units + unit);
    v = q.x; b = q.y; len = q.z; slot = q.w;
  } else {
    v = unit;
    b = a.indptr[v];
    len = a.indptr[v + 1] - b;
  }
  const int row = (slot >= 0) ? a.long_rows[v] : v;
  if (len > 24) __builtin_amdgcn_s_setprio(2);

  const int k0 = c * 4;                    // channels [k0, k0+4) of the H*F row
  const bool kin = k0 < HF;
  // head of each of the lane's 4 channels: all equal when F % 4 == 0 (the VEC path); with
  // an odd F a lane's channels can straddle two heads, so the softmax state is per channel
  int hq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) hq[q] = (k0 + q < HF) ? (k0 + q) / F : 0;
  const int hl = hq[0];
  const int nchunk = (H + 3) / 4;
  const __amdgpu_buffer_rsrc_t rft =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ft), 0, (int)a.ft_bytes, 0x00020000);
  const bool ft_buf = VEC && a.ft_bytes != 0;

  constexpr int NS = VEC ? 1 : 4;          // softmax states per lane
  float m[NS], l[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) { m[q] = -INFINITY; l[q] = 0.f; }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};

  for (int i0 = 0; i0 < len; i0 += LPE) {
    const int nb = min(LPE, len - i0);
    // ---- phase 1: lane c computes the H logits of edge i0 + c ------------------------------
    wave_sync();
    int u = 0;
    if (c < nb) {
      const int p = b + i0 + c;
      u = a.indices[p];
      const int64_t ed = a.eid ? a.eid[p] : p;
      const uint32_t n = a.pos_lo + (a.nidx ? (uint32_t)a.nidx[p] : (uint32_t)p);
      for (int cc = 0; cc < nchunk; ++cc) {
        float w[4];
        head_w4(a, key, n, ed, (uint32_t)cc, w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int h = 4 * cc + j;
          if (h < H) {
            const float s = a.el[(int64_t)u * H + h] + a.er[(int64_t)row * H + h];
            const float lr = s > 0.f ? s : a.neg_slope * s;
            const float ns = a.nscale ? a.nscale[(int64_t)row * H + h] : 1.0f;
            const float lg = (w[j] * ns) * lr;
            logit[c * H + h] = lg;
          }
        }
      }
    }
    wave_sync();
    // ---- phase 2: fold the batch into this lane's head --------------------------------------
    if (kin) {
#pragma unroll
      for (int sI = 0; sI < NS; ++sI) {
        float bm = -INFINITY;
        for (int i = 0; i < nb; ++i) bm = fmaxf(bm, logit[i * H + hq[sI]]);
        const float mn = fmaxf(m[sI], bm);
        const float corr = __expf(m[sI] - mn);     // 0 on the first batch (m = -inf)
        l[sI] *= corr;
        if constexpr (VEC) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] *= corr;
        } else {
          acc[sI] *= corr;
        }
        m[sI] = mn;
      }
    }
    for (int i = 0; i < nb; i += 4) {
      int ui[4];
      float fv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ui[j] = __builtin_amdgcn_ds_bpermute((team_lane0 + i + j) << 2, u);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i + j < nb && kin) {
          if (ft_buf) bufrow4(rft, ui[j], (uint32_t)HF * 4u, (uint32_t)k0 * 4u, fv[j]);
          else loadrow4(a.ft + (int64_t)ui[j] * HF + k0, k0, HF, VEC, fv[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i + j < nb && kin) {
          if constexpr (VEC) {
            const float pe = __expf(logit[(i + j) * H + hl] - m[0]);
            l[0] += pe;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_fmaf(pe, fv[j][q], acc[q]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float pe = __expf(logit[(i + j) * H + hq[q]] - m[q]);
              l[q] += pe;
              acc[q] = __builtin_fmaf(pe, fv[j][q], acc[q]);
            }
          }
        }
      }
    }
  }

  if (slot < 0) {
    // ---- whole row: normalise and store ------------------------------------------------------
    if (kin) {
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float lq = l[VEC ? 0 : q];
        o[q] = acc[q] * ((lq > 0.f) ? 1.0f / lq : 0.f);
      }
      store4_out(a.out + (int64_t)row * HF, k0, HF, VEC, o);
    }
    // the lane holding a head's first channel publishes that head's statistics: the attention
    // values and the backward pass are computed from them (no [E, H] tensor leaves this kernel)
    if (a.stats) {
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (k0 + q < HF && (k0 + q) % F == 0) {
          a.stats[(int64_t)row * 2 * H + hq[q]] = m[q];
          a.stats[(int64_t)row * 2 * H + H + hq[q]] = l[q];
        }
    }
  } else {
    // ---- segment: publish (acc, m, l) write-through, take a ticket ---------------------------
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)a.ws_bytes, 0x00020000);
    const uint32_t base = (uint32_t)slot * ((uint32_t)a.ws_stride * 4u);
    if (kin) {
      store4_sc1(rws, base + (uint32_t)k0 * 4u, k0, HF, VEC, acc);
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (k0 + q < HF && (k0 + q) % F == 0) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[q]), rws, (int)(base + (uint32_t)(HF + hq[q]) * 4u), 0, 16);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l[q]), rws, (int)(base + (uint32_t)(HF + H + hq[q]) * 4u), 0, 16);
        }
    }
    const int s0 = a.long_seg_ptr[v], s1 = a.long_seg_ptr[v + 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (c == 0)
      ticket = __hip_atomic_fetch_add(a.seg_counters + v, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_ds_bpermute(team_lane0 << 2, ticket);
    if (ticket != (s1 - s0) - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (c == 0) a.seg_counters[v] = 0;
    // merge the segment states in segment order
    float M[NS], L[NS];
    float A[4] = {0.f, 0.f, 0.f, 0.f};
    if (kin) {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        M[q] = -INFINITY; L[q] = 0.f;
        for (int s = s0; s < s1; ++s) M[q] = fmaxf(M[q], a.ws[(int64_t)s * a.ws_stride + HF + hq[q]]);
      }
      for (int s = s0; s < s1; ++s) {
        const float* wr = a.ws + (int64_t)s * a.ws_stride;
        float t[4];
        load4(wr, k0, HF, VEC, t);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const float sc = __expf(wr[HF + hq[q]] - M[q]);
          L[q] += wr[HF + H + hq[q]] * sc;
          if constexpr (VEC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) A[r] = __builtin_fmaf(t[r], sc, A[r]);
          } else {
            A[q] = __builtin_fmaf(t[q], sc, A[q]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float lq = L[VEC ? 0 : q];
        A[q] *= (lq > 0.f) ? 1.0f / lq : 0.f;
      }
      store4_out(a.out + (int64_t)row * HF, k0, HF, VEC, A);
    }
    if (a.stats && kin) {
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (k0 + q < HF && (k0 + q) % F == 0) {
          a.stats[(int64_t)row * 2 * H + hq[q]] = M[q];
          a.stats[(int64_t)row * 2 * H + H + hq[q]] = L[q];
        }
    }
  }
}

// ---- workgroup-cooperative forward (the default with a block plan, F % 4 == 0, H <= 16) --------
// gat_fwd_kernel gives a whole wave to one destination row: on a graph of mean in-degree 7 its
// edge-parallel phase runs 7 of 64 lanes through a chain of dependent round trips (unit -> ids -> el ->
// draw) while nothing is being gathered: 283 us at cfg5 for 1.5 GB, 5.4 TB/s, 5 waves per SIMD, each
// alive ~14 us for one row.  Here a WORKGROUP takes a batch of consecutive units of the plan (stag_plan
// block_ptr: <= 256 edges, <= 32 units):
//   phase 1  thread t <-> edge t of the batch: draws its H weights, forms its H noisy logits -> LDS
//            (every lane busy, one chain of round trips per batch instead of one per row);
//   phase 1b thread <-> (unit, head): max and sum of exp over the unit's logits in LDS; the logits are
//            replaced by p = exp(e - m) in place;
//   phase 2  a team of LPE lanes per unit (the block's units dealt round robin): a plain weighted gather
//            — ids and weights come from LDS, NR rows in flight — then normalise / publish.
// A unit's arithmetic does not depend on which batch it rides in: m = max over its edges, l = sum of
// exp in edge order, acc = fma(p_i, row_i, acc) in edge order — so shards reproduce the whole graph.
constexpr int kBlkEdges = STAG_BLOCK_EDGES, kBlkUnits = STAG_BLOCK_UNITS, kBlkMaxH = 16;
constexpr int kBlkThreads = STAG_BLOCK_EDGES;     // a thread per edge of the batch
static_assert(kBlkUnits <= 64 && kBlkThreads % 64 == 0 && kBlkThreads <= 1024, "the unit prefix is one wave's scan");
#ifndef STAG_GAT_DBG
#define STAG_GAT_DBG 0    // 16: per-stage timestamps into the stats buffer (tools/gat_trace.py)
#endif
#ifndef STAG_GAT_NR
#define STAG_GAT_NR 4     // rows in flight per team in phase 2
#endif
#ifndef STAG_GAT_BRANCHFREE
#define STAG_GAT_BRANCHFREE 1
#endif
// Rows in flight per team in the FORWARD's gather.  Round 3: with the loads of a round really issued together
// (STAG_GAT_BRANCHFREE) more of them is monotonically WORSE on cfg5 — 1: 222.2, 2: 238.1, 3: 249.6, 4: 252.8 us; the
// branchy form that serialised its "4 in flight" by accident: 228.5 us — and so is any other number of workgroups per CU
// than 4 (3: 246.9, 5: 239.0, 6: 253.9, 8: 255.8 us): a gather of 1-KB rows out of a 173 MB table wants ~16 rows in
// flight per CU and no more.  (128-thread workgroups over 128-edge batches, 4 / 6 / 8 / 12 per CU: 314.6 / 239.6 / 220.8 / 249.9 us —
// the same 16 waves per CU, the same time.)
#ifndef STAG_GAT_NR_FWD
#define STAG_GAT_NR_FWD 1
#endif
#ifndef STAG_GAT_NR_FWD_LOCAL
#define STAG_GAT_NR_FWD_LOCAL 2
#endif
// Round 4, re-measured on the kernel as it stands (three builds beside the shipped one in one call): on the one-GPU cfg5
// launch the count no longer matters — 1: 221.0-221.7, 2: 219.8-221.3, 3: 219.7, 4: 220.3 us — and on a SHARD-sized launch
// (an eighth of the graph: 900 batches, not fabric-bound: its time is the longest unit's chain of row round trips) it
// decides: 52.1 | 42.1 | 38.8 | 37.9 us for 1 | 2 | 3 | 4.  Rows of one chunk per lane (H * F <= 256: every BASELINE
// config) therefore keep 4 in flight; wider rows (2 or 4 chunks per lane: 4 rows would be 32-64 more registers) stay
// as they were.
#ifndef STAG_GAT_NR_FWD_NARROW
#define STAG_GAT_NR_FWD_NARROW 4
#endif
// (the one-gather backward keeps its branchy NR = 4 loop: branch-free loads there change nothing — cfg5 training step
//  632.5 against 632.1 us — and fewer rows per round cost it: NR = 2 640.5, NR = 1 658 us)

// (80 SGPRs: a CU admits 8 workgroups of 256 threads only up to that count — MI355X_MICROARCH.md,
//  "Residency"; the argument block alone would take ~100, the rest spill to lanes of a VGPR.)
#ifndef STAG_GAT_SGPR
#define STAG_GAT_SGPR 0
#endif
#if STAG_GAT_SGPR
#define STAG_GAT_SGPR_ATTR __attribute__((amdgpu_num_sgpr(STAG_GAT_SGPR)))
#else
#define STAG_GAT_SGPR_ATTR
#endif
#ifndef STAG_GAT_LDS_MIN
// Bytes of LDS a workgroup asks for at least: caps the workgroups a CU admits (160 KB / bytes).  The
// gather of 1-KB rows from a table beyond L2 is fastest with FEWER workgroups resident than registers
// and the kernel's own LDS would allow (cfg5, us per forward: 6 per CU 250, 5: 235, 4: 224.5, 3: 252,
// 2: 341; forward + backward with the two backward kernels at 4: 759, 5: 722, 6: 733).
#define STAG_GAT_LDS_MIN 40000      // forward: 4 workgroups per CU
#endif
#ifndef STAG_GAT_LDS_MIN_ONE
// the one-gather backward: no cap beyond its registers' (80 VGPRs: 6 waves per SIMD).  Round 4, 32000 | 16000 B: the
// one-GPU training step 621-628 | 618-624 us (no difference), the remote-rows source pass of a shard of eight 62.9 | 53.2 us
// (85 k rows of 1.5 edges: latency x concurrency, not the fabric); 48000: 689 us and 74.9 us.
#define STAG_GAT_LDS_MIN_ONE 16000
#endif
#ifndef STAG_GAT_LDS_MIN_BWD
#define STAG_GAT_LDS_MIN_BWD 32000  // backward passes: 5
#endif
// NRF: rows in flight per team in the gather — STAG_GAT_NR_FWD (1) when the rows come out of the Infinity Cache (cfg5: a
// 173 MB table, uniformly random sources), STAG_GAT_NR_FWD_LOCAL (2) when the batches are XCD-local (stag_plan_blocks_xcd*:
// the caller says so by a non-NULL plan->xcd_order) and the rows come out of an XCD's L2, where the gather is bound by
// latency, not by the fabric (round 4; tools/bench_configs.py --lib, us, 1 | 2 | 4 rows: PPI batch 4 x 256 356.6 | 334.2 |
// 337.3; cfg5 221.8 | 238.0 | 252.0).
template <int LPE, int CPL, int NRF = STAG_GAT_NR_FWD>
__global__ __launch_bounds__(kBlkThreads) STAG_GAT_SGPR_ATTR void gat_fwd_block_kernel(const GatArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int H = a.H, F = a.F, HF = a.HF;
  float* s_w = lds;                                   // [kBlkEdges][H] logits, then p = exp(e - m)
  float* s_m = s_w + kBlkEdges * H;                   // [kBlkUnits][H]
  float* s_l = s_m + kBlkUnits * H;                   // [kBlkUnits][H]
  int* s_u = reinterpret_cast<int*>(s_l + kBlkUnits * H);   // [kBlkEdges] source row of each edge
  int* s_start = s_u + kBlkEdges;                     // [kBlkUnits + 1] first edge slot of each unit
  int4* s_unit = reinterpret_cast<int4*>(s_start + kBlkUnits + 4);   // [kBlkUnits] (row, start, len, slot); 16-B aligned
  const int t = threadIdx.x;
#if STAG_GAT_DBG & 16
  uint64_t* trace = reinterpret_cast<uint64_t*>(a.stats) + (int64_t)blockIdx.x * 8;
  if (t == 0) trace[0] = wall_clock64();
#define GAT_TS(i) if (t == 0) trace[i] = wall_clock64();
#else
#define GAT_TS(i)
#endif
  const int ub = a.block_ptr[blockIdx.x], nu = a.block_ptr[blockIdx.x + 1] - ub;

  // ---- the batch: unit records, edge-slot prefix -----------------------------------------------
  if (t < kBlkUnits) {
    int4 q = make_int4(0, 0, 0, -1);
    if (t < nu) q = *reinterpret_cast<const int4*>(a.units + ub + t);
    s_unit[t] = q;
    int incl = q.z;                                    // inclusive scan of the lengths over lanes 0..31
#pragma unroll
    for (int d = 1; d < kBlkUnits; d <<= 1) {
      const int up = __shfl_up(incl, d, kBlkUnits);          // lanes 0..kBlkUnits-1 of wave 0
      if (t >= d) incl += up;
    }
    s_start[t + 1] = incl;
    if (t == 0) s_start[0] = 0;
  }
  __syncthreads();
  GAT_TS(1)
  const int ne = s_start[nu];                          // edges of the batch (<= kBlkEdges by the plan)

  // ---- phase 1: thread t <-> edge slot t ---------------------------------------------------------
  uint32_t kmask = 0;                                  // attention dropout: bit h = head h of my edge survives
  if (t < ne) {
    int lo = 0, hi = nu;                               // unit j with s_start[j] <= t < s_start[j+1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_start[mid] <= t) lo = mid; else hi = mid;
    }
    const int4 q = s_unit[lo];
    const int row = (q.w >= 0) ? a.long_rows[q.x] : q.x;
    const int p = q.y + (t - s_start[lo]);
    const int u = a.indices[p];
    s_u[t] = u;
    const int64_t ed = a.eid ? a.eid[p] : p;
    const uint32_t n = a.pos_lo + (a.nidx ? (uint32_t)a.nidx[p] : (uint32_t)p);
    const PhiloxKey key = resolve_epoch(a.key);
    const int nchunk = (H + 3) / 4;
    const bool h4 = (H & 3) == 0 && a.hvec;           // el / er / nscale rows as dwordx4 (one L1 lookup per 4 heads)
    for (int cc = 0; cc < nchunk; ++cc) {
      float w[4], sl4[4], sr4[4], ns4[4] = {1.f, 1.f, 1.f, 1.f};
      if (h4) {
        load4(a.el + (int64_t)u * H, 4 * cc, H, true, sl4);
        load4(a.er + (int64_t)row * H, 4 * cc, H, true, sr4);
        if (a.nscale) load4(a.nscale + (int64_t)row * H, 4 * cc, H, true, ns4);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int h = 4 * cc + j;
          sl4[j] = h < H ? a.el[(int64_t)u * H + h] : 0.f;
          sr4[j] = h < H ? a.er[(int64_t)row * H + h] : 0.f;
          if (a.nscale && h < H) ns4[j] = a.nscale[(int64_t)row * H + h];
        }
      }
      head_w4(a, key, n, ed, (uint32_t)cc, w);
      if (a.drop_keep > 0.f) kmask |= drop_keep4(a, resolve_epoch(a.drop_key), n, (uint32_t)cc) << (4 * cc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int h = 4 * cc + j;
        if (h < H) {
          const float sL = sl4[j] + sr4[j];
          const float lr = sL > 0.f ? sL : a.neg_slope * sL;
          s_w[t * H + h] = (w[j] * ns4[j]) * lr;
        }
      }
    }
  }
  GAT_TS(2)
  __syncthreads();
  GAT_TS(3)

  // ---- phase 1b: thread <-> (unit, head): softmax statistics, logits -> p ---------------------------
  for (int i = t; i < nu * H; i += kBlkThreads) {
    const int j = i / H, h = i - j * H;
    const int e0 = s_start[j], e1 = s_start[j + 1];
    float m = -INFINITY;
    for (int e = e0; e < e1; ++e) m = fmaxf(m, s_w[e * H + h]);
    float l = 0.f;
    for (int e = e0; e < e1; ++e) {
      const float pe = __expf(s_w[e * H + h] - m);
      s_w[e * H + h] = pe;
      l += pe;
    }
    s_m[i] = m;
    s_l[i] = l;
  }
  GAT_TS(4)
  __syncthreads();
  GAT_TS(5)
  if (a.drop_keep > 0.f) {       // (uniform) the softmax statistics saw every edge; the weighted sum sees the survivors
    if (t < ne) {
      for (int h = 0; h < H; ++h)
        s_w[t * H + h] = ((kmask >> h) & 1u) ? s_w[t * H + h] * a.drop_scale : 0.f;
    }
    __syncthreads();
  }

  // ---- phase 2: a team per unit, weighted gather -------------------------------------------------------
  // lane c owns CPL chunks of 4 channels: [4 (c + LPE j), +4), j < CPL  (H*F <= 256: one; up to 1024: 2 or 4)
  constexpr int TEAMS = kBlkThreads / LPE, NR = STAG_GAT_BRANCHFREE ? NRF : (CPL >= 4 ? 2 : STAG_GAT_NR);
  const int team = t / LPE, c = t % LPE;
  const int team_lane0 = (int)(t & 63) - c;
  int k0[CPL], hl[CPL];
  bool kin[CPL];
#pragma unroll
  for (int cj = 0; cj < CPL; ++cj) {
    lane_chunk(c + LPE * cj, H, F, a.lphp, k0[cj], kin[cj], hl[cj]);
  }
  const __amdgpu_buffer_rsrc_t rft =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ft), 0, (int)a.ft_bytes, 0x00020000);
  const bool ft_buf = a.ft_bytes != 0;
  for (int j = team; j < nu; j += TEAMS) {
    const int4 q = s_unit[j];
    const int e0 = s_start[j], e1 = s_start[j + 1];
    float acc[CPL][4];
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) acc[cj][0] = acc[cj][1] = acc[cj][2] = acc[cj][3] = 0.f;
#if STAG_GAT_BRANCHFREE
    // The NR gathers of a round are issued back to back with NO control flow between them: a round's source ids and
    // weights come out of LDS first (positions past the unit's end are clamped to its last edge and carry weight 0:
    // a repeated row, which the L1 serves), lanes without a chunk load chunk 0 and store nothing.  With the
    // `if (e + r < e1)` / `if (kin)` branches around each load the compiler could not count the loads in flight and put
    // `s_waitcnt vmcnt(0)` in front of every next LDS read: the "NR rows in flight" went out one at a time.
    for (int e = e0; e < e1; e += NR) {
      float fv[NR][CPL][4], pe[NR][CPL];
      int u[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int er = min(e + r, e1 - 1);
        u[r] = s_u[er];
#pragma unroll
        for (int cj = 0; cj < CPL; ++cj) {
          const float wv = s_w[er * H + (kin[cj] ? hl[cj] : 0)];
          pe[r][cj] = (e + r < e1) ? wv : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int cj = 0; cj < CPL; ++cj) {
          if (ft_buf) bufrow4(rft, u[r], (uint32_t)HF * 4u, (uint32_t)k0[cj] * 4u, fv[r][cj]);
          else loadrow4(a.ft + (int64_t)u[r] * HF + k0[cj], k0[cj], HF, true, fv[r][cj]);
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int cj = 0; cj < CPL; ++cj) {
#pragma unroll
          for (int x = 0; x < 4; ++x) acc[cj][x] = __builtin_fmaf(pe[r][cj], fv[r][cj][x], acc[cj][x]);
        }
      }
    }
#else
    for (int e = e0; e < e1; e += NR) {
      float fv[NR][CPL][4];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {
          const int u = s_u[e + r];
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            if (kin[cj]) {
              if (ft_buf) bufrow4(rft, u, (uint32_t)HF * 4u, (uint32_t)k0[cj] * 4u, fv[r][cj]);
              else loadrow4(a.ft + (int64_t)u * HF + k0[cj], k0[cj], HF, true, fv[r][cj]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            if (kin[cj]) {
              const float pe = s_w[(e + r) * H + hl[cj]];
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[cj][x] = __builtin_fmaf(pe, fv[r][cj][x], acc[cj][x]);
            }
          }
        }
      }
    }
#endif
    if (q.w < 0) {
      // ---- whole row: normalise and store --------------------------------------------------------
#pragma unroll
      for (int cj = 0; cj < CPL; ++cj) {
        if (!kin[cj]) continue;
        const float m = s_m[j * H + hl[cj]], l = s_l[j * H + hl[cj]];
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        float o[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) o[x] = acc[cj][x] * inv;
        store4_out(a.out + (int64_t)q.x * HF, k0[cj], HF, true, o);
        if (a.stats && k0[cj] % F == 0 && !(STAG_GAT_DBG & 16)) {
          a.stats[(int64_t)q.x * 2 * H + hl[cj]] = m;
          a.stats[(int64_t)q.x * 2 * H + H + hl[cj]] = l;
        }
      }
      continue;
    }
    // ---- segment: publish (acc, m, l) write-through, take a ticket; the last arriver merges ----------
    const int v = q.x, slot = q.w, row = a.long_rows[v];
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)a.ws_bytes, 0x00020000);
    const uint32_t base = (uint32_t)slot * ((uint32_t)a.ws_stride * 4u);
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) {
      if (!kin[cj]) continue;
      store4_sc1(rws, base + (uint32_t)k0[cj] * 4u, k0[cj], HF, true, acc[cj]);
      if (k0[cj] % F == 0) {
        const float m = s_m[j * H + hl[cj]], l = s_l[j * H + hl[cj]];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), rws, (int)(base + (uint32_t)(HF + hl[cj]) * 4u), 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l), rws, (int)(base + (uint32_t)(HF + H + hl[cj]) * 4u), 0, 16);
      }
    }
    const int s0 = a.long_seg_ptr[v], s1 = a.long_seg_ptr[v + 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (c == 0)
      ticket = __hip_atomic_fetch_add(a.seg_counters + v, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_ds_bpermute(team_lane0 << 2, ticket);
    if (ticket != (s1 - s0) - 1) continue;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (c == 0) a.seg_counters[v] = 0;
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) {
      if (!kin[cj]) continue;
      float M = -INFINITY, L = 0.f;
      float A[4] = {0.f, 0.f, 0.f, 0.f};
      // The merge is a chain on the launch's critical path when the launch is small (a shard of an 8-way partition: the
      // 13k-edge hub's 205 segment states, one L2 round trip each, were 97 of the shard's 145 us).  The states are fetched
      // kGatMergeNF (12) at a time — independent loads, clamped to the last segment instead of branching — and folded in
      // segment order as before, so the row's bits do not change.  12 states are 121 VGPRs: free, because the kernel's LDS
      // request (STAG_GAT_LDS_MIN) caps a SIMD at 4 waves, which registers allow up to 128 (one-GPU cfg5 with 4 | 8 | 12
      // at a time: 223.8 | 223.5 | 223.1 us; the hub's shard of eight: 70.2 | 57.4 | 53.0 us).
      constexpr int NFM = CPL == 1 ? kGatMergeNF : 1;     // (wider rows, CPL chunks per lane: their register budget has no room, one state at a time as before)
      // the row's maximum per head.  Where the team's lanes are exactly H heads x F/4 lanes (cfg5: 8 x 8 = 64), the F/4
      // lanes of a head take every (F/4)-th segment each and exchange their maxima: 205 states in 4 round trips
      // instead of 26 (a maximum does not care about the order)
      const int lph = a.lphp;
      if (CPL == 1 && lph >= 2 && lph * 4 == F && H * lph == LPE) {
        const int g = c & (lph - 1);
        for (int sg = s0 + g; sg < s1; sg += 2 * NFM * lph) {
          float mm[2 * NFM];
#pragma unroll
          for (int i = 0; i < 2 * NFM; ++i)
            mm[i] = a.ws[(int64_t)min(sg + i * lph, s1 - 1) * a.ws_stride + HF + hl[cj]];
#pragma unroll
          for (int i = 0; i < 2 * NFM; ++i) M = fmaxf(M, mm[i]);
        }
        for (int d = 1; d < lph; d <<= 1) M = fmaxf(M, __shfl_xor(M, d));
      } else {
        for (int sg = s0; sg < s1; sg += 2 * NFM) {
          float mm[2 * NFM];
#pragma unroll
          for (int i = 0; i < 2 * NFM; ++i)
            mm[i] = a.ws[(int64_t)min(sg + i, s1 - 1) * a.ws_stride + HF + hl[cj]];
#pragma unroll
          for (int i = 0; i < 2 * NFM; ++i) M = fmaxf(M, mm[i]);
        }
      }
      for (int sg = s0; sg < s1; sg += NFM) {
        float tt[NFM][4], ms[NFM], ls[NFM];
#pragma unroll
        for (int i = 0; i < NFM; ++i) {
          const float* wr = a.ws + (int64_t)min(sg + i, s1 - 1) * a.ws_stride;
          load4(wr, k0[cj], HF, true, tt[i]);
          ms[i] = wr[HF + hl[cj]];
          ls[i] = wr[HF + H + hl[cj]];
        }
#pragma unroll
        for (int i = 0; i < NFM; ++i) {
          if (sg + i < s1) {
            const float sc = __expf(ms[i] - M);
            L += ls[i] * sc;
#pragma unroll
            for (int x = 0; x < 4; ++x) A[x] = __builtin_fmaf(tt[i][x], sc, A[x]);
          }
        }
      }
      const float inv = (L > 0.f) ? 1.0f / L : 0.f;
#pragma unroll
      for (int x = 0; x < 4; ++x) A[x] *= inv;
      store4_out(a.out + (int64_t)row * HF, k0[cj], HF, true, A);
      if (a.stats && k0[cj] % F == 0 && !(STAG_GAT_DBG & 16)) {
        a.stats[(int64_t)row * 2 * H + hl[cj]] = M;
        a.stats[(int64_t)row * 2 * H + H + hl[cj]] = L;
      }
    }
  }
  GAT_TS(6)
}

// Attention values a[eid, h] = exp(logit - m[v,h]) / l[v,h] (get_attention=True,
// stag/zoo/gat.py:146-147) from the row statistics of gat_fwd_kernel: 8 lanes per unit of the
// plan, a lane per edge; the noisy logit is redrawn from its counters.
__global__ __launch_bounds__(256) void gat_attn_kernel(const GatArgs a) {
  const PhiloxKey key = resolve_epoch(a.key);
  const int c = threadIdx.x & 7;
  const int unit = blockIdx.x * 32 + (threadIdx.x >> 3);
  if (unit >= a.n_units) return;
  int v, b, len, slot = -1;
  if (a.units) {
    const int4 q = *reinterpret_cast<const int4*>(a.units + unit);
    v = q.x; b = q.y; len = q.z; slot = q.w;
  } else {
    v = unit; b = a.indptr[v]; len = a.indptr[v + 1] - b;
  }
  const int row = (slot >= 0) ? a.long_rows[v] : v;
  const int H = a.H, nchunk = (H + 3) / 4;
  const float* st = a.stats + (int64_t)row * 2 * H;
  for (int p = b + c; p < b + len; p += 8) {
    const int u = a.indices[p];
    const int64_t ed = a.eid ? a.eid[p] : p;
    const uint32_t n = a.pos_lo + (a.nidx ? (uint32_t)a.nidx[p] : (uint32_t)p);
    for (int cc = 0; cc < nchunk; ++cc) {
      float w[4], at[4];
      head_w4(a, key, n, ed, (uint32_t)cc, w);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int h = 4 * cc + j;
        at[j] = 0.f;
        if (h < H) {
          const float e = a.el[(int64_t)u * H + h] + a.er[(int64_t)row * H + h];
          const float lr = e > 0.f ? e : a.neg_slope * e;
          const float ns = a.nscale ? a.nscale[(int64_t)row * H + h] : 1.0f;
          at[j] = __expf((w[j] * ns) * lr - st[h]) / st[H + h];
        }
      }
      if ((H & 3) == 0) {       // one 16-byte store per head chunk
        *reinterpret_cast<float4*>(a.attn + ed * H + 4 * cc) = make_float4(at[0], at[1], at[2], at[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * cc + j < H) a.attn[ed * H + 4 * cc + j] = at[j];
      }
    }
  }
}

// ---- backward, per-edge part ---------------------------------------------------------------
// Given G = dL/dout[v,h,:] and gdo[v,h] = <G[v,h,:], out[v,h,:]>:
//     da[p,h] = <G[v,h,:], ft[u_p,h,:]>
//     ds[p,h] = a[p,h] * (da[p,h] - gdo[v,h])                      softmax backward
//     de[p,h] = ds[p,h] * w[p,h] * lrelu'(el[u_p,h] + er[v,h])     -> d el / d er by segment sums
//     dw[p,h] = ds[p,h] * lrelu(el[u_p,h] + er[v,h]) * nscale      (explicit / materialised w)
// Edges are independent, so segments need no merge.  Phase 1 (edge-parallel) regenerates the
// weights and parks a, w*lrelu', lrelu in LDS; phase 2 (channel-parallel) gathers ft rows,
// takes the dot with G, reduces it over the F/4 lanes of a head and lets the head's first
// lane write.  Requires F % 4 == 0 and F/4 a power of two.
struct GatBwdArgs {
  GatArgs f;
  const float* g;      // [n_rows, H*F]
  const float* gdo;    // [n_rows, H] = <g[v,h,:], out[v,h,:]>, or null: computed here from `out`
  const float* out;    // [n_rows, H*F] forward output (used when gdo is null)
  float* de;           // [E, H] by edge id
  float* dw;           // [E, H] by edge id, or null
};

template <int LPE>
__global__ __launch_bounds__(256) void gat_bwd_edge_kernel(const GatBwdArgs ba) {
  extern __shared__ __align__(16) float lds[];
  const GatArgs& a = ba.f;
  const PhiloxKey key = resolve_epoch(a.key);
  constexpr int TEAMS_PER_BLOCK = 256 / LPE;
  const int H = a.H, F = a.F, HF = a.HF;
  const int team = threadIdx.x / LPE;
  const int c = threadIdx.x % LPE;
  const int team_lane0 = (int)(threadIdx.x & 63) - c;
  float* sa = lds + (size_t)team * (3 * LPE * H);   // [LPE][H] attention
  float* sc1 = sa + LPE * H;                         // [LPE][H] w * nscale * lrelu'
  float* sc2 = sc1 + LPE * H;                        // [LPE][H] lrelu * nscale
  const int unit = blockIdx.x * TEAMS_PER_BLOCK + team;
  if (unit >= a.n_units) return;

  int v, b, len, slot = -1;
  if (a.units) {
    const int4 q = *reinterpret_cast<const int4*>(a.units + unit);
    v = q.x; b = q.y; len = q.z; slot = q.w;
  } else {
    v = unit;
    b = a.indptr[v];
    len = a.indptr[v + 1] - b;
  }
  const int row = (slot >= 0) ? a.long_rows[v] : v;
  const int k0 = c * 4;
  const bool kin = k0 < HF;
  const int hl = kin ? k0 / F : 0;
  const int nchunk = (H + 3) / 4;
  const int lanes_per_head = F / 4;

  // sum over the F/4 lanes of a head: DPP inside a row of 16 lanes, bpermute only beyond
  auto head_sum = [&](float x) {
    if (lanes_per_head >= 2) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
    if (lanes_per_head >= 4) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
    if (lanes_per_head >= 8) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));
    if (lanes_per_head >= 16) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));
    if (lanes_per_head >= 32) x += __shfl_xor(x, 16);
    if (lanes_per_head >= 64) x += __shfl_xor(x, 32);
    return x;
  };
  float gv[4] = {0.f, 0.f, 0.f, 0.f};
  float gdo = 0.f;
  if (kin) load4(ba.g + (int64_t)row * HF, k0, HF, true, gv);
  if (ba.gdo) {
    if (kin) gdo = ba.gdo[(int64_t)row * H + hl];
  } else {                      // <g[v,h,:], out[v,h,:]> of the unit's own row
    float ov[4] = {0.f, 0.f, 0.f, 0.f};
    if (kin) load4(ba.out + (int64_t)row * HF, k0, HF, true, ov);
    gdo = head_sum((gv[0] * ov[0] + gv[1] * ov[1]) + (gv[2] * ov[2] + gv[3] * ov[3]));
  }

  for (int i0 = 0; i0 < len; i0 += LPE) {
    const int nb = min(LPE, len - i0);
    wave_sync();
    int u = 0, edl = 0;
    if (c < nb) {
      const int p = b + i0 + c;
      u = a.indices[p];
      const int64_t ed = a.eid ? a.eid[p] : p;
      edl = (int)ed;
      const uint32_t n = a.pos_lo + (a.nidx ? (uint32_t)a.nidx[p] : (uint32_t)p);
      for (int cc = 0; cc < nchunk; ++cc) {
        float w[4], at4[4] = {0.f, 0.f, 0.f, 0.f};
        head_w4(a, key, n, ed, (uint32_t)cc, w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int h = 4 * cc + j;
          if (h < H) {
            const float e = a.el[(int64_t)u * H + h] + a.er[(int64_t)row * H + h];
            const float ns = a.nscale ? a.nscale[(int64_t)row * H + h] : 1.0f;
            const float lr = e > 0.f ? e : a.neg_slope * e;
            // the attention value, from the row statistics of the forward pass
            const float at = __expf((w[j] * ns) * lr - a.stats[(int64_t)row * 2 * H + h]) /
                             a.stats[(int64_t)row * 2 * H + H + h];
            sa[c * H + h] = at;
            at4[j] = at;
            sc1[c * H + h] = (w[j] * ns) * (e > 0.f ? 1.0f : a.neg_slope);
            sc2[c * H + h] = lr * ns;
          }
        }
        if (a.attn) {      // by-product: the weights of the d ft aggregation, 16 bytes per head chunk
          if ((H & 3) == 0) {
            *reinterpret_cast<float4*>(a.attn + ed * H + 4 * cc) = make_float4(at4[0], at4[1], at4[2], at4[3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (4 * cc + j < H) a.attn[ed * H + 4 * cc + j] = at4[j];
          }
        }
      }
    }
    wave_sync();
    for (int i = 0; i < nb; i += 4) {      // four rows in flight
      int ui[4], ei[4];
      float fv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ui[j] = __builtin_amdgcn_ds_bpermute((team_lane0 + i + j) << 2, u);
        ei[j] = __builtin_amdgcn_ds_bpermute((team_lane0 + i + j) << 2, edl);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < nb && kin) load4(a.ft + (int64_t)ui[j] * HF, k0, HF, true, fv[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float dot = 0.f;
        if (i + j < nb && kin)
          dot = (gv[0] * fv[j][0] + gv[1] * fv[j][1]) + (gv[2] * fv[j][2] + gv[3] * fv[j][3]);
        dot = head_sum(dot);
        if (i + j < nb && kin && (k0 % F) == 0) {
          const float ds = sa[(i + j) * H + hl] * (dot - gdo);
          ba.de[(int64_t)ei[j] * H + hl] = ds * sc1[(i + j) * H + hl];
          if (ba.dw) ba.dw[(int64_t)ei[j] * H + hl] = ds * sc2[(i + j) * H + hl];
        }
      }
    }
  }
}

// ---- workgroup-cooperative backward (stag_gat_bwd) -------------------------------------------------
// Two passes over batches of units, each shaped like gat_fwd_block_kernel:
//   edge pass (destination-major batches): phase 1, thread <-> edge: the weight redrawn from its
//     counters, a = exp(e - m[v]) / l[v] from the forward's statistics, c1 = w ns lrelu'(s),
//     (c2 = lrelu(s) ns) -> LDS; phase 2, a team per unit: G[v] and gdo[v] = <G[v], out[v]> in registers,
//     per in-edge the ft row, <G[v,h,:], ft[u,h,:]> reduced over the head's lanes, ds = a (da - gdo),
//     de = ds c1 -> LDS; phase 3, thread <-> edge: (a, de) written BY FORWARD POSITION, [E, 2H], coalesced;
//     thread <-> (unit, head): d er[v,h] = sum of de over the unit (segments: a partial per segment).
//   source pass (source-major batches of the transposed CSR): phase 1, thread <-> out-edge: (a, de) of
//     its forward position (csr_t.nidx) -> LDS; d el[u,h] = sum of de; phase 2: d ft[u] = sum a G[v],
//     the forward kernel's weighted gather with G in the place of ft.
// Long rows (segments) leave partial sums in the plan's workspace; gat_seg_finish_kernel adds them in
// segment order (deterministic; nothing is accumulated with atomics).
struct GatBwdBlkArgs {
  GatArgs f;             // edge pass: the forward's arguments (graph, el, er, ft, noise, stats, plan batches)
  const float* g;        // [n_rows, H*F]
  const float* out;      // [n_rows, H*F] forward output
  float* ade;            // [E, 2H] by forward position: a[H] then de[H]
  float* d_er;           // [n_rows, H]
  float* dw;             // [E, H] by edge id, or null
  float* ws;             // segment partials of d er: [n_seg][H]
};

// the batch a workgroup owns: unit records and the edge-slot prefix in LDS; returns the unit count
__device__ __forceinline__ int blk_prologue(const stag_unit* units, const int32_t* block_ptr, int4* s_unit,
                                            int* s_start) {
  const int t = threadIdx.x;
  const int ub = block_ptr[blockIdx.x], nu = block_ptr[blockIdx.x + 1] - ub;
  if (t < kBlkUnits) {
    int4 q = make_int4(0, 0, 0, -1);
    if (t < nu) q = *reinterpret_cast<const int4*>(units + ub + t);
    s_unit[t] = q;
    int incl = q.z;
#pragma unroll
    for (int d = 1; d < kBlkUnits; d <<= 1) {
      const int up = __shfl_up(incl, d, kBlkUnits);          // lanes 0..kBlkUnits-1 of wave 0
      if (t >= d) incl += up;
    }
    s_start[t + 1] = incl;
    if (t == 0) s_start[0] = 0;
  }
  __syncthreads();
  return nu;
}

__device__ __forceinline__ int blk_unit_of(const int* s_start, int nu, int t) {
  int lo = 0, hi = nu;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (s_start[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

// sum over the F/4 lanes of a head (lanes_per_head a power of two): DPP inside a row of 16 lanes
__device__ __forceinline__ float gat_head_sum(float x, int lanes_per_head) {
  if (lanes_per_head >= 2) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
  if (lanes_per_head >= 4) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
  if (lanes_per_head >= 8) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));
  if (lanes_per_head >= 16) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));
  if (lanes_per_head >= 32) x += __shfl_xor(x, 16);
  if (lanes_per_head >= 64) x += __shfl_xor(x, 32);
  return x;
}

template <int LPE, int CPL>
__global__ __launch_bounds__(kBlkThreads) void gat_bwd_edge_block_kernel(const GatBwdBlkArgs ba) {
  extern __shared__ __align__(16) float lds[];
  const GatArgs& a = ba.f;
  const int H = a.H, F = a.F, HF = a.HF;
  float* s_a = lds;                                   // [kBlkEdges][H] attention
  float* s_c1 = s_a + kBlkEdges * H;                  // [kBlkEdges][H] w ns lrelu'(s), then de
  float* s_c2 = s_c1 + kBlkEdges * H;                 // [kBlkEdges][H] lrelu(s) ns, then dw (only when wanted)
  int* s_u = reinterpret_cast<int*>(s_c2 + (ba.dw ? kBlkEdges * H : 0));
  int* s_start = s_u + kBlkEdges;
  int4* s_unit = reinterpret_cast<int4*>(s_start + kBlkUnits + 4);
  const int t = threadIdx.x;
  const int nu = blk_prologue(a.units, a.block_ptr, s_unit, s_start);
  const int ne = s_start[nu];

  // ---- phase 1: thread <-> edge slot -------------------------------------------------------------
  int my_p = 0;
  int64_t my_ed = 0;
  if (t < ne) {
    const int j = blk_unit_of(s_start, nu, t);
    const int4 q = s_unit[j];
    const int row = (q.w >= 0) ? a.long_rows[q.x] : q.x;
    const int p = q.y + (t - s_start[j]);
    my_p = p;
    const int u = a.indices[p];
    s_u[t] = u;
    const int64_t ed = a.eid ? a.eid[p] : p;
    my_ed = ed;
    const uint32_t n = a.pos_lo + (a.nidx ? (uint32_t)a.nidx[p] : (uint32_t)p);
    const PhiloxKey key = resolve_epoch(a.key);
    const int nchunk = (H + 3) / 4;
    const bool h4 = (H & 3) == 0 && a.hvec;
    for (int cc = 0; cc < nchunk; ++cc) {
      float w[4], sl4[4], sr4[4], ns4[4] = {1.f, 1.f, 1.f, 1.f}, m4[4], l4[4];
      if (h4) {
        load4(a.el + (int64_t)u * H, 4 * cc, H, true, sl4);
        load4(a.er + (int64_t)row * H, 4 * cc, H, true, sr4);
        if (a.nscale) load4(a.nscale + (int64_t)row * H, 4 * cc, H, true, ns4);
        load4(a.stats + (int64_t)row * 2 * H, 4 * cc, H, true, m4);
        load4(a.stats + (int64_t)row * 2 * H + H, 4 * cc, H, true, l4);
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int h = 4 * cc + jj;
          const bool in = h < H;
          sl4[jj] = in ? a.el[(int64_t)u * H + h] : 0.f;
          sr4[jj] = in ? a.er[(int64_t)row * H + h] : 0.f;
          if (a.nscale && in) ns4[jj] = a.nscale[(int64_t)row * H + h];
          m4[jj] = in ? a.stats[(int64_t)row * 2 * H + h] : 0.f;
          l4[jj] = in ? a.stats[(int64_t)row * 2 * H + H + h] : 1.f;
        }
      }
      head_w4(a, key, n, ed, (uint32_t)cc, w);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int h = 4 * cc + jj;
        if (h < H) {
          const float sL = sl4[jj] + sr4[jj];
          const float lr = sL > 0.f ? sL : a.neg_slope * sL;
          const float wn = w[jj] * ns4[jj];
          s_a[t * H + h] = __expf(wn * lr - m4[jj]) / l4[jj];
          s_c1[t * H + h] = wn * (sL > 0.f ? 1.0f : a.neg_slope);
          if (ba.dw) s_c2[t * H + h] = lr * ns4[jj];
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: a team per unit: <G[v,h,:], ft[u,h,:]> per in-edge -> ds -> de ---------------------------
  constexpr int TEAMS = kBlkThreads / LPE, NR = CPL >= 4 ? 2 : STAG_GAT_NR;
  const int team = t / LPE, c = t % LPE;
  const int lph = a.lphp;
  int k0[CPL], hl[CPL];
  bool kin[CPL];
#pragma unroll
  for (int cj = 0; cj < CPL; ++cj) {
    lane_chunk(c + LPE * cj, H, F, a.lphp, k0[cj], kin[cj], hl[cj]);
  }
  const __amdgpu_buffer_rsrc_t rft =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ft), 0, (int)a.ft_bytes, 0x00020000);
  const bool ft_buf = a.ft_bytes != 0;
  for (int j = team; j < nu; j += TEAMS) {
    const int4 q = s_unit[j];
    const int row = (q.w >= 0) ? a.long_rows[q.x] : q.x;
    const int e0 = s_start[j], e1 = s_start[j + 1];
    if (e0 == e1) continue;
    float gv[CPL][4], gdo[CPL];
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) {
      float ov[4] = {0.f, 0.f, 0.f, 0.f};
      gv[cj][0] = gv[cj][1] = gv[cj][2] = gv[cj][3] = 0.f;
      if (kin[cj]) {
        load4(ba.g + (int64_t)row * HF, k0[cj], HF, true, gv[cj]);
        load4(ba.out + (int64_t)row * HF, k0[cj], HF, true, ov);
      }
      gdo[cj] = gat_head_sum((gv[cj][0] * ov[0] + gv[cj][1] * ov[1]) + (gv[cj][2] * ov[2] + gv[cj][3] * ov[3]), lph);
    }
    for (int e = e0; e < e1; e += NR) {
      float fv[NR][CPL][4];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {
          const int u = s_u[e + r];
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            if (kin[cj]) {
              if (ft_buf) bufrow4(rft, u, (uint32_t)HF * 4u, (uint32_t)k0[cj] * 4u, fv[r][cj]);
              else loadrow4(a.ft + (int64_t)u * HF + k0[cj], k0[cj], HF, true, fv[r][cj]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {                              // uniform over the team
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            float dot = 0.f;
            if (kin[cj])
              dot = (gv[cj][0] * fv[r][cj][0] + gv[cj][1] * fv[r][cj][1]) + (gv[cj][2] * fv[r][cj][2] + gv[cj][3] * fv[r][cj][3]);
            dot = gat_head_sum(dot, lph);
            if (kin[cj] && (k0[cj] % F) == 0) {
              const float ds = s_a[(e + r) * H + hl[cj]] * (dot - gdo[cj]);
              s_c1[(e + r) * H + hl[cj]] = ds * s_c1[(e + r) * H + hl[cj]];
              if (ba.dw) s_c2[(e + r) * H + hl[cj]] = ds * s_c2[(e + r) * H + hl[cj]];
            }
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 3: per-edge results by forward position (coalesced); d er per unit ------------------------
  if (t < ne) {
    float* dst = ba.ade + (int64_t)my_p * 2 * H;
    if ((H & 3) == 0) {
      for (int h = 0; h < H; h += 4) {
        *reinterpret_cast<float4*>(dst + h) = *reinterpret_cast<const float4*>(s_a + t * H + h);
        *reinterpret_cast<float4*>(dst + H + h) = *reinterpret_cast<const float4*>(s_c1 + t * H + h);
      }
    } else {
      for (int h = 0; h < H; ++h) { dst[h] = s_a[t * H + h]; dst[H + h] = s_c1[t * H + h]; }
    }
    if (ba.dw)
      for (int h = 0; h < H; ++h) ba.dw[my_ed * H + h] = s_c2[t * H + h];
  }
  for (int i = t; i < nu * H; i += kBlkThreads) {
    const int j = i / H, h = i - j * H;
    const int4 q = s_unit[j];
    float sum = 0.f;
    for (int e = s_start[j]; e < s_start[j + 1]; ++e) sum += s_c1[e * H + h];
    if (q.w < 0) ba.d_er[(int64_t)q.x * H + h] = sum;
    else ba.ws[(int64_t)q.w * H + h] = sum;            // a segment's share; gat_seg_finish_kernel adds them
  }
}

struct GatSrcBlkArgs {
  const int32_t* indices;      // csr_t: destination row of each transposed position
  const int32_t* nidx;         // csr_t: forward position of each transposed position
  const stag_unit* units;
  const int32_t* block_ptr;
  const int32_t* long_rows;
  const float* ade;            // [E, 2H] by forward position
  const float* g;              // [n_dst rows of the forward, H*F]
  int32_t H, F, HF;
  uint32_t g_bytes;
  float* d_ft;                 // [n_src, H*F]
  float* d_el;                 // [n_src, H]
  float* ws;                   // segment partials: [n_seg_t][HF + H]
};

template <int LPE, int CPL>
__global__ __launch_bounds__(kBlkThreads) void gat_bwd_src_block_kernel(const GatSrcBlkArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int H = a.H, F = a.F, HF = a.HF;
  float* s_a = lds;                                   // [kBlkEdges][H]
  float* s_de = s_a + kBlkEdges * H;                  // [kBlkEdges][H]
  int* s_v = reinterpret_cast<int*>(s_de + kBlkEdges * H);
  int* s_start = s_v + kBlkEdges;
  int4* s_unit = reinterpret_cast<int4*>(s_start + kBlkUnits + 4);
  const int t = threadIdx.x;
  const int nu = blk_prologue(a.units, a.block_ptr, s_unit, s_start);
  const int ne = s_start[nu];
  if (t < ne) {
    const int j = blk_unit_of(s_start, nu, t);
    const int4 q = s_unit[j];
    const int qq = q.y + (t - s_start[j]);
    s_v[t] = a.indices[qq];
    const float* src = a.ade + (int64_t)a.nidx[qq] * 2 * H;
    if ((H & 3) == 0) {
      for (int h = 0; h < H; h += 4) {
        *reinterpret_cast<float4*>(s_a + t * H + h) = *reinterpret_cast<const float4*>(src + h);
        *reinterpret_cast<float4*>(s_de + t * H + h) = *reinterpret_cast<const float4*>(src + H + h);
      }
    } else {
      for (int h = 0; h < H; ++h) { s_a[t * H + h] = src[h]; s_de[t * H + h] = src[H + h]; }
    }
  }
  __syncthreads();
  // d el[u,h] = sum of de over the out-edges, in transposed-position order
  for (int i = t; i < nu * H; i += kBlkThreads) {
    const int j = i / H, h = i - j * H;
    const int4 q = s_unit[j];
    float sum = 0.f;
    for (int e = s_start[j]; e < s_start[j + 1]; ++e) sum += s_de[e * H + h];
    if (q.w < 0) a.d_el[(int64_t)q.x * H + h] = sum;
    else a.ws[(int64_t)q.w * (HF + H) + HF + h] = sum;
  }
  // d ft[u,h,:] = sum over out-edges of a[e,h] G[v,h,:]
  constexpr int TEAMS = kBlkThreads / LPE, NR = CPL >= 4 ? 2 : STAG_GAT_NR;
  const int team = t / LPE, c = t % LPE;
  int k0[CPL], hl[CPL];
  bool kin[CPL];
#pragma unroll
  for (int cj = 0; cj < CPL; ++cj) {
    k0[cj] = (c + LPE * cj) * 4;
    kin[cj] = k0[cj] < HF;
    hl[cj] = kin[cj] ? k0[cj] / F : 0;
  }
  const __amdgpu_buffer_rsrc_t rg =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, (int)a.g_bytes, 0x00020000);
  const bool g_buf = a.g_bytes != 0;
  for (int j = team; j < nu; j += TEAMS) {
    const int4 q = s_unit[j];
    const int e0 = s_start[j], e1 = s_start[j + 1];
    float acc[CPL][4];
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) acc[cj][0] = acc[cj][1] = acc[cj][2] = acc[cj][3] = 0.f;
    for (int e = e0; e < e1; e += NR) {
      float fv[NR][CPL][4];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {
          const int v = s_v[e + r];
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            if (kin[cj]) {
              if (g_buf) bufrow4(rg, v, (uint32_t)HF * 4u, (uint32_t)k0[cj] * 4u, fv[r][cj]);
              else loadrow4(a.g + (int64_t)v * HF + k0[cj], k0[cj], HF, true, fv[r][cj]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            if (kin[cj]) {
              const float w = s_a[(e + r) * H + hl[cj]];
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[cj][x] = __builtin_fmaf(w, fv[r][cj][x], acc[cj][x]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) {
      if (!kin[cj]) continue;
      if (q.w < 0) store4_out(a.d_ft + (int64_t)q.x * HF, k0[cj], HF, true, acc[cj]);
      else store4(a.ws + (int64_t)q.w * (HF + H), k0[cj], HF, true, acc[cj]);
    }
  }
}

// ---- backward with ONE gather of [H*F] rows ---------------------------------------------------------------
// The two kernels above gather 1-KB rows twice: ft[u] by destination (for <G[v], ft[u]>) and G[v] by source (for
// d ft[u] = sum a G[v]).  The dot product is symmetric in where it is formed: on the SOURCE-major CSR ft[u] is the
// unit's own row and G[v] is the row the d ft sum gathers anyway.  What the source-major side lacks is the softmax
// correction sum_e' a[e',h] <G[v,h], ft[u',h]> = <G[v,h], out[v,h]> of the destination row — a per-node quantity,
// sdot [N, H], that one streaming pass over G and out provides (gat_rowdot_kernel) — and d er[v], whose terms it
// leaves in dsl [E, H] by forward position, where the in-edges of v are contiguous (gat_der_kernel).
//   gat_rowdot_kernel       sdot[v,h] = sum_f G[v,h,f] out[v,h,f]                       (2 N H F floats read)
//   gat_bwd_one_kernel      per batch of the source-major plan: phase 1 (thread per edge) a[e,h], the leaky-relu /
//                           noise factors and sdot[v,h] into LDS; phase 2 (team per source row) gathers G[v], forms
//                           the per-head dot with the own ft row, d s[e,h] = a (dot - sdot) c1, and d ft[u] += a G[v];
//                           phase 3 writes dsl (by forward position), dw (by edge id), d el[u]
//   gat_der_kernel          d er[v,h] = sum of dsl over the row's positions
struct GatBwd1Args {
  GatArgs f;             // graph fields: the SOURCE-major CSR (indices = destination, eid, nidx) and its block plan
  const float* g;        // [n_dst of the forward, H*F]
  uint32_t g_bytes;
  const float* pack;     // [n_dst of the forward, 4H]: er | m | l | sdot per destination (gat_rowdot_kernel)
  float* d_ft;           // [n_src, H*F]
  float* d_el;           // [n_src, H]
  float* dsl;            // [E, H] by forward position
  float* dw;             // [E, H] by edge id, or null
  float* ws;             // segment partials: [n_seg_t][HF + H]
  int32_t eid_is_pos;    // the forward CSR has no eid: edge id = forward position
  float* dp_part;        // [n_blocks][2][H] or null: this batch's share of the gradients of scalar / per-head noise
                         // parameters, sum_e dw[e,h] * dw/dp_i[e,h] (stag_gat_bwd_dp; vi=True, stag/layers.py:123-124)
};

constexpr int kRowdotRows = 4;
// sdot goes into a per-node record next to the other per-destination inputs of the edge phase — er, the softmax
// statistics m and l — so that an edge fetches ONE line of 4H floats instead of three separate rows:
// pack[v] = er[H] | m[H] | l[H] | sdot[H]
template <int LPE>
__global__ __launch_bounds__(256) void gat_rowdot_kernel(const float* g, const float* out, int n, int H, int F, int HF,
                                                         int lphp, const float* er, const float* stats, float* pack) {
  constexpr int R = kRowdotRows;
  const int c = threadIdx.x % LPE;
  const int row0 = (blockIdx.x * (256 / LPE) + threadIdx.x / LPE) * R;
  const int lph = lphp;
  for (int L0 = 0; L0 < H * lphp; L0 += LPE) {          // every lane makes every trip (the head sums are DPP)
    int k0, hd;
    bool in;
    lane_chunk(L0 + c, H, F, lphp, k0, in, hd);
    float gv[R][4], ov[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      gv[r][0] = gv[r][1] = gv[r][2] = gv[r][3] = 0.f;
      ov[r][0] = ov[r][1] = ov[r][2] = ov[r][3] = 0.f;
      if (in && row0 + r < n) {
        load4(g + (int64_t)(row0 + r) * HF, k0, HF, true, gv[r]);
        load4(out + (int64_t)(row0 + r) * HF, k0, HF, true, ov[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float d = gat_head_sum((gv[r][0] * ov[r][0] + gv[r][1] * ov[r][1]) + (gv[r][2] * ov[r][2] + gv[r][3] * ov[r][3]), lph);
      if (in && row0 + r < n && (k0 % F) == 0) {
        const int64_t row = row0 + r;
        const int h = k0 / F;
        float* rec = pack + row * 4 * H;
        rec[h] = er[row * H + h];
        rec[H + h] = stats[row * 2 * H + h];
        rec[2 * H + h] = stats[row * 2 * H + H + h];
        rec[3 * H + h] = d;
      }
    }
  }
}

template <int LPE, int CPL>
__global__ __launch_bounds__(kBlkThreads) void gat_bwd_one_kernel(const GatBwd1Args ba) {
  extern __shared__ __align__(16) float lds[];
  const GatArgs& a = ba.f;
  const int H = a.H, F = a.F, HF = a.HF;
  float* s_a = lds;                                   // [kBlkEdges][H] attention
  float* s_c1 = s_a + kBlkEdges * H;                  // [kBlkEdges][H] w ns lrelu'(s), then d s
  float* s_sd = s_c1 + kBlkEdges * H;                 // [kBlkEdges][H] sdot of the edge's destination
  float* s_c2 = s_sd + kBlkEdges * H;                 // [kBlkEdges][H] lrelu(s) ns, then dw (only when wanted)
  const bool want_dw = ba.dw != nullptr || ba.dp_part != nullptr;
  int* s_v = reinterpret_cast<int*>(s_c2 + (want_dw ? kBlkEdges * H : 0));
  int* s_start = s_v + kBlkEdges;
  int4* s_unit = reinterpret_cast<int4*>(s_start + kBlkUnits + 4);
  const int t = threadIdx.x;
  const int nu = blk_prologue(a.units, a.block_ptr, s_unit, s_start);
  const int ne = s_start[nu];

  // ---- phase 1: thread <-> out-edge slot -----------------------------------------------------------------
  int my_fp = 0;
  int64_t my_ed = 0;
  if (t < ne) {
    const int j = blk_unit_of(s_start, nu, t);
    const int4 q = s_unit[j];
    const int u = (q.w >= 0) ? a.long_rows[q.x] : q.x;          // the source node: this unit's row
    const int qq = q.y + (t - s_start[j]);
    const int v = a.indices[qq];
    s_v[t] = v;
    const int fp = a.nidx[qq];                                  // forward position: the noise index, the dsl slot
    my_fp = fp;
    const int64_t ed = (ba.eid_is_pos || !a.eid) ? (int64_t)fp : (int64_t)a.eid[qq];
    my_ed = ed;
    const uint32_t n = a.pos_lo + (uint32_t)fp;
    const PhiloxKey key = resolve_epoch(a.key);
    const int nchunk = (H + 3) / 4;
    const bool h4 = (H & 3) == 0 && a.hvec;
    for (int cc = 0; cc < nchunk; ++cc) {
      float w[4], sl4[4], sr4[4], ns4[4] = {1.f, 1.f, 1.f, 1.f}, m4[4], l4[4], sd4[4];
      const float* rec = ba.pack + (int64_t)v * 4 * H;          // er | m | l | sdot of the destination
      if (h4) {
        load4(a.el + (int64_t)u * H, 4 * cc, H, true, sl4);
        if (a.nscale) load4(a.nscale + (int64_t)v * H, 4 * cc, H, true, ns4);
        load4(rec, 4 * cc, H, true, sr4);
        load4(rec + H, 4 * cc, H, true, m4);
        load4(rec + 2 * H, 4 * cc, H, true, l4);
        load4(rec + 3 * H, 4 * cc, H, true, sd4);
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int h = 4 * cc + jj;
          const bool in = h < H;
          sl4[jj] = in ? a.el[(int64_t)u * H + h] : 0.f;
          if (a.nscale && in) ns4[jj] = a.nscale[(int64_t)v * H + h];
          sr4[jj] = in ? rec[h] : 0.f;
          m4[jj] = in ? rec[H + h] : 0.f;
          l4[jj] = in ? rec[2 * H + h] : 1.f;
          sd4[jj] = in ? rec[3 * H + h] : 0.f;
        }
      }
      head_w4(a, key, n, ed, (uint32_t)cc, w);
      const uint32_t kbits = a.drop_keep > 0.f ? drop_keep4(a, resolve_epoch(a.drop_key), n, (uint32_t)cc) : 15u;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int h = 4 * cc + jj;
        if (h < H) {
          const float sL = sl4[jj] + sr4[jj];
          const float lr = sL > 0.f ? sL : a.neg_slope * sL;
          const float wn = w[jj] * ns4[jj];
          const float av = __expf(wn * lr - m4[jj]) / l4[jj];
          // attention dropout: the sign carries the keep bit (a > 0): dropped edges are stored as -a
          s_a[t * H + h] = (a.drop_keep > 0.f && !((kbits >> jj) & 1u)) ? -av : av;
          s_c1[t * H + h] = wn * (sL > 0.f ? 1.0f : a.neg_slope);
          s_sd[t * H + h] = sd4[jj];
          if (want_dw) s_c2[t * H + h] = lr * ns4[jj];
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: a team per source row: gather G[v]; <G[v,h,:], ft[u,h,:]> -> d s; d ft[u] += a G[v] ----------
  constexpr int TEAMS = kBlkThreads / LPE, NR = CPL >= 4 ? 2 : STAG_GAT_NR;
  const int team = t / LPE, c = t % LPE;
  const int lph = a.lphp;
  int k0[CPL], hl[CPL];
  bool kin[CPL];
#pragma unroll
  for (int cj = 0; cj < CPL; ++cj) {
    lane_chunk(c + LPE * cj, H, F, a.lphp, k0[cj], kin[cj], hl[cj]);
  }
  const __amdgpu_buffer_rsrc_t rg =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ba.g), 0, (int)ba.g_bytes, 0x00020000);
  const bool g_buf = ba.g_bytes != 0;
  for (int j = team; j < nu; j += TEAMS) {
    const int4 q = s_unit[j];
    const int u = (q.w >= 0) ? a.long_rows[q.x] : q.x;
    const int e0 = s_start[j], e1 = s_start[j + 1];
    float fu[CPL][4], acc[CPL][4];
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) {
      fu[cj][0] = fu[cj][1] = fu[cj][2] = fu[cj][3] = 0.f;
      acc[cj][0] = acc[cj][1] = acc[cj][2] = acc[cj][3] = 0.f;
      if (kin[cj] && e0 < e1) load4(a.ft + (int64_t)u * HF, k0[cj], HF, true, fu[cj]);
    }
    for (int e = e0; e < e1; e += NR) {
      float fv[NR][CPL][4];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {
          const int v = s_v[e + r];
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            if (kin[cj]) {
              if (g_buf) bufrow4(rg, v, (uint32_t)HF * 4u, (uint32_t)k0[cj] * 4u, fv[r][cj]);
              else loadrow4(ba.g + (int64_t)v * HF + k0[cj], k0[cj], HF, true, fv[r][cj]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (e + r < e1) {                              // uniform over the team
#pragma unroll
          for (int cj = 0; cj < CPL; ++cj) {
            float dot = 0.f;
            if (kin[cj])
              dot = (fu[cj][0] * fv[r][cj][0] + fu[cj][1] * fv[r][cj][1]) + (fu[cj][2] * fv[r][cj][2] + fu[cj][3] * fv[r][cj][3]);
            dot = gat_head_sum(dot, lph);
            if (kin[cj]) {
              float w = s_a[(e + r) * H + hl[cj]];
              float ds;
              if (a.drop_keep > 0.f) {        // a' = a keep / q feeds d ft and <G, ft>; the correction stays with a
                const float aa = fabsf(w);
                w = w > 0.f ? w * a.drop_scale : 0.f;
                ds = w * dot - aa * s_sd[(e + r) * H + hl[cj]];
              } else {
                ds = w * (dot - s_sd[(e + r) * H + hl[cj]]);
              }
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[cj][x] = __builtin_fmaf(w, fv[r][cj][x], acc[cj][x]);
              if ((k0[cj] % F) == 0) {
                s_c1[(e + r) * H + hl[cj]] = ds * s_c1[(e + r) * H + hl[cj]];
                if (want_dw) s_c2[(e + r) * H + hl[cj]] = ds * s_c2[(e + r) * H + hl[cj]];
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int cj = 0; cj < CPL; ++cj) {
      if (!kin[cj]) continue;
      if (q.w < 0) store4_out(ba.d_ft + (int64_t)q.x * HF, k0[cj], HF, true, acc[cj]);
      else store4(ba.ws + (int64_t)q.w * (HF + H), k0[cj], HF, true, acc[cj]);
    }
  }
  __syncthreads();

  // ---- gradients of scalar / per-head noise parameters (vi=True): dw[e,h] is in LDS, the draw is redone with its
  //      derivatives (same counters), the products go where a and sdot were (both consumed), and 2H threads add the
  //      batch's edges in order: one [2][H] partial per batch, summed by two fixed-order launches (dp_stage*)
  if (ba.dp_part) {
    if (t < ne) {
      const uint32_t n = a.pos_lo + (uint32_t)my_fp;
      const PhiloxKey key = resolve_epoch(a.key);
      const uint32_t c1hi = a.pos_hi << 20;
      for (int cc = 0; cc < (H + 3) / 4; ++cc) {
        float pa[4], pb[4], w[4], d0[4], d1[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int h = 4 * cc + jj;
          const bool in = h < H;
          float q0 = a.p0s, q1 = a.p1s;
          if (a.pmode == STAG_PARAM_PER_CHANNEL) { q0 = in ? a.p0[h] : 0.f; q1 = (in && a.p1) ? a.p1[h] : 0.f; }
          if (a.pmode != STAG_PARAM_SCALAR && (a.relu & kFlagLogScale)) q1 = exp_scale(q1);
          pa[jj] = q0; pb[jj] = q1;
        }
        if (a.kind == kNormal) draw4_grad<kNormal>(n, (uint32_t)cc | c1hi, key, pa, pb, a.relu, w, d0, d1);
        else draw4_grad<kUniform>(n, (uint32_t)cc | c1hi, key, pa, pb, a.relu, w, d0, d1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int h = 4 * cc + jj;
          if (h < H) {
            s_a[t * H + h] = s_c2[t * H + h] * d0[jj];
            s_sd[t * H + h] = s_c2[t * H + h] * d1[jj];
          }
        }
      }
    }
    __syncthreads();
    if (t < 2 * H) {
      const float* src = t < H ? s_a : s_sd;
      const int h = t < H ? t : t - H;
      float sum = 0.f;
      for (int e = 0; e < ne; ++e) sum += src[e * H + h];
      ba.dp_part[(int64_t)blockIdx.x * 2 * H + t] = sum;
    }
  }

  // ---- phase 3: d s by forward position, dw by edge id; d el per unit ---------------------------------------
  if (t < ne) {
    float* dst = ba.dsl + (int64_t)my_fp * H;
    if ((H & 3) == 0) {
      for (int h = 0; h < H; h += 4)
        *reinterpret_cast<float4*>(dst + h) = *reinterpret_cast<const float4*>(s_c1 + t * H + h);
    } else {
      for (int h = 0; h < H; ++h) dst[h] = s_c1[t * H + h];
    }
    if (ba.dw)
      for (int h = 0; h < H; ++h) ba.dw[my_ed * H + h] = s_c2[t * H + h];
  }
  for (int i = t; i < nu * H; i += kBlkThreads) {
    const int j = i / H, h = i - j * H;
    const int4 q = s_unit[j];
    float sum = 0.f;
    for (int e = s_start[j]; e < s_start[j + 1]; ++e) sum += s_c1[e * H + h];
    if (q.w < 0) ba.d_el[(int64_t)q.x * H + h] = sum;
    else ba.ws[(int64_t)q.w * (HF + H) + HF + h] = sum;
  }
}

// d er[v,h] = sum over the positions of row v of dsl[p,h]: a thread per (unit of the forward plan, head), the
// unit's <= seg_len positions in order; segments leave partials for gat_seg_finish_kernel
__global__ __launch_bounds__(256) void gat_der_kernel(const stag_unit* units, int n_units, const int32_t* indptr,
                                                      const float* dsl, int H, float* d_er, float* ws) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t unit = i / H;
  const int h = (int)(i - unit * H);
  if (unit >= n_units) return;
  int row, b, len, slot = -1;
  if (units) {
    const int4 q = *reinterpret_cast<const int4*>(units + unit);
    row = q.x; b = q.y; len = q.z; slot = q.w;
  } else {
    row = (int)unit; b = indptr[row]; len = indptr[row + 1] - b;
  }
  float sum = 0.f;
  for (int p = b; p < b + len; ++p) sum += dsl[(int64_t)p * H + h];
  if (slot < 0) d_er[(int64_t)row * H + h] = sum;
  else ws[(int64_t)slot * H + h] = sum;
}

// out[long_rows[r]][k] = sum over the row's segments of ws[s][ws_off + k], k < width.  A workgroup per
// (long row, 16 columns): 16 slices of the segment list are summed side by side (slice i takes segments
// i, i+16, ... in order, Kahan), then the 16 slice sums in slice order — a fixed order, and a hub row
// of 200 segments costs 13 dependent loads instead of 200.
__global__ __launch_bounds__(256) void gat_seg_finish_kernel(const float* ws, int ws_stride, int ws_off, int width,
                                                             const int32_t* long_rows, const int32_t* long_seg_ptr,
                                                             float* out, int ld_out) {
  __shared__ float red[16][17];
  const int r = blockIdx.x, kx = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int k = blockIdx.y * 16 + kx;
  const int s0 = long_seg_ptr[r], s1 = long_seg_ptr[r + 1];
  float sum = 0.f, comp = 0.f;
  if (k < width) {
    for (int s = s0 + slice; s < s1; s += 16) {
      const float y = ws[(int64_t)s * ws_stride + ws_off + k] - comp;
      const float n = sum + y;
      comp = (n - sum) - y;
      sum = n;
    }
  }
  red[slice][kx] = sum;
  __syncthreads();
  if (slice == 0 && k < width) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i][kx];
    out[(int64_t)long_rows[r] * ld_out + k] = tot;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// lanes per head of the cooperative kernels: F / 4 rounded up to a power of two (F % 4 == 0)
static int lanes_per_head(int F) {
  int l = 1;
  while (l * 4 < F) l <<= 1;
  return l;
}

// attention dropout of a call: 0 = none / set, STAG_EINVAL for a bad probability
static int fill_drop(GatArgs& a, const stag_gat_drop* drop) {
  a.drop_keep = 0.f; a.drop_scale = 1.f;
  if (!drop || drop->keep_prob >= 1.0f) return STAG_OK;
  if (!(drop->keep_prob > 0.0f)) return STAG_EINVAL;
  a.drop_keep = drop->keep_prob;
  a.drop_scale = 1.0f / drop->keep_prob;
  a.drop_key.k0 = (uint32_t)(drop->seed & 0xFFFFFFFFull); a.drop_key.k1 = (uint32_t)(drop->seed >> 32);
  a.drop_key.o0 = (uint32_t)(drop->offset & 0xFFFFFFFFull); a.drop_key.o1 = (uint32_t)(drop->offset >> 32);
  a.drop_key.epoch = drop->epoch;
  return STAG_OK;
}
static bool drop_on(const stag_gat_drop* drop) { return drop && drop->keep_prob < 1.0f; }

extern "C" size_t stag_gat_workspace_bytes(int32_t n_seg, int32_t H, int32_t F) {
  if (n_seg <= 0 || H <= 0 || F <= 0) return 0;
  return (size_t)n_seg * (size_t)((H * F + 2 * H + 3) & ~3) * sizeof(float);   // rows padded to 16 bytes
}

extern "C" int stag_gat_fwd(const stag_csr* csr, const stag_plan* plan, const float* el,
                            const float* er, const float* ft, int32_t H, int32_t F, float neg_slope,
                            const stag_noise_spec* spec, const float* norm_scale, const stag_gat_drop* drop,
                            float* out, float* stats_out, void* stream) {
  if (!csr || !csr->indptr || csr->n_dst < 0 || csr->n_edges < 0) return STAG_EINVAL;
  if (!spec || spec->kind < STAG_NOISE_NONE || spec->kind > STAG_NOISE_BERNOULLI) return STAG_EINVAL;
  if (!out || H <= 0 || F <= 0) return STAG_EINVAL;
  const int64_t HF64 = (int64_t)H * F;
  // one team spans the H*F row: 256 channels on the one-unit-per-team kernel, 1024 (4 chunks per lane) on the
  // workgroup-cooperative one (which also wants H <= 16, F % 4 == 0 and a block plan: checked below)
  if (H > 64 || HF64 > 1024) return STAG_ENOSYS;
  if (spec->chunk_base != 0) return STAG_ENOSYS;  // heads are not channel-sharded
  if (spec->in_norm && !norm_scale) return STAG_EINVAL;   // the caller runs the row-sum pass first
  if (csr->n_dst == 0) return STAG_OK;
  if (csr->n_edges > 0 && (!csr->indices || !el || !er || !ft)) return STAG_EINVAL;
  if (spec->kind == STAG_NOISE_EXPLICIT && !spec->p0 && csr->n_edges > 0) return STAG_EINVAL;
  if (spec->kind >= STAG_NOISE_NORMAL && spec->param_mode != STAG_PARAM_SCALAR && csr->n_edges > 0 &&
      (!spec->p0 || (spec->kind != STAG_NOISE_BERNOULLI && !spec->p1)))
    return STAG_EINVAL;
  const int HF = (int)HF64;

  GatArgs a{};
  a.indptr = csr->indptr; a.indices = csr->indices; a.eid = csr->eid; a.nidx = csr->nidx;
  a.n_rows = csr->n_dst; a.el = el; a.er = er; a.ft = ft;
  a.nscale = spec->in_norm ? norm_scale : nullptr;
  a.H = H; a.F = F; a.HF = HF;
  a.neg_slope = neg_slope; a.kind = spec->kind; a.p0 = spec->p0; a.p1 = spec->p1;
  const bool logs = spec->kind == STAG_NOISE_NORMAL && spec->p1_log;
  a.p0s = spec->p0_scalar; a.p1s = logs ? expf(spec->p1_scalar) : spec->p1_scalar;
  a.pmode = spec->kind >= STAG_NOISE_NORMAL ? spec->param_mode : 0;
  a.relu = (spec->relu ? kFlagRelu : 0) | (logs ? kFlagLogScale : 0);
  if (spec->deriv != 0) return STAG_EINVAL;
  a.key.k0 = (uint32_t)(spec->seed & 0xFFFFFFFFull); a.key.k1 = (uint32_t)(spec->seed >> 32);
  a.key.o0 = (uint32_t)(spec->offset & 0xFFFFFFFFull); a.key.o1 = (uint32_t)(spec->offset >> 32);
  a.key.epoch = spec->epoch;
  a.pos_lo = (uint32_t)((uint64_t)spec->pos_base & 0xFFFFFFFFull);
  a.pos_hi = (uint32_t)((uint64_t)spec->pos_base >> 32);
  if (spec->kind >= STAG_NOISE_NORMAL && (uint64_t)a.pos_lo + (uint64_t)csr->n_edges > (1ull << 32))
    return STAG_ENOSYS;
  a.out = out; a.stats = stats_out;
  if (fill_drop(a, drop)) return STAG_EINVAL;
  const uint64_t ftb = (uint64_t)csr->n_src * (uint64_t)HF * 4u;
  a.ft_bytes = (ftb < (1ull << 32) && csr->n_src < (1 << 24)) ? (uint32_t)ftb : 0u;

  a.n_units = csr->n_dst;
  const bool use_plan = plan && plan->n_units > 0;
  if (use_plan) {
    if (!plan->units || !aligned16(plan->units)) return STAG_EINVAL;
    a.units = plan->units; a.n_units = plan->n_units;
    if (plan->n_seg > 0) {
      if (!plan->long_rows || !plan->long_seg_ptr || !plan->workspace || !plan->seg_counters)
        return STAG_EINVAL;
      const size_t need = stag_gat_workspace_bytes(plan->n_seg, H, F);
      if (plan->workspace_bytes < need) return STAG_ENOMEM;
      if (need >= (1ull << 32)) return STAG_ENOSYS;
      a.long_rows = plan->long_rows; a.long_seg_ptr = plan->long_seg_ptr; a.n_long = plan->n_long;
      a.seg_counters = plan->seg_counters; a.ws = plan->workspace;
      a.ws_stride = (HF + 2 * H + 3) & ~3; a.ws_bytes = (uint32_t)need; a.n_seg = plan->n_seg;
    }
  }
  bool vec = (F % 4 == 0) && aligned16(ft) && aligned16(out);
  if (a.ws) vec = vec && aligned16(a.ws) && (a.ws_stride % 4 == 0);

  int nchunk = (HF + 3) / 4;
  int lpe = 4;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  int cpl = nchunk <= 64 ? 1 : (nchunk <= 128 ? 2 : 4);   // chunks of 4 channels per lane
  const int tpb = 256 / lpe;
  const dim3 grid((a.n_units + tpb - 1) / tpb);
  const size_t lds_bytes = (size_t)256 * H * sizeof(float);   // [teams][LPE][H]
  hipStream_t s = (hipStream_t)stream;
  // the cooperative kernel gives a head lphp = F / 4 rounded up to a power of two lanes (the backward's head sums are
  // DPP butterflies, and the forward keeps the backward's mapping): a row takes H * lphp lanes, at most 256
  const int lphp = (F % 4 == 0) ? lanes_per_head(F) : 0;
  const bool blk_ok = use_plan && plan->block_ptr && plan->n_blocks > 0 && vec && H <= kBlkMaxH && plan->seg_len <= kBlkEdges &&
                      lphp > 0 && lphp <= 64 && H * lphp <= 256;
  if (HF > 256 && !blk_ok) return STAG_ENOSYS;
  if (a.drop_keep > 0.f && !blk_ok) return STAG_ENOSYS;      // attention dropout lives in the cooperative kernels
  if (blk_ok) {
    // workgroup-cooperative form: batches of units (stag_plan_blocks with STAG_BLOCK_EDGES / _UNITS)
    a.block_ptr = plan->block_ptr;
    a.lphp = lphp;
    nchunk = H * lphp;
    lpe = 4;
    while (lpe < nchunk && lpe < 64) lpe <<= 1;
    cpl = nchunk <= 64 ? 1 : (nchunk <= 128 ? 2 : 4);
    a.hvec = aligned16(el) && aligned16(er) && (!a.nscale || aligned16(a.nscale));
    size_t lds_blk = (size_t)(kBlkEdges * H + 2 * kBlkUnits * H) * sizeof(float) +
                     (size_t)(kBlkEdges + kBlkUnits + 4) * sizeof(int) + (size_t)kBlkUnits * sizeof(int4);
    if (lds_blk < STAG_GAT_LDS_MIN) lds_blk = STAG_GAT_LDS_MIN;
    const dim3 gb(plan->n_blocks);
    const bool local = plan->xcd_order != nullptr;      // the batches are XCD-local: the rows come out of an L2
    // rows of at most 256 floats (one chunk per lane): STAG_GAT_NR_FWD_NARROW rows in flight, whatever the launch
#define STAG_BLK_LAUNCH1(L) \
  hipLaunchKernelGGL((gat_fwd_block_kernel<L, 1, STAG_GAT_NR_FWD_NARROW>), gb, dim3(kBlkThreads), lds_blk, s, a)
#define STAG_BLK_LAUNCHW(Cc)                                                                                             \
  do {                                                                                                                   \
    if (local) hipLaunchKernelGGL((gat_fwd_block_kernel<64, Cc, STAG_GAT_NR_FWD_LOCAL>), gb, dim3(kBlkThreads), lds_blk, s, a); \
    else       hipLaunchKernelGGL((gat_fwd_block_kernel<64, Cc>), gb, dim3(kBlkThreads), lds_blk, s, a);                 \
  } while (0)
    if (cpl == 4) STAG_BLK_LAUNCHW(4);
    else if (cpl == 2) STAG_BLK_LAUNCHW(2);
    else switch (lpe) {
      case 64: STAG_BLK_LAUNCH1(64); break;
      case 32: STAG_BLK_LAUNCH1(32); break;
      case 16: STAG_BLK_LAUNCH1(16); break;
      case 8: STAG_BLK_LAUNCH1(8); break;
      default: STAG_BLK_LAUNCH1(4); break;
    }
#undef STAG_BLK_LAUNCH1
#undef STAG_BLK_LAUNCHW
    return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
  }
#define STAG_GAT_LAUNCH(L)                                                                         \
  do {                                                                                             \
    if (vec) hipLaunchKernelGGL((gat_fwd_kernel<L, true>), grid, dim3(256), lds_bytes, s, a);     \
    else     hipLaunchKernelGGL((gat_fwd_kernel<L, false>), grid, dim3(256), lds_bytes, s, a);    \
  } while (0)
  switch (lpe) {
    case 64: STAG_GAT_LAUNCH(64); break;
    case 32: STAG_GAT_LAUNCH(32); break;
    case 16: STAG_GAT_LAUNCH(16); break;
    case 8: STAG_GAT_LAUNCH(8); break;
    default: STAG_GAT_LAUNCH(4); break;
  }
#undef STAG_GAT_LAUNCH
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

// the argument block the attention / backward kernels share with the forward
static int fill_edge_args(GatArgs& a, const stag_csr* csr, const stag_plan* plan, const float* el,
                          const float* er, int32_t H, float neg_slope, const stag_noise_spec* spec,
                          const float* norm_scale, const float* stats) {
  a.indptr = csr->indptr; a.indices = csr->indices; a.eid = csr->eid; a.nidx = csr->nidx;
  a.n_rows = csr->n_dst; a.el = el; a.er = er;
  a.nscale = spec->in_norm ? norm_scale : nullptr;
  a.H = H; a.neg_slope = neg_slope;
  const bool logs = spec->kind == STAG_NOISE_NORMAL && spec->p1_log;
  a.kind = spec->kind; a.p0 = spec->p0; a.p1 = spec->p1; a.p0s = spec->p0_scalar;
  a.p1s = logs ? expf(spec->p1_scalar) : spec->p1_scalar;
  a.pmode = spec->kind >= STAG_NOISE_NORMAL ? spec->param_mode : 0;
  a.relu = (spec->relu ? kFlagRelu : 0) | (logs ? kFlagLogScale : 0);
  a.key.k0 = (uint32_t)(spec->seed & 0xFFFFFFFFull); a.key.k1 = (uint32_t)(spec->seed >> 32);
  a.key.o0 = (uint32_t)(spec->offset & 0xFFFFFFFFull); a.key.o1 = (uint32_t)(spec->offset >> 32);
  a.key.epoch = spec->epoch;
  a.pos_lo = (uint32_t)((uint64_t)spec->pos_base & 0xFFFFFFFFull);
  a.pos_hi = (uint32_t)((uint64_t)spec->pos_base >> 32);
  a.stats = const_cast<float*>(stats);
  a.n_units = csr->n_dst;
  if (plan && plan->n_units > 0) {
    if (!plan->units || !aligned16(plan->units)) return STAG_EINVAL;
    if (plan->n_seg > 0 && !plan->long_rows) return STAG_EINVAL;
    a.units = plan->units; a.n_units = plan->n_units; a.long_rows = plan->long_rows;
  }
  return STAG_OK;
}

extern "C" int stag_gat_attn(const stag_csr* csr, const stag_plan* plan, const float* el,
                             const float* er, int32_t H, float neg_slope, const stag_noise_spec* spec,
                             const float* norm_scale, const float* stats, float* attn_out,
                             void* stream) {
  if (!csr || !csr->indptr || csr->n_dst < 0 || csr->n_edges < 0) return STAG_EINVAL;
  if (!spec || spec->kind < STAG_NOISE_NONE || spec->kind > STAG_NOISE_BERNOULLI || spec->deriv) return STAG_EINVAL;
  if (!attn_out || !stats || H <= 0 || H > 64) return STAG_EINVAL;
  if (spec->chunk_base != 0) return STAG_ENOSYS;
  if (spec->in_norm && !norm_scale) return STAG_EINVAL;
  if (csr->n_dst == 0 || csr->n_edges == 0) return STAG_OK;
  if (!csr->indices || !el || !er) return STAG_EINVAL;
  if (spec->kind == STAG_NOISE_EXPLICIT && !spec->p0 && csr->n_edges > 0) return STAG_EINVAL;
  GatArgs a{};
  const int rc = fill_edge_args(a, csr, plan, el, er, H, neg_slope, spec, norm_scale, stats);
  if (rc) return rc;
  a.attn = attn_out;
  hipLaunchKernelGGL(gat_attn_kernel, dim3((a.n_units + 31) / 32), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

extern "C" int stag_gat_bwd_edge(const stag_csr* csr, const stag_plan* plan, const float* el,
                                 const float* er, const float* ft, const float* stats, const float* g,
                                 const float* out, int32_t H, int32_t F, float neg_slope,
                                 const stag_noise_spec* spec, const float* norm_scale, float* de,
                                 float* dw, float* attn_out, void* stream) {
  if (!csr || !csr->indptr || csr->n_dst < 0 || csr->n_edges < 0) return STAG_EINVAL;
  if (!spec || spec->kind < STAG_NOISE_NONE || spec->kind > STAG_NOISE_BERNOULLI || spec->deriv) return STAG_EINVAL;
  if (!de || H <= 0 || F <= 0) return STAG_EINVAL;
  const int64_t HF64 = (int64_t)H * F;
  const int lph = F / 4;
  if (H > 64 || HF64 > 256 || F % 4 != 0 || (lph & (lph - 1)) != 0) return STAG_ENOSYS;
  if (spec->chunk_base != 0) return STAG_ENOSYS;
  if (spec->in_norm && !norm_scale) return STAG_EINVAL;
  if (csr->n_dst == 0 || csr->n_edges == 0) return STAG_OK;
  if (!csr->indices || !el || !er || !ft || !stats || !g || !out) return STAG_EINVAL;
  if (!aligned16(ft) || !aligned16(g) || !aligned16(out)) return STAG_EINVAL;
  if (spec->kind == STAG_NOISE_EXPLICIT && !spec->p0 && csr->n_edges > 0) return STAG_EINVAL;
  const int HF = (int)HF64;
  GatBwdArgs ba{};
  GatArgs& a = ba.f;
  const int rc = fill_edge_args(a, csr, plan, el, er, H, neg_slope, spec, norm_scale, stats);
  if (rc) return rc;
  a.ft = ft; a.F = F; a.HF = HF; a.attn = attn_out;
  ba.g = g; ba.gdo = nullptr; ba.out = out; ba.de = de; ba.dw = dw;
  const int nchunk = (HF + 3) / 4;
  int lpe = 4;
  while (lpe < nchunk) lpe <<= 1;
  const int tpb = 256 / lpe;
  const dim3 grid((a.n_units + tpb - 1) / tpb);
  const size_t lds_bytes = (size_t)3 * 256 * H * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  switch (lpe) {
    case 64: hipLaunchKernelGGL(gat_bwd_edge_kernel<64>, grid, dim3(256), lds_bytes, s, ba); break;
    case 32: hipLaunchKernelGGL(gat_bwd_edge_kernel<32>, grid, dim3(256), lds_bytes, s, ba); break;
    case 16: hipLaunchKernelGGL(gat_bwd_edge_kernel<16>, grid, dim3(256), lds_bytes, s, ba); break;
    case 8: hipLaunchKernelGGL(gat_bwd_edge_kernel<8>, grid, dim3(256), lds_bytes, s, ba); break;
    default: hipLaunchKernelGGL(gat_bwd_edge_kernel<4>, grid, dim3(256), lds_bytes, s, ba); break;
  }
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

// One call = the whole backward of stag_gat_fwd w.r.t. el, er, ft (and explicit weights): the edge pass
// over the forward CSR and the source pass over its transpose (see gat_bwd_edge_block_kernel).
extern "C" size_t stag_gat_bwd_workspace_bytes(int32_t n_seg, int32_t n_seg_t, int32_t H, int32_t F) {
  if (H <= 0 || F <= 0) return 0;
  const size_t a = n_seg > 0 ? (size_t)n_seg * (size_t)H : 0;
  const size_t b = n_seg_t > 0 ? (size_t)n_seg_t * (size_t)(H * F + H) : 0;
  return (a > b ? a : b) * sizeof(float);
}

extern "C" int stag_gat_bwd_two_pass(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                                     const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                                     const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                                     float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                                     const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* dw,
                                     float* ade_ws, void* stream) {
  if (drop_on(drop)) return STAG_ENOSYS;               // attention dropout: stag_gat_bwd
  if (!csr || !csr_t || !csr->indptr || !csr_t->indptr || csr->n_dst < 0 || csr->n_edges < 0) return STAG_EINVAL;
  if (csr_t->n_edges != csr->n_edges || csr_t->n_dst != csr->n_src || csr_t->n_src != csr->n_dst) return STAG_EINVAL;
  if (!spec || spec->kind < STAG_NOISE_NONE || spec->kind > STAG_NOISE_BERNOULLI || spec->deriv) return STAG_EINVAL;
  if (!d_el || !d_er || !d_ft || !ade_ws || H <= 0 || F <= 0) return STAG_EINVAL;
  const int64_t HF64 = (int64_t)H * F;
  const int lph = F / 4;
  if (H > kBlkMaxH || HF64 > 1024 || F % 4 != 0 || lph > 64 || (lph & (lph - 1)) != 0) return STAG_ENOSYS;
  if (spec->chunk_base != 0) return STAG_ENOSYS;
  if (!plan || !plan_t || !plan->block_ptr || !plan_t->block_ptr || plan->n_blocks <= 0 || plan_t->n_blocks <= 0 ||
      plan->seg_len > kBlkEdges || plan_t->seg_len > kBlkEdges)
    return STAG_ENOSYS;                                   // needs the batch plans of both orientations
  if (spec->in_norm && !norm_scale) return STAG_EINVAL;
  if (csr->n_dst == 0 || csr->n_edges == 0) return STAG_OK;   // the caller zero-fills (no edge, no gradient)
  if (!csr->indices || !csr_t->indices || !csr_t->nidx || !el || !er || !ft || !stats || !g || !out) return STAG_EINVAL;
  if (!aligned16(ft) || !aligned16(g) || !aligned16(out) || !aligned16(d_ft) || !aligned16(ade_ws)) return STAG_EINVAL;
  if (spec->kind == STAG_NOISE_EXPLICIT && !spec->p0 && csr->n_edges > 0) return STAG_EINVAL;
  if (spec->kind >= STAG_NOISE_NORMAL && (uint64_t)((uint64_t)spec->pos_base & 0xFFFFFFFFull) + (uint64_t)csr->n_edges > (1ull << 32))
    return STAG_ENOSYS;
  const int HF = (int)HF64;
  const size_t need = stag_gat_bwd_workspace_bytes(plan->n_seg, plan_t->n_seg, H, F);
  if (need > 0 && (!plan->workspace || plan->workspace_bytes < need)) return STAG_ENOMEM;
  hipStream_t s = (hipStream_t)stream;

  GatBwdBlkArgs ba{};
  GatArgs& a = ba.f;
  int rc = fill_edge_args(a, csr, plan, el, er, H, neg_slope, spec, norm_scale, stats);
  if (rc) return rc;
  a.ft = ft; a.F = F; a.HF = HF; a.lphp = lph;
  const uint64_t ftb = (uint64_t)csr->n_src * (uint64_t)HF * 4u;
  a.ft_bytes = (ftb < (1ull << 32) && csr->n_src < (1 << 24)) ? (uint32_t)ftb : 0u;
  a.block_ptr = plan->block_ptr;
  a.hvec = aligned16(el) && aligned16(er) && aligned16(stats) && (!a.nscale || aligned16(a.nscale));
  ba.g = g; ba.out = out; ba.ade = ade_ws; ba.d_er = d_er; ba.dw = dw; ba.ws = plan->workspace;
  const int nchunk = (HF + 3) / 4;
  int lpe = 4;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  const int cpl = nchunk <= 64 ? 1 : (nchunk <= 128 ? 2 : 4);
  size_t lds_e = (size_t)kBlkEdges * H * (dw ? 3 : 2) * sizeof(float) +
                 (size_t)(kBlkEdges + kBlkUnits + 4) * sizeof(int) + (size_t)kBlkUnits * sizeof(int4);
  if (lds_e < STAG_GAT_LDS_MIN_BWD) lds_e = STAG_GAT_LDS_MIN_BWD;
  const dim3 ge(plan->n_blocks);
#define STAG_BLK_LAUNCH(L, Cc) hipLaunchKernelGGL((gat_bwd_edge_block_kernel<L, Cc>), ge, dim3(kBlkThreads), lds_e, s, ba)
    if (cpl == 4) STAG_BLK_LAUNCH(64, 4);
    else if (cpl == 2) STAG_BLK_LAUNCH(64, 2);
    else switch (lpe) {
      case 64: STAG_BLK_LAUNCH(64, 1); break;
      case 32: STAG_BLK_LAUNCH(32, 1); break;
      case 16: STAG_BLK_LAUNCH(16, 1); break;
      case 8: STAG_BLK_LAUNCH(8, 1); break;
      default: STAG_BLK_LAUNCH(4, 1); break;
    }
#undef STAG_BLK_LAUNCH
  if (plan->n_long > 0)
    hipLaunchKernelGGL(gat_seg_finish_kernel, dim3(plan->n_long, (H + 15) / 16), dim3(256), 0, s,
                       plan->workspace, H, 0, H, plan->long_rows, plan->long_seg_ptr, d_er, H);

  GatSrcBlkArgs sa{};
  sa.indices = csr_t->indices; sa.nidx = csr_t->nidx;
  sa.units = plan_t->units; sa.block_ptr = plan_t->block_ptr; sa.long_rows = plan_t->long_rows;
  sa.ade = ade_ws; sa.g = g; sa.H = H; sa.F = F; sa.HF = HF;
  const uint64_t gb = (uint64_t)csr->n_dst * (uint64_t)HF * 4u;
  sa.g_bytes = (gb < (1ull << 32) && csr->n_dst < (1 << 24)) ? (uint32_t)gb : 0u;
  sa.d_ft = d_ft; sa.d_el = d_el; sa.ws = plan->workspace;   // the edge pass's partials are consumed by now (stream order)
  if (!plan_t->units || !aligned16(plan_t->units)) return STAG_EINVAL;
  if (plan_t->n_seg > 0 && (!plan_t->long_rows || !plan_t->long_seg_ptr)) return STAG_EINVAL;
  size_t lds_s = (size_t)kBlkEdges * H * 2 * sizeof(float) + (size_t)(kBlkEdges + kBlkUnits + 4) * sizeof(int) +
                 (size_t)kBlkUnits * sizeof(int4);
  if (lds_s < STAG_GAT_LDS_MIN_BWD) lds_s = STAG_GAT_LDS_MIN_BWD;
  const dim3 gs(plan_t->n_blocks);
#define STAG_BLK_LAUNCH(L, Cc) hipLaunchKernelGGL((gat_bwd_src_block_kernel<L, Cc>), gs, dim3(kBlkThreads), lds_s, s, sa)
    if (cpl == 4) STAG_BLK_LAUNCH(64, 4);
    else if (cpl == 2) STAG_BLK_LAUNCH(64, 2);
    else switch (lpe) {
      case 64: STAG_BLK_LAUNCH(64, 1); break;
      case 32: STAG_BLK_LAUNCH(32, 1); break;
      case 16: STAG_BLK_LAUNCH(16, 1); break;
      case 8: STAG_BLK_LAUNCH(8, 1); break;
      default: STAG_BLK_LAUNCH(4, 1); break;
    }
#undef STAG_BLK_LAUNCH
  if (plan_t->n_long > 0) {
    hipLaunchKernelGGL(gat_seg_finish_kernel, dim3(plan_t->n_long, (HF + 15) / 16), dim3(256), 0, s,
                       plan->workspace, HF + H, 0, HF, plan_t->long_rows, plan_t->long_seg_ptr, d_ft, HF);
    hipLaunchKernelGGL(gat_seg_finish_kernel, dim3(plan_t->n_long, (H + 15) / 16), dim3(256), 0, s,
                       plan->workspace, HF + H, HF, H, plan_t->long_rows, plan_t->long_seg_ptr, d_el, H);
  }
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

extern "C" size_t stag_gat_bwd_scratch_bytes(int64_t n_dst, int64_t n_edges, int32_t H) {
  if (n_dst < 0 || n_edges < 0 || H <= 0) return 0;
  // two-pass form: a and de by forward position [E, 2H]; one-gather form: dsl [E, H] then the per-destination
  // records [n_dst, 4H]
  return ((size_t)2 * (size_t)n_edges + (size_t)4 * (size_t)n_dst) * (size_t)H * sizeof(float);
}

// the fixed-order reduction of per-block partials [gx][2][D] -> dp0 [D], dp1 [D] (api.hip: stag_agg_bwd_dp's stages)
namespace stag { int dp_reduce_partials(const float* part, int64_t gx, int32_t D, float* part2, float* dp0, float* dp1, hipStream_t s); }

static int gat_bwd_impl(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                        const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                        const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                        float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                        const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* dw,
                        float* scratch, void* stream, float* dp0, float* dp1, float* dp_ws, int stages = 7);

extern "C" int stag_gat_bwd(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                            const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                            const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                            float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                            const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* dw,
                            float* scratch, void* stream) {
  return gat_bwd_impl(csr, plan, csr_t, plan_t, el, er, ft, stats, g, out, H, F, neg_slope, spec, norm_scale, drop,
                      d_el, d_er, d_ft, dw, scratch, stream, nullptr, nullptr, nullptr);
}

extern "C" int stag_gat_bwd_stages(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                                   const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                                   const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                                   float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                                   const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* scratch,
                                   int32_t stages, void* stream) {
  if (stages <= 0 || (stages & ~(STAG_GAT_BWD_ROWDOT | STAG_GAT_BWD_SOURCE | STAG_GAT_BWD_DER))) return STAG_EINVAL;
  return gat_bwd_impl(csr, plan, csr_t, plan_t, el, er, ft, stats, g, out, H, F, neg_slope, spec, norm_scale, drop,
                      d_el, d_er, d_ft, nullptr, scratch, stream, nullptr, nullptr, nullptr, stages);
}

extern "C" size_t stag_gat_bwd_dp_workspace_bytes(int32_t n_blocks_t, int32_t H) {
  if (n_blocks_t <= 0 || H <= 0) return 0;
  return ((size_t)n_blocks_t * 2u * (size_t)H + (size_t)256 * 2u * (size_t)H) * sizeof(float);
}

extern "C" int stag_gat_bwd_dp(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                               const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                               const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                               float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                               const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft,
                               float* dp0, float* dp1, float* scratch, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!spec || !dp0 || !dp1 || !plan_t) return STAG_EINVAL;
  if (spec->kind != STAG_NOISE_NORMAL && spec->kind != STAG_NOISE_UNIFORM) return STAG_EINVAL;
  if (spec->param_mode != STAG_PARAM_SCALAR && spec->param_mode != STAG_PARAM_PER_CHANNEL) return STAG_EINVAL;
  if (spec->in_norm) return STAG_EINVAL;          // the in-norm factor is not differentiated here
  if (csr && (csr->n_dst == 0 || csr->n_edges == 0)) {
    if (hipMemsetAsync(dp0, 0, sizeof(float) * H, (hipStream_t)stream) != hipSuccess) return STAG_EIO;
    if (hipMemsetAsync(dp1, 0, sizeof(float) * H, (hipStream_t)stream) != hipSuccess) return STAG_EIO;
    return STAG_OK;
  }
  if (!workspace || workspace_bytes < stag_gat_bwd_dp_workspace_bytes(plan_t->n_blocks, H)) return STAG_ENOMEM;
  return gat_bwd_impl(csr, plan, csr_t, plan_t, el, er, ft, stats, g, out, H, F, neg_slope, spec, norm_scale, drop,
                      d_el, d_er, d_ft, nullptr, scratch, stream, dp0, dp1, static_cast<float*>(workspace));
}

static int gat_bwd_impl(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                        const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                        const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                        float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                        const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* dw,
                        float* scratch, void* stream, float* dp0, float* dp1, float* dp_ws, int stages) {
  if (!csr || !csr_t || !csr->indptr || !csr_t->indptr || csr->n_dst < 0 || csr->n_edges < 0) return STAG_EINVAL;
  if (csr_t->n_edges != csr->n_edges || csr_t->n_dst != csr->n_src || csr_t->n_src != csr->n_dst) return STAG_EINVAL;
  if (!spec || spec->kind < STAG_NOISE_NONE || spec->kind > STAG_NOISE_BERNOULLI || spec->deriv) return STAG_EINVAL;
  if (!d_el || !d_er || !d_ft || !scratch || H <= 0 || F <= 0) return STAG_EINVAL;
  const int64_t HF64 = (int64_t)H * F;
  if (H > kBlkMaxH || HF64 > 1024 || F % 4 != 0) return STAG_ENOSYS;
  const int lphp = lanes_per_head(F);                   // any F % 4 == 0: lanes past a head's channels idle
  const int lph = lphp;
  if (lphp > 64 || H * lphp > 256) return STAG_ENOSYS;
  if (spec->chunk_base != 0) return STAG_ENOSYS;
  if (!plan_t || !plan_t->block_ptr || plan_t->n_blocks <= 0 || plan_t->seg_len > kBlkEdges) return STAG_ENOSYS;
  if (spec->in_norm && !norm_scale) return STAG_EINVAL;
  if (csr->n_dst == 0 || csr->n_edges == 0) return STAG_OK;   // the caller zero-fills (no edge, no gradient)
  if (!csr->indices || !csr_t->indices || !csr_t->nidx || !el || !er || !ft || !stats || !g || !out) return STAG_EINVAL;
  if (csr->eid && !csr_t->eid) return STAG_EINVAL;            // edge ids of the transposed positions
  if (!aligned16(ft) || !aligned16(g) || !aligned16(out) || !aligned16(d_ft) || !aligned16(scratch)) return STAG_EINVAL;
  if (spec->kind == STAG_NOISE_EXPLICIT && !spec->p0 && csr->n_edges > 0) return STAG_EINVAL;
  if (spec->kind >= STAG_NOISE_NORMAL && (uint64_t)((uint64_t)spec->pos_base & 0xFFFFFFFFull) + (uint64_t)csr->n_edges > (1ull << 32))
    return STAG_ENOSYS;
  if (!plan_t->units || !aligned16(plan_t->units)) return STAG_EINVAL;
  if (plan_t->n_seg > 0 && (!plan_t->long_rows || !plan_t->long_seg_ptr)) return STAG_EINVAL;
  const bool fplan = plan && plan->n_units > 0;
  if (fplan && (!plan->units || !aligned16(plan->units))) return STAG_EINVAL;
  if (fplan && plan->n_seg > 0 && (!plan->long_rows || !plan->long_seg_ptr)) return STAG_EINVAL;
  const int HF = (int)HF64;
  const size_t need = stag_gat_bwd_workspace_bytes(fplan ? plan->n_seg : 0, plan_t->n_seg, H, F);
  // segment partials of BOTH orientations live in the forward plan's workspace, one after the other
  if (need > 0 && (!plan || !plan->workspace || plan->workspace_bytes < need)) return STAG_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  float* dsl = scratch;
  float* pack = scratch + (size_t)csr->n_edges * H;        // 16-byte aligned whenever H % 4 == 0 (the vector form)

  const int nchunk = H * lphp;                          // lanes a row takes
  int lpe = 4;
  while (lpe < nchunk && lpe < 64) lpe <<= 1;
  const int cpl = nchunk <= 64 ? 1 : (nchunk <= 128 ? 2 : 4);

  // 1. sdot[v,h] = <G[v,h,:], out[v,h,:]>
  if (stages & STAG_GAT_BWD_ROWDOT) {
    const int rl = lpe < lph ? lph : lpe;                       // a head's lanes inside one team
    const int64_t rpb = (int64_t)(256 / rl) * kRowdotRows;
    const dim3 gr((unsigned)((csr->n_dst + rpb - 1) / rpb));
    switch (rl) {
      case 64: hipLaunchKernelGGL((gat_rowdot_kernel<64>), gr, dim3(256), 0, s, g, out, csr->n_dst, H, F, HF, lphp, er, stats, pack); break;
      case 32: hipLaunchKernelGGL((gat_rowdot_kernel<32>), gr, dim3(256), 0, s, g, out, csr->n_dst, H, F, HF, lphp, er, stats, pack); break;
      case 16: hipLaunchKernelGGL((gat_rowdot_kernel<16>), gr, dim3(256), 0, s, g, out, csr->n_dst, H, F, HF, lphp, er, stats, pack); break;
      case 8: hipLaunchKernelGGL((gat_rowdot_kernel<8>), gr, dim3(256), 0, s, g, out, csr->n_dst, H, F, HF, lphp, er, stats, pack); break;
      default: hipLaunchKernelGGL((gat_rowdot_kernel<4>), gr, dim3(256), 0, s, g, out, csr->n_dst, H, F, HF, lphp, er, stats, pack); break;
    }
  }

  // 2. the source-major pass
  if (stages & STAG_GAT_BWD_SOURCE) {
  GatBwd1Args ba{};
  GatArgs& a = ba.f;
  int rc = fill_edge_args(a, csr_t, plan_t, el, er, H, neg_slope, spec, norm_scale, stats);
  if (rc) return rc;
  a.ft = ft; a.F = F; a.HF = HF; a.lphp = lphp;
  a.block_ptr = plan_t->block_ptr;
  if (fill_drop(a, drop)) return STAG_EINVAL;
  a.hvec = aligned16(el) && aligned16(pack) && (!a.nscale || aligned16(a.nscale));
  ba.g = g; ba.pack = pack; ba.d_ft = d_ft; ba.d_el = d_el; ba.dsl = dsl; ba.dw = dw;
  ba.dp_part = dp_ws;
  ba.ws = plan ? plan->workspace : nullptr;
  ba.eid_is_pos = csr->eid ? 0 : 1;
  const uint64_t gb = (uint64_t)csr->n_dst * (uint64_t)HF * 4u;
  ba.g_bytes = (gb < (1ull << 32) && csr->n_dst < (1 << 24)) ? (uint32_t)gb : 0u;
  size_t lds_s = (size_t)kBlkEdges * H * ((dw || dp_ws) ? 4 : 3) * sizeof(float) + (size_t)(kBlkEdges + kBlkUnits + 4) * sizeof(int) +
                 (size_t)kBlkUnits * sizeof(int4);
  if (lds_s < STAG_GAT_LDS_MIN_ONE) lds_s = STAG_GAT_LDS_MIN_ONE;
  const dim3 gs(plan_t->n_blocks);
#define STAG_BLK_LAUNCH(L, Cc) hipLaunchKernelGGL((gat_bwd_one_kernel<L, Cc>), gs, dim3(kBlkThreads), lds_s, s, ba)
    if (cpl == 4) STAG_BLK_LAUNCH(64, 4);
    else if (cpl == 2) STAG_BLK_LAUNCH(64, 2);
    else switch (lpe) {
      case 64: STAG_BLK_LAUNCH(64, 1); break;
      case 32: STAG_BLK_LAUNCH(32, 1); break;
      case 16: STAG_BLK_LAUNCH(16, 1); break;
      case 8: STAG_BLK_LAUNCH(8, 1); break;
      default: STAG_BLK_LAUNCH(4, 1); break;
    }
#undef STAG_BLK_LAUNCH
  if (plan_t->n_long > 0) {
    hipLaunchKernelGGL(gat_seg_finish_kernel, dim3(plan_t->n_long, (HF + 15) / 16), dim3(256), 0, s,
                       plan->workspace, HF + H, 0, HF, plan_t->long_rows, plan_t->long_seg_ptr, d_ft, HF);
    hipLaunchKernelGGL(gat_seg_finish_kernel, dim3(plan_t->n_long, (H + 15) / 16), dim3(256), 0, s,
                       plan->workspace, HF + H, HF, H, plan_t->long_rows, plan_t->long_seg_ptr, d_el, H);
  }
  }

  // 3. d er: the in-edges of a row are contiguous in dsl (the source pass's partials are consumed by now)
  if (stages & STAG_GAT_BWD_DER) {
  const int n_units_f = fplan ? plan->n_units : csr->n_dst;
  const int64_t nth = (int64_t)n_units_f * H;
  hipLaunchKernelGGL(gat_der_kernel, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, s,
                     fplan ? static_cast<const stag_unit*>(plan->units) : nullptr, n_units_f, csr->indptr, dsl, H, d_er,
                     plan ? plan->workspace : nullptr);
  if (fplan && plan->n_long > 0)
    hipLaunchKernelGGL(gat_seg_finish_kernel, dim3(plan->n_long, (H + 15) / 16), dim3(256), 0, s,
                       plan->workspace, H, 0, H, plan->long_rows, plan->long_seg_ptr, d_er, H);
  }
  if (dp_ws) {
    const int rc2 = stag::dp_reduce_partials(dp_ws, plan_t->n_blocks, H, dp_ws + (size_t)plan_t->n_blocks * 2u * (size_t)H, dp0, dp1, s);
    if (rc2) return rc2;
  }
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}
