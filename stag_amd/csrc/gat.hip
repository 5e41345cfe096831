// gat.hip — GAT edge attention with noisy logits, softmax and aggregation fused
// into one pass per destination row (stag/zoo/gat.py:114-126):
//     e[p,h]     = w[p,h] * leaky_relu(el[u_p,h] + er[v,h])
//     a[p,h]     = softmax_p(e[.,h])           (DGL edge_softmax: per dst, per head)
//     out[v,h,:] = sum_p a[p,h] * ft[u_p,h,:]
// Noise width is H (`sample_dimension`, stag/zoo/gat.py:11): one Philox block per
// (edge, 4 heads), so this path is gather-bound, not RNG-bound.
//
// One wave per destination row.  Per batch of 64 in-edges:
//   phase 1 (edge-parallel): lane i draws the H weights of edge i, forms the H
//           logits and parks them in LDS (no redundant RNG work across lanes);
//   phase 2 (channel-parallel): lane l owns channels [CPL*l, CPL*l+CPL) of head
//           h_l, walks the batch, and folds each edge into an online softmax
//           (running max m, running sum l, rescaled accumulator).
#include "../../include/stag_hip.h"
#include "noise.hpp"

using namespace stag;

namespace {

struct GatArgs {
  const int32_t* indptr;
  const int32_t* indices;
  const int32_t* eid;
  const int32_t* nidx;
  int32_t n_rows;
  const float* el;
  const float* er;
  const float* ft;
  int32_t H, F;
  float neg_slope;
  int32_t kind;
  const float* p0;
  const float* p1;
  float p0s, p1s;
  int32_t pmode, relu, in_norm;
  PhiloxKey key;
  int64_t pos_base;
  float* out;
  float* attn;
};

// the 4 weights of heads [4c, 4c+4) of the edge at position p
__device__ __forceinline__ void head_w4(const GatArgs& a, int p, int64_t ed, uint32_t chunk,
                                        float (&w)[4]) {
  const int h0 = (int)chunk * 4;
  float pa[4], pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int h = h0 + j;
    const bool in = h < a.H;
    float q0 = a.p0s, q1 = a.p1s;
    if (a.pmode == 1) { q0 = in ? a.p0[h] : 0.f; q1 = (in && a.p1) ? a.p1[h] : 0.f; }
    else if (a.pmode == 2) { q0 = a.p0[ed]; q1 = a.p1 ? a.p1[ed] : 0.f; }
    else if (a.pmode == 3) { q0 = in ? a.p0[ed * a.H + h] : 0.f; q1 = (in && a.p1) ? a.p1[ed * a.H + h] : 0.f; }
    pa[j] = q0; pb[j] = q1;
  }
  const int64_t gpos = a.nidx ? (int64_t)a.nidx[p] : a.pos_base + p;
  switch (a.kind) {
    case kNormal: draw4<kNormal>((uint32_t)gpos, ctr1_of(gpos, chunk), a.key, pa, pb, a.relu != 0, w); break;
    case kUniform: draw4<kUniform>((uint32_t)gpos, ctr1_of(gpos, chunk), a.key, pa, pb, a.relu != 0, w); break;
    case kBernoulli: draw4<kBernoulli>((uint32_t)gpos, ctr1_of(gpos, chunk), a.key, pa, pb, a.relu != 0, w); break;
    case kExplicit:
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = (h0 + j < a.H) ? a.p0[ed * a.H + h0 + j] : 0.f;
        w[j] = a.relu ? fmaxf(t, 0.f) : t;
      }
      break;
    default: w[0] = w[1] = w[2] = w[3] = 1.0f;
  }
}

// LDS hand-off between lanes of ONE wave: the LDS unit serves a wave's requests in
// order, so only the compiler has to be kept from reordering around the hand-off.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  return x;
}

// CPL channels per lane: 4 (dwordx4, needs F % 4 == 0) or 1
template <int CPL>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const GatArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int H = a.H, F = a.F, HF = H * F;
  float* logit = lds + (size_t)wave * (64 * H + 3 * H);   // [64][H]
  float* nscale = logit + 64 * H;                          // [H] in-norm factor
  float* stat_m = nscale + H;                              // [H]
  float* stat_l = stat_m + H;                              // [H]

  const int v = blockIdx.x * 4 + wave;
  if (v >= a.n_rows) return;   // whole wave leaves together: no block-level barrier is used
  const int b = a.indptr[v], e = a.indptr[v + 1];
  const int nchunk = (H + 3) / 4;

  // in-norm over the H-wide weights (stag/layers.py:8-36) needs the row sums first
  for (int h = lane; h < H; h += 64) nscale[h] = 1.0f;
  if (a.in_norm) {
    for (int c = 0; c < nchunk; ++c) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int p = b + lane; p < e; p += 64) {
        float w[4];
        head_w4(a, p, a.eid ? a.eid[p] : p, (uint32_t)c, w);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += w[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = wave_sum(s[j]);
        if (lane == 0 && 4 * c + j < H) nscale[4 * c + j] = (t != 0.f) ? (float)(e - b) / t : 1.f;
      }
    }
  }
  wave_sync();

  const int ntile = (HF + 64 * CPL - 1) / (64 * CPL);
  for (int tile = 0; tile < ntile; ++tile) {
    const int k0 = (tile * 64 + lane) * CPL;
    const bool kin = k0 < HF;
    const int hl = kin ? k0 / F : 0;
    float m = -INFINITY, l = 0.f;
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;

    for (int p0 = b; p0 < e; p0 += 64) {
      const int nb = min(64, e - p0);
      // ---- phase 1: logits of edge p0+lane into LDS -------------------------
      wave_sync();
      int u = 0;
      if (lane < nb) {
        const int p = p0 + lane;
        u = a.indices[p];
        const int64_t ed = a.eid ? a.eid[p] : p;
        for (int c = 0; c < nchunk; ++c) {
          float w[4];
          head_w4(a, p, ed, (uint32_t)c, w);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int h = 4 * c + j;
            if (h < H) {
              const float s = a.el[(int64_t)u * H + h] + a.er[(int64_t)v * H + h];
              const float lr = s > 0.f ? s : a.neg_slope * s;
              const float lg = (w[j] * nscale[h]) * lr;
              logit[lane * H + h] = lg;
              if (a.attn && tile == 0) a.attn[ed * H + h] = lg;   // normalised below
            }
          }
        }
      }
      wave_sync();
      // ---- phase 2: fold the batch into this lane's head ----------------------
      for (int i = 0; i < nb; ++i) {
        const int ui = __shfl(u, i);
        if (kin) {
          const float s = logit[i * H + hl];
          const float mn = fmaxf(m, s);
          const float corr = __expf(m - mn);
          const float pe = __expf(s - mn);
          l = l * corr + pe;
          m = mn;
          const float* fr = a.ft + (int64_t)ui * HF + k0;
          if constexpr (CPL == 4) {
            const float4 t = *reinterpret_cast<const float4*>(fr);
            acc[0] = acc[0] * corr + pe * t.x;
            acc[1] = acc[1] * corr + pe * t.y;
            acc[2] = acc[2] * corr + pe * t.z;
            acc[3] = acc[3] * corr + pe * t.w;
          } else {
            acc[0] = acc[0] * corr + pe * fr[0];
          }
        }
      }
    }
    if (kin) {
      const float inv = (l > 0.f) ? 1.0f / l : 0.f;
      float* o = a.out + (int64_t)v * HF + k0;
      if constexpr (CPL == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
      } else {
        o[0] = acc[0] * inv;
      }
      // the lane holding a head's first channel publishes its softmax statistics
      if (k0 % F == 0) { stat_m[hl] = m; stat_l[hl] = l; }
    }
  }
  // ---- attention values a[eid, h] (get_attention=True, stag/zoo/gat.py:146-147) --
  if (a.attn) {
    wave_sync();
    for (int p = b + lane; p < e; p += 64) {
      const int64_t ed = a.eid ? a.eid[p] : p;
      for (int h = 0; h < H; ++h) {
        const float lg = a.attn[ed * H + h];
        a.attn[ed * H + h] = __expf(lg - stat_m[h]) / stat_l[h];
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int stag_gat_fwd(const stag_csr* csr, const float* el, const float* er, const float* ft,
                            int32_t H, int32_t F, float neg_slope, const stag_noise_spec* spec,
                            float* out, float* attn_out, void* stream) {
  if (!csr || !csr->indptr || csr->n_dst < 0 || csr->n_edges < 0) return STAG_EINVAL;
  if (!spec || spec->kind < STAG_NOISE_NONE || spec->kind > STAG_NOISE_BERNOULLI) return STAG_EINVAL;
  if (!out || H <= 0 || F <= 0) return STAG_EINVAL;
  if (H > 64) return STAG_ENOSYS;   // LDS logit tile is [64][H] per wave
  if (csr->n_dst == 0) return STAG_OK;
  if (csr->n_edges > 0 && (!csr->indices || !el || !er || !ft)) return STAG_EINVAL;
  if (spec->kind == STAG_NOISE_EXPLICIT && !spec->p0) return STAG_EINVAL;
  if (spec->kind >= STAG_NOISE_NORMAL && spec->param_mode != STAG_PARAM_SCALAR &&
      (!spec->p0 || (spec->kind != STAG_NOISE_BERNOULLI && !spec->p1)))
    return STAG_EINVAL;

  GatArgs a{};
  a.indptr = csr->indptr; a.indices = csr->indices; a.eid = csr->eid; a.nidx = csr->nidx;
  a.n_rows = csr->n_dst; a.el = el; a.er = er; a.ft = ft; a.H = H; a.F = F;
  a.neg_slope = neg_slope; a.kind = spec->kind; a.p0 = spec->p0; a.p1 = spec->p1;
  a.p0s = spec->p0_scalar; a.p1s = spec->p1_scalar;
  a.pmode = spec->kind >= STAG_NOISE_NORMAL ? spec->param_mode : 0;
  a.relu = spec->relu; a.in_norm = spec->in_norm;
  a.key.k0 = (uint32_t)(spec->seed & 0xFFFFFFFFull); a.key.k1 = (uint32_t)(spec->seed >> 32);
  a.key.o0 = (uint32_t)(spec->offset & 0xFFFFFFFFull); a.key.o1 = (uint32_t)(spec->offset >> 32);
  a.pos_base = spec->pos_base; a.out = out; a.attn = attn_out;

  const size_t lds_bytes = 4u * (size_t)(64 * H + 3 * H) * sizeof(float);
  dim3 grid((csr->n_dst + 3) / 4);
  const bool vec = (F % 4 == 0) && aligned16(ft) && aligned16(out);
  if (vec) hipLaunchKernelGGL(gat_fwd_kernel<4>, grid, dim3(256), lds_bytes, (hipStream_t)stream, a);
  else     hipLaunchKernelGGL(gat_fwd_kernel<1>, grid, dim3(256), lds_bytes, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}
