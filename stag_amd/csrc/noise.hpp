// noise.hpp — counter-based edge noise for gfx950 (device side).
//
// Normative definition of the stream: include/stag_hip.h ("Noise stream").
// It stands in for `q_a.expand([E, Dn]).sample()` (stag/layers.py:117-127):
// every (edge, channel) gets an independent draw, but from Philox4x32-10 keyed by
// (seed, offset, CSR position, channel/4) instead of torch's global generator, so
// the [E, Dn] tensor never exists and any shard of the graph can redraw it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stag {

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u;
constexpr uint32_t kPhiloxW1 = 0xBB67AE85u;

struct PhiloxKey {   // wave-uniform: lives in SGPRs
  uint32_t k0, k1;   // lo32(seed), hi32(seed)
  uint32_t o0, o1;   // lo32(offset), hi32(offset)
  const uint64_t* epoch;   // device counter added to the offset at run time, or null
};

// The offset a launch really uses: spec.offset + *spec.epoch.  The epoch lives in device
// memory so that a captured hipGraph draws fresh noise on every replay (the host-side offsets
// of its kernel nodes are frozen at capture; a node that bumps the epoch is part of the graph).
__device__ __forceinline__ PhiloxKey resolve_epoch(PhiloxKey k) {
  if (k.epoch) {
    const uint64_t o = (((uint64_t)k.o1 << 32) | k.o0) + *k.epoch;   // uniform: scalar load + add
    k.o0 = (uint32_t)o;
    k.o1 = (uint32_t)(o >> 32);
  }
  return k;
}

// the key of the stream `d` offsets further on
__device__ __forceinline__ PhiloxKey key_plus(PhiloxKey k, uint64_t d) {
  const uint64_t o = (((uint64_t)k.o1 << 32) | k.o0) + d;
  k.o0 = (uint32_t)o;
  k.o1 = (uint32_t)(o >> 32);
  return k;
}

// One Philox4x32-10 block. c0 = lo32(gpos), c1 = chunk | hi(gpos) << 20.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, const PhiloxKey& key,
                                              uint32_t (&r)[4]) {
  uint32_t c2 = key.o0, c3 = key.o1;
  uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
    const uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
    // one v_bitop3_b32 (truth table 0x96 = a ^ b ^ c) per output word; the key is an SGPR
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += kPhiloxW0;   // scalar ALU: the key schedule is wave-uniform
    k1 += kPhiloxW1;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__device__ __forceinline__ void philox_at(int64_t gpos, uint32_t chunk, const PhiloxKey& key,
                                          uint32_t (&r)[4]) {
  const uint32_t c0 = (uint32_t)((uint64_t)gpos & 0xFFFFFFFFull);
  const uint32_t c1 = chunk | ((uint32_t)((uint64_t)gpos >> 32) << 20);
  philox4x32_10(c0, c1, key, r);
}

__device__ __forceinline__ uint32_t ctr1_of(int64_t gpos, uint32_t chunk) {
  return chunk | ((uint32_t)((uint64_t)gpos >> 32) << 20);
}

// f in [1,2): the low 23 bits of r become the mantissa.  One v_and_or_b32 (1.0f is an
// inline constant) instead of shift + int->float convert (v_cvt_f32_u32 is half rate).
__device__ __forceinline__ float f12(uint32_t r) {
  return __uint_as_float((r & 0x007FFFFFu) | 0x3F800000u);
}

// u in [0,1): 23 random bits, exact in fp32.
__device__ __forceinline__ float u01(uint32_t r) { return f12(r) - 1.0f; }

// Box-Muller on the hardware transcendentals: v_log_f32 is log2 and v_sin/v_cos take
// revolutions (and are periodic in them), so neither ln, nor 2*pi, nor the "- 1" of the
// angle costs an instruction: u1 = 2 - f12(ra) in (0,1], angle = f12(rb) revolutions.
// The three functions of 23 random bits a normal draw is made of.  stag_normal_tables (api.hip)
// tabulates exactly these for all 2^23 inputs: the CPU oracle can then redraw the device's normals
// bit for bit, and the tables themselves are checked exhaustively against libm.
__device__ __forceinline__ float bm_radius(uint32_t ra) {
  const float u1 = 2.0f - f12(ra);
  // -2 ln(u1) = (-2 ln 2) * log2(u1)
  return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
}
__device__ __forceinline__ float bm_cos(uint32_t rb) { return __builtin_amdgcn_cosf(f12(rb)); }
__device__ __forceinline__ float bm_sin(uint32_t rb) { return __builtin_amdgcn_sinf(f12(rb)); }

__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& za, float& zb) {
  const float rad = bm_radius(ra);
  za = rad * bm_cos(rb);
  zb = rad * bm_sin(rb);
}

enum : int { kNone = 0, kExplicit = 1, kNormal = 2, kUniform = 3, kBernoulli = 4 };

// The 4 draws of one (edge, chunk): a[j], b[j] are the two distribution parameters
// of channel 4*chunk + j (loc/scale, low/high, probs/-).
// (c0, c1) = Philox counter words 0 and 1: lo32(position), chunk | hi32(position) << 20.
// flags (wave-uniform): bit 0 = relu; bits 1-2 = derivative selector for the backward pass of
// a reparameterised draw (stag/layers.py:123-124, `rsample`): 0 -> w itself,
// 1 -> dw/dp0 (loc | low), 2 -> dw/dp1 (scale | high), each times 1[w > 0] under relu.
// bit 3 = the scale came in as its logarithm (spec.p1_log; NORMAL): the callers hand draw4 the exponentiated
// value, and the derivative w.r.t. the parameter is then the one w.r.t. the LOG: dw/dlog_scale = z * scale.
enum : int { kFlagRelu = 1, kDerivShift = 1, kDerivMask = 3, kFlagLogScale = 8 };

__device__ __forceinline__ float exp_scale(float log_scale) { return __expf(log_scale); }

template <int KIND>
__device__ __forceinline__ void draw4(uint32_t c0, uint32_t c1, const PhiloxKey& key,
                                      const float (&a)[4], const float (&b)[4], int flags,
                                      float (&w)[4]) {
  static_assert(KIND >= kNormal, "draw4 is for sampled noise");
  uint32_t r[4];
  philox4x32_10(c0, c1, key, r);
  float t[4];   // the parameter-free draw: z ~ N(0,1) or u ~ U[0,1)
  if constexpr (KIND == kNormal) {
    box_muller(r[0], r[1], t[0], t[1]);
    box_muller(r[2], r[3], t[2], t[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __builtin_fmaf(b[j], t[j], a[j]);
  } else if constexpr (KIND == kUniform) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = u01(r[j]); w[j] = __builtin_fmaf(b[j] - a[j], t[j], a[j]); }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = u01(r[j]) < a[j] ? 1.0f : 0.0f;
  }
  const int deriv = (flags >> kDerivShift) & kDerivMask;
  if (deriv == 0) {
    if (flags & kFlagRelu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = fmaxf(w[j], 0.0f);
    }
    return;
  }
  if constexpr (KIND != kBernoulli) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mask = ((flags & kFlagRelu) && !(w[j] > 0.0f)) ? 0.0f : 1.0f;
      float d;
      if constexpr (KIND == kNormal)                                           // w = loc + scale z
        d = (deriv == 1) ? 1.0f : ((flags & kFlagLogScale) ? t[j] * b[j] : t[j]);
      else d = (deriv == 1) ? 1.0f - t[j] : t[j];                             // w = low + (high-low) u
      w[j] = d * mask;
    }
  }
}

// Backward of a reparameterised draw in ONE call: w (after relu) and both parameter derivatives
// d0 = dw/dp0 (loc | low), d1 = dw/dp1 (scale | high), each times 1[w > 0] under relu —
// exactly what draw4 returns for deriv = 0, 1, 2, from a single Philox block.
template <int KIND>
__device__ __forceinline__ void draw4_grad(uint32_t c0, uint32_t c1, const PhiloxKey& key,
                                           const float (&a)[4], const float (&b)[4], int flags,
                                           float (&w)[4], float (&d0)[4], float (&d1)[4]) {
  static_assert(KIND == kNormal || KIND == kUniform, "only reparameterised draws have derivatives");
  uint32_t r[4];
  philox4x32_10(c0, c1, key, r);
  float t[4];
  if constexpr (KIND == kNormal) {
    box_muller(r[0], r[1], t[0], t[1]);
    box_muller(r[2], r[3], t[2], t[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __builtin_fmaf(b[j], t[j], a[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = u01(r[j]); w[j] = __builtin_fmaf(b[j] - a[j], t[j], a[j]); }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float mask = ((flags & kFlagRelu) && !(w[j] > 0.0f)) ? 0.0f : 1.0f;
    if (flags & kFlagRelu) w[j] = fmaxf(w[j], 0.0f);
    d0[j] = ((KIND == kNormal) ? 1.0f : 1.0f - t[j]) * mask;
    d1[j] = (KIND == kNormal && (flags & kFlagLogScale)) ? t[j] * b[j] * mask : t[j] * mask;
  }
}

}  // namespace stag
