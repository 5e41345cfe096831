// ubench_boxmuller.hip — where does the error of the hardware Box-Muller come from?
// z = sqrt(-2 ln u1) * cos|sin(2 pi u2) as noise.hpp computes it (v_log_f32, v_sqrt_f32, v_sin/v_cos_f32 in
// revolutions) against the same expression in fp64, over 2^24 Philox-drawn pairs; variants replace one
// hardware step at a time by its correctly rounded value, and try cheap refinements.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_boxmuller.hip -o tools/_bin/ubench_boxmuller
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../stag_amd/csrc/noise.hpp"
using namespace stag;

constexpr int NV = 8;
struct Stats { double sum2[NV]; double mx[NV]; };

__device__ inline void bm_variant(int v, uint32_t ra, uint32_t rb, float& za, float& zb) {
  const float f1 = f12(ra), th = f12(rb);
  const float u1 = 2.0f - f1;
  const double u1d = (double)u1, u2d = (double)th - 1.0;
  float lg = __builtin_amdgcn_logf(u1);                     // log2
  float c = __builtin_amdgcn_cosf(th), s = __builtin_amdgcn_sinf(th);
  float rad;
  switch (v) {
    case 0: break;                                           // the kernel's path
    case 1: lg = (float)log2(u1d); break;                    // exact log
    case 2: c = (float)cos(2.0 * M_PI * u2d); s = (float)sin(2.0 * M_PI * u2d); break;   // exact trig
    case 3: {                                                // exact sqrt argument chain (log + mul + sqrt in fp64)
      rad = (float)sqrt(-2.0 * log(u1d));
      za = rad * c; zb = rad * s; return;
    }
    case 4: {                                                // trig on the reduced angle: fold into [-1/8, 1/8] rev
      // cos/sin of t revolutions via the octant: hardware error is absolute, smaller arguments do not help
      // unless the unit is relative-accurate near 0 — measured here
      const float t = th - 1.0f;                             // [0,1)
      const float q = rintf(t * 4.0f);                       // nearest quarter
      const float r = t - q * 0.25f;                         // [-1/8, 1/8]
      const float cr = __builtin_amdgcn_cosf(r), sr = __builtin_amdgcn_sinf(r);
      const int qi = (int)q & 3;
      c = qi == 0 ? cr : qi == 1 ? -sr : qi == 2 ? -cr : sr;
      s = qi == 0 ? sr : qi == 1 ? cr : qi == 2 ? -sr : -cr;
      break;
    }
    case 5: {                                                // one Newton step on log2: y += (u - 2^y) / (u ln 2)
      const float e = __builtin_amdgcn_exp2f(lg);
      lg = lg + (u1 - e) * (1.4426950408889634f / u1);
      break;
    }
    case 6: {                                                // renormalise (c, s) to unit length: first-order fix of
      const float n2 = c * c + s * s;                        // the radial error only
      const float k = 1.5f - 0.5f * n2;
      c *= k; s *= k; break;
    }
    case 7: {                                                // exact everything but rounded to fp32 at the end: floor
      const double r = sqrt(-2.0 * log(u1d));
      za = (float)(r * cos(2.0 * M_PI * u2d)); zb = (float)(r * sin(2.0 * M_PI * u2d)); return;
    }
  }
  rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * lg);
  za = rad * c; zb = rad * s;
}

__global__ void kern(uint64_t seed, int64_t n, Stats* out) {
  __shared__ double s2[NV][256];
  __shared__ double smx[NV][256];
  double a2[NV] = {}, amx[NV] = {};
  PhiloxKey key{(uint32_t)seed, (uint32_t)(seed >> 32), 0, 0, nullptr};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox_at(i, 0, key, r);
    for (int pair = 0; pair < 2; ++pair) {
      const uint32_t ra = r[2 * pair], rb = r[2 * pair + 1];
      const double u1 = (double)(2.0f - f12(ra)), u2 = (double)f12(rb) - 1.0;
      const double R = sqrt(-2.0 * log(u1));
      const double ca = R * cos(2.0 * M_PI * u2), cb = R * sin(2.0 * M_PI * u2);
      for (int v = 0; v < NV; ++v) {
        float za, zb;
        bm_variant(v, ra, rb, za, zb);
        const double ea = fabs((double)za - ca), eb = fabs((double)zb - cb);
        a2[v] += ea * ea + eb * eb;
        amx[v] = fmax(amx[v], fmax(ea, eb));
      }
    }
  }
  for (int v = 0; v < NV; ++v) { s2[v][threadIdx.x] = a2[v]; smx[v][threadIdx.x] = amx[v]; }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int v = 0; v < NV; ++v) {
      double t = 0, m = 0;
      for (int j = 0; j < 256; ++j) { t += s2[v][j]; m = fmax(m, smx[v][j]); }
      out[blockIdx.x].sum2[v] = t; out[blockIdx.x].mx[v] = m;
    }
}

int main() {
  const int64_t n = 1 << 23;      // Philox blocks: 2^24 pairs, 2^25 normals
  const int blocks = 1024;
  Stats* d;
  hipMalloc(&d, blocks * sizeof(Stats));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, 0x5747A6ull, n, d);
  std::vector<Stats> h(blocks);
  hipMemcpy(h.data(), d, blocks * sizeof(Stats), hipMemcpyDeviceToHost);
  const char* names[NV] = {"hardware (kernel path)", "exact log2", "exact sin/cos", "exact radius (fp64 log+sqrt)",
                           "trig on octant-reduced angle", "log2 + one Newton step (v_exp)",
                           "(c,s) renormalised", "fp64 everything, rounded once (floor)"};
  for (int v = 0; v < NV; ++v) {
    double t = 0, m = 0;
    for (auto& s : h) { t += s.sum2[v]; m = fmax(m, s.mx[v]); }
    printf("%-40s rms %.3e  max %.3e\n", names[v], sqrt(t / (4.0 * n)), m);
  }
  return 0;
}
