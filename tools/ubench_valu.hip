// ubench_valu.hip — instruction-throughput probes for the RNG inner loop on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu
// Each kernel runs ITER iterations of NOP independent ops per lane; reports cycles per
// wave-instruction per SIMD at full occupancy (8 waves/SIMD) and with 1 wave/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include "../stag_amd/csrc/noise.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;

template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed) {
  uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9E3779B9u, a2 = a0 + 77u, a3 = a1 + 1234567u;
  float f0 = (float)(a0 & 0xFFFF) * 1e-5f + 0.1f, f1 = f0 + 0.3f, f2 = f0 + 0.7f, f3 = f0 + 0.9f;
  uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
  for (int i = 0; i < ITER; ++i) {
    if constexpr (OP == 0) {   // v_mad_u64_u32 (full 64-bit product)
      q0 = (uint64_t)(uint32_t)q0 * 0xD2511F53u + (q0 >> 32);
      q1 = (uint64_t)(uint32_t)q1 * 0xCD9E8D57u + (q1 >> 32);
      q2 = (uint64_t)(uint32_t)q2 * 0xD2511F53u + (q2 >> 32);
      q3 = (uint64_t)(uint32_t)q3 * 0xCD9E8D57u + (q3 >> 32);
    } else if constexpr (OP == 1) {   // v_mul_hi_u32
      a0 = __umulhi(a0, 0xD2511F53u) + 1u; a1 = __umulhi(a1, 0xCD9E8D57u) + 1u;
      a2 = __umulhi(a2, 0xD2511F53u) + 1u; a3 = __umulhi(a3, 0xCD9E8D57u) + 1u;
    } else if constexpr (OP == 2) {   // v_mul_lo_u32
      a0 = a0 * 0xD2511F53u; a1 = a1 * 0xCD9E8D57u; a2 = a2 * 0xD2511F53u; a3 = a3 * 0xCD9E8D57u;
    } else if constexpr (OP == 3) {   // v_log_f32
      f0 = __builtin_amdgcn_logf(f0) ; f1 = __builtin_amdgcn_logf(f1); f2 = __builtin_amdgcn_logf(f2); f3 = __builtin_amdgcn_logf(f3);
      f0 = f0*f0+1.f; f1=f1*f1+1.f; f2=f2*f2+1.f; f3=f3*f3+1.f;
    } else if constexpr (OP == 4) {   // v_sin_f32
      f0 = __builtin_amdgcn_sinf(f0); f1 = __builtin_amdgcn_sinf(f1); f2 = __builtin_amdgcn_sinf(f2); f3 = __builtin_amdgcn_sinf(f3);
    } else if constexpr (OP == 5) {   // v_fma_f32
      f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
      f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
    } else if constexpr (OP == 6) {   // v_xor
      a0 ^= a1 + 1; a1 ^= a2; a2 ^= a3; a3 ^= a0;
    } else if constexpr (OP == 7) {   // one full philox4x32-10 block
      stag::PhiloxKey k{seed, 1u, 2u, 3u};
      uint32_t r[4];
      stag::philox4x32_10(a0, a1, k, r);
      a0 = r[0] ^ r[2]; a1 = r[1] ^ r[3];
    } else if constexpr (OP == 8) {   // philox + 4 normals (draw4)
      stag::PhiloxKey k{seed, 1u, 2u, 3u};
      float pa[4] = {1.f, 1.f, 1.f, 1.f}, pb[4] = {.5f, .5f, .5f, .5f}, w[4];
      stag::draw4<stag::kNormal>((int64_t)a0, a1 & 1023u, k, pa, pb, false, w);
      f0 += w[0]; f1 += w[1]; f2 += w[2]; f3 += w[3];
      a0 += 1;
    } else if constexpr (OP == 9) {   // philox + 4 uniforms
      stag::PhiloxKey k{seed, 1u, 2u, 3u};
      float pa[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {1.f, 1.f, 1.f, 1.f}, w[4];
      stag::draw4<stag::kUniform>((int64_t)a0, a1 & 1023u, k, pa, pb, false, w);
      f0 += w[0]; f1 += w[1]; f2 += w[2]; f3 += w[3];
      a0 += 1;
    } else if constexpr (OP == 10) {  // v_sqrt_f32
      f0 = __builtin_amdgcn_sqrtf(f0) + 1.f; f1 = __builtin_amdgcn_sqrtf(f1) + 1.f; f2 = __builtin_amdgcn_sqrtf(f2) + 1.f; f3 = __builtin_amdgcn_sqrtf(f3) + 1.f;
    } else if constexpr (OP == 12) {  // v_cvt_f32_u32
      f0 = (float)a0; a0 = __float_as_uint(f0) + 3u; f1 = (float)a1; a1 = __float_as_uint(f1) + 3u;
      f2 = (float)a2; a2 = __float_as_uint(f2) + 3u; f3 = (float)a3; a3 = __float_as_uint(f3) + 3u;
    } else if constexpr (OP == 13) {  // v_pk_fma_f32
      typedef float f2_t __attribute__((ext_vector_type(2)));
      f2_t x = {f0, f1}, y = {f2, f3}, k = {1.0001f, 0.9999f};
      x = __builtin_elementwise_fma(x, k, y); y = __builtin_elementwise_fma(y, k, x);
      x = __builtin_elementwise_fma(x, k, y); y = __builtin_elementwise_fma(y, k, x);
      f0 = x.x; f1 = x.y; f2 = y.x; f3 = y.y;
    } else if constexpr (OP == 14) {  // v_bitop3_b32
      a0 = __builtin_amdgcn_bitop3_b32(a0, a1, a2, 0x96); a1 = __builtin_amdgcn_bitop3_b32(a1, a2, a3, 0x96);
      a2 = __builtin_amdgcn_bitop3_b32(a2, a3, a0, 0x96); a3 = __builtin_amdgcn_bitop3_b32(a3, a0, a1, 0x96);
    } else if constexpr (OP == 15) {  // v_lshl_add_u64
      q0 = (q0 << 2) + q1; q1 = (q1 << 2) + q2; q2 = (q2 << 2) + q3; q3 = (q3 << 2) + q0;
    } else if constexpr (OP == 16) {  // ds_bpermute_b32
      a0 = __builtin_amdgcn_ds_bpermute(a1 & 252, a0); a1 = __builtin_amdgcn_ds_bpermute(a2 & 252, a1);
      a2 = __builtin_amdgcn_ds_bpermute(a3 & 252, a2); a3 = __builtin_amdgcn_ds_bpermute(a0 & 252, a3);
    } else if constexpr (OP == 17) {  // v_cndmask_b32 (vcc from v_cmp)
      a0 = (a1 > a2) ? a0 + 1 : a3; a1 = (a2 > a3) ? a1 + 1 : a0; a2 = (a3 > a0) ? a2 + 1 : a1; a3 = (a0 > a1) ? a3 + 1 : a2;
    } else if constexpr (OP == 18) {  // v_mul_lo_u32 (variable operands)
      a0 = a0 * a1 + 1u; a1 = a1 * a2 + 1u; a2 = a2 * a3 + 1u; a3 = a3 * a0 + 1u;
    } else if constexpr (OP == 11) {  // v_mul_u32_u24 pair
      a0 = __umul24(a0, a1) + 1; a1 = __umul24(a1, a2) + 1;
      a2 = __umul24(a2, a3) + 1; a3 = __umul24(a3, a0) + 1;
    }
  }
  uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)q0 ^ (uint32_t)q1 ^ (uint32_t)q2 ^ (uint32_t)q3 ^
               __float_as_uint(f0) ^ __float_as_uint(f1) ^ __float_as_uint(f2) ^ __float_as_uint(f3);
  if (r == 0x12345678u) out[0] = r;   // keep everything live
}

__global__ void accuracy(const uint32_t* in, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float za, zb;
  stag::box_muller(in[2 * i], in[2 * i + 1], za, zb);
  out[2 * i] = za; out[2 * i + 1] = zb;
}

template <int OP>
int run(const char* name, int ops_per_iter, uint32_t* d) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {8, 1}) {
    dim3 grid(256 * wps), block(256);   // 4 waves per block -> wps blocks per CU = wps waves per SIMD
    hipLaunchKernelGGL(probe<OP>, grid, block, 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<OP>, grid, block, 0, 0, d, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD = wps * ITER * ops_per_iter ; cycles at 2.4 GHz
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cyc per wave-op per SIMD (assuming 2.4 GHz)\n", name, wps, ms,
           cyc / ((double)wps * ITER * ops_per_iter));
  }
  return 0;
}

int main() {
  uint32_t* d; CHECK(hipMalloc(&d, 1024));
  run<5>("v_fma_f32", 4, d);
  run<6>("v_xor_b32(+add)", 4, d);
  run<0>("v_mad_u64_u32", 4, d);
  run<1>("v_mul_hi_u32(+add)", 4, d);
  run<2>("v_mul_lo_u32", 4, d);
  run<11>("v_mul_u32_u24(+add)", 4, d);
  run<3>("v_log_f32(+fma)", 4, d);
  run<4>("v_sin_f32", 4, d);
  run<10>("v_sqrt_f32(+add)", 4, d);
  run<12>("v_cvt_f32_u32(+add)", 4, d);
  run<13>("v_pk_fma_f32", 4, d);
  run<14>("v_bitop3_b32", 4, d);
  run<15>("v_lshl_add_u64", 4, d);
  run<16>("ds_bpermute_b32(+and)", 4, d);
  run<17>("cmp+add+cndmask", 4, d);
  run<18>("v_mul_lo_u32(+add)", 4, d);
  run<7>("philox4x32_10 block", 1, d);
  run<8>("philox + 4 normals", 1, d);
  run<9>("philox + 4 uniforms", 1, d);

  // accuracy of the hardware Box-Muller against double
  const int n = 1 << 20;
  std::vector<uint32_t> h(2 * n);
  uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
  h[0] = 0; h[1] = 0; h[2] = 0xFFFFFFFFu; h[3] = 0xFFFFFFFFu;   // extremes
  uint32_t* din; float* dout;
  CHECK(hipMalloc(&din, 8 * n)); CHECK(hipMalloc(&dout, 8 * n));
  CHECK(hipMemcpy(din, h.data(), 8 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(accuracy, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
  std::vector<float> z(2 * n);
  CHECK(hipMemcpy(z.data(), dout, 8 * n, hipMemcpyDeviceToHost));
  double maxabs = 0, maxrel = 0;
  for (int i = 0; i < n; ++i) {
    double u1 = ((double)(h[2 * i] >> 8) + 1.0) * 0x1p-24, u2 = (double)(h[2 * i + 1] >> 8) * 0x1p-24;
    double rad = sqrt(-2.0 * log(u1)), a = 6.283185307179586 * u2;
    double ea = fabs(z[2 * i] - rad * cos(a)), eb = fabs(z[2 * i + 1] - rad * sin(a));
    maxabs = fmax(maxabs, fmax(ea, eb));
  }
  printf("box_muller hw vs double: max abs err %.3e over %d pairs\n", maxabs, n);
  return 0;
}
