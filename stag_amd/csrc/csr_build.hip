// csr_build.hip — COO -> destination-major CSR on the device (graph preprocessing, not the hot
// path).  The order inside a row is ascending original edge id, exactly
// what the CPU oracle's counting sort produces, so the Philox positions agree.
//   1. stable LSD radix sort of (dst, edge id) pairs          rocPRIM radix_sort_pairs
//   2. indices[p] = src[eid[p]]                                gather kernel
//   3. in/out degree histograms (integer atomics), indptr = exclusive scan of in-degrees
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/stag_hip.h"

namespace {

__global__ void iota_kernel(int32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}

__global__ void zero_kernel(int32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

__global__ void gather_hist_kernel(const int32_t* src, const int32_t* dst, const int32_t* eid, int64_t E,
                                   int32_t* indices, int32_t* in_deg, int32_t* out_deg) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  const int32_t e = eid[p];
  const int32_t u = src[e];
  indices[p] = u;
  atomicAdd(in_deg + dst[e], 1);
  if (out_deg) atomicAdd(out_deg + u, 1);
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

unsigned key_bits(int32_t n) {
  unsigned b = 1;
  while (b < 31 && (1u << b) < (unsigned)n) ++b;
  return b;
}

}  // namespace

extern "C" size_t stag_csr_build_workspace_bytes(int32_t n_dst, int64_t E) {
  if (E <= 0 || n_dst <= 0) return 256;
  size_t sort_tmp = 0, scan_tmp = 0;
  int32_t* null = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, null, null, null, null, (size_t)E, 0u, key_bits(n_dst));
  (void)rocprim::exclusive_scan(nullptr, scan_tmp, null, null, 0, (size_t)n_dst + 1, rocprim::plus<int32_t>());
  return align_up((size_t)E * 4) * 2 + align_up(((size_t)n_dst + 1) * 4) + align_up(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
}

extern "C" int stag_csr_build(const int32_t* src, const int32_t* dst, int32_t n_src, int32_t n_dst,
                              int64_t E, int32_t* indptr, int32_t* indices, int32_t* eid,
                              int32_t* out_deg, void* workspace, size_t workspace_bytes, void* stream) {
  if (E < 0 || n_src < 0 || n_dst < 0 || !indptr || E > 0x7FFFFFFFll) return STAG_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n_dst == 0 && E > 0) return STAG_EINVAL;
  if (E == 0) {
    hipLaunchKernelGGL(zero_kernel, dim3((n_dst + 1 + 255) / 256), dim3(256), 0, s, indptr, (int64_t)n_dst + 1);
    if (out_deg && n_src > 0)
      hipLaunchKernelGGL(zero_kernel, dim3((n_src + 255) / 256), dim3(256), 0, s, out_deg, (int64_t)n_src);
    return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
  }
  if (!src || !dst || !indices || !eid || !workspace) return STAG_EINVAL;
  if (workspace_bytes < stag_csr_build_workspace_bytes(n_dst, E)) return STAG_ENOMEM;
  char* w = static_cast<char*>(workspace);
  int32_t* keys_out = reinterpret_cast<int32_t*>(w); w += align_up((size_t)E * 4);
  int32_t* iota = reinterpret_cast<int32_t*>(w);     w += align_up((size_t)E * 4);
  int32_t* deg = reinterpret_cast<int32_t*>(w);      w += align_up(((size_t)n_dst + 1) * 4);
  void* tmp = w;
  size_t tmp_bytes = workspace_bytes - (size_t)(w - static_cast<char*>(workspace));
  const unsigned eb = (unsigned)((E + 255) / 256);
  hipLaunchKernelGGL(iota_kernel, dim3(eb), dim3(256), 0, s, iota, E);
  hipLaunchKernelGGL(zero_kernel, dim3((n_dst + 1 + 255) / 256), dim3(256), 0, s, deg, (int64_t)n_dst + 1);
  if (out_deg)
    hipLaunchKernelGGL(zero_kernel, dim3((n_src + 255) / 256), dim3(256), 0, s, out_deg, (int64_t)n_src);
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, dst, keys_out, iota, eid, (size_t)E, 0u, key_bits(n_dst), s) != hipSuccess)
    return STAG_EIO;
  hipLaunchKernelGGL(gather_hist_kernel, dim3(eb), dim3(256), 0, s, src, dst, eid, E, indices, deg, out_deg);
  if (rocprim::exclusive_scan(tmp, tmp_bytes, deg, indptr, 0, (size_t)n_dst + 1, rocprim::plus<int32_t>(), s) != hipSuccess)
    return STAG_EIO;
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}
