// csr_build.hip — COO -> destination-major CSR on the device (graph preprocessing, not the hot
// path).  The order inside a row is ascending original edge id, exactly
// what the CPU oracle's counting sort produces, so the Philox positions agree.
//   1. stable LSD radix sort of (dst, edge id) pairs          rocPRIM radix_sort_pairs
//   2. indices[p] = src[eid[p]]                                gather kernel
//   3. in/out degree histograms (integer atomics), indptr = exclusive scan of in-degrees
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
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

// ---- the launch plan on the device (stag_plan_device) -----------------------------------------------------
// The same plan stag_plan_count / stag_plan_fill build on the host, array for array: long rows (more than seg_len
// edges) first, most edges first, cut into balanced segments; then the whole rows, longest first; ties in row order.
// A minibatch graph is planned where it was built: no indptr read-back, no host loops, no uploads — one 16-byte
// read-back of the counts at the end.
//   1. key[r] = long ? (0x7FFFFFFF - deg) : (1 << 31) | (seg_len - deg);  stable radix sort of (key, row)
//   2. units per sorted row (segments | 1), exclusive scan -> first unit of every sorted row
//   3. fill: one thread per sorted row writes its unit records, long_rows, long_seg_ptr
namespace {

__global__ void plan_keys_kernel(const int32_t* indptr, int32_t n, int32_t seg_len, uint32_t* keys, int32_t* rows) {
  const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int32_t deg = indptr[r + 1] - indptr[r];
  keys[r] = deg > seg_len ? (0x7FFFFFFFu - (uint32_t)deg) : (0x80000000u | (uint32_t)(seg_len - deg));
  rows[r] = r;
}

// nu[i] = units of sorted row i; counts[0..2] += (long rows, segments, whole rows longer than STAG_HEAVY_LEN)
__global__ void plan_units_kernel(const int32_t* indptr, const int32_t* rows, int32_t n, int32_t seg_len,
                                  int32_t* nu, int32_t* counts) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t is_long = 0, nseg = 0, heavy = 0;
  if (i < n) {
    const int32_t v = rows[i];
    const int32_t deg = indptr[v + 1] - indptr[v];
    if (deg > seg_len) { is_long = 1; nseg = (deg + seg_len - 1) / seg_len; nu[i] = nseg; }
    else { nu[i] = 1; heavy = deg > STAG_HEAVY_LEN ? 1 : 0; }
  }
  // integer sums: any order gives the same counts
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    is_long += __shfl_xor(is_long, d); nseg += __shfl_xor(nseg, d); heavy += __shfl_xor(heavy, d);
  }
  if ((threadIdx.x & 63) == 0) {
    if (is_long) atomicAdd(counts + 0, is_long);
    if (nseg) atomicAdd(counts + 1, nseg);
    if (heavy) atomicAdd(counts + 2, heavy);
  }
}

__global__ void plan_fill_kernel(const int32_t* indptr, const int32_t* rows, const int32_t* uoff, int32_t n,
                                 int32_t seg_len, stag_unit* units, int32_t* long_rows, int32_t* long_seg_ptr,
                                 const int32_t* counts) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) long_seg_ptr[counts[0]] = counts[1];
  if (i >= n) return;
  const int32_t v = rows[i];
  const int32_t b = indptr[v], deg = indptr[v + 1] - b;
  const int32_t u0 = uoff[i];
  if (deg > seg_len) {                       // sorted position = index among the long rows (they come first)
    const int32_t nseg = (deg + seg_len - 1) / seg_len;
    const int32_t base = deg / nseg, rem = deg % nseg;      // balanced: lengths differ by at most 1
    long_rows[i] = v;
    long_seg_ptr[i] = u0;
    int32_t p = b;
    for (int32_t k = 0; k < nseg; ++k) {
      const int32_t l = base + (k < rem ? 1 : 0);
      units[u0 + k] = stag_unit{i, p, l, u0 + k};
      p += l;
    }
  } else {
    units[u0] = stag_unit{v, b, deg, -1};
  }
}

}  // namespace

extern "C" size_t stag_plan_device_workspace_bytes(int32_t n_dst) {
  if (n_dst <= 0) return 256;
  size_t sort_tmp = 0, scan_tmp = 0;
  uint32_t* ku = nullptr;
  int32_t* null = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, ku, ku, null, null, (size_t)n_dst, 0u, 32u);
  (void)rocprim::exclusive_scan(nullptr, scan_tmp, null, null, 0, (size_t)n_dst, rocprim::plus<int32_t>());
  // keys in/out, rows in/out, units per row, unit offsets, 4 counters, rocPRIM's temporary storage
  return align_up((size_t)n_dst * 4) * 6 + 256 + align_up(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
}

extern "C" int stag_plan_device(const int32_t* indptr, int32_t n_dst, int64_t n_edges, int32_t seg_len,
                                stag_unit* units, int64_t units_capacity, int32_t* long_rows, int32_t* long_seg_ptr,
                                int64_t long_capacity, int32_t* counts_out_host, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!indptr || n_dst < 0 || n_edges < 0 || seg_len <= 0 || seg_len > (1 << 20) || !counts_out_host) return STAG_EINVAL;
  counts_out_host[0] = counts_out_host[1] = counts_out_host[2] = counts_out_host[3] = 0;
  if (n_dst == 0) return STAG_OK;
  // upper bounds the caller sized the arrays with: a long row has more than seg_len edges
  const int64_t max_long = n_edges / ((int64_t)seg_len + 1);
  const int64_t max_units = (int64_t)n_dst + n_edges / seg_len + 1;
  if (!units || !long_rows || !long_seg_ptr || units_capacity < max_units || long_capacity < max_long + 1) return STAG_EINVAL;
  if (!workspace || workspace_bytes < stag_plan_device_workspace_bytes(n_dst)) return STAG_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  char* w = static_cast<char*>(workspace);
  const size_t slot = align_up((size_t)n_dst * 4);
  uint32_t* keys = reinterpret_cast<uint32_t*>(w);
  uint32_t* keys_s = reinterpret_cast<uint32_t*>(w + slot);
  int32_t* rows = reinterpret_cast<int32_t*>(w + 2 * slot);
  int32_t* rows_s = reinterpret_cast<int32_t*>(w + 3 * slot);
  int32_t* nu = reinterpret_cast<int32_t*>(w + 4 * slot);
  int32_t* uoff = reinterpret_cast<int32_t*>(w + 5 * slot);
  int32_t* counts = reinterpret_cast<int32_t*>(w + 6 * slot);
  void* tmp = w + 6 * slot + 256;
  size_t tmp_bytes = workspace_bytes - (6 * slot + 256);
  const dim3 grid((n_dst + 255) / 256), block(256);
  if (hipMemsetAsync(counts, 0, 16, s) != hipSuccess) return STAG_EIO;
  hipLaunchKernelGGL(plan_keys_kernel, grid, block, 0, s, indptr, n_dst, seg_len, keys, rows);
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_s, rows, rows_s, (size_t)n_dst, 0u, 32u, s) != hipSuccess)
    return STAG_EIO;
  hipLaunchKernelGGL(plan_units_kernel, grid, block, 0, s, indptr, rows_s, n_dst, seg_len, nu, counts);
  if (rocprim::exclusive_scan(tmp, tmp_bytes, nu, uoff, 0, (size_t)n_dst, rocprim::plus<int32_t>(), s) != hipSuccess)
    return STAG_EIO;
  hipLaunchKernelGGL(plan_fill_kernel, grid, block, 0, s, indptr, rows_s, uoff, n_dst, seg_len, units, long_rows,
                     long_seg_ptr, counts);
  int32_t c[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(c, counts, 16, hipMemcpyDeviceToHost, s) != hipSuccess) return STAG_EIO;
  if (hipStreamSynchronize(s) != hipSuccess) return STAG_EIO;
  const int64_t n_long = c[0], n_seg = c[1], n_heavy = (int64_t)c[1] + c[2];
  const int64_t n_units = (int64_t)n_dst - n_long + n_seg;
  if (n_units > 0x7FFFFFFFll) return STAG_EINVAL;
  counts_out_host[0] = (int32_t)n_units; counts_out_host[1] = (int32_t)n_long;
  counts_out_host[2] = (int32_t)n_seg; counts_out_host[3] = (int32_t)(n_heavy > 0x7FFFFFFFll ? 0x7FFFFFFFll : n_heavy);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

// ---- the XCD-aware order of a plan's units on the device (stag_plan_xcd_device_count / _fill) ---------------
// stag_plan_xcd's stable partition as a stable radix sort of (fine stripe key, unit index); the stripe starts by
// lower-bound search; then every record goes to (its XCD stripe's base) + (its rank inside that stripe) — the fine
// stripes of one XCD stripe are neighbours in the sorted order.
namespace {

constexpr int kXcdMaxKeys = 2 * STAG_XCD_STRIPES * STAG_XCD_FINE_MAX;      // 256

__global__ void xcd_keys_kernel(const stag_unit* units, int32_t n_units, int32_t n_heavy, int64_t E, int32_t S,
                                uint32_t* keys, int32_t* idx) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_units) return;
  int64_t k = (int64_t)units[i].start * S / E;
  k = k < 0 ? 0 : k >= S ? S - 1 : k;
  keys[i] = (uint32_t)((i >= n_heavy ? S : 0) + (int)k);
  idx[i] = i;
}

// the same with a range table (stag_plan_xcd_device_count_ranges): key of the range the unit's first edge lies in
__global__ void xcd_keys_ranges_kernel(const stag_unit* units, int32_t n_units, int32_t n_heavy, const int64_t* cuts,
                                       const int32_t* rkeys, int32_t R, int32_t S, uint32_t* keys, int32_t* idx) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_units) return;
  const int64_t start = units[i].start;
  int32_t lo = 0, hi = R;
  while (hi - lo > 1) {
    const int32_t mid = lo + (hi - lo) / 2;
    if (cuts[mid] <= start) lo = mid; else hi = mid;
  }
  int k = rkeys[lo];
  k = k < 0 ? 0 : k >= S ? S - 1 : k;
  keys[i] = (uint32_t)((i >= n_heavy ? S : 0) + k);
  idx[i] = i;
}

// starts[k] = first sorted position with key >= k, k in [0, 2 S]
__global__ void xcd_starts_kernel(const uint32_t* keys_s, int32_t n_units, int32_t S, int32_t* starts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > 2 * S) return;
  int32_t lo = 0, hi = n_units;
  while (lo < hi) {
    const int32_t mid = lo + (hi - lo) / 2;
    if (keys_s[mid] < (uint32_t)t) lo = mid + 1; else hi = mid;
  }
  starts[t] = lo;
}

__global__ void xcd_clear_kernel(int32_t* xcd, int64_t n_rec, const int32_t* starts, int32_t fine, int32_t sh, int32_t sl) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < STAG_XCD_HEADER) {
    int32_t v = 0;
    if (i < 2 * STAG_XCD_STRIPES) v = starts[(i + 1) * fine] - starts[i * fine];
    else if (i == 2 * STAG_XCD_STRIPES) v = sh;
    else if (i == 2 * STAG_XCD_STRIPES + 1) v = sl;
    else if (i == 2 * STAG_XCD_STRIPES + 2) v = fine;
    xcd[i] = v;
  }
  if (i < n_rec) reinterpret_cast<int4*>(xcd + STAG_XCD_HEADER)[i] = make_int4(-1, 0, 0, -1);
}

__global__ void xcd_fill_kernel(const stag_unit* units, const uint32_t* keys_s, const int32_t* idx_s, int32_t n_units,
                                const int32_t* starts, int32_t fine, int32_t sh, int32_t sl, int32_t* xcd) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_units) return;
  const int x = (int)keys_s[i] / fine;               // XCD stripe: heavy [0, 8), the others [8, 16)
  const int64_t base = x < STAG_XCD_STRIPES ? (int64_t)x * sh
                                            : (int64_t)STAG_XCD_STRIPES * sh + (int64_t)(x - STAG_XCD_STRIPES) * sl;
  reinterpret_cast<int4*>(xcd + STAG_XCD_HEADER)[base + (i - starts[x * fine])] = reinterpret_cast<const int4*>(units)[idx_s[i]];
}

struct XcdWs {
  uint32_t *keys, *keys_s;
  int32_t *idx, *idx_s, *starts;
  void* tmp;
  size_t tmp_bytes;
};
constexpr size_t kXcdStartsBytes = 2048;     // >= (kXcdMaxKeys + 1) ints
XcdWs xcd_ws(void* workspace, size_t workspace_bytes, int32_t n_units) {
  char* w = static_cast<char*>(workspace);
  const size_t slot = align_up((size_t)(n_units > 0 ? n_units : 1) * 4);
  XcdWs x;
  x.keys_s = reinterpret_cast<uint32_t*>(w);                 // what _fill reads comes first
  x.idx_s = reinterpret_cast<int32_t*>(w + slot);
  x.starts = reinterpret_cast<int32_t*>(w + 2 * slot);
  x.keys = reinterpret_cast<uint32_t*>(w + 2 * slot + kXcdStartsBytes);
  x.idx = reinterpret_cast<int32_t*>(w + 3 * slot + kXcdStartsBytes);
  x.tmp = w + 4 * slot + kXcdStartsBytes;
  x.tmp_bytes = workspace_bytes - (4 * slot + kXcdStartsBytes);
  return x;
}

}  // namespace

extern "C" size_t stag_plan_xcd_device_workspace_bytes(int32_t n_units) {
  if (n_units <= 0) n_units = 1;
  size_t sort_tmp = 0;
  uint32_t* ku = nullptr;
  int32_t* null = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, ku, ku, null, null, (size_t)n_units, 0u, 9u);
  return align_up((size_t)n_units * 4) * 4 + kXcdStartsBytes + align_up(sort_tmp);
}

static int xcd_device_count_impl(const stag_unit* units, int32_t n_units, int32_t n_heavy, int64_t n_edges,
                                 const int64_t* cuts, const int32_t* rkeys, int32_t n_ranges,
                                 int32_t fine, int32_t* strides_out_host, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (n_units < 0 || n_heavy < 0 || n_heavy > n_units || n_edges < 0 || !strides_out_host || fine < 1 ||
      fine > STAG_XCD_FINE_MAX || (n_units > 0 && !units) || ((uintptr_t)units & 15)) return STAG_EINVAL;
  if (!workspace || workspace_bytes < stag_plan_xcd_device_workspace_bytes(n_units)) return STAG_ENOMEM;
  strides_out_host[0] = strides_out_host[1] = 0;
  if (n_units == 0) return STAG_OK;
  hipStream_t s = (hipStream_t)stream;
  const XcdWs x = xcd_ws(workspace, workspace_bytes, n_units);
  const int S = STAG_XCD_STRIPES * fine;
  unsigned bits = 1;
  while ((1u << bits) < (unsigned)(2 * S)) ++bits;
  const dim3 grid((unsigned)((n_units + 255) / 256)), block(256);
  if (cuts)
    hipLaunchKernelGGL(xcd_keys_ranges_kernel, grid, block, 0, s, units, n_units, n_heavy, cuts, rkeys, n_ranges, S, x.keys, x.idx);
  else
    hipLaunchKernelGGL(xcd_keys_kernel, grid, block, 0, s, units, n_units, n_heavy, n_edges > 0 ? n_edges : (int64_t)1, S,
                       x.keys, x.idx);
  size_t tmp_bytes = x.tmp_bytes;
  if (rocprim::radix_sort_pairs(x.tmp, tmp_bytes, x.keys, x.keys_s, x.idx, x.idx_s, (size_t)n_units, 0u, bits, s) != hipSuccess)
    return STAG_EIO;
  hipLaunchKernelGGL(xcd_starts_kernel, dim3((2 * S + 1 + 63) / 64), dim3(64), 0, s, x.keys_s, n_units, S, x.starts);
  int32_t h[kXcdMaxKeys + 1];
  if (hipMemcpyAsync(h, x.starts, sizeof(int32_t) * (2 * S + 1), hipMemcpyDeviceToHost, s) != hipSuccess) return STAG_EIO;
  if (hipStreamSynchronize(s) != hipSuccess) return STAG_EIO;
  for (int k = 0; k < STAG_XCD_STRIPES; ++k) {
    strides_out_host[0] = std::max(strides_out_host[0], h[(k + 1) * fine] - h[k * fine]);
    strides_out_host[1] = std::max(strides_out_host[1], h[S + (k + 1) * fine] - h[S + k * fine]);
  }
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

extern "C" int stag_plan_xcd_device_count(const stag_unit* units, int32_t n_units, int32_t n_heavy, int64_t n_edges,
                                          int32_t fine, int32_t* strides_out_host, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  return xcd_device_count_impl(units, n_units, n_heavy, n_edges, nullptr, nullptr, 0, fine, strides_out_host, workspace,
                               workspace_bytes, stream);
}

extern "C" int stag_plan_xcd_device_count_ranges(const stag_unit* units, int32_t n_units, int32_t n_heavy,
                                                 const int64_t* cuts, const int32_t* keys, int32_t n_ranges, int32_t fine,
                                                 int32_t* strides_out_host, void* workspace, size_t workspace_bytes,
                                                 void* stream) {
  if (n_ranges < 1 || !cuts || !keys) return STAG_EINVAL;
  return xcd_device_count_impl(units, n_units, n_heavy, 0, cuts, keys, n_ranges, fine, strides_out_host, workspace,
                               workspace_bytes, stream);
}

extern "C" int stag_plan_xcd_device_fill(const stag_unit* units, int32_t n_units, const int32_t* strides, int32_t fine,
                                         int32_t* xcd, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_units <= 0 || !units || !strides || !xcd || ((uintptr_t)xcd & 15) || ((uintptr_t)units & 15) ||
      strides[0] < 0 || strides[1] < 0 || fine < 1 || fine > STAG_XCD_FINE_MAX) return STAG_EINVAL;
  if (!workspace || workspace_bytes < stag_plan_xcd_device_workspace_bytes(n_units)) return STAG_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  const XcdWs x = xcd_ws(workspace, workspace_bytes, n_units);
  const int64_t n_rec = (int64_t)STAG_XCD_STRIPES * ((int64_t)strides[0] + strides[1]);
  if (n_rec < n_units || n_rec > 0x7FFFFFFFll) return STAG_EINVAL;
  const int64_t n_clear = n_rec > STAG_XCD_HEADER ? n_rec : STAG_XCD_HEADER;
  hipLaunchKernelGGL(xcd_clear_kernel, dim3((unsigned)((n_clear + 255) / 256)), dim3(256), 0, s, xcd, n_rec, x.starts, fine,
                     strides[0], strides[1]);
  hipLaunchKernelGGL(xcd_fill_kernel, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, s, units, x.keys_s, x.idx_s,
                     n_units, x.starts, fine, strides[0], strides[1], xcd);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

// ---- stripe locality of a square CSR (what "auto" decides the XCD-aware order by) ---------------------------
// edges whose source row starts in the same eighth of the CSR as the edge itself lies in: integer count, any order
namespace {
__global__ void stripe_locality_kernel(const int32_t* indptr, const int32_t* indices, int32_t n_dst, int64_t E,
                                       unsigned long long* same) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int hit = 0;
  if (p < E) {
    const int32_t u = indices[p];
    if (u >= 0 && u < n_dst) {
      int64_t ks = (int64_t)indptr[u] * STAG_XCD_STRIPES / E;
      ks = ks >= STAG_XCD_STRIPES ? STAG_XCD_STRIPES - 1 : ks;
      hit = (p * STAG_XCD_STRIPES / E) == ks;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) hit += __shfl_xor(hit, d);
  if ((threadIdx.x & 63) == 0 && hit) atomicAdd(same, (unsigned long long)hit);
}
}  // namespace

extern "C" int stag_stripe_locality(const int32_t* indptr, const int32_t* indices, int32_t n_dst, int64_t n_edges,
                                    int64_t* same_out_host, void* workspace, void* stream) {
  if (!same_out_host || n_dst < 0 || n_edges < 0 || !workspace) return STAG_EINVAL;
  *same_out_host = 0;
  if (n_edges == 0 || n_dst == 0) return STAG_OK;
  if (!indptr || !indices) return STAG_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* same = static_cast<unsigned long long*>(workspace);
  if (hipMemsetAsync(same, 0, 8, s) != hipSuccess) return STAG_EIO;
  hipLaunchKernelGGL(stripe_locality_kernel, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, s, indptr, indices, n_dst,
                     n_edges, same);
  unsigned long long h = 0;
  if (hipMemcpyAsync(&h, same, 8, hipMemcpyDeviceToHost, s) != hipSuccess) return STAG_EIO;
  if (hipStreamSynchronize(s) != hipSuccess) return STAG_EIO;
  *same_out_host = (int64_t)h;
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}

// ---- laying int32 arrays end to end with offsets (stag_concat_jobs) -------------------------------------------
// A block-diagonal batch's CSR views are its parts' arrays one after the other, shifted by the nodes / edges before the
// part.  ONE launch copies every piece of every array: a table of jobs on the device, a block per 1024 elements of a job.
namespace {
__global__ void concat_jobs_kernel(const stag_concat_job* jobs, const int64_t* chunk_start, int32_t n_jobs) {
  // the job of this block: chunk_start[j] <= blockIdx.x < chunk_start[j + 1]   (block-uniform binary search)
  int lo = 0, hi = n_jobs;
  const int64_t b = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= b) lo = mid; else hi = mid;
  }
  const stag_concat_job j = jobs[lo];
  const int64_t first = (b - chunk_start[lo]) * 1024;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = first + k * 256 + threadIdx.x;
    if (i >= j.count) return;
    j.dst[i] = (j.kind == STAG_CONCAT_FILL ? 0 : j.src[i]) + j.add;
  }
}
}  // namespace

extern "C" int stag_concat_jobs(const stag_concat_job* jobs, const int64_t* chunk_start, int32_t n_jobs, int64_t n_chunks,
                                void* stream) {
  if (n_jobs < 0 || n_chunks < 0 || n_chunks > 0x7FFFFFFFll) return STAG_EINVAL;
  if (n_jobs == 0 || n_chunks == 0) return STAG_OK;
  if (!jobs || !chunk_start) return STAG_EINVAL;
  hipLaunchKernelGGL(concat_jobs_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, jobs, chunk_start, n_jobs);
  return hipGetLastError() == hipSuccess ? STAG_OK : STAG_EIO;
}
