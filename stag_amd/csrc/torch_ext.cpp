// torch_ext.cpp — PyTorch-ROCm front end of the hot calls: TORCH_LIBRARY ops `stag::agg_fwd` and
// `stag::agg_bwd` over the C ABI of include/stag_hip.h (the boundary stays that header; this file only
// marshals tensors into its structs, allocates outputs with the caching allocator and picks the current
// HIP stream).  Replaces the ctypes marshalling on the per-layer path (stag_amd/ops.py falls back to ctypes
// when this module is not built); the ops are visible to the dispatcher and carry Meta kernels, so a
// traced / compiled graph sees them as ordinary custom ops.
//
// Reference call these stand for: `base_layer.forward(graph=, feat=, edge_weight=)` -> DGL
// `update_all(u_mul_e, sum | mean)` (stag/layers.py:109-113, stag/zoo/gcn.py:94-96).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>    // PyTorch-ROCm: HIP devices answer to DeviceType::CUDA
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/stag_hip.h"

namespace {

using at::Tensor;
using OptT = const c10::optional<Tensor>&;

template <class T>
const T* ptr_of(OptT t) {
  return (t.has_value() && t->defined() && t->numel() > 0) ? static_cast<const T*>(t->data_ptr()) : nullptr;
}

void check_rc(int rc, const char* what) {
  TORCH_CHECK(rc == STAG_OK, what, ": ", stag_strerror(rc), " (rc=", rc, ")");
}

// graph = (indptr, indices, eid?, nidx?), n_src; plan = (units?, long_rows?, long_seg_ptr?, block_ptr?, xcd?,
// counters?) + plan_ints = [seg_len, n_units, n_long, n_seg, n_heavy, n_blocks, xcd_stride_heavy, xcd_stride_light]
struct Graph {
  stag_csr csr;
  stag_plan plan;
  bool has_plan;
};

Graph make_graph(const Tensor& indptr, const Tensor& indices, OptT eid, OptT nidx, int64_t n_src, OptT units,
                 OptT long_rows, OptT long_seg_ptr, OptT block_ptr, OptT xcd, OptT counters, at::IntArrayRef plan_ints) {
  TORCH_CHECK(indptr.is_cuda() && indptr.scalar_type() == at::kInt && indptr.is_contiguous(), "indptr: int32 on the device");
  TORCH_CHECK(indices.scalar_type() == at::kInt && indices.is_contiguous(), "indices: int32, contiguous");
  Graph g{};
  g.csr.n_dst = (int32_t)(indptr.numel() - 1);
  g.csr.n_src = (int32_t)n_src;
  g.csr.n_edges = indices.numel();
  g.csr.indptr = indptr.data_ptr<int32_t>();
  g.csr.indices = indices.numel() ? indices.data_ptr<int32_t>() : nullptr;
  g.csr.eid = ptr_of<int32_t>(eid);
  g.csr.nidx = ptr_of<int32_t>(nidx);
  g.has_plan = units.has_value() && units->defined() && plan_ints.size() == 8 && plan_ints[1] > 0;
  if (g.has_plan) {
    g.plan.seg_len = (int32_t)plan_ints[0];
    g.plan.n_units = (int32_t)plan_ints[1];
    g.plan.n_long = (int32_t)plan_ints[2];
    g.plan.n_seg = (int32_t)plan_ints[3];
    g.plan.n_heavy = (int32_t)plan_ints[4];
    g.plan.n_blocks = (int32_t)plan_ints[5];
    g.plan.units = static_cast<const stag_unit*>(units->data_ptr());
    g.plan.long_rows = ptr_of<int32_t>(long_rows);
    g.plan.long_seg_ptr = ptr_of<int32_t>(long_seg_ptr);
    g.plan.block_ptr = ptr_of<int32_t>(block_ptr);
    g.plan.xcd_order = ptr_of<int32_t>(xcd);
    g.plan.xcd_stride_heavy = (int32_t)plan_ints[6];
    g.plan.xcd_stride_light = (int32_t)plan_ints[7];
    g.plan.seg_counters = const_cast<int32_t*>(ptr_of<int32_t>(counters));
  }
  return g;
}

// noise_ints = [kind, param_mode, relu, in_norm, deriv, group, chunk_base, p1_log]; noise_u64 = [seed, offset, pos_base]
// (64-bit patterns carried in int64); noise_floats = [p0_scalar, p1_scalar]
stag_noise_spec make_spec(at::IntArrayRef ni, at::IntArrayRef nu, at::ArrayRef<double> nf, OptT p0, OptT p1, OptT epoch) {
  TORCH_CHECK(ni.size() == 8 && nu.size() == 3 && nf.size() == 2, "noise descriptor: 8 ints, 3 x 64 bit, 2 floats");
  stag_noise_spec s{};
  s.kind = (int32_t)ni[0]; s.param_mode = (int32_t)ni[1]; s.relu = (int32_t)ni[2]; s.in_norm = (int32_t)ni[3];
  s.deriv = (int32_t)ni[4]; s.group = (int32_t)ni[5]; s.chunk_base = (int32_t)ni[6]; s.p1_log = (int32_t)ni[7];
  s.seed = (uint64_t)nu[0]; s.offset = (uint64_t)nu[1]; s.pos_base = nu[2];
  s.p0_scalar = (float)nf[0]; s.p1_scalar = (float)nf[1];
  s.p0 = ptr_of<float>(p0); s.p1 = ptr_of<float>(p1);
  s.epoch = reinterpret_cast<const uint64_t*>(ptr_of<int64_t>(epoch));
  return s;
}

std::tuple<Tensor, Tensor> agg_fwd(const Tensor& indptr, const Tensor& indices, OptT eid, OptT nidx, int64_t n_src,
                                   OptT units, OptT long_rows, OptT long_seg_ptr, OptT block_ptr, OptT xcd, OptT counters,
                                   at::IntArrayRef plan_ints, const Tensor& x, bool broadcast_x, at::IntArrayRef noise_ints,
                                   at::IntArrayRef noise_u64, at::ArrayRef<double> noise_floats, OptT p0, OptT p1,
                                   OptT epoch, int64_t reduce, OptT src_scale, OptT dst_scale, bool want_norm_scale) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous(), "x: fp32, contiguous, on the device");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Graph g = make_graph(indptr, indices, eid, nidx, n_src, units, long_rows, long_seg_ptr, block_ptr, xcd, counters, plan_ints);
  const stag_noise_spec spec = make_spec(noise_ints, noise_u64, noise_floats, p0, p1, epoch);
  const int64_t D = broadcast_x ? x.numel() : x.size(1);
  Tensor out = at::empty({(int64_t)g.csr.n_dst, D}, x.options());
  Tensor ns = want_norm_scale ? at::empty({(int64_t)g.csr.n_dst, D}, x.options()) : Tensor();
  Tensor ws;
  if (g.has_plan) {
    const size_t nbytes = stag_plan_workspace_bytes(g.plan.n_seg, (int32_t)D, spec.in_norm);
    if (nbytes) {
      ws = at::empty({(int64_t)(nbytes / 4)}, x.options());
      g.plan.workspace = ws.data_ptr<float>();
      g.plan.workspace_bytes = nbytes;
    }
  }
  hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream();
  check_rc(stag_agg_fwd(&g.csr, g.has_plan ? &g.plan : nullptr, x.data_ptr<float>(), broadcast_x ? 0 : x.stride(0),
                        (int32_t)D, &spec, (int32_t)reduce, ptr_of<float>(src_scale), ptr_of<float>(dst_scale),
                        out.data_ptr<float>(), D, want_norm_scale ? ns.data_ptr<float>() : nullptr, stream),
           "stag_agg_fwd");
  return {out, want_norm_scale ? ns : at::empty({0}, x.options())};
}

std::tuple<Tensor, Tensor> agg_fwd_meta(const Tensor& indptr, const Tensor&, OptT, OptT, int64_t, OptT, OptT, OptT, OptT, OptT,
                                        OptT, at::IntArrayRef, const Tensor& x, bool broadcast_x, at::IntArrayRef, at::IntArrayRef,
                                        at::ArrayRef<double>, OptT, OptT, OptT, int64_t, OptT, OptT, bool want_norm_scale) {
  const int64_t D = broadcast_x ? x.numel() : x.size(1), n = indptr.numel() - 1;
  return {at::empty({n, D}, x.options()), at::empty({want_norm_scale ? n : 0, want_norm_scale ? D : 0}, x.options())};
}

// dx and (want_dp) the two parameter-derivative aggregates over the source-major CSR (stag_agg_bwd)
std::tuple<Tensor, Tensor, Tensor> agg_bwd(const Tensor& indptr, const Tensor& indices, OptT eid, OptT nidx, int64_t n_src,
                                           OptT units, OptT long_rows, OptT long_seg_ptr, OptT block_ptr, OptT xcd, OptT counters,
                                           at::IntArrayRef plan_ints, const Tensor& g_in, at::IntArrayRef noise_ints,
                                           at::IntArrayRef noise_u64, at::ArrayRef<double> noise_floats, OptT p0, OptT p1,
                                           OptT epoch, OptT g_scale, OptT row_scale, bool want_dp) {
  TORCH_CHECK(g_in.is_cuda() && g_in.scalar_type() == at::kFloat && g_in.is_contiguous(), "g: fp32, contiguous, on the device");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g_in.device());
  Graph g = make_graph(indptr, indices, eid, nidx, n_src, units, long_rows, long_seg_ptr, block_ptr, xcd, counters, plan_ints);
  const stag_noise_spec spec = make_spec(noise_ints, noise_u64, noise_floats, p0, p1, epoch);
  const int64_t D = g_in.size(1);
  Tensor dx = at::empty({(int64_t)g.csr.n_dst, D}, g_in.options());
  Tensor t0 = want_dp ? at::empty_like(dx) : at::empty({0}, g_in.options());
  Tensor t1 = want_dp ? at::empty_like(dx) : at::empty({0}, g_in.options());
  Tensor ws;
  if (g.has_plan) {
    const size_t nbytes = stag_plan_workspace_bytes(g.plan.n_seg, (int32_t)((want_dp ? 3 : 1) * D), 0);
    if (nbytes) {
      ws = at::empty({(int64_t)(nbytes / 4)}, g_in.options());
      g.plan.workspace = ws.data_ptr<float>();
      g.plan.workspace_bytes = nbytes;
    }
  }
  hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(g_in.device().index()).stream();
  check_rc(stag_agg_bwd(&g.csr, g.has_plan ? &g.plan : nullptr, g_in.data_ptr<float>(), g_in.stride(0), (int32_t)D, &spec,
                        ptr_of<float>(g_scale), ptr_of<float>(row_scale), dx.data_ptr<float>(),
                        want_dp ? t0.data_ptr<float>() : nullptr, want_dp ? t1.data_ptr<float>() : nullptr, D, stream),
           "stag_agg_bwd");
  return {dx, t0, t1};
}

std::tuple<Tensor, Tensor, Tensor> agg_bwd_meta(const Tensor& indptr, const Tensor&, OptT, OptT, int64_t, OptT, OptT, OptT, OptT,
                                                OptT, OptT, at::IntArrayRef, const Tensor& g_in, at::IntArrayRef, at::IntArrayRef,
                                                at::ArrayRef<double>, OptT, OptT, OptT, OptT, OptT, bool want_dp) {
  const int64_t D = g_in.size(1), n = indptr.numel() - 1;
  Tensor dx = at::empty({n, D}, g_in.options());
  return {dx, want_dp ? at::empty_like(dx) : at::empty({0}, g_in.options()),
          want_dp ? at::empty_like(dx) : at::empty({0}, g_in.options())};
}

}  // namespace

TORCH_LIBRARY(stag, m) {
  m.def("abi_version() -> int", []() -> int64_t { return stag_abi_version(); });
  m.def("agg_fwd(Tensor indptr, Tensor indices, Tensor? eid, Tensor? nidx, int n_src, Tensor? units, "
        "Tensor? long_rows, Tensor? long_seg_ptr, Tensor? block_ptr, Tensor? xcd, Tensor? counters, int[] plan_ints, Tensor x, "
        "bool broadcast_x, int[] noise_ints, int[] noise_u64, float[] noise_floats, Tensor? p0, Tensor? p1, "
        "Tensor? epoch, int reduce, Tensor? src_scale, Tensor? dst_scale, bool want_norm_scale) -> (Tensor, Tensor)");
  m.def("agg_bwd(Tensor indptr, Tensor indices, Tensor? eid, Tensor? nidx, int n_src, Tensor? units, "
        "Tensor? long_rows, Tensor? long_seg_ptr, Tensor? block_ptr, Tensor? xcd, Tensor? counters, int[] plan_ints, Tensor g, "
        "int[] noise_ints, int[] noise_u64, float[] noise_floats, Tensor? p0, Tensor? p1, Tensor? epoch, "
        "Tensor? g_scale, Tensor? row_scale, bool want_dp) -> (Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(stag, CUDA, m) {      // the HIP backend answers to the CUDA dispatch key in PyTorch-ROCm
  m.impl("agg_fwd", &agg_fwd);
  m.impl("agg_bwd", &agg_bwd);
}

TORCH_LIBRARY_IMPL(stag, Meta, m) {
  m.impl("agg_fwd", &agg_fwd_meta);
  m.impl("agg_bwd", &agg_bwd_meta);
}
