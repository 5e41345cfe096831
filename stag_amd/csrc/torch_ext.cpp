// torch_ext.cpp — PyTorch-ROCm front end of the hot calls: TORCH_LIBRARY ops `stag::agg_fwd`, `stag::agg_bwd` and (round 4)
// `stag::agg_fwd_mc`, `stag::agg_bwd_dp`, `stag::gat_fwd`, `stag::gat_bwd` over the C ABI of include/stag_hip.h (the boundary stays that header; this file only
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

// ---- (round 4) the rest of the hot surface: Monte-Carlo batches, the one-pass parameter gradients, GAT ----------------

hipStream_t stream_of(const Tensor& t) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

void give_workspace(Graph& g, size_t nbytes, const Tensor& like, Tensor& keep) {
  if (g.has_plan && nbytes) {
    keep = at::empty({(int64_t)((nbytes + 3) / 4)}, like.options());
    g.plan.workspace = keep.data_ptr<float>();
    g.plan.workspace_bytes = nbytes;
  }
}

// [S, N, D]: S samples of the same gathered rows, sample s drawn at offset + s * offset_stride (stag_agg_fwd_mc;
// StagModel's Monte-Carlo loop, stag/models.py:45-55)
Tensor agg_fwd_mc(const Tensor& indptr, const Tensor& indices, OptT eid, OptT nidx, int64_t n_src, OptT units, OptT long_rows,
                  OptT long_seg_ptr, OptT block_ptr, OptT xcd, OptT counters, at::IntArrayRef plan_ints, const Tensor& x,
                  at::IntArrayRef noise_ints, at::IntArrayRef noise_u64, at::ArrayRef<double> noise_floats, OptT p0, OptT p1,
                  OptT epoch, int64_t n_samples, int64_t offset_stride, int64_t reduce, OptT src_scale, OptT dst_scale) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous() && x.dim() == 2, "x: [N, D] fp32, contiguous");
  TORCH_CHECK(n_samples >= 1, "n_samples >= 1");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Graph g = make_graph(indptr, indices, eid, nidx, n_src, units, long_rows, long_seg_ptr, block_ptr, xcd, counters, plan_ints);
  const stag_noise_spec spec = make_spec(noise_ints, noise_u64, noise_floats, p0, p1, epoch);
  const int64_t D = x.size(1), n = g.csr.n_dst;
  Tensor out = at::empty({n_samples, n, D}, x.options());
  Tensor ws;
  // a pass carries up to 4 samples (2 with in-norm: sums and weight sums): the segment partials of one pass
  give_workspace(g, g.has_plan ? stag_plan_workspace_bytes(g.plan.n_seg, (int32_t)(4 * D), 0) : 0, x, ws);
  check_rc(stag_agg_fwd_mc(&g.csr, g.has_plan ? &g.plan : nullptr, x.data_ptr<float>(), x.stride(0), (int32_t)D, &spec,
                           (int32_t)n_samples, offset_stride, (int32_t)reduce, ptr_of<float>(src_scale),
                           ptr_of<float>(dst_scale), out.data_ptr<float>(), D, n * D, stream_of(x)),
           "stag_agg_fwd_mc");
  return out;
}

Tensor agg_fwd_mc_meta(const Tensor& indptr, const Tensor&, OptT, OptT, int64_t, OptT, OptT, OptT, OptT, OptT, OptT,
                       at::IntArrayRef, const Tensor& x, at::IntArrayRef, at::IntArrayRef, at::ArrayRef<double>, OptT, OptT,
                       OptT, int64_t n_samples, int64_t, int64_t, OptT, OptT) {
  return at::empty({n_samples, indptr.numel() - 1, x.size(1)}, x.options());
}

// dx and the FINISHED gradients of scalar / per-channel parameters from one pass over the source-major CSR
// (stag_agg_bwd_dp; `vi=True`, stag/layers.py:123-124).  x: the units' own rows (None: ones in their place)
std::tuple<Tensor, Tensor, Tensor> agg_bwd_dp(const Tensor& indptr, const Tensor& indices, OptT eid, OptT nidx, int64_t n_src,
                                              OptT units, OptT long_rows, OptT long_seg_ptr, OptT block_ptr, OptT xcd,
                                              OptT counters, at::IntArrayRef plan_ints, const Tensor& g_in, OptT x,
                                              at::IntArrayRef noise_ints, at::IntArrayRef noise_u64,
                                              at::ArrayRef<double> noise_floats, OptT p0, OptT p1, OptT epoch, OptT g_scale,
                                              OptT row_scale, bool want_dx) {
  TORCH_CHECK(g_in.is_cuda() && g_in.scalar_type() == at::kFloat && g_in.is_contiguous() && g_in.dim() == 2, "g: [M, D] fp32");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g_in.device());
  Graph g = make_graph(indptr, indices, eid, nidx, n_src, units, long_rows, long_seg_ptr, block_ptr, xcd, counters, plan_ints);
  const stag_noise_spec spec = make_spec(noise_ints, noise_u64, noise_floats, p0, p1, epoch);
  const int64_t D = g_in.size(1), n = g.csr.n_dst;
  const bool has_x = x.has_value() && x->defined() && x->numel() > 0;
  if (has_x) TORCH_CHECK(x->scalar_type() == at::kFloat && x->stride(1) == 1 && x->size(0) == n && x->size(1) == D, "x: [n_dst, D] fp32");
  Tensor dx = want_dx ? at::empty({n, D}, g_in.options()) : at::empty({0}, g_in.options());
  Tensor dp0 = at::empty({D}, g_in.options()), dp1 = at::empty({D}, g_in.options());
  Tensor ws, ws2;
  give_workspace(g, g.has_plan ? stag_plan_workspace_bytes(g.plan.n_seg, (int32_t)D, 0) : 0, g_in, ws);
  const size_t wbytes = stag_agg_bwd_dp_workspace_bytes(g.has_plan ? g.plan.n_units : n, (int32_t)D);
  ws2 = at::empty({(int64_t)(wbytes / 4 + 1)}, g_in.options());
  check_rc(stag_agg_bwd_dp(&g.csr, g.has_plan ? &g.plan : nullptr, g_in.data_ptr<float>(), g_in.stride(0), (int32_t)D, &spec,
                           ptr_of<float>(g_scale), ptr_of<float>(row_scale), has_x ? x->data_ptr<float>() : nullptr,
                           has_x ? x->stride(0) : 0, want_dx ? dx.data_ptr<float>() : nullptr, D, dp0.data_ptr<float>(),
                           dp1.data_ptr<float>(), ws2.data_ptr<float>(), wbytes, stream_of(g_in)),
           "stag_agg_bwd_dp");
  return {dx, dp0, dp1};
}

std::tuple<Tensor, Tensor, Tensor> agg_bwd_dp_meta(const Tensor& indptr, const Tensor&, OptT, OptT, int64_t, OptT, OptT, OptT,
                                                   OptT, OptT, OptT, at::IntArrayRef, const Tensor& g_in, OptT,
                                                   at::IntArrayRef, at::IntArrayRef, at::ArrayRef<double>, OptT, OptT, OptT,
                                                   OptT, OptT, bool want_dx) {
  const int64_t D = g_in.size(1), n = indptr.numel() - 1;
  return {want_dx ? at::empty({n, D}, g_in.options()) : at::empty({0}, g_in.options()), at::empty({D}, g_in.options()),
          at::empty({D}, g_in.options())};
}

// attention dropout inside the GAT kernels: drop_floats = [keep_prob] (empty: none), drop_u64 = [seed, offset]
bool make_drop(stag_gat_drop& d, at::ArrayRef<double> drop_floats, at::IntArrayRef drop_u64, OptT drop_epoch) {
  if (drop_floats.empty()) return false;
  TORCH_CHECK(drop_floats.size() == 1 && drop_u64.size() == 2, "attention dropout: [keep_prob], [seed, offset]");
  d.keep_prob = (float)drop_floats[0];
  d.seed = (uint64_t)drop_u64[0]; d.offset = (uint64_t)drop_u64[1];
  d.epoch = reinterpret_cast<const uint64_t*>(ptr_of<int64_t>(drop_epoch));
  return true;
}

// out [M, H, F] and the softmax statistics [M, 2H] of the fused noisy-logit edge softmax + aggregation (stag_gat_fwd;
// stag/zoo/gat.py:109-126)
std::tuple<Tensor, Tensor> gat_fwd(const Tensor& indptr, const Tensor& indices, OptT eid, OptT nidx, int64_t n_src, OptT units,
                                   OptT long_rows, OptT long_seg_ptr, OptT block_ptr, OptT xcd, OptT counters,
                                   at::IntArrayRef plan_ints, const Tensor& el, const Tensor& er, const Tensor& ft,
                                   double neg_slope, at::IntArrayRef noise_ints, at::IntArrayRef noise_u64,
                                   at::ArrayRef<double> noise_floats, OptT p0, OptT p1, OptT epoch, OptT norm_scale,
                                   at::ArrayRef<double> drop_floats, at::IntArrayRef drop_u64, OptT drop_epoch, bool want_stats) {
  TORCH_CHECK(ft.is_cuda() && ft.scalar_type() == at::kFloat && ft.is_contiguous() && ft.dim() == 3, "ft: [N, H, F] fp32");
  TORCH_CHECK(el.is_contiguous() && er.is_contiguous() && el.scalar_type() == at::kFloat && er.scalar_type() == at::kFloat, "el, er: fp32, contiguous");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(ft.device());
  Graph g = make_graph(indptr, indices, eid, nidx, n_src, units, long_rows, long_seg_ptr, block_ptr, xcd, counters, plan_ints);
  const stag_noise_spec spec = make_spec(noise_ints, noise_u64, noise_floats, p0, p1, epoch);
  const int64_t H = ft.size(1), F = ft.size(2), n = g.csr.n_dst;
  stag_gat_drop drop{};
  const bool has_drop = make_drop(drop, drop_floats, drop_u64, drop_epoch);
  Tensor out = at::empty({n, H, F}, ft.options());
  Tensor stats = want_stats ? at::empty({n, 2 * H}, ft.options()) : at::empty({0}, ft.options());
  Tensor ws;
  give_workspace(g, g.has_plan ? stag_gat_workspace_bytes(g.plan.n_seg, (int32_t)H, (int32_t)F) : 0, ft, ws);
  check_rc(stag_gat_fwd(&g.csr, g.has_plan ? &g.plan : nullptr, el.data_ptr<float>(), er.data_ptr<float>(),
                        ft.data_ptr<float>(), (int32_t)H, (int32_t)F, (float)neg_slope, &spec, ptr_of<float>(norm_scale),
                        has_drop ? &drop : nullptr, out.data_ptr<float>(), want_stats ? stats.data_ptr<float>() : nullptr,
                        stream_of(ft)),
           "stag_gat_fwd");
  return {out, stats};
}

std::tuple<Tensor, Tensor> gat_fwd_meta(const Tensor& indptr, const Tensor&, OptT, OptT, int64_t, OptT, OptT, OptT, OptT, OptT,
                                        OptT, at::IntArrayRef, const Tensor&, const Tensor&, const Tensor& ft, double,
                                        at::IntArrayRef, at::IntArrayRef, at::ArrayRef<double>, OptT, OptT, OptT, OptT,
                                        at::ArrayRef<double>, at::IntArrayRef, OptT, bool want_stats) {
  const int64_t n = indptr.numel() - 1, H = ft.size(1), F = ft.size(2);
  return {at::empty({n, H, F}, ft.options()), want_stats ? at::empty({n, 2 * H}, ft.options()) : at::empty({0}, ft.options())};
}

// the whole backward with ONE gather of the [H*F] rows (stag_gat_bwd): d el [n_src, H], d er [n_dst, H], d ft [n_src, H, F]
// and (want_dw) dw [E, H] by edge id.  The second graph / plan is the source-major orientation (its block plan required).
std::tuple<Tensor, Tensor, Tensor, Tensor> gat_bwd(
    const Tensor& indptr, const Tensor& indices, OptT eid, OptT nidx, int64_t n_src, OptT units, OptT long_rows,
    OptT long_seg_ptr, OptT block_ptr, OptT xcd, OptT counters, at::IntArrayRef plan_ints, const Tensor& indptr_t,
    const Tensor& indices_t, OptT eid_t, OptT nidx_t, OptT units_t, OptT long_rows_t, OptT long_seg_ptr_t, OptT block_ptr_t,
    OptT counters_t, at::IntArrayRef plan_ints_t, const Tensor& el, const Tensor& er, const Tensor& ft, const Tensor& stats,
    const Tensor& grad, const Tensor& out, double neg_slope, at::IntArrayRef noise_ints, at::IntArrayRef noise_u64,
    at::ArrayRef<double> noise_floats, OptT p0, OptT p1, OptT epoch, OptT norm_scale, at::ArrayRef<double> drop_floats,
    at::IntArrayRef drop_u64, OptT drop_epoch, bool want_dw) {
  TORCH_CHECK(ft.is_cuda() && ft.is_contiguous() && ft.dim() == 3 && grad.is_contiguous() && out.is_contiguous() &&
              stats.is_contiguous(), "ft [N, H, F], grad, out, stats: fp32, contiguous, on the device");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(ft.device());
  Graph g = make_graph(indptr, indices, eid, nidx, n_src, units, long_rows, long_seg_ptr, block_ptr, xcd, counters, plan_ints);
  Graph t = make_graph(indptr_t, indices_t, eid_t, nidx_t, indptr.numel() - 1, units_t, long_rows_t, long_seg_ptr_t,
                       block_ptr_t, c10::nullopt, counters_t, plan_ints_t);
  TORCH_CHECK(t.has_plan, "stag_gat_bwd needs the block plan of the source-major orientation");
  const stag_noise_spec spec = make_spec(noise_ints, noise_u64, noise_floats, p0, p1, epoch);
  const int64_t H = ft.size(1), F = ft.size(2), E = g.csr.n_edges;
  stag_gat_drop drop{};
  const bool has_drop = make_drop(drop, drop_floats, drop_u64, drop_epoch);
  Tensor d_el = at::empty({(int64_t)t.csr.n_dst, H}, ft.options());
  Tensor d_er = at::empty({(int64_t)g.csr.n_dst, H}, ft.options());
  Tensor d_ft = at::empty({(int64_t)t.csr.n_dst, H, F}, ft.options());
  Tensor dw = want_dw ? at::empty({E, H}, ft.options()) : at::empty({0}, ft.options());
  Tensor scratch = at::empty({(int64_t)(stag_gat_bwd_scratch_bytes(g.csr.n_dst, E, (int32_t)H) / 4 + 1)}, ft.options());
  Tensor ws;
  stag_plan dummy{};
  if (!g.has_plan) g.plan = dummy;
  const size_t nbytes = stag_gat_bwd_workspace_bytes(g.has_plan ? g.plan.n_seg : 0, t.plan.n_seg, (int32_t)H, (int32_t)F);
  if (nbytes) {
    ws = at::empty({(int64_t)(nbytes / 4 + 1)}, ft.options());
    g.plan.workspace = ws.data_ptr<float>();
    g.plan.workspace_bytes = nbytes;
  }
  check_rc(stag_gat_bwd(&g.csr, &g.plan, &t.csr, &t.plan, el.data_ptr<float>(), er.data_ptr<float>(), ft.data_ptr<float>(),
                        stats.data_ptr<float>(), grad.data_ptr<float>(), out.data_ptr<float>(), (int32_t)H, (int32_t)F,
                        (float)neg_slope, &spec, ptr_of<float>(norm_scale), has_drop ? &drop : nullptr, d_el.data_ptr<float>(),
                        d_er.data_ptr<float>(), d_ft.data_ptr<float>(), want_dw ? dw.data_ptr<float>() : nullptr,
                        scratch.data_ptr<float>(), stream_of(ft)),
           "stag_gat_bwd");
  return {d_el, d_er, d_ft, dw};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> gat_bwd_meta(
    const Tensor& indptr, const Tensor& indices, OptT, OptT, int64_t, OptT, OptT, OptT, OptT, OptT, OptT, at::IntArrayRef,
    const Tensor& indptr_t, const Tensor&, OptT, OptT, OptT, OptT, OptT, OptT, OptT, at::IntArrayRef, const Tensor&,
    const Tensor&, const Tensor& ft, const Tensor&, const Tensor&, const Tensor&, double, at::IntArrayRef, at::IntArrayRef,
    at::ArrayRef<double>, OptT, OptT, OptT, OptT, at::ArrayRef<double>, at::IntArrayRef, OptT, bool want_dw) {
  const int64_t H = ft.size(1), F = ft.size(2), nd = indptr.numel() - 1, ns = indptr_t.numel() - 1;
  return {at::empty({ns, H}, ft.options()), at::empty({nd, H}, ft.options()), at::empty({ns, H, F}, ft.options()),
          want_dw ? at::empty({indices.numel(), H}, ft.options()) : at::empty({0}, ft.options())};
}

}  // namespace

#define STAG_GRAPH_ARGS "Tensor indptr, Tensor indices, Tensor? eid, Tensor? nidx, int n_src, Tensor? units, Tensor? long_rows, " \
                        "Tensor? long_seg_ptr, Tensor? block_ptr, Tensor? xcd, Tensor? counters, int[] plan_ints, "
#define STAG_NOISE_ARGS "int[] noise_ints, int[] noise_u64, float[] noise_floats, Tensor? p0, Tensor? p1, Tensor? epoch, "

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
  m.def("agg_fwd_mc(" STAG_GRAPH_ARGS "Tensor x, " STAG_NOISE_ARGS
        "int n_samples, int offset_stride, int reduce, Tensor? src_scale, Tensor? dst_scale) -> Tensor");
  m.def("agg_bwd_dp(" STAG_GRAPH_ARGS "Tensor g, Tensor? x, " STAG_NOISE_ARGS
        "Tensor? g_scale, Tensor? row_scale, bool want_dx) -> (Tensor, Tensor, Tensor)");
  m.def("gat_fwd(" STAG_GRAPH_ARGS "Tensor el, Tensor er, Tensor ft, float neg_slope, " STAG_NOISE_ARGS
        "Tensor? norm_scale, float[] drop_floats, int[] drop_u64, Tensor? drop_epoch, bool want_stats) -> (Tensor, Tensor)");
  m.def("gat_bwd(" STAG_GRAPH_ARGS "Tensor indptr_t, Tensor indices_t, Tensor? eid_t, Tensor? nidx_t, Tensor? units_t, "
        "Tensor? long_rows_t, Tensor? long_seg_ptr_t, Tensor? block_ptr_t, Tensor? counters_t, int[] plan_ints_t, "
        "Tensor el, Tensor er, Tensor ft, Tensor stats, Tensor grad, Tensor out, float neg_slope, " STAG_NOISE_ARGS
        "Tensor? norm_scale, float[] drop_floats, int[] drop_u64, Tensor? drop_epoch, bool want_dw) "
        "-> (Tensor, Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(stag, CUDA, m) {      // the HIP backend answers to the CUDA dispatch key in PyTorch-ROCm
  m.impl("agg_fwd", &agg_fwd);
  m.impl("agg_bwd", &agg_bwd);
  m.impl("agg_fwd_mc", &agg_fwd_mc);
  m.impl("agg_bwd_dp", &agg_bwd_dp);
  m.impl("gat_fwd", &gat_fwd);
  m.impl("gat_bwd", &gat_bwd);
}

TORCH_LIBRARY_IMPL(stag, Meta, m) {
  m.impl("agg_fwd", &agg_fwd_meta);
  m.impl("agg_bwd", &agg_bwd_meta);
  m.impl("agg_fwd_mc", &agg_fwd_mc_meta);
  m.impl("agg_bwd_dp", &agg_bwd_dp_meta);
  m.impl("gat_fwd", &gat_fwd_meta);
  m.impl("gat_bwd", &gat_bwd_meta);
}
