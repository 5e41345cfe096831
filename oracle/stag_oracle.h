/*
 * stag_oracle.h — CPU twins (`*_cpu`) of the entry points in include/stag_hip.h.
 * TEST INFRASTRUCTURE ONLY (see stag_oracle.c). Same structs, HOST pointers,
 * no stream argument, no launch plan.
 */
#ifndef STAG_ORACLE_H
#define STAG_ORACLE_H
#include "../include/stag_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

void stag_philox4x32_10_cpu(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int stag_philox_raw_cpu(uint64_t seed, uint64_t offset, int64_t pos0, int64_t n_pos,
                        int32_t n_chunk, uint32_t* out);
int stag_noise_materialize_cpu(const stag_csr* csr, const stag_noise_spec* spec,
                               int32_t Dn, float* w_out, int64_t ldw);
int stag_agg_fwd_cpu(const stag_csr* csr, const float* x, int64_t ldx, int32_t D,
                     const stag_noise_spec* spec, int32_t reduce,
                     const float* src_scale, const float* dst_scale, float* out,
                     int64_t ldo, float* norm_scale_out);
int stag_agg_ref_dataflow_cpu(const stag_csr* csr, const int32_t* coo_src,
                              const int32_t* coo_dst, const float* x, int64_t ldx,
                              int32_t D, const stag_noise_spec* spec, float* w_buf,
                              float* m_buf, float* out, int64_t ldo);
int stag_agg_bwd_w_cpu(const stag_csr* csr, const float* x, int64_t ldx,
                       const float* g, int64_t ldg, int32_t D,
                       const float* src_scale, const stag_noise_spec* spec, int32_t reduce_k,
                       float* dw, int64_t ldw);
int stag_csr_build_cpu(const int32_t* src, const int32_t* dst, int32_t n_src,
                       int32_t n_dst, int64_t E, int32_t* indptr, int32_t* indices,
                       int32_t* eid, int32_t* in_deg, int32_t* out_deg);
int stag_segment_reduce_cpu(const float* x, int64_t ldx, int32_t D,
                            const int32_t* offsets, int32_t n_seg, int32_t reduce,
                            float* out, int64_t ldo);
int stag_gat_fwd_cpu(const stag_csr* csr, const float* el, const float* er,
                     const float* ft, int32_t H, int32_t F, float neg_slope,
                     const stag_noise_spec* spec, float* out, float* attn_out);
int stag_gat_fwd_drop_cpu(const stag_csr* csr, const float* el, const float* er,
                          const float* ft, int32_t H, int32_t F, float neg_slope,
                          const stag_noise_spec* spec, const float* keep, float keep_prob,
                          float* out, float* attn_out);

/* twin of stag_gat_bwd (and of stag_gat_bwd_two_pass): d el, d er, d ft and, for EXPLICIT weights, dw [E, H] —
 * what autograd returns for stag/zoo/gat.py:109-126; keep / keep_prob as in stag_gat_fwd_drop_cpu (NULL: no dropout) */
int stag_gat_bwd_cpu(const stag_csr* csr, const float* el, const float* er, const float* ft,
                     const float* gout, int32_t H, int32_t F, float neg_slope,
                     const stag_noise_spec* spec, const float* keep, float keep_prob,
                     float* d_el, float* d_er, float* d_ft, float* dw);

#ifdef __cplusplus
}
#endif
#endif
