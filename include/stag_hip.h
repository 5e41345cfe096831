/*
 * stag_hip.h — C ABI of the MI355X-native stochastic-aggregation path.
 *
 * This library replaces exactly one thing in yuanqing-wang/stag: the work done
 * between `StagLayer.forward` and DGL's sparse kernels, i.e.
 *
 *     w   = q_a.expand([E, Dn]).sample()            stag/layers.py:115-129
 *     w   = relu(w) ; w = _in_norm(graph, w)        stag/layers.py:98-105, 8-36
 *     out = update_all(u_mul_e('h', w), sum|mean)   stag/zoo/gcn.py:94-96,
 *                                                   stag/zoo/graph_sage.py:71-73
 *
 * as ONE fused pass: the noise is never materialised, it is regenerated from a
 * counter-based Philox4x32-10 stream inside the gather/segmented-reduce kernel.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in
 *     `_host`; nothing is allocated, freed or kept by the library;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and the call returns without synchronising;
 *   - return value: 0 on success, a negative errno-style code otherwise
 *     (STAG_E*); no C++ exception crosses this boundary;
 *   - all feature matrices are row-major fp32, a row per node; `ld*` is the row
 *     stride in floats;
 *   - edge data handed over by the caller (explicit weights, per-edge
 *     parameters) is indexed by ORIGINAL edge id, as DGL edge frames are
 *     (`graph.edata[...]`, stag/zoo/gcn.py:61-63); `csr.eid` maps a CSR
 *     position to that id.
 *
 * Noise stream (normative; the CPU oracle in oracle/ restates it)
 *   One Philox4x32-10 call yields the noise of 4 consecutive channels of one
 *   edge:
 *       gpos = spec.pos_base + (csr.nidx ? csr.nidx[position] : position)
 *       ctr  = { lo32(gpos), chunk | (hi32(gpos) << 20), lo32(offset), hi32(offset) }
 *       key  = { lo32(seed), hi32(seed) }
 *       (r0,r1,r2,r3) = philox4x32_10(ctr, key)      channel k = 4*chunk + j uses r_j
 *   f12(r) = the fp32 in [1,2) with mantissa (r & 0x7FFFFF)   (23 random bits, exact)
 *   UNIFORM   u_j = f12(r_j) - 1 in [0,1)                    w = fma(high-low, u_j, low)
 *   BERNOULLI u_j as above                                   w = u_j < probs ? 1 : 0
 *   NORMAL    pairs (r0,r1) and (r2,r3): u1 = 2 - f12(r_a) in (0,1],
 *             u2 = f12(r_b) - 1 in [0,1), rad = sqrt(-2 ln u1),
 *             z_a = rad * cos(2 pi u2), z_b = rad * sin(2 pi u2)
 *                                                            w = fma(scale, z, loc)
 *   then optional relu, then optional in-degree renormalisation.
 */
#ifndef STAG_HIP_H
#define STAG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAG_ABI_VERSION 19

#define STAG_OK 0
#define STAG_EINVAL (-22)   /* bad argument (shape, enum, NULL where required) */
#define STAG_ENOMEM (-12)   /* workspace too small */
#define STAG_EIO (-5)       /* HIP runtime reported an error at launch */
#define STAG_ENOSYS (-38)   /* combination not implemented */

/* kind of per-edge multiplicative weight (stag/layers.py:56-64 `q_a`) */
enum {
  STAG_NOISE_NONE = 0,      /* w = 1  (base_layer.forward(edge_weight=None)) */
  STAG_NOISE_EXPLICIT = 1,  /* w given by caller: p0 = w[E, Dn], rows by edge id */
  STAG_NOISE_NORMAL = 2,    /* p0 = loc,  p1 = scale   */
  STAG_NOISE_UNIFORM = 3,   /* p0 = low,  p1 = high    */
  STAG_NOISE_BERNOULLI = 4  /* p0 = probs              */
};

/* how p0/p1 broadcast to [E, Dn] (`q_a.expand([E, Dn])`, stag/layers.py:117-119) */
enum {
  STAG_PARAM_SCALAR = 0,       /* p0_scalar / p1_scalar                          */
  STAG_PARAM_PER_CHANNEL = 1,  /* p0[Dn], p1[Dn]   (citation_rc: Normal(ones(D))) */
  STAG_PARAM_PER_EDGE1 = 2,    /* p0[E,1], p1[E,1] (AmortizedDistribution(.,1))   */
  STAG_PARAM_PER_EDGE = 3      /* p0[E,Dn], p1[E,Dn] (AmortizedDistribution(.,D)) */
};

enum { STAG_REDUCE_SUM = 0, STAG_REDUCE_MEAN = 1 };

/* Destination-major CSR of one (shard of a) graph. Replaces the DGL graph
 * handle that `update_all` receives (stag/zoo/gcn.py:94-96). */
typedef struct stag_csr {
  int32_t n_dst;          /* M: rows (destination nodes of this shard)              */
  int32_t n_src;          /* N: rows of the gathered matrix                          */
  int64_t n_edges;        /* E: edges of this shard                                  */
  const int32_t* indptr;  /* [M+1]                                                   */
  const int32_t* indices; /* [E] source row of each CSR position                     */
  const int32_t* eid;     /* [E] original edge id of each position; NULL = identity  */
  const int32_t* nidx;    /* [E] LOCAL noise index of each position, in [0, E): the backward
                             pass walks the transposed graph but must redraw the forward
                             pass's noise, so nidx = forward position of the same edge.
                             NULL = the position itself.  The global index of the draw is
                             spec.pos_base + nidx[position]; one call must not straddle a
                             2^32 boundary of the global index space (STAG_ENOSYS)         */
} stag_csr;

typedef struct stag_noise_spec {
  int32_t kind;       /* STAG_NOISE_*  */
  int32_t param_mode; /* STAG_PARAM_*  */
  const float* p0;
  const float* p1;
  float p0_scalar;
  float p1_scalar;
  int32_t relu;    /* w <- max(w, 0)                        stag/layers.py:98-99   */
  int32_t in_norm; /* w <- w * indeg / sum_in(w) per dst    stag/layers.py:8-36    */
  int32_t deriv;   /* 0: the weight w.  Backward of a reparameterised (`rsample`,
                      stag/layers.py:123-124) NORMAL / UNIFORM draw: 1: dw/dp0, 2: dw/dp1
                      (times 1[w > 0] under relu) — same counters, so the [E, Dn] noise is
                      regenerated, never stored.  Requires in_norm == 0.                  */
  int32_t group;   /* EXPLICIT only: channels sharing one weight column, p0 = w[E, D/group]
                      (GAT: a[e,h] over the F features of head h); 0 or 1 = one per channel */
  uint64_t seed;
  uint64_t offset;
  int64_t pos_base; /* global CSR position of this shard's position 0 (node-range shards) */
  int32_t chunk_base; /* first global channel / 4 of this shard's channel 0 (channel shards:
                         every GPU holds the whole CSR and D/P of the channels, no exchange) */
  int32_t p1_log;   /* NORMAL only, 0 | 1: p1 (p1_scalar) holds log(scale) — how a vi=True ParametrizedDistribution and an
                       AmortizedDistribution store it (stag/distributions.py:108-121, 235-242).  The kernels
                       exponentiate it where they load it, so the [E, Dn] exp pass and the tensor autograd would keep
                       for it never exist; every derivative w.r.t. p1 (deriv = 2, the dp1 / dw1 outputs) is then the
                       one w.r.t. the LOG: dw/dlog_scale = z * scale.  param_mode SCALAR, PER_EDGE1, PER_EDGE
                       (PER_CHANNEL: STAG_ENOSYS — a [Dn] row is exponentiated by the caller; it would cost the
                       hot kernel a wave per SIMD).                                                                */
  const uint64_t* epoch; /* NULL, or a DEVICE counter read at run time: the launch draws with
                            offset + *epoch.  Lets a captured hipGraph draw fresh noise on every
                            replay: its kernel nodes keep the offsets they were captured with, and
                            a node of the same graph advances the counter.                    */
} stag_noise_spec;

/* Launch plan, built once per graph on the host (stag_plan_count / stag_plan_fill).
 * The aggregation kernel walks UNITS: a unit is a whole destination row, or one
 * segment (<= seg_len edges, balanced) of a row longer than seg_len.  units[0, n_seg) are
 * the segments (slot == index, hub rows first); units[n_seg, n_units) the whole rows sorted
 * by length, longest first: lanes that share a wave then run equal trip counts and the
 * heavy units are dispatched first.  Segment sums go to `workspace`; the segment that
 * finishes last (an arrival counter per long row) adds them in segment order, in the
 * same launch, so the result does not depend on scheduling. */
typedef struct stag_unit {
  int32_t row;   /* slot < 0: destination row; slot >= 0: index into long_rows    */
  int32_t start; /* first CSR position                                           */
  int32_t len;   /* number of edges                                              */
  int32_t slot;  /* -1: the unit is the whole row; >= 0: segment id (workspace)  */
} stag_unit;

typedef struct stag_plan {
  int32_t seg_len;
  int32_t n_units;             /* rows not split + segments                        */
  int32_t n_long;              /* rows with in-degree > seg_len                    */
  int32_t n_seg;               /* segments over all long rows                      */
  const stag_unit* units;      /* [n_units], 16-byte aligned                       */
  const int32_t* long_rows;    /* [n_long]   row ids, largest in-degree first      */
  const int32_t* long_seg_ptr; /* [n_long+1] segment-id range of each long row     */
  int32_t* seg_counters;       /* [n_long * ceil(D / 256)] arrival counters: ZERO on
                                  entry, left zero by every completed call         */
  float* workspace;            /* >= stag_plan_workspace_bytes(); one call at a
                                  time may use a plan's workspace and counters     */
  size_t workspace_bytes;
  int32_t n_heavy;             /* units[0, n_heavy): all segments, then the whole rows longer
                                  than STAG_HEAVY_LEN edges (stag_plan_count reports it); the
                                  kernel spreads each of them over more lanes.  0 is valid.  */
  int32_t n_blocks;            /* 0: no block plan                                          */
  const int32_t* block_ptr;    /* [n_blocks+1] unit offsets (stag_plan_blocks): consecutive units
                                  batched per workgroup, at most STAG_BLOCK_EDGES edges and
                                  STAG_BLOCK_UNITS units each.  The GAT kernels draw the weights and
                                  form the logits of a whole batch edge-parallel, then gather.
                                  NULL: those kernels take one unit per team instead.        */
  const int32_t* xcd_order;    /* NULL, or the XCD-aware order of the same unit records (stag_plan_xcd): workgroups
                                  go to the chip's 8 XCDs round-robin and every XCD has its own 4 MB L2, so workgroup
                                  b takes its units from stripe b mod 8 — the destination rows that hold the
                                  (b mod 8)-th eighth of the edges.  A graph whose sources lie near its destinations (a
                                  block-diagonal batch: scripts/ppi_mle, scripts/molhiv_mle) then gathers from an eighth
                                  of the table per XCD.  16-byte aligned; a header of STAG_XCD_HEADER ints (units per
                                  heavy stripe [0, 8), per other stripe [8, 16), the two strides, fine), then 8 heavy stripes
                                  of xcd_stride_heavy records and 8 stripes of xcd_stride_light records, each padded
                                  with null records {-1, 0, 0, -1}.  Used by every aggregation launch that walks units
                                  (stag_agg_fwd with any parameter mode, stag_agg_fwd_mc, stag_agg_bwd, stag_agg_bwd_edge;
                                  v19: round 3 limited it to one output and scalar / per-channel parameters) except
                                  stag_agg_bwd_dp, whose block partials are added in plan order; every result is
                                  bit-identical with and without it.                                               */
  int32_t xcd_stride_heavy;    /* records per heavy stripe = units in the longest one; per other stripe (stag_plan_xcd) */
  int32_t xcd_stride_light;
} stag_plan;

int stag_abi_version(void);
const char* stag_strerror(int code);

/* ---- host-side planning (plain C++ on host arrays, no GPU call) ---------- */
#define STAG_HEAVY_LEN 16
int stag_plan_count(const int32_t* indptr_host, int32_t n_dst, int32_t seg_len,
                    int32_t* n_units_out, int32_t* n_long_out, int32_t* n_seg_out,
                    int32_t* n_heavy_out /* may be NULL */);
int stag_plan_fill(const int32_t* indptr_host, int32_t n_dst, int32_t seg_len,
                   stag_unit* units_host, int32_t* long_rows_host,
                   int32_t* long_seg_ptr_host);
size_t stag_plan_workspace_bytes(int32_t n_seg, int32_t D, int32_t in_norm);
/* Batches of consecutive units for the workgroup-cooperative kernels: block_ptr_host[n_blocks+1]
 * (NULL: count only), greedy in plan order: a batch closes before it would exceed max_edges edges
 * (a single longer unit gets a batch of its own) or max_units units.                          */
#ifndef STAG_BLOCK_EDGES
#define STAG_BLOCK_EDGES 256
#endif
#ifndef STAG_BLOCK_UNITS
#define STAG_BLOCK_UNITS 32
#endif
int stag_plan_blocks(const stag_unit* units_host, int32_t n_units, int32_t max_edges,
                     int32_t max_units, int32_t* block_ptr_host, int32_t* n_blocks_out);

/* The XCD-aware form of stag_plan_blocks for the workgroup-cooperative GAT kernels (which walk `units` through
 * `block_ptr`, one batch per workgroup): the unit records re-ordered (units_out_host[n_units]) and batched so that batch b
 * belongs to stripe b mod 8 of the destination rows (stag_plan.xcd_order's stripes, `fine` finer row ranges inside each,
 * batched one after the other); a stripe that has run out of batches gets empty ones (block_ptr[b] == block_ptr[b + 1]:
 * the kernels return at once).  Hand the two arrays to the GAT entry points as plan.units / plan.block_ptr /
 * plan.n_blocks (the other fields as they are): every output is the same — the forward bit for bit, the block partials
 * of stag_gat_bwd_dp added in the new batch order.  units_out_host == block_ptr_host == NULL: count only.
 * On the PPI-sized batch the forward takes 100 instead of 140 us (H*F = 256), 395 instead of 510 us (4 x 256).      */
/* (v19) A plan handed to stag_gat_fwd with these batches should also carry a non-NULL xcd_order (any device pointer: the GAT
 * kernels do not read it): it tells the forward that its rows come out of an XCD's L2, where two rows in flight per team
 * beat one (PPI batch, 4 x 256: 357 -> 334 us); on rows from the Infinity Cache one is best (cfg5: 222 against 238 us).  */
int stag_plan_blocks_xcd(const stag_unit* units_host, int32_t n_units, int64_t n_edges, int32_t fine, int32_t max_edges,
                         int32_t max_units, stag_unit* units_out_host, int32_t* block_ptr_host, int32_t* n_blocks_out);

/* The XCD-aware order of a plan's units (stag_plan.xcd_order), from host unit records.  Two calls: xcd_host == NULL
 * reports strides_out[2] = xcd_stride_heavy, xcd_stride_light; the second fills xcd_host[stag_plan_xcd_ints(strides)]
 * (upload it 16-byte aligned).  n_edges: the edges of the CSR the plan belongs to.  fine (1 ... STAG_XCD_FINE_MAX;
 * stag_plan_xcd_fine(n_dst) proposes one): inside its stripe an XCD walks `fine` finer row ranges one after the other —
 * about STAG_XCD_FINE_ROWS rows each, so that the rows one of them gathers fit the XCD's 4 MB L2 at D <= 256.
 * On the device, from device unit records: stag_plan_xcd_device_count (a stable radix sort of the stripe keys;
 * one read-back of the stripe sizes: it synchronises `stream`) leaves the sorted order in `workspace`
 * (>= stag_plan_xcd_device_workspace_bytes(n_units)) and reports the strides; stag_plan_xcd_device_fill, given the
 * same workspace, strides and fine, writes the array.  Both builders produce the same ints.                       */
#define STAG_XCD_STRIPES 8
#define STAG_XCD_HEADER 32
#define STAG_XCD_FINE_ROWS 2048
#define STAG_XCD_FINE_MAX 16
size_t stag_plan_xcd_ints(int32_t stride_heavy, int32_t stride_light);  /* STAG_XCD_HEADER + 4 * 8 * (the two strides) */
int32_t stag_plan_xcd_fine(int32_t n_dst);
int stag_plan_xcd(const stag_unit* units_host, int32_t n_units, int32_t n_heavy, int64_t n_edges, int32_t fine,
                  int32_t* xcd_host, int32_t* strides_out);
/* (v19) The same order with the stripes given as a RANGE TABLE instead of equal eighths of the CSR: cuts[n_ranges + 1]
 * ascending CSR positions, keys[n_ranges] in [0, 8 * fine) = stripe * fine + the stripe's fine range; a unit takes the key
 * of the range its first edge lies in.  For a block-diagonal batch (`dgl.batch`, scripts/ppi_mle/run.py:12-14) the ranges
 * are whole GRAPHS, bin-packed to the 8 stripes by edge count, a stripe's graphs packed into fine ranges that fit an
 * XCD's 4 MB L2 at the launch's row width (stag_amd/graph.py: CsrView.xcd_ranges) — a graph never straddles two XCDs, and
 * an XCD gathers from one L2-sized set of graphs at a time.  n_heavy = 0 puts every unit into the second (light) family of
 * stripes: one pass over each fine range instead of a heavy pass and a light pass (the wide shapes, which have no slotted
 * loop for heavy units).  Device form: cuts / keys are device arrays.  Results are bit-identical to every other order. */
int stag_plan_xcd_ranges(const stag_unit* units_host, int32_t n_units, int32_t n_heavy, const int64_t* cuts_host,
                         const int32_t* keys_host, int32_t n_ranges, int32_t fine, int32_t* xcd_host, int32_t* strides_out);
int stag_plan_blocks_xcd_ranges(const stag_unit* units_host, int32_t n_units, const int64_t* cuts_host,
                                const int32_t* keys_host, int32_t n_ranges, int32_t fine, int32_t max_edges, int32_t max_units,
                                stag_unit* units_out_host, int32_t* block_ptr_host, int32_t* n_blocks_out);
int stag_plan_xcd_device_count_ranges(const stag_unit* units, int32_t n_units, int32_t n_heavy, const int64_t* cuts,
                                      const int32_t* keys, int32_t n_ranges, int32_t fine, int32_t* strides_out_host,
                                      void* workspace, size_t workspace_bytes, void* stream);
size_t stag_plan_xcd_device_workspace_bytes(int32_t n_units);
int stag_plan_xcd_device_count(const stag_unit* units, int32_t n_units, int32_t n_heavy, int64_t n_edges, int32_t fine,
                               int32_t* strides_out_host, void* workspace, size_t workspace_bytes, void* stream);
int stag_plan_xcd_device_fill(const stag_unit* units, int32_t n_units, const int32_t* strides, int32_t fine, int32_t* xcd,
                              void* workspace, size_t workspace_bytes, void* stream);

/* int32 arrays laid end to end with offsets, every piece of every array in ONE launch: what a block-diagonal batch
 * (`dgl.batch`, scripts/ppi_mle/run.py:12-14) needs to take its CSR views from its parts' — row pointers shifted by the
 * edges before the part, column ids by its nodes, edge ids and forward positions by its edges (stag_amd/graph.py: batch).
 * `jobs` and `chunk_start` are DEVICE arrays: job j writes dst[i] = src[i] + add for i < count (FILL: dst[i] = add),
 * chunk_start[j] = sum over earlier jobs of ceil(count / 1024), chunk_start[n_jobs] = n_chunks.                        */
#define STAG_CONCAT_I32 0
#define STAG_CONCAT_FILL 2
typedef struct stag_concat_job {
  const int32_t* src;
  int32_t* dst;
  int64_t count;
  int32_t add;
  int32_t kind;
} stag_concat_job;
int stag_concat_jobs(const stag_concat_job* jobs, const int64_t* chunk_start, int32_t n_jobs, int64_t n_chunks, void* stream);

/* How many of a SQUARE CSR's edges (n_src == n_dst, device arrays) have their source row in the same eighth of the CSR —
 * the stripe of stag_plan_xcd — as the edge itself: what one XCD's L2 can hope to find again when it walks one stripe.
 * A block-diagonal batch: most of them; uniformly random sources: an eighth.  workspace: 8 bytes of device memory;
 * synchronises `stream` for the 8-byte read-back.                                                                  */
int stag_stripe_locality(const int32_t* indptr, const int32_t* indices, int32_t n_dst, int64_t n_edges,
                         int64_t* same_out_host, void* workspace, void* stream);

/* The same plan built ON THE DEVICE from a device indptr (rocPRIM sort + scan + one fill kernel), array for array
 * what stag_plan_count / stag_plan_fill produce on the host: a freshly batched minibatch graph (scripts/ppi_mle,
 * scripts/molhiv_mle build one per step) is planned where it was built — no indptr read-back, no host loops, no
 * uploads, ONE 16-byte read-back of the counts (the call synchronises `stream` for it).
 * units: device array of >= n_dst + n_edges / seg_len + 1 records; long_rows, long_seg_ptr: device arrays of
 * >= n_edges / (seg_len + 1) + 1 ints; counts_out_host[4] = n_units, n_long, n_seg, n_heavy;
 * workspace >= stag_plan_device_workspace_bytes(n_dst).  (stag_plan_blocks still takes host unit records.)   */
size_t stag_plan_device_workspace_bytes(int32_t n_dst);
int stag_plan_device(const int32_t* indptr, int32_t n_dst, int64_t n_edges, int32_t seg_len, stag_unit* units,
                     int64_t units_capacity, int32_t* long_rows, int32_t* long_seg_ptr, int64_t long_capacity,
                     int32_t* counts_out_host, void* workspace, size_t workspace_bytes, void* stream);

/* ---- graph preprocessing on the device: COO -> stable destination-major CSR ----------------
 * Position order inside a row = ascending original edge id (the order `graph.edata` frames
 * and the Philox positions are defined against).  indptr[n_dst+1], indices[E], eid[E];
 * out_deg[n_src] may be NULL; in-degrees are indptr differences.  Replaces the graph
 * construction the reference leaves to DGL (scripts/arxiv_mle/gcn/run.py:53-55).          */
size_t stag_csr_build_workspace_bytes(int32_t n_dst, int64_t n_edges);
int stag_csr_build(const int32_t* src, const int32_t* dst, int32_t n_src, int32_t n_dst,
                   int64_t n_edges, int32_t* indptr, int32_t* indices, int32_t* eid,
                   int32_t* out_deg, void* workspace, size_t workspace_bytes, void* stream);

/* ---- test hook: raw Philox words, out[n_pos][n_chunk][4] ------------------ */
int stag_philox_raw(uint64_t seed, uint64_t offset, int64_t pos0, int64_t n_pos,
                    int32_t n_chunk, uint32_t* out, void* stream);

/* ---- test hook: the three hardware functions a NORMAL draw is made of, over ALL 2^23 inputs ----
 *   rad[m] = sqrt(-2 ln(2 - f12(m)))   cosv[m] = cos(2 pi (f12(m) - 1))   sinv[m] = sin(...)
 * as the kernels evaluate them (v_log_f32 / v_sqrt_f32 / v_cos_f32 / v_sin_f32; a draw is
 * z_a = rad[r_a & 0x7FFFFF] * cosv[r_b & 0x7FFFFF], z_b = rad[..] * sinv[..], one fp32 multiply).
 * Each table is [2^23] floats (32 MB).  The CPU oracle loads them to redraw the device's normals bit
 * for bit; the tests also compare every entry with libm, so the approximations are pinned
 * exhaustively rather than by sampling.                                                       */
int stag_normal_tables(float* rad, float* cosv, float* sinv, void* stream);

/* ---- the hot path ---------------------------------------------------------
 * out[v, k] = dscale[v] * s[v, k] * sum_{p in row v} w[p, k] * sscale[u_p] * x[u_p, k]
 *   u_p     = csr.indices[p]
 *   w       = noise(spec) at (noise_index(p), k)           Dn == D
 *   s[v,k]  = in-norm factor (1 when spec.in_norm == 0)
 *   dscale  = dst_scale[v] (1 if NULL), times 1/max(indeg(v),1) for REDUCE_MEAN
 *   sscale  = src_scale[u] (1 if NULL)
 * Replaces: StagLayer.rsample_noise + relu + _in_norm (stag/layers.py:84-129)
 *           + GCN.forward's degree scaling and update_all (stag/zoo/gcn.py:67-75,
 *           94-96, 100-108) / GraphSAGE mean (stag/zoo/graph_sage.py:70-73).
 * `norm_scale_out` (may be NULL; written only when spec.in_norm): s[M, D], kept for
 * the backward pass.  `plan` may be NULL: one unit per row, in row order, no split.
 * ldx == 0 gathers one broadcast row (sum of edge data, stag/layers.py:12-15).   */
int stag_agg_fwd(const stag_csr* csr, const stag_plan* plan, const float* x,
                 int64_t ldx, int32_t D, const stag_noise_spec* spec,
                 int32_t reduce, const float* src_scale, const float* dst_scale,
                 float* out, int64_t ldo, float* norm_scale_out, void* stream);

/* n_samples Monte-Carlo draws of the same aggregation from ONE pass over the gathered rows
 * (up to 4 samples per launch): sample s is stag_agg_fwd with offset + s * offset_stride, bit for
 * bit, written to out + s * sample_stride (floats).  The n_samples loop of the reference
 * (stag/models.py:45-55, 67-68) re-runs the whole forward per sample; on the first layer, whose
 * input is the same for every sample, the gather is shared and only the draws repeat.
 * kind NORMAL | UNIFORM | BERNOULLI, param_mode SCALAR | PER_CHANNEL; the plan's workspace must hold
 * stag_plan_workspace_bytes(n_seg, 4 * D, 0) bytes.  (v17) spec.in_norm is allowed (`norm=True`,
 * stag/layers.py:8-36; scripts/arxiv_mle/gcn/run.py:70-74): every sample then carries its own
 * per-destination weight sums, two samples per pass (a segment's workspace row is
 * [sum_0 | sum_1 | wsum_0 | wsum_1]: the same 4 * D floats).                                     */
int stag_agg_fwd_mc(const stag_csr* csr, const stag_plan* plan, const float* x, int64_t ldx,
                    int32_t D, const stag_noise_spec* spec, int32_t n_samples,
                    int64_t offset_stride, int32_t reduce, const float* src_scale,
                    const float* dst_scale, float* out, int64_t ldo, int64_t sample_stride,
                    void* stream);

/* Backward of stag_agg_fwd with respect to x and to the noise parameters, ONE pass over the
 * source-major CSR (csr_t: rows = source nodes, indices = destination rows, nidx = the forward
 * CSR position of each edge, so the forward's noise is redrawn from its counters):
 *     dx[u,:]       = row_scale[u] * sum_{p: src_p = u} w[p,:]        * g_scale[v_p] * g[v_p,:]
 *     dp0_rows[u,:] = row_scale[u] * sum_{p: src_p = u} dw/dp0[p,:]   * g_scale[v_p] * g[v_p,:]
 *     dp1_rows[u,:] = ...          dw/dp1 ...
 * (dw/dp = 1 | z for Normal, 1-u | u for Uniform, times 1[w > 0] under relu; the reference gets
 * them from autograd through `rsample`, stag/layers.py:123-124.)  The gradient of a per-channel
 * parameter is then  dp[k] = sum_u x[u,k] * dp_rows[u,k]  (a column sum the caller does); for a
 * scalar parameter sum over k as well.  dp0_rows == dp1_rows == NULL: dx only (= stag_agg_fwd on
 * csr_t).  With the dp outputs: kind NORMAL | UNIFORM, param_mode SCALAR | PER_CHANNEL, in_norm 0;
 * the plan's workspace must hold stag_plan_workspace_bytes(n_seg, 3 * D, 0) bytes.
 * Replaces the three autograd passes over [E, D] tensors of the reference's backward.        */
int stag_agg_bwd(const stag_csr* csr_t, const stag_plan* plan_t, const float* g, int64_t ldg,
                 int32_t D, const stag_noise_spec* spec, const float* g_scale,
                 const float* row_scale, float* dx, float* dp0_rows, float* dp1_rows,
                 int64_t ldo, void* stream);

/* stag_agg_bwd with the parameter gradients FINISHED in the same pass (scalar | per-channel parameters, kind
 * NORMAL | UNIFORM, in_norm 0): on the source-major CSR the row of x an edge's term needs is the unit's own row, so
 *     dp_i[k] = sum_e dw/dp_i[e,k] * g_scale[v] g[v,k] * row_scale[u] x[u,k]
 * is summed where the edges are walked — every lane keeps its share, a block adds its teams and leaves one partial,
 * two small launches add the blocks in a fixed order — instead of two [n_dst, D] aggregates (stag_agg_bwd) and a
 * column-dot pass over them (stag_coldot).  dp0, dp1: [D] (a scalar parameter's gradient is their sum over k);
 * x == NULL: ones (the in-norm term of ops._AggregateVI); dx may be NULL; any D.
 * workspace >= stag_agg_bwd_dp_workspace_bytes(n_units, D), n_units = plan_t->n_units (no plan: csr_t->n_dst).     */
size_t stag_agg_bwd_dp_workspace_bytes(int64_t n_units, int32_t D);
int stag_agg_bwd_dp(const stag_csr* csr_t, const stag_plan* plan_t, const float* g, int64_t ldg,
                    int32_t D, const stag_noise_spec* spec, const float* g_scale,
                    const float* row_scale, const float* x, int64_t ldx, float* dx, int64_t ldo,
                    float* dp0, float* dp1, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of stag_agg_fwd for [E, 1] (amortised) parameters, param_mode PER_EDGE1, kind NORMAL | UNIFORM:
 * dx AND the gradient of every edge's parameter pair from ONE pass over the source-major CSR — the row of
 * x the pair needs is the unit's own row there, the row of g is the one the pass gathers anyway:
 *     dx[u,:]      = row_scale[u] * sum_{p: src_p = u} w[p,:] * g_scale[v_p] * g[v_p,:]
 *     dp0_edge[e]  = sum_k dw/dp0[e,k] * g_scale[v] * g[v,k] * row_scale[u] * x[u,k]       (e = csr_t->eid[p])
 *     dp1_edge[e]  = ... dw/dp1 ...   (with spec->p1_log: the gradient w.r.t. the log-scale, dw/dp1 * p1)
 * dx may be NULL (parameter gradients only); dp1_edge may be NULL.  The channel sum is one team sum, so the
 * row must fit one channel tile: D <= 256, else STAG_ENOSYS (the caller then runs stag_agg_bwd +
 * stag_agg_bwd_w, three gathers instead of one).  csr_t->eid and csr_t->nidx must be set.  in_norm 0.
 * The reference gets these from autograd through `rsample` on [E, D] tensors
 * (stag/layers.py:123-124, stag/distributions.py:229-242).                                          */
int stag_agg_bwd_edge(const stag_csr* csr_t, const stag_plan* plan_t, const float* g, int64_t ldg,
                      int32_t D, const stag_noise_spec* spec, const float* g_scale,
                      const float* row_scale, const float* x, int64_t ldx, float* dx, int64_t ldo,
                      float* dp0_edge, float* dp1_edge, void* stream);

/* w[eid, k] for every edge of the shard: what the reference keeps in
 * `self._edge_weight_sample` (stag/layers.py:107). relu and in_norm applied.
 * plan (may be NULL): the units bound what one team walks, so a hub row does not serialise.
 * norm_scale: [n_dst, Dn] device scratch, required when spec.in_norm (the row factors are
 * computed first, by the aggregation kernel on a broadcast row; with a plan that has segments
 * its workspace must hold stag_plan_workspace_bytes(n_seg, Dn, 1) bytes); NULL otherwise.    */
int stag_noise_materialize(const stag_csr* csr, const stag_plan* plan, const stag_noise_spec* spec,
                           int32_t Dn, float* w, int64_t ldw, float* norm_scale, void* stream);

/* Gradient w.r.t. per-edge weights or per-edge distribution parameters:
 *   dw[eid, k] = D[p, k] * sscale[u] * x[u, k] * g[v, k]      (g already carries dst scaling)
 * D = 1 for an explicit weight (spec NULL or kind NONE/EXPLICIT); D = dw/dp0 | dw/dp1 of the
 * regenerated draw when spec.deriv = 1 | 2 (AmortizedDistribution with vi=True).
 * dw1 != NULL (kind NORMAL | UNIFORM): BOTH derivatives from one pass and one Philox block —
 * dw takes dw/dp0, dw1 takes dw/dp1, spec.deriv is ignored.
 * reduce_k != 0: dw (dw1) is [E, 1] = sum over k (per-edge parameters of shape [E, 1]).
 * plan (may be NULL): as for stag_noise_materialize.                                         */
int stag_agg_bwd_w(const stag_csr* csr, const stag_plan* plan, const float* x, int64_t ldx,
                   const float* g, int64_t ldg, int32_t D,
                   const float* src_scale, const stag_noise_spec* spec, int32_t reduce_k,
                   float* dw, float* dw1, int64_t ldw, void* stream);

/* per-graph readout of a batched graph: out[b,:] = sum|mean of x[offsets[b]:offsets[b+1],:]
 * (dgl.sum_nodes / dgl.mean_nodes, stag/layers.py:165,177)                      */
int stag_segment_reduce(const float* x, int64_t ldx, int32_t D,
                        const int32_t* offsets, int32_t n_seg, int32_t reduce,
                        float* out, int64_t ldo, void* stream);

/* Column dots  out_i[k] = sum_n x[n,k] * t_i[n,k]  (i = 0, and 1 when t1/out1 are given): the
 * last step of a per-channel parameter gradient, dp_i[k] = sum_u x[u,k] * dp_i_rows[u,k]
 * (stag_agg_bwd).  Two launches, fixed summation order.  workspace: device memory of
 * stag_coldot_workspace_bytes(D) bytes.                                                      */
size_t stag_coldot_workspace_bytes(int32_t D);
int stag_coldot(const float* x, int64_t ldx, const float* t0, const float* t1, int64_t ldt,
                int64_t n_rows, int32_t D, float* out0, float* out1, void* workspace,
                size_t workspace_bytes, void* stream);

/* ---- amortised per-edge parameters with narrow heads ---------------------------------
 * AmortizedDistribution(in_features, 1) — what every scripts/<set>_rec/run.py builds (scripts/arxiv_rec/gcn/run.py:85);
 * hidden_features defaults to out_features = 1 (stag/distributions.py:158-159):
 *     h_e   = SiLU(W_e [feat[src_e] || feat[dst_e]] + b_e)        stag/distributions.py:178-183, 225-227
 *     par_c = W_c h_e + b_c   (c = loc, log_scale, ...)            stag/distributions.py:186-191, 229-231
 *     KL(N(loc, exp(log_scale)) || prior).mean()                   stag/layers.py:132-145
 * As dense torch these are GEMMs with 1-2 output columns and ~40 elementwise launches over [E, 1] tensors.
 * Three small kernels instead, every reduction in a fixed order:
 *
 * stag_node_project_fwd   y [n_rows, C] = x [n_rows, K] . w [K, C] + b [C]   (b may be NULL), C <= 16.
 *     W_e [feat_src || feat_dst] = feat W_src^T [src] + feat W_dst^T [dst]: with w = [W_src^T | W_dst^T]
 *     (C = 2 hidden) ONE pass over feat yields both projected tables.
 * stag_node_project_bwd   dx [n_rows, K] = gy . w^T (NULL: skipped), dw [K, C] = x^T . gy, db [C] = column sums of
 *     gy, one pass over x; workspace >= stag_amort_workspace_bytes((K + 1) * C).
 * stag_edge_mlp_fwd       par[c * n_edges + e] = b_c + sum_j SiLU(ps[src_e, j] + pd[dst_e, j]) wh[j, c];
 *     ps / pd: the two projected tables, row stride ldp floats, hidden <= 8 columns each; wh [hidden, n_par],
 *     n_par <= 4; one planar [n_edges] array per parameter (an [E, 1] tensor each, by edge id).
 * stag_edge_mlp_bwd       from gpar (planar like par): dpre [n_edges, hidden] (d / d of the pre-activation; its
 *     sums over the out-edges of a source / in-edges of a destination are the gradients of ps / pd — an
 *     aggregation of explicit rows, stag_agg_fwd), dwh [hidden, n_par], dbh [n_par];
 *     workspace >= stag_amort_workspace_bytes(36).
 * stag_normal_kl_fwd/bwd  kl_mean = mean_i KL(N(loc_i, exp(log_scale_i)) || N(p_loc, p_scale)) and its gradients
 *     (torch.distributions.kl._kl_normal_normal); p_loc, p_scale, kl_mean, g (the incoming gradient of kl_mean)
 *     are ONE-element device arrays: no host round trip; dloc / dlog_scale [n] and dp_loc / dp_scale [1] may be
 *     NULL; workspace >= stag_amort_workspace_bytes(2).                                                       */
size_t stag_amort_workspace_bytes(int32_t n_values);
/* Per-head dots of a [n_rows, G * F] matrix with C <= 2 vectors w [C][G * F]:
 *     y[c][n][g] = sum_f x[n, g F + f] * w[c][g F + f]
 * GAT's el / er = (ft * attn_l).sum(-1), (ft * attn_r).sum(-1) (stag/zoo/gat.py:109-110) from ONE pass over ft; the
 * backward (dx [n_rows, G F] = sum_c gy[c][n][g] w[c][k]; dw [C][G F] = sum_n gy[c][n][g] x[n, k]) from one pass too.
 * F % 4 == 0, 4 <= F <= 256 (a head takes F / 4 rounded up to a power of two lanes: the rest idle); rows 16-byte
 * aligned; workspace >= stag_amort_workspace_bytes(C * G * F).                                                  */
int stag_head_dot_fwd(const float* x, int64_t ldx, int64_t n_rows, int32_t G, int32_t F, const float* w, int32_t C,
                      float* y, void* stream);
int stag_head_dot_bwd(const float* x, int64_t ldx, int64_t n_rows, int32_t G, int32_t F, const float* w, int32_t C,
                      const float* gy, float* dx, int64_t lddx, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream);
int stag_node_project_fwd(const float* x, int64_t ldx, int64_t n_rows, int32_t K, const float* w,
                          const float* b, int32_t C, float* y, void* stream);
int stag_node_project_bwd(const float* x, int64_t ldx, int64_t n_rows, int32_t K, const float* w, int32_t C,
                          const float* gy, float* dx, int64_t lddx, float* dw, float* db, void* workspace,
                          size_t workspace_bytes, void* stream);
int stag_edge_mlp_fwd(const int32_t* src, const int32_t* dst, int64_t n_edges, const float* ps,
                      const float* pd, int64_t ldp, int32_t hidden, const float* wh, const float* bh,
                      int32_t n_par, float* par, void* stream);
int stag_edge_mlp_bwd(const int32_t* src, const int32_t* dst, int64_t n_edges, const float* ps,
                      const float* pd, int64_t ldp, int32_t hidden, const float* wh, int32_t n_par,
                      const float* gpar, float* dpre, float* dwh, float* dbh, void* workspace,
                      size_t workspace_bytes, void* stream);
int stag_normal_kl_fwd(const float* loc, const float* log_scale, int64_t n, const float* p_loc,
                       const float* p_scale, float* kl_mean, void* workspace, size_t workspace_bytes,
                       void* stream);
int stag_normal_kl_bwd(const float* loc, const float* log_scale, int64_t n, const float* p_loc,
                       const float* p_scale, const float* g, float* dloc, float* dlog_scale, float* dp_loc,
                       float* dp_scale, void* workspace, size_t workspace_bytes, void* stream);

/* Attention dropout of a GAT call (stag/zoo/gat.py:122: `attn_drop(edge_softmax(...))`, 0.6 in the reference's GAT
 * scripts): a[e,h] -> a[e,h] * keep[e,h] / keep_prob after the softmax, keep[e,h] = u < keep_prob with u the uniform
 * of the mask's OWN Philox stream (seed, offset (+ *epoch)) at (global forward position, head) — regenerated in the
 * backward, never stored.  NULL or keep_prob >= 1: no dropout.  Workgroup-cooperative kernels only (other shapes:
 * STAG_ENOSYS).                                                                                              */
typedef struct stag_gat_drop {
  float keep_prob;          /* 1 - p, in (0, 1] */
  uint64_t seed, offset;
  const uint64_t* epoch;    /* device counter added to offset when the kernel runs, or NULL (as stag_noise_spec) */
} stag_gat_drop;

/* GAT edge attention with noisy logits + softmax + aggregation, one launch:
 *   e[p,h]  = w[p,h] * leaky_relu(el[u_p,h] + er[v,h])     stag/zoo/gat.py:114-119
 *   a[p,h]  = softmax over the in-edges of v                stag/zoo/gat.py:122
 *   out[v,h,:] = sum_p a[p,h] * ft[u_p,h,:]                 stag/zoo/gat.py:125-126
 * ft is [N, H*F] row-major (H*F <= 256, H <= 64).  stats_out (may be NULL) receives the softmax
 * statistics of every destination row, [M, 2H] = max logit m[H] then sum l[H] of exp(e - m):
 * everything that depends on a[p,h] later (stag_gat_attn, stag_gat_bwd_edge) is computed from
 * them, so no [E, H] tensor leaves this kernel.
 * `plan` as for stag_agg_fwd, with a workspace of stag_gat_workspace_bytes() (long rows
 * are merged from per-segment softmax states).  When spec.in_norm is set, `norm_scale`
 * [M, H] must hold indeg / sum_in(w) per destination and head (stag/layers.py:8-36):
 * obtain the sums with stag_agg_fwd over a broadcast row of ones (D = H, ldx = 0).      */
size_t stag_gat_workspace_bytes(int32_t n_seg, int32_t H, int32_t F);
int stag_gat_fwd(const stag_csr* csr, const stag_plan* plan, const float* el, const float* er,
                 const float* ft, int32_t H, int32_t F, float neg_slope,
                 const stag_noise_spec* spec, const float* norm_scale, const stag_gat_drop* drop,
                 float* out, float* stats_out, void* stream);

/* The attention values a[eid, h] = exp(e[p,h] - m[v,h]) / l[v,h] (get_attention=True,
 * stag/zoo/gat.py:146-147) from the statistics of stag_gat_fwd; same spec (the noisy logits are
 * redrawn from their counters).                                                               */
int stag_gat_attn(const stag_csr* csr, const stag_plan* plan, const float* el, const float* er,
                  int32_t H, float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                  const float* stats, float* attn_out, void* stream);

/* Per-edge part of the GAT backward (the rest is stag_agg_fwd on the transposed CSR):
 *   da = <g[v,h,:], ft[u,h,:]>,  ds = a * (da - gdo[v,h]),  gdo[v,h] = <g[v,h,:], out[v,h,:]>
 *   (`out` is the forward's output [M, H*F]; gdo is formed in the kernel)
 *   de[eid,h] = ds * w * lrelu'(el[u,h] + er[v,h])      sum over in-edges -> d er, over out-edges -> d el
 *   dw[eid,h] = ds * lrelu(...) * norm_scale            (NULL: not wanted)
 * a is recomputed from `stats`; attn_out (may be NULL) receives a[eid, h] as a by-product:
 * d ft[u,h,:] = sum_{out-edges} a[e,h] * g[v,h,:]  is stag_agg_fwd on the transposed CSR with
 * EXPLICIT weights a[E,H] and spec.group = F.  Requires F % 4 == 0, F/4 a power of two.  */
int stag_gat_bwd_edge(const stag_csr* csr, const stag_plan* plan, const float* el,
                      const float* er, const float* ft, const float* stats, const float* g,
                      const float* out, int32_t H, int32_t F, float neg_slope,
                      const stag_noise_spec* spec, const float* norm_scale, float* de,
                      float* dw, float* attn_out, void* stream);

/* ---- multi-GPU: the halo exchange of a node-range partition on RCCL over xGMI ---------
 * The reference is single-process; BASELINE.json's north_star partitions the node range over the GPUs of
 * a node.  One process per GPU; RCCL is bound at run time (the copy already in the process if there is
 * one), so the library loads without it and these entry points then return STAG_ENOSYS.
 *   stag_comm_unique_id  rank 0 makes a 128-byte id and hands it to the other ranks (any side channel);
 *   stag_comm_init       every rank, its device current: joins the communicator (collective);
 *   stag_halo_allgather  x_full[world * n_floats] = the ranks' x_local[n_floats] in rank order
 *                        (equal-size padded row shards; backward = the caller's reduce-scatter);
 *   stag_halo_exchange   all-to-all-v of exactly the rows each rank's edges reference: peer p receives
 *                        send[off_p, off_p + send_counts[p]) and recv[...] fills with what p sends, in peer
 *                        order; counts in FLOATS, host arrays of length world, own entry 0; ONE RCCL group,
 *                        so every xGMI link carries its pair at the same time.
 * All enqueue on `stream` and return.                                                               */
int stag_comm_unique_id(void* id_out_host /* 128 bytes */);
int stag_comm_init(const void* id_host, int32_t rank, int32_t world, void** comm_out);
int stag_comm_destroy(void* comm);
int stag_halo_allgather(void* comm, const float* x_local, int64_t n_floats, float* x_full, void* stream);
int stag_halo_exchange(void* comm, const float* send, const int64_t* send_counts_host, float* recv,
                       const int64_t* recv_counts_host, void* stream);
/* (v17) The same exchange for SEVERAL row tables at once — GAT's ft [n, H*F] and el [n, H] travel to the same peers
 * (stag/zoo/gat.py:109-114 reads both at the source of every edge) — in ONE RCCL group: no packed [ft | el] copy on
 * either side.  send[t] / recv[t]: table t's rows, widths[t] floats each; send_rows / recv_rows: ROWS per peer, host
 * arrays of length world, the same for every table. */
int stag_halo_exchange_multi(void* comm, int32_t n_tables, const float* const* send, float* const* recv,
                             const int32_t* widths, const int64_t* send_rows_host,
                             const int64_t* recv_rows_host, void* stream);
/* (v17) out[i, :] = x[idx[i], :] for i < n: the rows a rank sends to its peers, written straight into the
 * persistent send buffer of the exchange (no per-step index_select allocation).  idx: int32 device array. */
int stag_gather_rows(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t width,
                     float* out, int64_t ldo, void* stream);

/* The whole backward of stag_gat_fwd in one call, for shapes with F % 4 == 0, H <= 16, H*F <= 1024,
 * H * lanes_per_head <= 256 (lanes_per_head = F / 4 rounded up to a power of two <= 64: the lanes past a head's
 * channels idle; stag_gat_bwd_two_pass wants F / 4 itself a power of two) and a block plan (stag_plan.block_ptr)
 * on the source-major orientation.
 *
 * stag_gat_bwd — ONE gather of the [H*F] rows (the forward has one, autograd through DGL's ops has four):
 *   1. sdot[v,h] = <g[v,h,:], out[v,h,:]>, the softmax's correction term, from one streaming pass;
 *   2. source pass (csr_t, plan_t; csr_t.nidx = forward position, csr_t.eid = edge id of each transposed
 *      position): per out-edge of u the weight is redrawn from its counters and a[e,h] rebuilt from `stats`;
 *      the team that owns u gathers g[v], forms <g[v,h,:], ft[u,h,:]> with its own row of ft, and gets
 *      d s[e,h] = a (dot - sdot[v,h]) w ns lrelu'(.), d ft[u,h,:] = sum_out a g[v,h,:], d el[u,h] = sum_out d s;
 *      d s goes to scratch by FORWARD position, dw (may be NULL: [E, H] by edge id) = d s-factor * lrelu * ns;
 *   3. d er[v,h] = sum of d s over the row's positions, which are contiguous there.
 * With `drop` (attention dropout) the mask is redrawn from its counters: d ft and the dot use a keep / keep_prob,
 * the softmax correction keeps a (sdot is unchanged: sum_e a_e d a_e = <g, out> with or without the mask).
 * stag_gat_bwd_two_pass — the form it replaces (no attention dropout: STAG_ENOSYS): an edge pass over csr (gathers ft[u]; a, de to scratch, d er)
 * and a source pass over csr_t (gathers g[v]; d ft, d el): two gathers; needs block plans on both orientations.
 * scratch: stag_gat_bwd_scratch_bytes(n_dst, n_edges, H) bytes, 16-byte aligned.
 * Long rows leave per-segment partials in plan->workspace (>= stag_gat_bwd_workspace_bytes(); the forward
 * plan's workspace serves both orientations, one after the other) and a small launch adds them in segment
 * order: no atomics, results do not depend on scheduling.
 * STAG_ENOSYS for other shapes / plans: use stag_gat_bwd_edge + stag_agg_fwd on the transposed CSR.
 * Replaces DGL's backward of u_add_v / edge_softmax / u_mul_e (stag/zoo/gat.py:114-126).          */
size_t stag_gat_bwd_workspace_bytes(int32_t n_seg, int32_t n_seg_t, int32_t H, int32_t F);
size_t stag_gat_bwd_scratch_bytes(int64_t n_dst, int64_t n_edges, int32_t H);
int stag_gat_bwd(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                 const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                 const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                 float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                 const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* dw, float* scratch,
                 void* stream);
/* (v19) stag_gat_bwd one STAGE at a time, for a caller that has something to do between them — a node-range shard
 * (BASELINE configs[4]; stag_amd/partition.py: _ShardGat) sends the gradient rows of the REMOTE sources back to their
 * owners while the local ones are still being computed:
 *     stages = ROWDOT                      step 1 of stag_gat_bwd (sdot and the per-destination records in `scratch`)
 *     stages = SOURCE, plan_t = sub-plan   step 2 over the units of that sub-plan only: writes d_ft / d_el of ITS rows
 *                                          (the segments of long rows must all be in one sub-plan) and d s of its edges
 *     stages = DER                         step 3 (d er), once every SOURCE call has been issued on the stream
 * with the SAME scratch, plan->workspace and output arrays in every call; plan_t is validated in every call (pass any
 * of the sub-plans for ROWDOT / DER).  ROWDOT | SOURCE | DER with the whole plan_t is stag_gat_bwd (dw = NULL).
 * The arithmetic of a unit does not depend on the call it rides in: results are bit-identical to stag_gat_bwd. */
#define STAG_GAT_BWD_ROWDOT 1
#define STAG_GAT_BWD_SOURCE 2
#define STAG_GAT_BWD_DER 4
int stag_gat_bwd_stages(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                        const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                        const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                        float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                        const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* scratch,
                        int32_t stages, void* stream);
/* (v17) stag_gat_bwd for a REPARAMETERISED draw whose parameters carry gradients (`vi=True`: `rsample`,
 * stag/layers.py:123-124, through the logits of stag/zoo/gat.py:117-119): the same pass also returns the FINISHED
 *   dp_i[h] = sum_e dL/dw[e,h] * dw/dp_i[e,h]      (p0 = loc | low, p1 = scale | high; d/dlog under spec.p1_log;
 *                                                   times 1[w > 0] under relu)
 * for SCALAR / PER_CHANNEL parameters of a NORMAL / UNIFORM spec without in-norm: the source pass has dL/dw of its
 * batch in LDS and redoes the draw with its derivatives; every batch leaves one [2][H] partial, two small launches
 * add them in a fixed order.  No [E, H] tensor — weights, their gradient — exists at any point.
 * workspace: stag_gat_bwd_dp_workspace_bytes(plan_t->n_blocks, H).  dp0, dp1: [H] (scalar parameters: sum them). */
size_t stag_gat_bwd_dp_workspace_bytes(int32_t n_blocks_t, int32_t H);
int stag_gat_bwd_dp(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                    const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                    const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                    float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                    const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* dp0, float* dp1,
                    float* scratch, void* workspace, size_t workspace_bytes, void* stream);
int stag_gat_bwd_two_pass(const stag_csr* csr, const stag_plan* plan, const stag_csr* csr_t,
                          const stag_plan* plan_t, const float* el, const float* er, const float* ft,
                          const float* stats, const float* g, const float* out, int32_t H, int32_t F,
                          float neg_slope, const stag_noise_spec* spec, const float* norm_scale,
                          const stag_gat_drop* drop, float* d_el, float* d_er, float* d_ft, float* dw,
                          float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STAG_HIP_H */
