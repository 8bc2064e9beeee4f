/*
 * petal_mi355x.h -- C ABI of the MI355X-native exact k-NN engine that drops in
 * behind petal-neighbors' BallTree::{euclidean, query, query_radius,
 * query_nearest} and distance::{Euclidean, pairwise}.
 *
 * petal-neighbors (Rust, /root/reference) has no FFI of its own: its boundary
 * is the public Rust API.  Each entry point below cites the reference item it
 * replaces (file:line relative to the reference checkout).  INTEGRATION.md
 * shows the Rust `extern "C"` block + safe wrapper a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no exceptions, no C++ or torch types.
 *   - every function returns PN_OK (0) or a PN_ERR_* code; pn_last_error()
 *     returns a thread-local message for the last failure on this thread.
 *   - strides are in ELEMENTS, not bytes.
 *   - "host" entry points take host pointers (what an ndarray hands over), do
 *     the PCIe transfers themselves and return when the results are in the
 *     caller's buffers; "_device" entry points take pointers into the index's
 *     GPU memory and a hipStream_t (as void*; NULL = HIP's default stream),
 *     ENQUEUE all their work on that stream and return without waiting for it
 *     and without reading anything back: results are ready in stream order
 *     (synchronise the stream, or keep enqueueing consumers on it).  Even the
 *     second tier for queries the first tier could not prove is enqueued
 *     unconditionally and sized by a count that stays on the device.
 *   - results are written into caller-allocated buffers; the only
 *     library-allocated result (radius CSR indices) is released with pn_free.
 *   - all query functions are re-entrant on a shared `const pn_index*`
 *     (BallTree queries take &self and `Euclidean: Sync`, src/distance.rs:19):
 *     every call works in its own pooled workspace, calls from several host
 *     threads run side by side; create/destroy/set_option must not race with
 *     queries on the same handle.
 *   - statistics (pn_index_get_stats) are collected lazily: that call waits
 *     for the device and then reads the counters the kernels kept.
 *   - there is NO CPU fallback: if no usable GPU is present every compute entry
 *     point fails with PN_ERR_DEVICE.
 *
 * Result contract
 *   - distances are bit-identical to the reference's sequential, unfused
 *     fold + sqrt (src/distance.rs:26-35) in the index's element type.
 *   - neighbours are ordered by (distance, index); NaN distances sort last
 *     (ordered-float semantics used at src/ball_tree.rs:396-421).  The
 *     reference's order inside groups of exactly equal distances is
 *     unspecified (it depends on tree shape and heap internals); this library
 *     returns ascending index there.
 */
#ifndef PETAL_MI355X_H
#define PETAL_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 4): + pn_query_radius_device_*, pn_sharded_query_radius_device_*, pn_bf16_selftest,
 * pn_debug_seed_model_feedback, PN_OPT_BF16_WAVES, PN_OPT_SEED_MODEL, pn_info.seed_model (the former reserved word);
 * PN_OPT_MFMA_STRUCTURE = 1 is now PN_ERR_INVALID.  Everything of version 2 is unchanged. */
#define PN_ABI_VERSION 3

/* ---- error codes.  EMPTY / NOT_CONTIGUOUS are ArrayError (src/lib.rs:9-16). */
enum {
    PN_OK = 0,
    PN_ERR_EMPTY = 1,          /* "array is empty"                     src/ball_tree.rs:44-46 */
    PN_ERR_NOT_CONTIGUOUS = 2, /* "array is not contiguous in memory"  src/ball_tree.rs:47-49 */
    PN_ERR_INVALID = 3,        /* NULL pointer / bad argument (no reference counterpart) */
    PN_ERR_DEVICE = 4,         /* HIP failure or no GPU */
    PN_ERR_NOMEM = 5,
    PN_ERR_UNSUPPORTED = 6,
    PN_ERR_EMPTY_MATRIX = 7,   /* zero columns with >= 2 rows: the reference panics "empty matrix" (src/ball_tree.rs:582) */
    PN_ERR_COMM = 8            /* RCCL failure, or RCCL not loadable (row-sharded corpora only) */
};

/* engine selection, PN_OPT_ENGINE */
enum {
    PN_ENGINE_AUTO = 0,  /* bf16 filter -> f32 MFMA filter -> exact scan, each tier taking what the previous
                            one could not prove; tiers that cannot serve the index are skipped */
    PN_ENGINE_EXACT = 1, /* exact VALU scan only (always bit-exact by construction) */
    PN_ENGINE_MFMA = 2,  /* force the f32 MFMA filter path (f32 only); verified, falls back per query */
    PN_ENGINE_BF16 = 3   /* force the bf16 MFMA filter as first tier (f32, D <= 1024); verified, falls back per query */
};

enum {
    PN_OPT_ENGINE = 1,
    PN_OPT_SEGMENTS = 2,    /* corpus row segments per query tile; 0 = auto */
    PN_OPT_INDEX_BASE = 3,  /* added to every returned index (row-sharded corpora) */
    PN_OPT_PROFILE = 4,     /* 1: bracket the dominant kernel with hipEvents on its stream (hot_ms); 2: also the whole
                               call (last_call_ms) -- every event record costs the stream a few microseconds */
    PN_OPT_FILTER_SLOTS = 5, /* k' kept by the MFMA filter per (query, segment); 0 = auto */
    PN_OPT_MFMA_STRUCTURE = 6, /* f32 MFMA tier: 0 auto; 2 = persistent partition, LDS candidate buffers, 1 workgroup/CU;
                                 3 = persistent partition, HBM candidate buffers, 2 workgroups/CU (1, the first
                                 structure -- a (query tile x segment) grid -- was retired in round 4: PN_ERR_INVALID) */
    PN_OPT_EXCHANGE_ALWAYS = 7, /* pn_sharded_set_option only.  A handle with ONE shard answers straight into the
                                  caller's buffers (nothing to exchange); 1 sends it through the packed buffer, the
                                  all-gather and the merge all the same (tests: RCCL at world size 1) */
    PN_OPT_SHARED_THRESHOLDS = 8, /* bf16 tier, plans with several row segments per query: 1 (default) = the segments
                                  of a query tighten each other's thresholds while the filter runs (refresher
                                  workgroups in the idle workgroup slots); 0 = off; n >= 2 = on with the shared
                                  threshold at the n-th smallest bound of the union (experiments).  Never changes a
                                  result: only how many candidates the filter keeps. */
    PN_OPT_BF16_WAVES = 9,       /* bf16 tier, narrow rows, main pass of a k-NN call: 0 (default) = the library picks -- the
                                  8-wave kernel (one 32-query column block per wave, four waves per SIMD) for short runs,
                                  the 4-wave kernel (two column blocks per wave, two waves per SIMD) for long ones; 4 / 8
                                  = always that kernel where it applies.  Never changes a result (A/B measurements,
                                  tests of both kernels). */
    PN_OPT_SEED_MODEL = 10       /* bf16 tier, indexes of narrow rows (D <= 128) whose seed model was accepted at build
                                  (pn_info.seed_model): 1 (default) = k-NN calls with k <= 128 take their starting
                                  thresholds from the model (no scout launch); 0 = always scout.  Never changes a result:
                                  a threshold only decides which tier answers a query. */
};

typedef struct pn_index pn_index;

typedef struct pn_info {
    uint64_t n_points; /* BallTree::num_points  src/ball_tree.rs:351-353 */
    uint64_t dim;
    uint64_t row_stride_device; /* padded row length in HBM (elements) */
    int32_t elem_bytes;         /* 4 = f32, 8 = f64 */
    int32_t device;
    int32_t mfma_eligible; /* 1 when the f32 MFMA filter path can serve this index */
    int32_t bf16_eligible; /* 1 when the bf16 MFMA filter path can serve this index */
    int32_t bf16_layout;   /* 0 none; 1 = five extra columns per row; 2 = row norm as the accumulator's initial value and
                              a per-query error constant (rows of homogeneous norm, D mod 16 in {0, 12..15}) */
    int32_t seed_model;    /* 1 when the index's seed model was accepted at build: uniform-like
                              corpora; clustered ones keep the scout launch */
} pn_info;

typedef struct pn_stats {
    uint64_t queries;          /* k-NN queries served */
    uint64_t fallback_queries; /* queries whose MFMA-filter result failed verification and were re-run exactly */
    uint64_t candidates;       /* candidates re-ranked exactly */
    uint64_t hot_launches;     /* launches of the dominant kernel inside profiled calls */
    double hot_ms;             /* their summed hipEvent duration (PN_OPT_PROFILE=1) */
    double last_call_ms;       /* hipEvent duration of the whole last profiled *_device call */
    uint64_t radius_results;
    uint64_t evaluations;      /* candidates whose exact distance was actually computed (the others were proven
                                  farther than the k-th from their filter bound alone) */
    double shard_ms;           /* pn_sharded_* under PN_OPT_PROFILE: summed hipEvent time of the local half of the calls
                                  (every local shard's filter + re-rank + local merge) ... */
    double exchange_ms;        /* ... and of their exchange half (pack is in the local half; ncclAllGather + final merge) */
    uint64_t reserved[1];
} pn_stats;

const char *pn_last_error(void);
const char *pn_strerror(int code);
int pn_abi_version(void);
int pn_device_count(int *count);

/* ---- construction: BallTree::new / BallTree::euclidean
 * (src/ball_tree.rs:38-63, 367-373).  Validation is the reference's: zero
 * rows -> PN_ERR_EMPTY; inner (column) stride != 1 with more than one column
 * -> PN_ERR_NOT_CONTIGUOUS (only the inner stride is checked, as at :47); the
 * row stride is free.  The host array is borrowed for the duration of the
 * call only; the index keeps a zero-padded copy in HBM (plus row norms).
 * The ball tree itself is not built: the walk is replaced by a batched scan. */
int pn_index_create_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                        ptrdiff_t col_stride, int device, pn_index **out);
int pn_index_create_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                        ptrdiff_t col_stride, int device, pn_index **out);
/* BallTree::new(points, Cosine) (src/ball_tree.rs:38 with src/distance.rs:76-122).  Cosine distance is not a metric
 * (no triangle inequality), so the reference's ball-pruned walk may skip true neighbours under it and its answers
 * depend on the tree's shape; this engine does not walk a tree: every query function on a Cosine index returns the EXACT
 * answer -- the k smallest (Cosine::distance, index) / every row with Cosine::distance < r -- in the reference's
 * arithmetic (three sequential sums, 1 - dot / (|a| |b|), zip-truncated dot product).  Where the reference's walk prunes
 * nothing the two agree bit for bit; where it prunes wrongly this engine returns the true nearest rows.
 * MEASURED DEVIATION (tests/test_gpu_tree_chain.py, the oracle's faithful walk under Cosine vs this index, 256 queries
 * each, uniform [-0.5, 0.5) data, profiles/r04_tree_chain.log): 60 000 x 16, k = 10: the walk misses a true neighbour on
 * 77.7 % of the queries; 20 000 x 3, k = 5: 64.8 %; 200 000 x 128, k = 10: 0 % (at that dimension the walk prunes nothing).
 * Wherever the answers differ, every entry of this index's answer is at most the walk's entry of the same rank.
 * Engines: k-NN on an index whose rows all have a squared norm inside [2^-100, 2^100] is served by the bf16 MFMA filter
 * over the rows NORMALISED in f64 (|q/|q| - p/|p||^2 = 2 (1 - cos): the Euclidean tier's images and kernels, DESIGN.md
 * 4.8) + a re-rank that evaluates Cosine::distance itself + a per-query proof; query_radius with 0 < r < 1 by the same
 * filter against each query's fixed bound + the Cosine::distance check of its survivors (1M x 128, 10^4 queries: 2.1 ms
 * against the exact scan's 126); unproven queries, queries of another length than the rows, other radii and every other
 * index take the exact scan.  Results never depend on the filter. */
int pn_index_create_cosine_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                               ptrdiff_t col_stride, int device, pn_index **out);
int pn_index_create_cosine_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                               ptrdiff_t col_stride, int device, pn_index **out);
/* same, from rows already resident on `device` (row-major, inner stride 1) */
int pn_index_create_device_f32(const float *d_points, size_t n_rows, size_t n_cols, size_t row_stride,
                               int device, void *stream, pn_index **out);
int pn_index_create_device_f64(const double *d_points, size_t n_rows, size_t n_cols, size_t row_stride,
                               int device, void *stream, pn_index **out);
void pn_index_destroy(pn_index *index);
int pn_index_info(const pn_index *index, pn_info *out);
int pn_index_set_option(pn_index *index, int option, int64_t value);
int pn_index_get_stats(const pn_index *index, pn_stats *out, int reset);

/* ---- k-NN: BallTree::query (src/ball_tree.rs:102-121, 203-243) for a batch of
 * queries (the reference takes one point per call; nq = 1 is that call).
 * q_cols is the length of each query vector: like the reference's `zip`
 * (src/distance.rs:27-28) the distance runs over min(q_cols, dim) coordinates.
 * Writes nq rows of kout = min(k, n_points) results, ascending by distance;
 * k = 0 writes nothing and succeeds (src/ball_tree.rs:106-108).  Never fails on
 * NaN coordinates (CHANGELOG.md:113-115).
 * One point per call is a supported pattern: corpora of at most 4096 rows answer a call of <= 64 queries with one
 * kernel launch (query and answer through mapped pinned memory); a handful of queries against a large corpus are spread
 * over every CU.  f64 indexes take the same first tier as f32 ones (bf16 filter, then f64 re-rank and proof).
 * Scratch: every call works in a pooled per-handle workspace that grows to the largest call seen -- for a filter-tier
 * batch of 10^4 queries ~0.4 GB (candidate buffers + the second tier's worst case), allocated on the first such call
 * (hipMalloc: no device-wide wait); a buffer that has to grow is retired behind the workspace's end-of-use event, never
 * freed inside a call.  Each concurrent host thread (and each shard sharing a GPU) has its own workspace. */
int pn_query_f32(const pn_index *index, const float *queries, size_t nq, size_t q_cols,
                 ptrdiff_t q_row_stride, size_t k, uint64_t *idx_out, float *dist_out);
int pn_query_f64(const pn_index *index, const double *queries, size_t nq, size_t q_cols,
                 ptrdiff_t q_row_stride, size_t k, uint64_t *idx_out, double *dist_out);
int pn_query_device_f32(const pn_index *index, const float *d_queries, size_t nq, size_t q_cols,
                        size_t q_row_stride, size_t k, uint64_t *d_idx_out, float *d_dist_out,
                        void *stream);
int pn_query_device_f64(const pn_index *index, const double *d_queries, size_t nq, size_t q_cols,
                        size_t q_row_stride, size_t k, uint64_t *d_idx_out, double *d_dist_out,
                        void *stream);

/* ---- 1-NN: BallTree::query_nearest (src/ball_tree.rs:80-86, 149-196). */
int pn_query_nearest_f32(const pn_index *index, const float *queries, size_t nq, size_t q_cols,
                         ptrdiff_t q_row_stride, uint64_t *idx_out, float *dist_out);
int pn_query_nearest_f64(const pn_index *index, const double *queries, size_t nq, size_t q_cols,
                         ptrdiff_t q_row_stride, uint64_t *idx_out, double *dist_out);

/* ---- radius: BallTree::query_radius (src/ball_tree.rs:137-142, 250-294).
 * Returns, per query, { i : distance(q, p_i) < r } (the leaf test at :277 is
 * strict) in ascending index order (the reference's order is unspecified; its
 * tests sort, :667, :777).  CSR output: offsets[nq + 1] is caller-allocated,
 * *idx_out is allocated by the library (release with pn_free). */
int pn_query_radius_f32(const pn_index *index, const float *queries, size_t nq, size_t q_cols,
                        ptrdiff_t q_row_stride, float radius, uint64_t *offsets, uint64_t **idx_out);
int pn_query_radius_f64(const pn_index *index, const double *queries, size_t nq, size_t q_cols,
                        ptrdiff_t q_row_stride, double radius, uint64_t *offsets, uint64_t **idx_out);
void pn_free(void *p);
/* The same for queries ALREADY IN HBM (round 4), everything enqueued on `stream`, nothing read back: the caller supplies
 * d_idx and its capacity (entries).  d_offsets [nq + 1] always receives the complete CSR offsets (per-query counts scanned
 * on the device); the rows of query q land at d_idx[d_offsets[q] ...] wherever that position is below `capacity`; and
 * d_total[0] (nullable) = d_offsets[nq].  A total above the capacity means "call again with a larger buffer" -- every
 * offset is right and the first `capacity` entries are in place even then; capacity = 0 (d_idx may be NULL) only counts.
 * Same lists, same order as pn_query_radius_* (strict '<', ascending index).  q_row_stride in elements. */
int pn_query_radius_device_f32(const pn_index *index, const float *d_queries, size_t nq, size_t q_cols,
                               size_t q_row_stride, float radius, uint64_t *d_offsets, uint64_t *d_idx, size_t capacity,
                               uint64_t *d_total, void *stream);
int pn_query_radius_device_f64(const pn_index *index, const double *d_queries, size_t nq, size_t q_cols,
                               size_t q_row_stride, double radius, uint64_t *d_offsets, uint64_t *d_idx, size_t capacity,
                               uint64_t *d_total, void *stream);

/* ---- distance::pairwise(x, &Euclidean) (src/distance.rs:58-74): n x n
 * symmetric matrix, zero diagonal, n < 2 -> zeros. Host in, host out. */
int pn_pairwise_f32(const float *x, size_t n_rows, size_t n_cols, ptrdiff_t row_stride, int device,
                    float *out);
int pn_pairwise_f64(const double *x, size_t n_rows, size_t n_cols, ptrdiff_t row_stride, int device,
                    double *out);
/* the same with rows and the n x n result in HBM (extension; row_stride >= n_cols elements), enqueued on `stream` and
 * complete when the call returns.  Only the pairs i < j are evaluated, each written twice (src/distance.rs:66-72). */
int pn_pairwise_device_f32(const float *d_x, size_t n_rows, size_t n_cols, size_t row_stride, int device, float *d_out,
                           void *stream);
int pn_pairwise_device_f64(const double *d_x, size_t n_rows, size_t n_cols, size_t row_stride, int device,
                           double *d_out, void *stream);

/* ---- Metric<A> for Euclidean (src/distance.rs:21-55): scalar, host-side by
 * design (one pair per call is not GPU work); bit-exact with the batched path. */
float pn_euclidean_f32(const float *a, const float *b, size_t len);
double pn_euclidean_f64(const double *a, const double *b, size_t len);
float pn_reuclidean_f32(const float *a, const float *b, size_t len);
double pn_reuclidean_f64(const double *a, const double *b, size_t len);
float pn_rdistance_to_distance_f32(float d);
double pn_rdistance_to_distance_f64(double d);
float pn_distance_to_rdistance_f32(float d);
double pn_distance_to_rdistance_f64(double d);

/* ---- Metric<A> for Cosine (src/distance.rs:76-122): distance = 1 - dot / (|a| |b|) with the reference's three
 * sequential sums (the dot product over the shorter length, each norm over its own vector); rdistance and both
 * conversions are the identity there.  pn_pairwise_cosine_*: distance::pairwise(x, &Cosine) on the GPU, same
 * contract as pn_pairwise_*.  BallTree::new(points, Cosine): pn_index_create_cosine_* above. */
float pn_cosine_f32(const float *a, size_t len_a, const float *b, size_t len_b);
double pn_cosine_f64(const double *a, size_t len_a, const double *b, size_t len_b);
int pn_pairwise_cosine_f32(const float *x, size_t n_rows, size_t n_cols, ptrdiff_t row_stride, int device,
                           float *out);
int pn_pairwise_cosine_f64(const double *x, size_t n_rows, size_t n_cols, ptrdiff_t row_stride, int device,
                           double *out);

/* ---- row-sharded corpora: merge `n_parts` per-shard results
 * (each nq x k_part, already carrying global indices via PN_OPT_INDEX_BASE and
 * sorted by (distance, index)) into the global top k_out.  Device pointers.
 * Part p's indices start at d_idx_parts + p * idx_part_stride (elements), its
 * distances at d_dist_parts + p * dist_part_stride, each laid out [query][k_part]
 * -- so one all-gather of a packed per-rank buffer {idx[nq][k] | dist[nq][k]}
 * can be merged in place.  Slots with index UINT64_MAX are treated as absent. */
int pn_merge_topk_device_f32(const uint64_t *d_idx_parts, const float *d_dist_parts, size_t n_parts,
                             size_t idx_part_stride, size_t dist_part_stride, size_t nq, size_t k_part,
                             size_t k_out, uint64_t *d_idx_out, float *d_dist_out, int device, void *stream);
int pn_merge_topk_device_f64(const uint64_t *d_idx_parts, const double *d_dist_parts, size_t n_parts,
                             size_t idx_part_stride, size_t dist_part_stride, size_t nq, size_t k_part,
                             size_t k_out, uint64_t *d_idx_out, double *d_dist_out, int device, void *stream);

/* ---- tree introspection: BallTree::{num_nodes, children_of, points_of, radius_of, compare_nodes,
 * node_distance_lower_bound} (src/ball_tree.rs:296-353), public in the reference for downstream dual-tree algorithms.
 * Queries on this engine never use a tree; the first call of any function below builds -- once, on the host, from the
 * index's own copy of the points -- the implicit complete binary ball tree exactly as the reference does
 * (src/ball_tree.rs:38-63, 445-461, 504-613: node i has children 2i+1 / 2i+2, median split on the column of maximum
 * spread, sequential-mean centroid, radius = largest distance to it UNDER THE INDEX'S METRIC -- Node::init and
 * node_distance_lower_bound call metric.distance, src/ball_tree.rs:309, 459: on a pn_index_create_cosine_* index
 * radius_of, compare_nodes and node_distance_lower_bound are Cosine::distance values, the split and the permutation do
 * not depend on the metric), so every answer equals the reference's node for node.  O(n d log n) time and (nodes x d) extra host memory, paid only by callers of this API.
 * A node number >= num_nodes is PN_ERR_INVALID (the reference panics).
 *   children_of:   *is_some = 0 for a leaf (None), else 1 with (*left, *right) = (2n+1, 2n+2)
 *   points_of:     *idx points INTO the tree's permutation (valid until pn_index_destroy), *count entries
 *   compare_nodes: *ordering = -1 Less / 0 Equal / 1 Greater by radius, 2 = None (a NaN radius)
 *   pn_tree_centroid_of: the node's centroid (n_cols elements of the index's type); private in the reference, exported
 *                  for tests */
int pn_tree_num_nodes(const pn_index *index, uint64_t *out);
int pn_tree_children_of(const pn_index *index, uint64_t node, int *is_some, uint64_t *left, uint64_t *right);
int pn_tree_points_of(const pn_index *index, uint64_t node, const uint64_t **idx, uint64_t *count);
int pn_tree_radius_of_f32(const pn_index *index, uint64_t node, float *out);
int pn_tree_radius_of_f64(const pn_index *index, uint64_t node, double *out);
int pn_tree_compare_nodes(const pn_index *index, uint64_t x, uint64_t y, int *ordering);
int pn_tree_node_distance_lower_bound_f32(const pn_index *index, uint64_t n1, uint64_t n2, float *out);
int pn_tree_node_distance_lower_bound_f64(const pn_index *index, uint64_t n1, uint64_t n2, double *out);
int pn_tree_centroid_of(const pn_index *index, uint64_t node, void *out_n_cols_elements);

/* ---- row-sharded corpora with the exchange behind the ABI (north_star: "the corpus shards by row
 * across the 8 GPUs of one node, per-shard (idx, dist) top-k merged by one RCCL allgather over xGMI").  Shard g of G
 * holds rows [g ceil(N/G), min(N, (g+1) ceil(N/G))); queries are replicated; each shard answers on its rows with global
 * indices; ONE ncclAllGather per query batch of the packed per-GPU buffer {idx[nq][k'] | dist[nq][k']}; a merge kernel
 * ordered by (distance, index).  Results are identical for every number of shards.  RCCL is loaded at run time; if
 * it cannot be, these entry points fail with PN_ERR_COMM (everything else keeps working).
 *
 *   pn_sharded_create_f32             BallTree::new over `n_devices` row shards driven by THIS process: shard g lives
 *                                     on devices[g] (ncclCommInitAll over the distinct devices; a device named several
 *                                     times holds several shards, merged locally before the exchange).  Same
 *                                     validation and errors as pn_index_create_f32.
 *   pn_comm_unique_id +               one process per GPU: rank 0 gets a communicator id (PN_COMM_ID_BYTES bytes),
 *   pn_sharded_create_rank_device_f32 the host program carries it to the other ranks, every rank passes it with ITS
 *                                     rows (already on `device`, row-major, inner stride 1): n_local must be the
 *                                     library's shard size for (n_total, rank, world) -- 0 for a rank beyond the corpus,
 *                                     which still takes part in every exchange.  Collective: all ranks must call it.
 *   pn_sharded_query_f32              host queries in, host results out (BallTree::query semantics, pn_query_f32's
 *                                     argument meaning); with one process per GPU every rank must call it with the SAME
 *                                     queries and every rank receives the full answer.
 *   pn_sharded_query_device_f32       the same, queries/results in HBM of the GPU this process drives, enqueued on
 *                                     `stream` (handles that drive exactly one GPU); batches above 131 072 queries are
 *                                     cut into chunks whose exchange + merge overlap the next chunk's filter.
 *   pn_sharded_query_radius_f32       BallTree::query_radius over all shards, CSR like pn_query_radius_f32.
 * One query at a time per handle (the communicator and the exchange buffers are per-handle state, serialised inside). */
#define PN_COMM_ID_BYTES 128
typedef struct pn_sharded pn_sharded;
typedef struct pn_sharded_info_t {
    uint64_t n_points;        /* rows of the whole corpus */
    uint64_t dim;
    uint64_t local_first_row; /* first row held by this process */
    uint64_t local_rows;      /* rows held by this process */
    int32_t n_shards;         /* row shards over all processes */
    int32_t world;            /* ranks of the RCCL communicator = GPUs */
    int32_t local_shards;     /* shards held by this process */
    int32_t rank;             /* this process's rank (0 with one process) */
    int32_t mfma_eligible;    /* every local shard can be served by the f32 MFMA filter / the bf16 filter (pn_info) */
    int32_t bf16_eligible;
} pn_sharded_info_t;
int pn_comm_unique_id(void *id_out /* PN_COMM_ID_BYTES */);
int pn_sharded_create_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                          ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out);
int pn_sharded_create_rank_device_f32(const float *d_rows, size_t n_local, size_t n_cols, size_t row_stride,
                                      uint64_t n_total, int rank, int world, const void *comm_id, int device,
                                      void *stream, pn_sharded **out);
/* the same over an f64 corpus (BallTree<f64, Euclidean>): every shard is an f64 index (pn_index_create_f64: bf16 filter,
 * f64 re-rank and proof), distances travel and merge as f64 */
int pn_sharded_create_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                          ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out);
int pn_sharded_create_rank_device_f64(const double *d_rows, size_t n_local, size_t n_cols, size_t row_stride,
                                      uint64_t n_total, int rank, int world, const void *comm_id, int device,
                                      void *stream, pn_sharded **out);
/* BallTree::new(points, Cosine) over row shards driven by THIS process (every shard a pn_index_create_cosine_* index;
 * queries through pn_sharded_query_* / pn_sharded_query_radius_* of the same element type) */
int pn_sharded_create_cosine_f32(const float *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                 ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out);
int pn_sharded_create_cosine_f64(const double *points, size_t n_rows, size_t n_cols, ptrdiff_t row_stride,
                                 ptrdiff_t col_stride, const int *devices, int n_devices, pn_sharded **out);
void pn_sharded_destroy(pn_sharded *sharded);
int pn_sharded_info(const pn_sharded *sharded, pn_sharded_info_t *out);
int pn_sharded_set_option(pn_sharded *sharded, int option, int64_t value); /* forwarded to every local shard */
int pn_sharded_get_stats(const pn_sharded *sharded, pn_stats *out, int reset); /* summed over the local shards */
int pn_sharded_query_f32(const pn_sharded *sharded, const float *queries, size_t nq, size_t q_cols,
                         ptrdiff_t q_row_stride, size_t k, uint64_t *idx_out, float *dist_out);
int pn_sharded_query_device_f32(const pn_sharded *sharded, const float *d_queries, size_t nq, size_t q_cols,
                                size_t q_row_stride, size_t k, uint64_t *d_idx_out, float *d_dist_out, void *stream);
int pn_sharded_query_radius_f32(const pn_sharded *sharded, const float *queries, size_t nq, size_t q_cols,
                                ptrdiff_t q_row_stride, float radius, uint64_t *offsets, uint64_t **idx_out);
int pn_sharded_query_f64(const pn_sharded *sharded, const double *queries, size_t nq, size_t q_cols,
                         ptrdiff_t q_row_stride, size_t k, uint64_t *idx_out, double *dist_out);
int pn_sharded_query_device_f64(const pn_sharded *sharded, const double *d_queries, size_t nq, size_t q_cols,
                                size_t q_row_stride, size_t k, uint64_t *d_idx_out, double *d_dist_out, void *stream);
int pn_sharded_query_radius_f64(const pn_sharded *sharded, const double *queries, size_t nq, size_t q_cols,
                                ptrdiff_t q_row_stride, double radius, uint64_t *offsets, uint64_t **idx_out);
/* pn_query_radius_device_* on a sharded handle with ONE shard (what a rank of a one-GPU job holds); several shards:
 * PN_ERR_UNSUPPORTED -- their ragged exchange goes through the host entry point above. */
int pn_sharded_query_radius_device_f32(const pn_sharded *sharded, const float *d_queries, size_t nq, size_t q_cols,
                                       size_t q_row_stride, float radius, uint64_t *d_offsets, uint64_t *d_idx,
                                       size_t capacity, uint64_t *d_total, void *stream);
int pn_sharded_query_radius_device_f64(const pn_sharded *sharded, const double *d_queries, size_t nq, size_t q_cols,
                                       size_t q_row_stride, double radius, uint64_t *d_offsets, uint64_t *d_idx,
                                       size_t capacity, uint64_t *d_total, void *stream);

/* ---- diagnostic: the first-tier filter's lower bounds themselves.  bounds_out[q * n_rows + i] = L'(q, p_i)
 * for the first n_rows corpus rows (clamped to n_points), with L' + qnorm_out[q] <= |q - p_i|^2 in real
 * arithmetic; qnorm_out[q] <= |q - mu|^2 where mu (mu_out, n_cols floats, nullable) is the translation vector the
 * tier works with -- the corpus mean (petal-neighbors_amd/csrc/bf16_filter.hip states the bound).  Host pointers;
 * q_cols must equal the index dimension.  Not part of the reference's interface: tests use it to check the
 * inequality and to measure the matrix core's accumulation error against the allowance the proof makes. */
int pn_bf16_bounds_f32(const pn_index *index, const float *queries, size_t nq, size_t q_cols,
                       ptrdiff_t q_row_stride, size_t n_rows, float *bounds_out, double *qnorm_out, float *mu_out);
/* ---- diagnostic: the bf16 tier's hardware self-test (round 4).  The bound allows the matrix core's f32 accumulation an
 * error of 2^-13 * sum|terms| -- a measured property of gfx950, not a documented one.  The library checks it itself the
 * first time an index on `device` is given its bf16 tier (synthetic chains of 8 and 65 MFMA steps against terms rebuilt
 * in f64) and refuses the tier (bf16_eligible = 0, pn_last_error says why) when the error exceeds 2 % of the allowance.
 * This entry runs the check (again) and returns the measured fraction of the allowance in *ratio_out. */
int pn_bf16_selftest(int device, float *ratio_out);
/* ---- diagnostic: the handle's seed-model feedback (round 4).  Every finished bf16-tier call tells the
 * handle whether its starting thresholds -- from the index's seed model or from a scout launch -- left queries unproven;
 * the handle then aims higher, goes back to the scout, or (after 64 scouted calls, at most three times) gives the model
 * another try.  This entry feeds one such observation (model_seed: the call was seeded by the model; `unproven` of `nq`
 * queries went to the next tier) into the handle exactly as a finished call would and returns the state in out[4] =
 * { off, widen steps, scouted calls since off, retries used }: tests drive the state machine without having to
 * construct corpora and batches that defeat the model. */
int pn_debug_seed_model_feedback(pn_index *index, int model_seed, uint64_t unproven, uint64_t nq, int32_t *out4);

/* ---- synthetic data (bench / tests): uniform [0,1) with exactly 24 random
 * bits, x[i] = (mix32(seed, first_counter + i) >> 8) * 2^-24, generated in
 * HBM; bit-identical to oracle_fill_uniform_f32. */
int pn_fill_uniform_device_f32(float *d_out, uint64_t count, uint64_t seed, uint64_t first_counter,
                               int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PETAL_MI355X_H */
