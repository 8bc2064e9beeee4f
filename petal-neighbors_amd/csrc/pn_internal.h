// pn_internal.h -- shared declarations between the translation units of
// libpetal_mi355x.so (gfx950 only).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>

namespace pn {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device): the attribute belongs to
// the device's copy of the kernel, and index handles on different devices -- or threads -- may launch the same
// instantiation.  One static instance per launcher.
struct LdsAttrOnce {
    std::atomic<uint64_t> done{0};
    hipError_t ensure(const void *kern, size_t bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
        return e;
    }
};

// ---- tiling constants of the exact scan (exact_scan.hip)
constexpr int kTileQ = 64;    // queries per workgroup tile
constexpr int kTileP = 64;    // corpus rows per tile step
constexpr int kChunkK = 32;   // coordinates staged per LDS chunk
constexpr int kRowAlign = 8;  // device rows are zero-padded to a multiple of 8 elements
constexpr int kRowPad = 256;  // device row COUNT is padded (zero rows) to a multiple of this

// ---- sortable integer keys.  Distances are >= +0 or NaN, so the IEEE bit
// pattern read as an unsigned integer is monotone; NaN is canonicalised to the
// quiet-NaN pattern, which is greater than +inf: exactly ordered-float's total
// order used by Neighbor (reference src/ball_tree.rs:396-421).
template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    using type = uint32_t;
    static constexpr type kMax = 0xFFFFFFFFu;
    static constexpr type kNaN = 0x7FC00000u;
};
template <> struct KeyOf<double> {
    using type = uint64_t;
    static constexpr type kMax = 0xFFFFFFFFFFFFFFFFull;
    static constexpr type kNaN = 0x7FF8000000000000ull;
};

// Per-(segment, query) candidate buffers in HBM, filled by the scan kernels and
// consumed by the select kernel.  Layout: [seg][query][slot].
struct CandBuf {
    void *keys;      // KeyOf<T>::type (exact) or float bits of the lower bound (MFMA filter)
    uint32_t *idx;   // corpus row (local to this index)
    uint32_t *cnt;   // entries in use            [seg][query]
    void *tau;       // last threshold (same type as keys) [seg][query]
    size_t nq_pad;   // queries padded to a multiple of kTileQ
    int nseg;
    int cap;         // slots per (segment, query); multiple of 64
    int idx_stride = 1;  // 2: keys and idx interleaved as (key, row) pairs, idx = keys + 1 (bf16 filter)
    int final_keep = 0;  // bf16 filter, 64-slot buffers: a buffer holding at most this many entries ends its run uncut (0: k')
    int bf16_waves = 0;  // PN_OPT_BF16_WAVES: 0 = the 8-wave main-pass kernel where it applies, 4 = the 4-wave kernel
};

inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- exact_scan.hip
// lo_key/lo_idx (nullable, [nq_pad]): only entries strictly after (lo_key[q], lo_idx[q]) take part
// nq_dev (nullable) / nq_off: device-driven query count -- only the first min(nq, max(*nq_dev - nq_off, 0)) queries
// exist; the grid is sized for nq and the surplus query tiles exit at once (second tier behind a filter)
// pnorm / qnorm (nullable, together): the index's metric is Cosine -- distance = 1 - dot / (qnorm[q] pnorm[row])
// qsel (nullable, f32 Euclidean only): query r of the launch is row qsel[nq_off + r] of Q -- the flagged queries of a
// filter tier read in place (no gather launch)
hipError_t launch_exact_knn_f32(const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq,
                                size_t ldq, int kp, size_t seg_len, const CandBuf &cb, const void *lo_key,
                                const uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, const float *pnorm,
                                const float *qnorm, hipStream_t s, const uint32_t *qsel = nullptr);
hipError_t launch_exact_knn_f64(const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq,
                                size_t ldq, int kp, size_t seg_len, const CandBuf &cb, const void *lo_key,
                                const uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, const double *pnorm,
                                const double *qnorm, hipStream_t s, const uint32_t *qsel = nullptr);
// radius: count pass (fill == nullptr) then fill pass.  counts/offsets are [query][seg].
hipError_t launch_exact_radius_f32(const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq,
                                   size_t ldq, float r, size_t seg_len, int nseg, uint32_t *counts,
                                   const uint64_t *offsets, uint64_t *fill, uint64_t index_base, const float *pnorm,
                                   const float *qnorm, hipStream_t s, const uint32_t *qsel = nullptr,
                                   const uint32_t *nq_dev = nullptr, uint64_t capacity = ~0ull);
hipError_t launch_exact_radius_f64(const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq,
                                   size_t ldq, double r, size_t seg_len, int nseg, uint32_t *counts,
                                   const uint64_t *offsets, uint64_t *fill, uint64_t index_base, const double *pnorm,
                                   const double *qnorm, hipStream_t s, const uint32_t *qsel = nullptr,
                                   const uint32_t *nq_dev = nullptr, uint64_t capacity = ~0ull);
// qsel / nq_dev (nullable): query r of the launch is row qsel[r] of Q, only the first *nq_dev listed queries exist (grid
// sized for nq); counts / offsets indexed by r.  capacity: fill positions at or beyond it are not written.

// ---- radius_device.hip: the device-side plumbing of pn_query_radius_device_* (no host round trip)
// list the queries the exact scan must answer: need[q] = over[q] | bad[q] (either nullable); sel[0 .. *nsel) <- those q
// ascending, pos[q] <- position in sel or 0xFFFFFFFF; nkept[q] <- 0 for listed queries.  nsel zeroed by the caller.
hipError_t launch_rad_list(const uint32_t *over, const uint32_t *bad, int nq, uint32_t *nkept, uint32_t *sel,
                           uint32_t *pos, uint32_t *nsel, hipStream_t s);
// per-query result counts: fin[q] = pos[q] valid ? sum_seg counts_x[pos[q] * nseg + seg] : nkept[q]
// (pos / nkept nullable: every query listed at its own position / none kept)
hipError_t launch_rad_counts(const uint32_t *nkept, const uint32_t *pos, const uint32_t *counts_x, int nseg, int nq,
                             uint32_t *fin, hipStream_t s);
// offsets[0 .. n] <- exclusive prefix sums of in[0 .. n) (64-bit); scratch: >= (n / 4096 + 2) * 8 bytes; total (nullable)
// <- offsets[n]
hipError_t launch_exclusive_scan_u32(const uint32_t *in, size_t n, uint64_t *offsets, uint64_t *scratch, uint64_t *total,
                                     hipStream_t s);
// offs_x[r * nseg + seg] <- offsets[sel ? sel[r] : r] + sum_{t < seg} counts_x[r * nseg + t], r < (nsel ? *nsel : nq)
hipError_t launch_rad_seg_offsets(const uint64_t *offsets, const uint32_t *sel, const uint32_t *nsel, int nq,
                                  const uint32_t *counts_x, int nseg, uint64_t *offs_x, hipStream_t s);
// out[offsets[q] + e] <- index_base + kept[q * kept_stride + e], e < nkept[q], positions below capacity only
hipError_t launch_radius_gather_cap(const uint32_t *kept, const uint32_t *nkept, const uint64_t *offsets, int nq,
                                    size_t kept_stride, uint64_t index_base, uint64_t *out, uint64_t capacity,
                                    hipStream_t s);
// Cosine's norms: norms[i] = sqrt(sequential sum of x_i[k]^2, k < dim) in T
hipError_t launch_cosine_norms_f32(const float *X, size_t n, int dim, size_t ld, float *norms, hipStream_t s);
hipError_t launch_cosine_norms_f64(const double *X, size_t n, int dim, size_t ld, double *norms, hipStream_t s);
hipError_t launch_exact_pairwise_f32(const float *X, size_t n, int dim, size_t ld, float *out, hipStream_t s);
hipError_t launch_exact_pairwise_f64(const double *X, size_t n, int dim, size_t ld, double *out, hipStream_t s);
// distance::pairwise under the Cosine metric; norms: n elements of scratch
hipError_t launch_cosine_pairwise_f32(const float *X, size_t n, int dim, size_t ld, float *norms, float *out,
                                      hipStream_t s);
hipError_t launch_cosine_pairwise_f64(const double *X, size_t n, int dim, size_t ld, double *norms, double *out,
                                      hipStream_t s);

// ---- select.hip
// Exact mode: keys are exact distance keys; picks the kout smallest (key, idx).
// writes results r < kout of query q at [q * out_stride + out_off + r]; when lo_key != nullptr also
// records the last (key, row) written per query as the next round's lower bound
// osel (nullable): the results of query q go to output row osel[nq_off + q] instead of row q (scatter in place)
hipError_t launch_select_exact_f32(const CandBuf &cb, int nq, int kout, uint64_t index_base, uint64_t *idx_out,
                                   float *dist_out, size_t out_stride, size_t out_off, void *lo_key,
                                   uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, bool signed_keys,
                                   hipStream_t s, const uint32_t *osel = nullptr);
hipError_t launch_select_exact_f64(const CandBuf &cb, int nq, int kout, uint64_t index_base, uint64_t *idx_out,
                                   double *dist_out, size_t out_stride, size_t out_off, void *lo_key,
                                   uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, bool signed_keys,
                                   hipStream_t s, const uint32_t *osel = nullptr);
hipError_t launch_select_exact_groups_f32(const CandBuf &cb, int groups, int kp, int nq, int kout, uint64_t index_base,
                                          uint64_t *idx_out, float *dist_out, size_t out_group_stride,
                                          const uint32_t *nq_dev, uint32_t nq_off, hipStream_t s, bool signed_keys = false);
hipError_t launch_select_exact_groups_f64(const CandBuf &cb, int groups, int kp, int nq, int kout, uint64_t index_base,
                                          uint64_t *idx_out, double *dist_out, size_t out_group_stride,
                                          const uint32_t *nq_dev, uint32_t nq_off, hipStream_t s, bool signed_keys = false);
// MFMA mode: keys are f32 lower bounds L; recomputes every candidate's distance
// in the reference's operation order, selects, and verifies the filter's
// exclusions (flags[q] = 1 -> the query must be re-run exactly).
// qn / qbad (nullable, [nq]): the thresholds bound |q-p|^2 - |q|^2 (bf16 filter), qn[q] <= |q|^2 is added back;
// qbad[q] != 0 flags the query outright
// results of query q go to idx_out/dist_out[q * out_stride + r]; n_flagged: per-call count of flagged queries (device,
// zeroed by the caller); sel (nullable, [nq]): the flagged queries are LISTED here as they are found (sel[0 ..
// *n_flagged), any order) -- the second tier's work list, no listing launch; stats (nullable): the index's running
// device counters, kPnStatWords words: [0] fallback queries, then kPnStatSlots pairs {candidates, exact evaluations}
// that the queries add to by q mod kPnStatSlots (10^4 waves adding to ONE address cost as much as the whole kernel)
constexpr int kPnStatSlots = 64, kPnStatWords = 4 + 2 * kPnStatSlots;
hipError_t launch_select_rerank_f32(const CandBuf &cb, const float *P, size_t n, int dim, size_t ldp,
                                    const float *Q, int nq, size_t ldq, int kout, uint64_t index_base,
                                    uint64_t *idx_out, float *dist_out, size_t out_stride, uint32_t *flags,
                                    uint32_t *n_flagged, const double *qn, const uint32_t *qbad, uint32_t *sel,
                                    unsigned long long *stats, hipStream_t s, int first_eval = 0, int cell_max = 0);
// first_eval: candidates (smallest bounds first) evaluated in the first round, >= kout (0: kout); cell_max: entries a
// (segment, query) cell holds at most (0: cb.cap) -- sizes the kernel's LDS
// f64 indexes behind the bf16 filter (qn required): distances in the reference's f64 fold, proof with u = 2^-53
hipError_t launch_select_rerank_f64(const CandBuf &cb, const double *P, size_t n, int dim, size_t ldp,
                                    const double *Q, int nq, size_t ldq, int kout, uint64_t index_base,
                                    uint64_t *idx_out, double *dist_out, size_t out_stride, uint32_t *flags,
                                    uint32_t *n_flagged, const double *qn, const uint32_t *qbad, uint32_t *sel,
                                    unsigned long long *stats, hipStream_t s, int first_eval = 0, int cell_max = 0);
// Cosine indexes behind the bf16 filter (select.hip, cos_proof_lb): cnorm [n] / qnorm [nq] = the rows' / queries' norms in
// the index's type (cosine_norms_kernel), candidates evaluated by Cosine::distance, order-preserving keys of ALL floats
hipError_t launch_select_rerank_cos_f32(const CandBuf &cb, const float *P, size_t n, int dim, size_t ldp, const float *Q,
                                        int nq, size_t ldq, int kout, uint64_t index_base, uint64_t *idx_out,
                                        float *dist_out, size_t out_stride, uint32_t *flags, uint32_t *n_flagged,
                                        const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *stats,
                                        hipStream_t s, int first_eval, int cell_max, const float *cnorm, const float *qnorm);
hipError_t launch_select_rerank_cos_f64(const CandBuf &cb, const double *P, size_t n, int dim, size_t ldp, const double *Q,
                                        int nq, size_t ldq, int kout, uint64_t index_base, uint64_t *idx_out,
                                        double *dist_out, size_t out_stride, uint32_t *flags, uint32_t *n_flagged,
                                        const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *stats,
                                        hipStream_t s, int first_eval, int cell_max, const double *cnorm, const double *qnorm);
// osel (nullable): the merged result of query q goes to row osel[q] of the outputs (row stride out_stride, 0 = k_out);
// host_count (nullable, mapped pinned memory): block 0 copies *nq_dev there (the count a LATER call looks at)
// Small corpora, a few queries per call: the whole call in one launch (one wave per query over all rows); Q and the
// outputs may be mapped pinned host memory.  radius_mode: out row q = {count, rows ascending ...} (out_stride >= n + 1)
size_t tiny_query_lds_bytes(size_t n, int dim_eff, int elem_bytes);  // must be <= 64 KiB
hipError_t launch_tiny_query_f32(const float *P, size_t n, int dim_eff, size_t ldp, const float *Q, size_t ldq, int nq,
                                 int kout, bool radius_mode, float radius, uint64_t index_base, uint64_t *idx_out,
                                 float *dist_out, size_t out_stride, hipStream_t s);
hipError_t launch_tiny_query_f64(const double *P, size_t n, int dim_eff, size_t ldp, const double *Q, size_t ldq, int nq,
                                 int kout, bool radius_mode, double radius, uint64_t index_base, uint64_t *idx_out,
                                 double *dist_out, size_t out_stride, hipStream_t s);
hipError_t launch_merge_topk_f32(const uint64_t *idx_parts, const float *dist_parts, int n_parts,
                                 size_t idx_part_stride, size_t dist_part_stride, int nq, int k_part, int k_out,
                                 uint64_t *idx_out, float *dist_out, hipStream_t s, const uint32_t *nq_dev = nullptr,
                                 const uint32_t *osel = nullptr, size_t out_stride = 0, uint32_t *host_count = nullptr,
                                 bool signed_keys = false);
hipError_t launch_merge_topk_f64(const uint64_t *idx_parts, const double *dist_parts, int n_parts,
                                 size_t idx_part_stride, size_t dist_part_stride, int nq, int k_part, int k_out,
                                 uint64_t *idx_out, double *dist_out, hipStream_t s, const uint32_t *nq_dev = nullptr,
                                 const uint32_t *osel = nullptr, size_t out_stride = 0, uint32_t *host_count = nullptr,
                                 bool signed_keys = false);
// gather rows sel[off + i] of src into dst row i / scatter result rows i back to query sel[off + i], for
// i < min(max_rows, *nsel - off): the count stays on the device
template <typename T>  // T = float | double (explicitly instantiated)
hipError_t launch_gather_rows(const T *src, size_t ld, const uint32_t *sel, const uint32_t *nsel, uint32_t off,
                              uint32_t max_rows, T *dst, hipStream_t s);
template <typename T>
hipError_t launch_radius_check(const uint32_t *rcnt, const uint32_t *ridx, size_t nq_pad, int nseg, uint32_t cap,
                               const T *P, size_t ldp, const T *Q, int nq, int dim, T r, uint32_t *kept,
                               uint32_t *nkept, uint32_t *overflow, int ridx_stride, uint32_t *over_q, hipStream_t s,
                               const T *cnorm = nullptr, const T *qnorm = nullptr);  // (both: Cosine::distance < r)
hipError_t launch_gather_rows_f32(const float *src, size_t ld, const uint32_t *sel, const uint32_t *nsel, uint32_t off,
                                  uint32_t max_rows, float *dst, hipStream_t s);
hipError_t launch_scatter_results_f32(const uint64_t *idx_in, const float *dist_in, const uint32_t *sel,
                                      const uint32_t *nsel, uint32_t off, uint32_t max_rows, int kout, uint64_t *idx_out,
                                      float *dist_out, size_t out_stride, hipStream_t s);
hipError_t launch_compact_flags(const uint32_t *flags, int nq, uint32_t *sel, uint32_t *nsel, hipStream_t s);

// ---- pack.hip
// dst[r][c] = (r < n && c < cols) ? src[r*row_stride + c] : 0  for r < n_pad, c < ld
hipError_t launch_pack_rows_f32(const float *src, size_t n, size_t cols, size_t row_stride, float *dst,
                                size_t n_pad, size_t ld, hipStream_t s);
hipError_t launch_pack_rows_f64(const double *src, size_t n, size_t cols, size_t row_stride, double *dst,
                                size_t n_pad, size_t ld, hipStream_t s);
hipError_t launch_fill_uniform_f32(float *out, uint64_t count, uint64_t seed, uint64_t first, hipStream_t s);
// scaled squared row norms for the MFMA lower bound (see mfma_filter_v2.hip)
hipError_t launch_row_norms_f32(const float *X, size_t n_pad, size_t n, int dim, size_t ld, float alpha,
                                float *norm_out, uint32_t *nonfinite_flag, hipStream_t s);

// ---- mfma_filter_v2.hip (the f32 MFMA tier; round 4: the first structure, a (query tile x segment) grid in
// mfma_filter.hip, is retired -- nothing selected it but PN_OPT_MFMA_STRUCTURE = 1)
bool mfma_supported(int dim, size_t ld);
// persistent balanced partition, k' <= 32 with LDS buffers / k' <= 224 with HBM buffers
int mfma_v2_max_segments(size_t q_tiles, int n_wg);
// gcand != nullptr: candidate buffers in HBM (mfma_v2_gcand_bytes(n_wg) bytes), 2 workgroups per CU
size_t mfma_v2_gcand_bytes(int n_wg, int kp);
int mfma_v2_cap_for(int kp);
hipError_t launch_mfma_filter_v2_f32(const float *P, const float *pnorm, size_t n, size_t ldp, const float *Q,
                                     const float *qnorm, size_t ldq, int kp, const CandBuf &cb, int n_wg,
                                     uint32_t *gcand, hipStream_t s);

// radius filter (mfma_filter_v2.hip) + exact check / ascending-row ordering of its survivors (select.hip)
hipError_t launch_mfma_radius_f32(const float *P, const float *pnorm, size_t n, size_t ldp, const float *Q,
                                  const float *qnorm, size_t nq_pad, float tau_excl, uint32_t cap, uint32_t *rcnt,
                                  uint32_t *ridx, int n_wg, hipStream_t s);
// per query: exact distance of every listed row, keep dist < r, order by row; kept[q][*], nkept[q]; *overflow += 1
// when some list of the query exceeded `cap`
// ridx_stride: element stride of the row lists (2 when they are the rows of (key, row) pairs)
hipError_t launch_radius_check_f32(const uint32_t *rcnt, const uint32_t *ridx, size_t nq_pad, int nseg, uint32_t cap,
                                   const float *P, size_t ldp, const float *Q, int nq, int dim, float r,
                                   uint32_t *kept, uint32_t *nkept, uint32_t *overflow, int ridx_stride,
                                   uint32_t *over_q /* nullable: per-query overflow flags */, hipStream_t s);
hipError_t launch_radius_gather(const uint32_t *kept, const uint32_t *nkept, const uint64_t *offsets, int nq,
                                size_t kept_stride, uint64_t index_base, uint64_t *out, hipStream_t s);

// ---- bf16_filter.hip: first-tier filter (bf16 MFMA lower bound of |q-p|^2 - |q|^2), D <= 128
bool bf16_supported(int dim);
bool bf16_is_wide(int dim);  // 128 < D: K-chunked kernel (launch_bf16_wide_filter), images in the chunked layout
// ci: the "norm in the accumulator" layout (bf16_filter.hip, bf16_ci_dim): no extra columns, row norms as the
// chain's initial value, the error products as a per-query constant folded into qn
int bf16_ks_for(int dim, bool ci);
bool bf16_ci_candidate(int dim);
size_t bf16_image_bytes(size_t n, int dim, bool ci);        // corpus tile images
size_t bf16_query_bytes(size_t nq_pad, int dim, bool ci);   // packed query rows
int bf16_cap_for(int kp);                          // slots per (segment, query): 64 / 128 / 256
int bf16_query_tile();                             // queries per workgroup (256)
// mu: translation vector [dim]; sums: [dim + 1] per-dimension f64 sums of the corpus + the sum of all squares
// (zeroed by the caller)
template <typename T>  // T = float | double (explicitly instantiated): the index's element type
hipError_t launch_bf16_column_sums(const T *P, size_t n, int dim, size_t ld, double *sums, hipStream_t s);
// out[0] <- the largest |delivered - exact| / (2^-13 sum|terms|) over synthetic chains of 8 and 65 MFMA steps: the matrix
// core's accumulation error against the allowance the bf16 bound makes for it (bf16_filter.hip, bf16_selftest_kernel)
hipError_t launch_bf16_selftest(float *out, hipStream_t s);
// mu [dim] <- the per-dimension mean (f32) when translating by it shrinks the sum of squared norms 16x (wide rows: 2x), else
// zero; words[0] <- 1 / 0 accordingly.  never: always zero (diagnostic builds)
hipError_t launch_bf16_decide_mu(const double *sums, size_t n, int dim, bool never, float *mu, uint32_t *words, hipStream_t s);
// out4 (zeroed by the caller): max Bp, max Dp, sum Bp, sum Dp over the rows
template <typename T>
hipError_t launch_bf16_row_stats(const T *P, const float *mu, size_t n, int dim, size_t ld, double *out4,
                                 hipStream_t s);
// Cosine indexes: rows normalised in f64 (bf16_filter.hip, cos_normalize_rows_kernel); out [n_out][ld_out], rows >= n_valid
// zero; a row with a squared norm outside [2^-100, 2^100] raises *bad, or (nan_rows) becomes NaNs
template <typename T>
hipError_t launch_cos_normalize_rows(const T *X, size_t n_valid, size_t n_out, int dim, size_t ld_in, double *out,
                                     size_t ld_out, uint32_t *bad, bool nan_rows, hipStream_t s);
template <typename T>
hipError_t launch_bf16_pack_corpus(const T *P, const float *mu, size_t n, int dim, size_t ld, void *img,
                                   uint32_t *bad, bool ci, hipStream_t s);
// Qp / ldq / misc (nullable, narrow rows only: bf16_pack_fused_supported): Q is the CALLER's array (row stride ld); the
// kernel also writes the zero-padded f32 copy Qp[nq_pad][ldq] and zeroes the 16 counter words at misc -- the three
// launches at the head of a call in one
bool bf16_pack_fused_supported(int dim);
// Seed model of an index (round 4; index.hip, seed_model_build): starting thresholds of a k-NN call from per-dimension
// moments of the corpus instead of a scout launch.  m1 / a / b: [dim] f32 (M1_k, 4 (M2_k - M1_k^2), 4 (M3_k - M2_k M1_k)
// of u = p - mu), c0 = sum M2_k, v0 = sum (M4_k - M2_k^2), z = the calibrated number of standard deviations below the mean;
// seed_out [nq_pad] receives the sortable keys (nullptr: no seeds).
struct Bf16SeedModel {
    const float *m1 = nullptr, *a = nullptr, *b = nullptr;
    double c0 = 0.0, v0 = 0.0, z = 0.0;
    uint32_t *seed_out = nullptr;
};
template <typename T>
hipError_t launch_bf16_pack_queries(const T *Q, const float *mu, size_t nq, size_t nq_pad, int dim, size_t ld,
                                    void *B, double *qn, uint32_t *qbad, bool ci, double bmax, double dmax,
                                    hipStream_t s, T *Qp = nullptr, size_t ldq = 0, uint32_t *misc = nullptr,
                                    const Bf16SeedModel *sm = nullptr);
// out[j * dim + k] += sum over rows of (p_k - mu_k)^(j + 1), j = 0 .. 3 (zeroed by the caller)
template <typename T>
hipError_t launch_bf16_column_moments(const T *P, const float *mu, size_t n, int dim, size_t ld, double *out, hipStream_t s);
// split: row parts per query tile (>= 1); cb.nseg >= bf16_segments(q_tiles, n_wg, split); scout_max: cap on the
// tiles of a run that are contracted first, without buffers, to seed the threshold (0 = no scouting)
int bf16_segments(size_t q_tiles, int n_wg, int split);
// tau_init (nullable, [nq_pad] sortable keys): starting thresholds instead of the scout pass.  radius = true:
// fixed thresholds (tau_init required, cb.cap == 256), buffers overflow (count = cap + 1) instead of compacting
// scout_out (nullable): scout-only launch -- no buffers are touched; every run leaves its lanes' smallest block
// minima in scout_out[cell][2][bf16_scout_list()] (pre-filled with +inf by the caller)
// shp (nullable; main pass of an aligned k-NN plan with tau_init only): shared thresholds -- n_refresh extra "refresher"
// workgroups merge what the segments of a query hold and lower the query's word in tau_init while the pass runs
// (bf16_filter.hip, "Shared thresholds").  pcnt: [nseg][nq_pad] words (zeroed whenever the geometry changes), done: one
// word, ZEROED before the launch; epoch 1 .. 4095, different from the previous launch on pcnt; rank: r-th smallest.
struct Bf16Shared {
    uint32_t *pcnt, *done;
    uint32_t epoch, rank;
    int n_refresh;  // 0: no shared thresholds (the seed fields below may still be set)
    // Seed in the kernel (narrow rows, <= 32 segments): the scout-only launch is given `seed_words` ([nq_pad], set to
    // +inf there: the words the refreshers lower), the main launch `seed_lists` (the scout launch's lists) + the rank
    // and the number of real queries, and derives its starting thresholds itself -- no bf16_seed_kernel launch between
    const float *seed_lists = nullptr;
    uint32_t seed_rank = 0, seed_nq = 0;
    uint32_t *seed_words = nullptr;
};
bool bf16_seed_in_kernel(int nseg);
bool bf16_shared_supported(int cap);
hipError_t launch_bf16_filter(const void *img, size_t n, int dim, const void *B, int kp, const CandBuf &cb, int n_wg,
                              int split, int scout_max, const uint32_t *tau_init, bool radius, float *scout_out,
                              bool ci, hipStream_t s, const Bf16Shared *shp = nullptr);
// wide rows: n_wg persistent workgroups (one per CU) over equal slices of the (query tile, row tile) list;
// cb.nseg >= bf16_wide_segments(q_tiles, n_wg) (a workgroup's two row halves are two segments); cells without a writer
// must read "empty" unless n_wg is a multiple of q_tiles; scout_max in 256-row tiles
// tau_init (nullable): starting thresholds per query; radius: fixed thresholds, overflow instead of compaction (cap 256);
// scout_out (nullable): scout-only launch, lists as for launch_bf16_filter
int bf16_wide_segments(size_t q_tiles, int n_wg);
hipError_t launch_bf16_wide_filter(const void *img, size_t n, int dim, const void *B, int kp, const CandBuf &cb,
                                   int n_wg, int scout_max, const uint32_t *tau_init, bool radius, float *scout_out,
                                   hipStream_t s);
int bf16_scout_list();
int bf16_cell_max(int kp, int cap, int nseg, bool wide);  // entries a cell holds at most after a k-NN launch = CandBuf::final_keep to launch with
// out[q] = key just above the rank-th smallest value over the lists of q's nseg cells
// nq (0: nq_pad): the queries beyond it are padding and get the threshold -inf
hipError_t launch_bf16_seed(const float *lists, size_t nq_pad, int nseg, int rank, uint32_t *out, hipStream_t s,
                            size_t nq = 0);
hipError_t launch_bf16_radius_tau(const double *qn, size_t nq_pad, double tau_r, uint32_t *out, hipStream_t s);
// diagnostic: out[q][row] = L'(q, row), q < nq, row < n_rows
hipError_t launch_bf16_bound(const void *img, const void *B, size_t n_rows, size_t nq, int dim, float *out, bool ci,
                             hipStream_t s);

}  // namespace pn

#include "host_tree.h"  // tree.cpp: the reference's ball tree, built on the host only when its introspection API is used

// ---- index.hip, for the other translation units
struct pn_index;
namespace pn {
int set_error(int code, const char *fmt, ...);
int query_device_strided_f32(const pn_index *ix, const float *d_q, size_t nq, size_t q_cols, size_t q_stride, size_t k,
                             uint64_t *d_idx, float *d_dist, size_t out_stride, hipStream_t s);
int query_device_strided_f64(const pn_index *ix, const double *d_q, size_t nq, size_t q_cols, size_t q_stride, size_t k,
                             uint64_t *d_idx, double *d_dist, size_t out_stride, hipStream_t s);
int merge_topk_device_keys_f32(const uint64_t *pi, const float *pd, size_t np, size_t is, size_t ds, size_t nq, size_t kp,
                               size_t ko, uint64_t *oi, float *od, int device, void *stream, bool signed_keys);
int merge_topk_device_keys_f64(const uint64_t *pi, const double *pd, size_t np, size_t is, size_t ds, size_t nq, size_t kp,
                               size_t ko, uint64_t *oi, double *od, int device, void *stream, bool signed_keys);
}  // namespace pn
