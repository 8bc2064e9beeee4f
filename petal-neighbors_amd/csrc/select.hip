// select.hip -- final selection kernels (gfx950).
//
// After a scan kernel has left, per (segment, query), at most kp candidates in
// HBM, one wave per query gathers them into LDS, ranks them by the total order
// (distance key, index) -- Neighbor's order, reference src/ball_tree.rs:396-421,
// with ascending index inside equal-distance groups -- and writes the first
// kout.  In MFMA mode the candidates carry only a lower bound, so the kernel
// first recomputes every candidate's distance in the reference's operation
// order (src/distance.rs:26-35) and then PROVES that nothing the filter dropped
// could belong to the answer; queries it cannot prove are flagged for the exact
// engine.  Compiled with -ffp-contract=off.
#include "pn_internal.h"

namespace pn {

__device__ __forceinline__ uint32_t sel_key(float d) { return (d != d) ? KeyOf<float>::kNaN : __float_as_uint(d); }
__device__ __forceinline__ uint64_t sel_key(double d) {
    return (d != d) ? KeyOf<double>::kNaN : (uint64_t)__double_as_longlong(d);
}
__device__ __forceinline__ float key_to_dist(uint32_t k) { return __uint_as_float(k); }
__device__ __forceinline__ double key_to_dist(uint64_t k) { return __longlong_as_double((long long)k); }
// dist_key_signed of exact_scan.hip (the order-preserving map of ALL floats, -0 as +0, NaN above +inf): a Cosine index's
// distances can lie a few ulp below zero, so the shard merge orders them by these keys
__device__ __forceinline__ uint32_t sel_key_signed(float d) {
    if (d != d) return 0xFFC00000u;
    const uint32_t b = __float_as_uint(d == 0.0f ? 0.0f : d);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ uint64_t sel_key_signed(double d) {
    if (d != d) return 0xFFF8000000000000ull;
    const uint64_t b = (uint64_t)__double_as_longlong(d == 0.0 ? 0.0 : d);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
// inverse of dist_key_signed (exact_scan.hip): keys of a Cosine index
__device__ __forceinline__ float key_to_dist_signed(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__device__ __forceinline__ double key_to_dist_signed(uint64_t k) {
    return __longlong_as_double((long long)((k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

// rank of entry e among n LDS entries under (key, idx); one lane per entry
template <typename KeyT, typename IdxT>
__device__ __forceinline__ uint32_t rank_of(const KeyT *skey, const IdxT *sidx, uint32_t n, KeyT k, IdxT ix) {
    uint32_t r = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const KeyT bk = skey[j];
        const IdxT bi = sidx[j];
        r += ((bk < k) || (bk == k && bi < ix)) ? 1u : 0u;
    }
    return r;
}

// Shrinks n LDS entries to the (at most kout, barring duplicate entries) smallest under (key, idx), in place and
// in their original relative order, so that rank_of runs over kout entries instead of n: a radix select on the
// key (one ballot per 64 entries per bit), then on the index among the keys equal to the kout-th one.  One wave.
__device__ __forceinline__ uint32_t kth_smallest_lds(const uint32_t *a, uint32_t n, uint32_t k, int lane);
template <typename KeyT>
__device__ __forceinline__ KeyT kth_smallest_keys(const KeyT *a, uint32_t n, uint32_t k, int lane);
template <typename KeyT, typename IdxT>
__device__ __forceinline__ uint32_t prune_to_topk(KeyT *skey, IdxT *sidx, uint32_t n, uint32_t kout, int lane) {
    if (n <= kout || n <= 128) return n;
    KeyT T = 0;  // becomes the kout-th smallest key
    if (n <= 512u)
        T = kth_smallest_keys<KeyT>(skey, n, kout, lane);  // keys in registers
    else
    for (int b = (int)sizeof(KeyT) * 8 - 1; b >= 0; --b) {
        const KeyT cand = T | ((KeyT)1 << b);
        uint32_t c = 0;
        for (uint32_t e0 = 0; e0 < n; e0 += 64) {
            const uint32_t e = e0 + lane;
            c += (uint32_t)__popcll(__ballot(e < n && skey[e] < cand));
        }
        if (c < kout) T = cand;
    }
    uint32_t n_less = 0, n_eq = 0;
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {
        const uint32_t e = e0 + lane;
        const KeyT k = e < n ? skey[e] : (KeyT)0;
        n_less += (uint32_t)__popcll(__ballot(e < n && k < T));
        n_eq += (uint32_t)__popcll(__ballot(e < n && k == T));
    }
    const uint32_t need = kout - n_less;  // >= 1 entries with key == T belong to the answer
    IdxT I = ~(IdxT)0;
    if (n_eq > need) {
        I = 0;  // becomes the need-th smallest index among key == T
        for (int b = (int)sizeof(IdxT) * 8 - 1; b >= 0; --b) {
            const IdxT cand = I | ((IdxT)1 << b);
            uint32_t c = 0;
            for (uint32_t e0 = 0; e0 < n; e0 += 64) {
                const uint32_t e = e0 + lane;
                c += (uint32_t)__popcll(__ballot(e < n && skey[e] == T && sidx[e] < cand));
            }
            if (c < need) I = cand;
        }
    }
    uint32_t w = 0;
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {  // stable compaction; writes land at or below the chunk just read
        const uint32_t e = e0 + lane;
        KeyT k = 0;
        IdxT ix = 0;
        bool keep = false;
        if (e < n) {
            k = skey[e];
            ix = sidx[e];
            keep = (k < T) || (k == T && ix <= I);
        }
        const unsigned long long m = __ballot(keep);
        const uint32_t pos = w + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            skey[pos] = k;
            sidx[pos] = ix;
        }
        w += (uint32_t)__popcll(m);
    }
    return w;
}

// ---------------------------------------------------------------------------
// exact mode: block = 64 threads = one query
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void select_exact_kernel(const typename KeyOf<T>::type *__restrict__ ckey,
                                                          const uint32_t *__restrict__ cidx,
                                                          const uint32_t *__restrict__ ccnt, size_t nq_pad, int nseg,
                                                          int cap, int kout, uint64_t index_base,
                                                          uint64_t *__restrict__ idx_out, T *__restrict__ dist_out,
                                                          size_t out_stride, size_t out_off,
                                                          typename KeyOf<T>::type *__restrict__ lo_key,
                                                          uint32_t *__restrict__ lo_idx,
                                                          const uint32_t *__restrict__ nq_dev, uint32_t nq_off,
                                                          int signed_keys, size_t out_group_stride,
                                                          const uint32_t *__restrict__ osel) {
    using KeyT = typename KeyOf<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    if (nq_dev) {  // device-driven query count (see launch_exact_knn_*)
        const uint32_t tot = *nq_dev;
        if (q >= (size_t)(tot > nq_off ? tot - nq_off : 0u)) return;
    }
    const size_t qo = osel ? (size_t)osel[nq_off + q] : q;  // output row (second tier: the flagged query's own row)
    // blockIdx.y = segment group (two-level selection over many segments): group g selects among segments
    // [g nseg, (g+1) nseg) into its own output part
    ckey += (size_t)blockIdx.y * (size_t)nseg * nq_pad * (size_t)cap;
    cidx += (size_t)blockIdx.y * (size_t)nseg * nq_pad * (size_t)cap;
    ccnt += (size_t)blockIdx.y * (size_t)nseg * nq_pad;
    idx_out += (size_t)blockIdx.y * out_group_stride;
    dist_out += (size_t)blockIdx.y * out_group_stride;
    int total_cap = 0;
    for (int s = 0; s < nseg; ++s) total_cap += (int)ccnt[(size_t)s * nq_pad + q];
    KeyT *skey = reinterpret_cast<KeyT *>(smem);
    uint32_t *sidx = reinterpret_cast<uint32_t *>(smem + sizeof(KeyT) * (size_t)total_cap);
    uint32_t n = 0;
    for (int s = 0; s < nseg; ++s) {
        const uint32_t c = ccnt[(size_t)s * nq_pad + q];
        const size_t base = ((size_t)s * nq_pad + q) * (size_t)cap;
        for (uint32_t e = lane; e < c; e += 64) {
            skey[n + e] = ckey[base + e];
            sidx[n + e] = cidx[base + e];
        }
        n += c;
    }
    __syncthreads();
    const uint32_t n_all = n;
    n = prune_to_topk<KeyT, uint32_t>(skey, sidx, n, (uint32_t)kout, lane);
    __syncthreads();
    for (uint32_t e = lane; e < n; e += 64) {
        const KeyT k = skey[e];
        const uint32_t ix = sidx[e];
        const uint32_t r = rank_of<KeyT, uint32_t>(skey, sidx, n, k, ix);
        if (r < (uint32_t)kout) {
            idx_out[qo * out_stride + out_off + r] = index_base + ix;
            dist_out[qo * out_stride + out_off + r] = signed_keys ? key_to_dist_signed(k) : key_to_dist(k);
            if (lo_key && r == (uint32_t)kout - 1) {  // next round resumes strictly after this entry
                lo_key[q] = k;
                lo_idx[q] = ix;
            }
        }
    }
    for (uint32_t r = n_all + lane; r < (uint32_t)kout; r += 64) {  // cannot happen for kout = min(k, n_points)
        idx_out[qo * out_stride + out_off + r] = ~0ull;
        dist_out[qo * out_stride + out_off + r] = key_to_dist(KeyOf<T>::kNaN);
    }
}

template <typename T>
static hipError_t launch_select_exact(const CandBuf &cb, int nq, int kout, uint64_t index_base, uint64_t *idx_out,
                                      T *dist_out, int kp_bound, size_t out_stride, size_t out_off, void *lo_key,
                                      uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, bool signed_keys,
                                      hipStream_t s, int groups = 1, size_t out_group_stride = 0,
                                      const uint32_t *osel = nullptr) {
    using KeyT = typename KeyOf<T>::type;
    // cb.nseg = segments per group; every cell holds at most kp_bound entries
    const size_t sh = (size_t)cb.nseg * (size_t)kp_bound * (sizeof(KeyT) + sizeof(uint32_t));
    if (sh > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL((select_exact_kernel<T>), dim3((unsigned)nq, (unsigned)groups), dim3(64), sh, s,
                       static_cast<const KeyT *>(cb.keys), cb.idx, cb.cnt, cb.nq_pad, cb.nseg, cb.cap, kout,
                       index_base, idx_out, dist_out, out_stride, out_off, static_cast<KeyT *>(lo_key), lo_idx, nq_dev,
                       nq_off, signed_keys ? 1 : 0, out_group_stride, osel);
    return hipGetLastError();
}
// two-level selection, first level: `groups` groups of cb.nseg segments each (cb describes group 0, the groups follow
// each other in the candidate buffers); group g's top kout of every query to idx_out/dist_out + g * out_group_stride.
// kp: entries a cell holds at most (the scan kernel's final compaction leaves <= its kp)
hipError_t launch_select_exact_groups_f32(const CandBuf &cb, int groups, int kp, int nq, int kout, uint64_t index_base,
                                          uint64_t *idx_out, float *dist_out, size_t out_group_stride,
                                          const uint32_t *nq_dev, uint32_t nq_off, hipStream_t s, bool signed_keys) {
    return launch_select_exact<float>(cb, nq, kout, index_base, idx_out, dist_out, kp, (size_t)kout, 0, nullptr, nullptr,
                                      nq_dev, nq_off, signed_keys, s, groups, out_group_stride);
}
hipError_t launch_select_exact_groups_f64(const CandBuf &cb, int groups, int kp, int nq, int kout, uint64_t index_base,
                                          uint64_t *idx_out, double *dist_out, size_t out_group_stride,
                                          const uint32_t *nq_dev, uint32_t nq_off, hipStream_t s, bool signed_keys) {
    return launch_select_exact<double>(cb, nq, kout, index_base, idx_out, dist_out, kp, (size_t)kout, 0, nullptr, nullptr,
                                       nq_dev, nq_off, signed_keys, s, groups, out_group_stride);
}
hipError_t launch_select_exact_f32(const CandBuf &cb, int nq, int kout, uint64_t index_base, uint64_t *idx_out,
                                   float *dist_out, size_t out_stride, size_t out_off, void *lo_key,
                                   uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, bool signed_keys,
                                   hipStream_t s, const uint32_t *osel) {
    return launch_select_exact<float>(cb, nq, kout, index_base, idx_out, dist_out, cb.cap, out_stride, out_off,
                                      lo_key, lo_idx, nq_dev, nq_off, signed_keys, s, 1, 0, osel);
}
hipError_t launch_select_exact_f64(const CandBuf &cb, int nq, int kout, uint64_t index_base, uint64_t *idx_out,
                                   double *dist_out, size_t out_stride, size_t out_off, void *lo_key,
                                   uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, bool signed_keys,
                                   hipStream_t s, const uint32_t *osel) {
    return launch_select_exact<double>(cb, nq, kout, index_base, idx_out, dist_out, cb.cap, out_stride, out_off,
                                       lo_key, lo_idx, nq_dev, nq_off, signed_keys, s, 1, 0, osel);
}

// ---------------------------------------------------------------------------
// MFMA mode: exact re-rank + verification.  block = 64 threads = one query.
//
// Candidates: per segment up to `cap` (idx, L) pairs with L <= d2(q, p) in real
// arithmetic for every corpus row p (mfma_filter_v2.hip proves this bound), and
// tau_seg such that every row of the segment NOT in the buffer has L >= tau_seg
// (tau_seg = +inf when nothing was ever dropped).
//
// Reference value: s_ref = fl-fold >= d2 * (1 - (D+3)u) - D*2^-149  (u = 2^-24):
// three roundings per term, D additions, gradual underflow.  Distances are
// f32 sqrt(s_ref), correctly rounded, hence monotone in s_ref.
// A dropped row is provably outside the answer when
//     tau_seg * (1 - (D+4)u) - 1e-37  >  ((d_k + succ(d_k)) / 2)^2
// (everything at or below the right-hand side may still round to d_k and tie).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float exact_distance_f32(const float *__restrict__ q, const float *__restrict__ p,
                                                    int dim) {
#pragma clang fp contract(off)
    float s = 0.0f;
    int k = 0;
    for (; k + 4 <= dim; k += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(q + k);
        const float4 b = *reinterpret_cast<const float4 *>(p + k);
        float d;
        d = a.x - b.x; s = s + d * d;
        d = a.y - b.y; s = s + d * d;
        d = a.z - b.z; s = s + d * d;
        d = a.w - b.w; s = s + d * d;
    }
    for (; k < dim; ++k) {
        const float d = q[k] - p[k];
        s = s + d * d;
    }
    return sqrtf(s);
}

// The same fold with the row PREFETCHED: 16 independent 16-byte loads are issued before the first of them is
// consumed (64 coordinates per round), and the query is read from LDS.  The plain loop above waits for memory once
// per four coordinates; with a handful of lanes active per wave that latency, not arithmetic or bandwidth, was the
// whole cost of the re-rank kernel.  len must be a multiple of 4 (device rows are zero padded to a multiple of 8:
// the padding adds (0-0)*(0-0) = +0, which leaves every partial sum unchanged).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef PN_DIAG_RR_PREFETCH
#define PN_DIAG_RR_PREFETCH 16
#endif
__device__ __forceinline__ float exact_distance_prefetched_f32(const float *qs, const float *__restrict__ p, int len) {
#pragma clang fp contract(off)
    float s = 0.0f;
    constexpr int NV = PN_DIAG_RR_PREFETCH;  // 16-byte loads in flight per round
    for (int k0 = 0; k0 < len; k0 += 4 * NV) {
        f32x4_t v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (k0 + 4 * i < len) v[i] = *reinterpret_cast<const f32x4_t *>(p + k0 + 4 * i);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (k0 + 4 * i < len) {
                const f32x4_t a = *reinterpret_cast<const f32x4_t *>(qs + k0 + 4 * i);
                float d;
                d = a.x - v[i].x; s = s + d * d;
                d = a.y - v[i].y; s = s + d * d;
                d = a.z - v[i].z; s = s + d * d;
                d = a.w - v[i].w; s = s + d * d;
            }
        }
    }
    return sqrtf(s);
}

// k-th smallest (1-based) of n LDS keys, one wave, array left untouched: radix select.  Requires 1 <= k <= n.
// W keys per lane held in registers: one round per bit of compare + ballot + scalar popcount, no LDS round trip in them
// (a round over LDS is latency-bound: ~100 cycles per 64 keys and bit; the re-rank of the headline batch runs three such
// selections over ~150 candidates per query, a 125 k-row shard's over ~220 with a tail beyond 256)
template <typename KeyT, int W>
__device__ __forceinline__ KeyT kth_smallest_regs(const KeyT *a, uint32_t n, uint32_t k, int lane) {
    KeyT v[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t e = (uint32_t)lane + 64u * (uint32_t)i;
        v[i] = e < n ? a[e] : ~(KeyT)0;  // padding: never below a candidate threshold
    }
    KeyT T = 0;
    for (int b = (int)sizeof(KeyT) * 8 - 1; b >= 0; --b) {
        const KeyT cand = T | ((KeyT)1 << b);
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) c += (uint32_t)__popcll(__ballot(v[i] < cand));
        if (c < k) T = cand;
    }
    return T;
}
// ... of n LDS words: in registers up to 512 of them, else one ballot per 64 words and bit over LDS
__device__ __forceinline__ uint32_t kth_smallest_lds(const uint32_t *a, uint32_t n, uint32_t k, int lane) {
    uint32_t T = 0;
#ifndef PN_DIAG_SELECT_LDS
    if (n <= 128u) return kth_smallest_regs<uint32_t, 2>(a, n, k, lane);
    if (n <= 192u) return kth_smallest_regs<uint32_t, 3>(a, n, k, lane);
    if (n <= 256u) return kth_smallest_regs<uint32_t, 4>(a, n, k, lane);
    if (n <= 384u) return kth_smallest_regs<uint32_t, 6>(a, n, k, lane);
    if (n <= 512u) return kth_smallest_regs<uint32_t, 8>(a, n, k, lane);
#endif
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        uint32_t c = 0;
        for (uint32_t e0 = 0; e0 < n; e0 += 64) {
            const uint32_t e = e0 + lane;
            c += (uint32_t)__popcll(__ballot(e < n && a[e] < cand));
        }
        if (c < k) T = cand;
    }
    return T;
}

// ---- element-type generic pieces of the re-rank (T = float: the f32 functions above; T = double: an f64 index served
// by the bf16 filter -- the bound is a statement about real vectors, only this kernel's arithmetic follows the type)
template <typename T> __device__ __forceinline__ T exact_distance_seq(const T *__restrict__ q, const T *__restrict__ p, int dim);
template <> __device__ __forceinline__ float exact_distance_seq<float>(const float *__restrict__ q, const float *__restrict__ p, int dim) {
    return exact_distance_f32(q, p, dim);
}
template <> __device__ __forceinline__ double exact_distance_seq<double>(const double *__restrict__ q, const double *__restrict__ p, int dim) {
#pragma clang fp contract(off)
    double s = 0.0;
    for (int k = 0; k < dim; ++k) {  // Euclidean::distance, src/distance.rs:26-35: sub, mul, add separately rounded, ascending k
        const double d = q[k] - p[k];
        s = s + d * d;
    }
    return sqrt(s);
}
template <typename T> __device__ __forceinline__ T exact_distance_prefetched(const T *qs, const T *__restrict__ p, int len);
template <> __device__ __forceinline__ float exact_distance_prefetched<float>(const float *qs, const float *__restrict__ p, int len) {
    return exact_distance_prefetched_f32(qs, p, len);
}
// f64: 16 independent 16-byte loads (32 coordinates) in flight per round; the query from LDS; len a multiple of 8
typedef double f64x2_t __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ double exact_distance_prefetched<double>(const double *qs, const double *__restrict__ p, int len) {
#pragma clang fp contract(off)
    double s = 0.0;
    for (int k0 = 0; k0 < len; k0 += 32) {
        f64x2_t v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (k0 + 2 * i < len) v[i] = *reinterpret_cast<const f64x2_t *>(p + k0 + 2 * i);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (k0 + 2 * i < len) {
                const f64x2_t a = *reinterpret_cast<const f64x2_t *>(qs + k0 + 2 * i);
                double d;
                d = a.x - v[i].x; s = s + d * d;
                d = a.y - v[i].y; s = s + d * d;
            }
        }
    }
    return sqrt(s);
}
// k-th smallest (1-based) of n LDS keys of either width
template <typename KeyT>
__device__ __forceinline__ KeyT kth_smallest_keys(const KeyT *a, uint32_t n, uint32_t k, int lane) {
    if constexpr (sizeof(KeyT) == 4) {
        return (KeyT)kth_smallest_lds(reinterpret_cast<const uint32_t *>(a), n, k, lane);
    } else {
        if (n <= 256u) return kth_smallest_regs<KeyT, 4>(a, n, k, lane);
        if (n <= 512u) return kth_smallest_regs<KeyT, 8>(a, n, k, lane);
        KeyT Tk = 0;
        for (int b = (int)sizeof(KeyT) * 8 - 1; b >= 0; --b) {
            const KeyT cand = Tk | ((KeyT)1 << b);
            uint32_t c = 0;
            for (uint32_t e0 = 0; e0 < n; e0 += 64) {
                const uint32_t e = e0 + lane;
                c += (uint32_t)__popcll(__ballot(e < n && a[e] < cand));
            }
            if (c < k) Tk = cand;
        }
        return Tk;
    }
}
// The proof's two sides (header of this section).  rhs: everything at or below it may still round to d_k (key kk, finite,
// >= +0) and tie; lb: lower bound of the REFERENCE's folded sum for a row whose bound of d2 - |q|^2 is L (tagged: within
// 2^-19 of the bound itself), qadd <= |q|^2.
//   f32 distances: the sides are evaluated in f64, 29 bits to spare;
//   f64 distances: the same inequalities with u = 2^-53, evaluated in f64 itself, so each side is pushed outward by a few
//     ulp for its own roundings -- rhs = succ(d_k)^2 (1 + 2^-51) >= ((d_k + succ d_k) / 2)^2, lb times (1 - 2^-50).
template <typename T> __device__ __forceinline__ double proof_rhs(typename KeyOf<T>::type kk);
template <> __device__ __forceinline__ double proof_rhs<float>(uint32_t kk) {
    const double dk = (double)__uint_as_float(kk), dn = (double)__uint_as_float(kk + 1);  // succ(d_k): next float up
    const double mid = 0.5 * (dk + dn);
    return mid * mid * (1.0 + 4.5e-16);
}
template <> __device__ __forceinline__ double proof_rhs<double>(uint64_t kk) {
    const double dn = __longlong_as_double((long long)(kk + 1));  // succ(d_k)
    return dn * dn * (1.0 + 4.440892098500626e-16);
}
template <typename T> __device__ __forceinline__ double proof_lb(double L, double qadd, int dim);
template <> __device__ __forceinline__ double proof_lb<float>(double L, double qadd, int dim) {
    return ((L - fabs(L) * 1.9073486328125e-06) + qadd) * (1.0 - (double)(dim + 4) * 5.9604644775390625e-08) - 1e-37;
}
template <> __device__ __forceinline__ double proof_lb<double>(double L, double qadd, int dim) {
    return ((L - fabs(L) * 1.9073486328125e-06) + qadd) * (1.0 - (double)(dim + 4) * 1.1102230246251565e-16) *
               (1.0 - 8.881784197001252e-16) - 1e-300;
}

// ---- Cosine indexes behind the bf16 filter (round 4).  The filter ran on the rows and queries NORMALISED in f64
// (p~ = p / |p|, q~ = q / |q|: index.hip) and bounds |q~ - p~|^2, which in real arithmetic is 2 (1 - cos) = twice the
// Cosine distance; the candidates are evaluated here in the REFERENCE's arithmetic (src/distance.rs:85-107: dot, the two
// squared norms each a sequential fold of separately rounded products, norm1 * norm2, one division, one subtraction),
// with |p| from the index (cosine_norms_kernel: the same fold) and |q| from the call's query norms.
template <typename T>
__device__ __forceinline__ T exact_cosine_prefetched(const T *qs, const T *__restrict__ p, int len, T qnorm, T pnorm) {
#pragma clang fp contract(off)
    constexpr int VE = 16 / (int)sizeof(T);
    typedef T tv_ __attribute__((ext_vector_type(VE)));
    T dot = (T)0;
    for (int k0 = 0; k0 < len; k0 += 16 * VE) {  // 16 independent 16-byte loads in flight per round (len a multiple of 8)
        tv_ v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (k0 + VE * i < len) v[i] = *reinterpret_cast<const tv_ *>(p + k0 + VE * i);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (k0 + VE * i < len) {
                const tv_ a = *reinterpret_cast<const tv_ *>(qs + k0 + VE * i);
#pragma unroll
                for (int j = 0; j < VE; ++j) {
                    const T pr = a[j] * v[i][j];  // (zero padding: + 0 changes nothing)
                    dot = dot + pr;
                }
            }
        }
    }
    const T den = qnorm * pnorm;
    return (T)1 - dot / den;
}
// Lower bound of the REFERENCE's Cosine::distance d_ref(q, p) for a row whose (tagged) filter bound is L, qadd <= |q~|^2
// (minus E(q) in the CI layout), D = dim, u = the type's unit roundoff:
//   |q~ - p~|^2 >= (L - |L| 2^-19) + qadd                                   (the filter's guarantee, as for Euclidean)
//   | |q~ - p~|^2 - 2 d* | <= eps1 = (4.5 D + 45) 2^-53                      (d* = 1 - cos in real arithmetic; q~, p~ are
//        the f64-computed normalisations: every coordinate within (D/2 + 5) 2^-53 relative of q_k / |q|)
//   | d_ref - d* | <= E' = (2 D + 16) u                                     (dot: gamma_D |q||p|; the norms' folds and sqrt,
//        their product, the division, the subtraction; norms^2 in [2^-100, 2^100] keep underflow below D 2^-50 u)
// so d_ref >= ((L - |L| 2^-19) + qadd - eps1) / 2 - E', evaluated in f64 and pushed down for its own roundings.
template <typename T> __device__ __forceinline__ double cos_proof_lb(double L, double qadd, int dim) {
    const double u = sizeof(T) == 4 ? 5.9604644775390625e-08 : 1.1102230246251565e-16;
    const double x = (L - fabs(L) * 1.9073486328125e-06) + qadd;
    const double eps1 = (4.5 * (double)dim + 45.0) * 1.1102230246251565e-16;
    const double e = (2.0 * (double)dim + 16.0) * u;
    return 0.5 * (x - eps1) - e - (fabs(x) + 1.0) * 1.0e-15;
}

#ifdef PN_DIAG_RR_STAMP  // diagnostic build only: where a re-rank wave's time goes (cycle sums per phase)
// per-query phase durations [16384][8], written once per wave at its end (an atomic per phase on a shared counter
// would itself be what the waves wait for)
__device__ uint32_t g_rrdbg[16384 * 8];
#define RR_STAMP(i)                                                                   \
    do {                                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
        rr_d_[i] = (uint32_t)(t_ - rr_t_);                                            \
        rr_t_ = t_;                                                                   \
    } while (0)
extern "C" int pn_debug_read_rr(unsigned long long *out, int nq) {
    static uint32_t h[16384 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rrdbg), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (int q = 0; q < nq && q < 16384; ++q)
        for (int i = 0; i < 8; ++i) out[i] += h[q * 8 + i];
    return 0;
}
#else
#define RR_STAMP(i) ((void)0)
#endif
#ifndef PN_DIAG_RR_WAVES
#define PN_DIAG_RR_WAVES 1
#endif
template <typename T, bool COS = false>
__global__ __launch_bounds__(64, PN_DIAG_RR_WAVES) void select_rerank_kernel(
    const uint32_t *__restrict__ ctau, const uint32_t *__restrict__ cidx, const uint32_t *__restrict__ ccnt,
    size_t nq_pad, int nseg, int cap, const T *__restrict__ P, size_t ldp, const T *__restrict__ Q,
    size_t ldq, int dim, uint32_t n_rows, int kout, uint64_t index_base, uint64_t *__restrict__ idx_out,
    T *__restrict__ dist_out, size_t out_stride, uint32_t *__restrict__ flags, uint32_t *__restrict__ n_flagged,
    const double *__restrict__ qn, const uint32_t *__restrict__ qbad,
    int idx_stride, const uint32_t *__restrict__ ckey, uint32_t *__restrict__ sel,
    unsigned long long *__restrict__ stats, uint32_t first_eval, const T *__restrict__ cnorm = nullptr,
    const T *__restrict__ qnorm = nullptr) {
    // COS: a Cosine index (cnorm = the rows' norms, qnorm = the queries'): candidates are evaluated by Cosine::distance,
    // keys are the order-preserving map of ALL floats (a distance can lie a few ulp below zero), the proof is cos_proof_lb
    using KeyT = typename KeyOf<T>::type;
    constexpr KeyT KMAX = KeyOf<T>::kMax;                       // an entry whose distance has not been evaluated
    const KeyT KINF = COS ? sel_key_signed((T)__builtin_huge_val()) : sel_key((T)__builtin_huge_val());  // keys at or above it are inf / NaN
    auto eval_row = [&](const T *qv, uint32_t ix, int ln) -> KeyT {
        if constexpr (COS) return sel_key_signed(exact_cosine_prefetched<T>(qv, P + (size_t)ix * ldp, ln, qnorm[blockIdx.x], cnorm[ix]));
        else return sel_key(exact_distance_prefetched<T>(qv, P + (size_t)ix * ldp, ln));
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ KeyT kth_key;
    __shared__ uint32_t seg_off[65], seg_cnt[64];
    __shared__ uint32_t elist[64];  // slots of the candidates one evaluation round takes, a lane each
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
#ifdef PN_DIAG_RR_STAMP
    unsigned long long rr_t_ = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t rr_d_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // Gather phase with few dependent memory round trips (a wave per query spends its time waiting, not computing):
    // the lanes read the cells' counts and thresholds side by side, a wave scan turns the counts into offsets, and
    // every lane then finds the cell of "its" candidate by itself -- instead of walking the cells one after another.
    uint32_t total_cap = 0;
    uint32_t min_key = 0xFF800000u;  // sortable(+inf): nothing dropped
    for (int s0 = 0; s0 < nseg; s0 += 64) {
        const int sg = s0 + lane;
        const uint32_t c = sg < nseg ? ccnt[(size_t)sg * nq_pad + q] : 0u;
        const uint32_t t = sg < nseg ? ctau[(size_t)sg * nq_pad + q] : 0xFF800000u;  // order-preserving keys (f2s)
        uint32_t m = t, a = c;
        for (int d = 32; d > 0; d >>= 1) {
            const uint32_t om = (uint32_t)__shfl_xor((int)m, d);
            m = om < m ? om : m;
            a += (uint32_t)__shfl_xor((int)a, d);
        }
        min_key = m < min_key ? m : min_key;
        total_cap += a;
    }
    const float min_tau = __uint_as_float((min_key & 0x80000000u) ? (min_key & 0x7FFFFFFFu) : ~min_key);
    RR_STAMP(0);  // counts + thresholds read and reduced
    // LDS: the query row FIRST (zero padded like the corpus rows; its 16-byte reads need a 16-byte aligned base -- behind
    // arrays of total_cap entries it was misaligned for three counts in four, and a misaligned ds_read_b128 is replayed),
    // then the exact keys, the rows, the filter's keys
    const int len = (int)((dim + 7) / 8 * 8);  // <= ldp, ldq: both are padded to a multiple of 8 with zeros
    T *qs = reinterpret_cast<T *>(smem);
    KeyT *skey = reinterpret_cast<KeyT *>(qs + len);
    uint32_t *sidx = reinterpret_cast<uint32_t *>(skey + total_cap);
    uint32_t *sfk = sidx + total_cap;  // the filter's keys (when given)
    if (lane == 0) kth_key = KeyOf<T>::kNaN;
    const T *qrow = Q + q * ldq;
    for (int k = lane; k < len; k += 64) qs[k] = qrow[k];
    const double u = 5.9604644775390625e-08;  // 2^-24
    uint32_t n = 0, evaluated = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) {
        // offsets of up to 64 cells (inclusive wave scan), then one candidate per lane and round
        const int sg = s0 + lane;
        const uint32_t c = sg < nseg ? ccnt[(size_t)sg * nq_pad + q] : 0u;
        uint32_t inc = c;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, d);
            if (lane >= d) inc += up;
        }
        __syncthreads();  // previous chunk's readers are done with seg_off / seg_cnt
        seg_off[lane] = inc - c;
        seg_cnt[lane] = c;
        const uint32_t chunk_total = (uint32_t)__shfl((int)inc, 63);
        if (lane == 0) seg_off[64] = chunk_total;
        __syncthreads();
        const int nsc = nseg - s0 < 64 ? nseg - s0 : 64;
        if (ckey) {
            // four rounds of 64 candidates at a time: all their (row, key) loads are in flight together -- a round
            // that waits for its own loads before the next one starts costs a memory round trip per 64 candidates
            for (uint32_t e0 = 0; e0 < chunk_total; e0 += 256) {
                uint32_t ixr[4], fkr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t e = e0 + 64u * (uint32_t)r + (uint32_t)lane;
                    ixr[r] = 0xFFFFFFFFu;
                    fkr[r] = 0xFFFFFFFFu;
                    if (e < chunk_total) {
                        int lo = 0, hi = nsc - 1;  // last cell whose offset is <= e (empty cells are skipped by <=)
                        while (lo < hi) {
                            const int mid = (lo + hi + 1) >> 1;
                            if (seg_off[mid] <= e) lo = mid; else hi = mid - 1;
                        }
                        const size_t src = (((size_t)(s0 + lo) * nq_pad + q) * (size_t)cap + (e - seg_off[lo])) * (size_t)idx_stride;
                        if (idx_stride == 2) {  // (key, row) pairs of the bf16 filter: one 8-byte load
                            const uint2 pr = *reinterpret_cast<const uint2 *>(ckey + src);
                            fkr[r] = pr.x;
                            ixr[r] = pr.y;
                        } else {
                            ixr[r] = cidx[src];
                            fkr[r] = ckey[src];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t e = e0 + 64u * (uint32_t)r + (uint32_t)lane;
                    if (e < chunk_total) {
                        sidx[n + e] = ixr[r];
                        sfk[n + e] = fkr[r];
                        skey[n + e] = KMAX;
                    }
                }
            }
        } else
        for (uint32_t e0 = 0; e0 < chunk_total; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < chunk_total) {
                int lo = 0, hi = nsc - 1;  // last cell whose offset is <= e (cells with zero entries are skipped by <=)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (seg_off[mid] <= e) lo = mid; else hi = mid - 1;
                }
                const size_t src = (((size_t)(s0 + lo) * nq_pad + q) * (size_t)cap + (e - seg_off[lo])) * (size_t)idx_stride;
                const uint32_t ix = cidx[src];
                sidx[n + e] = ix;
                // a row number beyond the corpus (never produced; defensive) sorts behind everything
                skey[n + e] = ix < n_rows ? sel_key(exact_distance_seq<T>(qrow, P + (size_t)ix * ldp, dim)) : KMAX;
            }
        }
        n += chunk_total;
    }
    RR_STAMP(1);  // candidates gathered into LDS
    if (!ckey) {
        evaluated = n;
    } else {
        // The filter's keys are lower bounds of the candidates' squared distances (minus |q|^2 with qn), so most
        // candidates never need their exact distance: evaluate the kout with the smallest bounds, take the kout-th
        // exact distance d among them, and evaluate further only the candidates whose bound does not PROVE them
        // farther than d (same inequality as the proof below).  Unevaluated entries keep the key 0xFFFFFFFF.
        // (round 2: the first round takes the first_eval >= kout smallest bounds -- the plan's estimate of how many
        // candidates need their distance anyway (C2: 24 of 136; 22.8 were evaluated in two rounds before) -- so that the
        // second round, a second pair of dependent memory round trips, is empty for most queries)
        __syncthreads();
        const uint32_t m1 = first_eval > (uint32_t)kout ? first_eval : (uint32_t)kout;
        bool all = n <= m1;
        uint32_t K1 = 0xFFFFFFFFu;
        if (!all) K1 = kth_smallest_lds(sfk, n, m1, lane);
        RR_STAMP(2);  // k-th smallest bound
        // The entries to evaluate are scattered over the n slots (24 of ~150 on the headline batch): evaluated where
        // they lie, every 64-slot chunk that holds one costs a whole round -- dependent row fetches and the D-step
        // chain -- for a handful of busy lanes.  So their positions are first packed into lists of 64 (ballot + prefix
        // count per chunk, no memory traffic), and a list is evaluated with one candidate per lane: one round for the
        // headline batch's first evaluation instead of three.  `pred` must not depend on what the evaluation writes.
        KeyT lane_key = KMAX;      // the key this lane evaluated in the latest list
        uint32_t list_total = 0;   // entries the latest call evaluated
        auto eval_packed = [&](auto pred) {
            // (up to 64 chunks: a lane remembers its slots' verdicts in a mask, so `pred` -- f64 arithmetic in the
            // second round -- runs once per slot)
            const bool cached = n <= 4096u;
            unsigned long long mine = 0ull;
            uint32_t total = 0;
            for (uint32_t c = 0, e0 = 0; e0 < n; e0 += 64, ++c) {
                const bool go = pred(e0 + (uint32_t)lane);
                if (cached && go) mine |= 1ull << c;
                total += (uint32_t)__popcll(__ballot(go));
            }
            for (uint32_t b0 = 0; b0 < total; b0 += 64) {
                uint32_t w = 0;
                for (uint32_t c = 0, e0 = 0; e0 < n && w < b0 + 64u; e0 += 64, ++c) {
                    const uint32_t e = e0 + (uint32_t)lane;
                    const bool go = cached ? ((mine >> c) & 1ull) != 0ull : pred(e);
                    const unsigned long long m = __ballot(go);
                    const uint32_t pos = w + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (go && pos >= b0 && pos < b0 + 64u) elist[pos - b0] = e;
                    w += (uint32_t)__popcll(m);
                }
                __syncthreads();
                lane_key = KMAX;
                if (b0 + (uint32_t)lane < total) {
                    const uint32_t e = elist[lane];
                    const uint32_t ix = sidx[e];
                    if (ix < n_rows) {
                        lane_key = eval_row(qs, ix, len);
                        skey[e] = lane_key;
                    }
                }
                __syncthreads();
            }
            evaluated += total;
            list_total = total;
        };
        eval_packed([&](uint32_t e) { return e < n && (all || sfk[e] <= K1); });
        RR_STAMP(3);  // first evaluation round
        if (!all) {
            // the kout-th smallest exact distance so far: a single list's keys are still in the lanes' registers
            KeyT dk1;
            if (list_total <= 64u) {
                dk1 = 0;
                for (int b = (int)sizeof(KeyT) * 8 - 1; b >= 0; --b) {
                    const KeyT cand = dk1 | ((KeyT)1 << b);
                    if ((uint32_t)__popcll(__ballot(lane_key < cand)) < (uint32_t)kout) dk1 = cand;
                }
            } else {
                dk1 = kth_smallest_keys<KeyT>(skey, n, (uint32_t)kout, lane);
            }
            RR_STAMP(4);  // k-th smallest exact distance so far
            const bool prune = dk1 < KINF;  // finite: else every candidate is evaluated
            double rhs = 0.0;
            if (prune) rhs = COS ? (double)key_to_dist_signed(dk1) : proof_rhs<T>(dk1);
            const double qadd = qn ? qn[q] : 0.0;
            // (the first round evaluated exactly the entries with a bound <= K1: the rest are the unevaluated ones)
            eval_packed([&](uint32_t e) {
                if (e >= n) return false;
                const uint32_t fk = sfk[e];
                if (fk <= K1) return false;
                if (!prune) return true;
                const double L = (double)__uint_as_float((fk & 0x80000000u) ? (fk & 0x7FFFFFFFu) : ~fk);
                // not provably farther than the kout-th exact distance so far
                return !((COS ? cos_proof_lb<T>(L, qadd, dim) : proof_lb<T>(L, qadd, dim)) > rhs);
            });
        }
    }
    __syncthreads();
    RR_STAMP(5);  // second evaluation round
    const uint32_t n_all = n;
    if (ckey) {
        // only evaluated candidates can be among the answers: move them to the front (stable, in place: a chunk's
        // writes land at or below what it has just read), so that selection and ranking run over ~2k entries, not n
        uint32_t w = 0;
        for (uint32_t e0 = 0; e0 < n; e0 += 64) {
            const uint32_t e = e0 + lane;
            const KeyT k = e < n ? skey[e] : KMAX;
            const uint32_t ix = e < n ? sidx[e] : 0u;
            const bool keep = k != KMAX;
            const unsigned long long m = __ballot(keep);
            const uint32_t pos = w + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (keep) {
                skey[pos] = k;
                sidx[pos] = ix;
            }
            w += (uint32_t)__popcll(m);
        }
        n = w;
        __syncthreads();
    }
    n = prune_to_topk<KeyT, uint32_t>(skey, sidx, n, (uint32_t)kout, lane);
    __syncthreads();
    RR_STAMP(6);  // cut to the k smallest
    for (uint32_t e = lane; e < n; e += 64) {
        const KeyT k = skey[e];
        const uint32_t ix = sidx[e];
        const uint32_t r = rank_of<KeyT, uint32_t>(skey, sidx, n, k, ix);
        if (r < (uint32_t)kout) {
            idx_out[q * out_stride + r] = index_base + ix;
            dist_out[q * out_stride + r] = COS ? key_to_dist_signed(k) : key_to_dist(k);
            if (r == (uint32_t)kout - 1) kth_key = k;
        }
    }
    __syncthreads();
    RR_STAMP(7);  // ranked and written
#ifdef PN_DIAG_RR_STAMP
    if (lane < 8 && q < 16384) {
        uint32_t v_ = 0;
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) v_ = lane == i_ ? rr_d_[i_] : v_;
        g_rrdbg[q * 8 + lane] = v_;
    }
#endif
    if (lane == 0) {
        bool ok = n_all >= (uint32_t)kout;
        const KeyT kk = kth_key;
        if (ok && kk >= KINF) ok = false;  // k-th distance is inf/NaN: let the exact engine order it
        if (ok && min_tau < __uint_as_float(0x7F800000u)) {
            if constexpr (COS) {
                // every dropped row's reference distance is >= lb; strictly above the k-th one found: it cannot tie either
                ok = (min_tau == min_tau) && qn && (cos_proof_lb<T>((double)min_tau, qn[q], dim) > (double)key_to_dist_signed(kk));
            } else {
                const double rhs = proof_rhs<T>(kk);
                // bf16 filter: the thresholds are TAGGED bounds of d2 - |q|^2 (bf16_filter.hip): within 2^-19 relative
                // of the bound itself; qn[q] <= |q|^2
                const double lb = qn ? proof_lb<T>((double)min_tau, qn[q], dim)
                                     : (double)min_tau * (1.0 - (double)(dim + 4) * u) - 1e-37;
                ok = (min_tau == min_tau) && (lb > rhs);
            }
        }
        if (qbad && qbad[q]) ok = false;
        flags[q] = ok ? 0u : 1u;
        if (!ok) {  // the second tier's work list is built here (any order: answers go back by query number)
            const uint32_t slot = atomicAdd(n_flagged, 1u);
            if (sel) sel[slot] = (uint32_t)q;
            if (stats) atomicAdd(stats, 1ull);
        }
        // statistics: kPnStatSlots pairs of running counters, a query adds to pair q mod kPnStatSlots (10^4 waves
        // adding to ONE address took as long as everything else in this kernel together); the host sums the pairs
        if (stats) {
            unsigned long long *sp = stats + 4 + 2 * (q % (size_t)kPnStatSlots);
            atomicAdd(sp, (unsigned long long)n_all);
            atomicAdd(sp + 1, (unsigned long long)evaluated);
        }
    }
}

template <typename T>
static hipError_t launch_select_rerank(const CandBuf &cb, const T *P, size_t n, int dim, size_t ldp, const T *Q, int nq,
                                       size_t ldq, int kout, uint64_t index_base, uint64_t *idx_out, T *dist_out,
                                       size_t out_stride, uint32_t *flags, uint32_t *n_flagged, const double *qn,
                                       const uint32_t *qbad, uint32_t *sel, unsigned long long *stats, hipStream_t s,
                                       int first_eval, int cell_max, const T *cnorm = nullptr, const T *qnorm = nullptr) {
    using KeyT = typename KeyOf<T>::type;
    // with filter keys in the buffers (both MFMA tiers) candidates are evaluated lazily
    const uint32_t *ckey = static_cast<const uint32_t *>(cb.keys);
    // LDS for what the cells can HOLD (cell_max entries each: the filter's k' after its final cut), not for their
    // capacity: C2 12 x 13 entries = 1.9 KB instead of 9.2 KB, and the waves per CU are no longer set by LDS
    const size_t per_cell = cell_max > 0 && cell_max < cb.cap ? (size_t)cell_max : (size_t)cb.cap;
    const size_t sh = (size_t)cb.nseg * per_cell * (sizeof(KeyT) + 8) + ((size_t)dim + 8) * sizeof(T);
    if (sh > 64 * 1024) return hipErrorInvalidValue;
    if (cnorm) {  // a Cosine index: the bf16 filter's (key, row) pairs only
        if (!ckey || !qnorm || !qn) return hipErrorInvalidValue;
        hipLaunchKernelGGL((select_rerank_kernel<T, true>), dim3((unsigned)nq), dim3(64), sh, s,
                           static_cast<const uint32_t *>(cb.tau), cb.idx, cb.cnt, cb.nq_pad, cb.nseg, cb.cap, P, ldp, Q,
                           ldq, dim, (uint32_t)n, kout, index_base, idx_out, dist_out, out_stride, flags, n_flagged, qn,
                           qbad, cb.idx_stride, ckey, sel, stats, (uint32_t)(first_eval > 0 ? first_eval : 0), cnorm, qnorm);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((select_rerank_kernel<T, false>), dim3((unsigned)nq), dim3(64), sh, s,
                       static_cast<const uint32_t *>(cb.tau), cb.idx, cb.cnt, cb.nq_pad, cb.nseg, cb.cap, P, ldp, Q,
                       ldq, dim, (uint32_t)n, kout, index_base, idx_out, dist_out, out_stride, flags, n_flagged, qn,
                       qbad, cb.idx_stride, ckey, sel, stats, (uint32_t)(first_eval > 0 ? first_eval : 0), (const T *)nullptr,
                       (const T *)nullptr);
    return hipGetLastError();
}
hipError_t launch_select_rerank_f32(const CandBuf &cb, const float *P, size_t n, int dim, size_t ldp,
                                    const float *Q, int nq, size_t ldq, int kout, uint64_t index_base,
                                    uint64_t *idx_out, float *dist_out, size_t out_stride, uint32_t *flags,
                                    uint32_t *n_flagged, const double *qn, const uint32_t *qbad, uint32_t *sel,
                                    unsigned long long *stats, hipStream_t s, int first_eval, int cell_max) {
    return launch_select_rerank<float>(cb, P, n, dim, ldp, Q, nq, ldq, kout, index_base, idx_out, dist_out, out_stride,
                                       flags, n_flagged, qn, qbad, sel, stats, s, first_eval, cell_max);
}
// f64 indexes behind the bf16 filter: the candidates' distances in the reference's f64 fold, the proof with u = 2^-53
hipError_t launch_select_rerank_f64(const CandBuf &cb, const double *P, size_t n, int dim, size_t ldp,
                                    const double *Q, int nq, size_t ldq, int kout, uint64_t index_base,
                                    uint64_t *idx_out, double *dist_out, size_t out_stride, uint32_t *flags,
                                    uint32_t *n_flagged, const double *qn, const uint32_t *qbad, uint32_t *sel,
                                    unsigned long long *stats, hipStream_t s, int first_eval, int cell_max) {
    if (!qn) return hipErrorInvalidValue;  // (only the bf16 tier serves f64)
    return launch_select_rerank<double>(cb, P, n, dim, ldp, Q, nq, ldq, kout, index_base, idx_out, dist_out, out_stride,
                                        flags, n_flagged, qn, qbad, sel, stats, s, first_eval, cell_max);
}
// Cosine indexes behind the bf16 filter: candidates evaluated by Cosine::distance (cnorm / qnorm: the rows' and the queries'
// norms in the index's type), signed keys, cos_proof_lb
hipError_t launch_select_rerank_cos_f32(const CandBuf &cb, const float *P, size_t n, int dim, size_t ldp, const float *Q,
                                        int nq, size_t ldq, int kout, uint64_t index_base, uint64_t *idx_out,
                                        float *dist_out, size_t out_stride, uint32_t *flags, uint32_t *n_flagged,
                                        const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *stats,
                                        hipStream_t s, int first_eval, int cell_max, const float *cnorm, const float *qnorm) {
    return launch_select_rerank<float>(cb, P, n, dim, ldp, Q, nq, ldq, kout, index_base, idx_out, dist_out, out_stride,
                                       flags, n_flagged, qn, qbad, sel, stats, s, first_eval, cell_max, cnorm, qnorm);
}
hipError_t launch_select_rerank_cos_f64(const CandBuf &cb, const double *P, size_t n, int dim, size_t ldp, const double *Q,
                                        int nq, size_t ldq, int kout, uint64_t index_base, uint64_t *idx_out,
                                        double *dist_out, size_t out_stride, uint32_t *flags, uint32_t *n_flagged,
                                        const double *qn, const uint32_t *qbad, uint32_t *sel, unsigned long long *stats,
                                        hipStream_t s, int first_eval, int cell_max, const double *cnorm, const double *qnorm) {
    return launch_select_rerank<double>(cb, P, n, dim, ldp, Q, nq, ldq, kout, index_base, idx_out, dist_out, out_stride,
                                        flags, n_flagged, qn, qbad, sel, stats, s, first_eval, cell_max, cnorm, qnorm);
}

// ---------------------------------------------------------------------------
// Small corpora, a point (or a few) per call -- the reference's own call pattern (benches/ball_tree.rs:22-62: 64 x 10
// f64, BallTree::query / query_radius once per point): ONE launch is the whole call.  One wave per query: the lanes
// fold their rows' distances in the reference's order (Euclidean::distance, src/distance.rs:26-35), keys go to LDS;
// k-NN: the kout smallest under (key, row) -- the same selection and ranking as the exact engine's; radius: the rows
// with distance < r (strict, src/ball_tree.rs:277) in ascending order.  The query is read from, and the answer written
// to, mapped pinned host memory: no copy commands, the host waits for its stream once (index.hip, tiny_*).
// Radius output of query q: out[q * out_stride] = count, then the rows.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void tiny_query_kernel(const T *__restrict__ P, uint32_t n, int dim_eff, size_t ldp,
                                                        const T *__restrict__ Q, size_t ldq, int kout, int radius_mode,
                                                        T radius, uint64_t index_base, uint64_t *__restrict__ idx_out,
                                                        T *__restrict__ dist_out, size_t out_stride) {
    using KeyT = typename KeyOf<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    const int lenq = (dim_eff + 7) / 8 * 8;
    T *qs = reinterpret_cast<T *>(smem);
    KeyT *skey = reinterpret_cast<KeyT *>(qs + lenq);
    uint32_t *sidx = reinterpret_cast<uint32_t *>(skey + n);
    for (int k = lane; k < lenq; k += 64) qs[k] = k < dim_eff ? Q[q * ldq + k] : (T)0;
    __syncthreads();
    if (radius_mode) {
        uint32_t cnt = 0;
        uint64_t *out = idx_out + q * out_stride;
        for (uint32_t r0 = 0; r0 < n; r0 += 64) {
            const uint32_t r = r0 + (uint32_t)lane;
            bool in = false;
            if (r < n) in = exact_distance_seq<T>(qs, P + (size_t)r * ldp, dim_eff) < radius;
            const unsigned long long m = __ballot(in);
            if (in) out[1 + cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = index_base + r;
            cnt += (uint32_t)__popcll(m);
        }
        if (lane == 0) out[0] = cnt;
        return;
    }
    KeyT lmin = KeyOf<T>::kMax;  // this lane's smallest key
    for (uint32_t r = lane; r < n; r += 64) {
        const KeyT kk = sel_key(exact_distance_seq<T>(qs, P + (size_t)r * ldp, dim_eff));
        skey[r] = kk;
        sidx[r] = r;
        lmin = kk < lmin ? kk : lmin;
    }
    __syncthreads();
    uint32_t n_in = n;
    if (n > 256u && kout <= 32) {
        // Cut before selecting: the kout-th smallest of the 64 lanes' minima is an upper bound of the kout-th smallest key
        // (kout lanes hold a key at or below it), found on one register per lane; the keys at or below it -- a few times
        // kout of them -- are moved to the front (stable), and the general selection (a radix select over LDS, ~100
        // cycles per 64 keys and bit: 45 us at n = 1000 with 64-bit keys) runs over those only.
        KeyT Tm = 0;
        for (int b = (int)sizeof(KeyT) * 8 - 1; b >= 0; --b) {
            const KeyT cand = Tm | ((KeyT)1 << b);
            if ((uint32_t)__popcll(__ballot(lmin < cand)) < (uint32_t)kout) Tm = cand;
        }
        uint32_t w = 0;
        for (uint32_t e0 = 0; e0 < n; e0 += 64) {
            const uint32_t e = e0 + (uint32_t)lane;
            const KeyT kk = e < n ? skey[e] : KeyOf<T>::kMax;
            const uint32_t ix = e < n ? sidx[e] : 0u;
            const bool keep = e < n && kk <= Tm;
            const unsigned long long mm = __ballot(keep);
            const uint32_t pos = w + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
            if (keep) {
                skey[pos] = kk;
                sidx[pos] = ix;
            }
            w += (uint32_t)__popcll(mm);
        }
        n_in = w;
        __syncthreads();
    }
    const uint32_t m = prune_to_topk<KeyT, uint32_t>(skey, sidx, n_in, (uint32_t)kout, lane);
    __syncthreads();
    for (uint32_t e = lane; e < m; e += 64) {
        const KeyT k = skey[e];
        const uint32_t ix = sidx[e];
        const uint32_t rk = rank_of<KeyT, uint32_t>(skey, sidx, m, k, ix);
        if (rk < (uint32_t)kout) {
            idx_out[q * out_stride + rk] = index_base + ix;
            dist_out[q * out_stride + rk] = key_to_dist(k);
        }
    }
}
template <typename T>
static hipError_t launch_tiny_query(const T *P, size_t n, int dim_eff, size_t ldp, const T *Q, size_t ldq, int nq,
                                    int kout, bool radius_mode, T radius, uint64_t index_base, uint64_t *idx_out,
                                    T *dist_out, size_t out_stride, hipStream_t s) {
    using KeyT = typename KeyOf<T>::type;
    const size_t sh = (size_t)((dim_eff + 7) / 8 * 8) * sizeof(T) + n * (sizeof(KeyT) + sizeof(uint32_t));
    if (sh > 64 * 1024 || n == 0 || n > 0xFFFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tiny_query_kernel<T>, dim3((unsigned)nq), dim3(64), sh, s, P, (uint32_t)n, dim_eff, ldp, Q, ldq, kout,
                       radius_mode ? 1 : 0, radius, index_base, idx_out, dist_out, out_stride);
    return hipGetLastError();
}
size_t tiny_query_lds_bytes(size_t n, int dim_eff, int elem_bytes) {
    return (size_t)((dim_eff + 7) / 8 * 8) * (size_t)elem_bytes + n * ((size_t)elem_bytes + 4);
}
hipError_t launch_tiny_query_f32(const float *P, size_t n, int dim_eff, size_t ldp, const float *Q, size_t ldq, int nq,
                                 int kout, bool radius_mode, float radius, uint64_t index_base, uint64_t *idx_out,
                                 float *dist_out, size_t out_stride, hipStream_t s) {
    return launch_tiny_query<float>(P, n, dim_eff, ldp, Q, ldq, nq, kout, radius_mode, radius, index_base, idx_out, dist_out,
                                    out_stride, s);
}
hipError_t launch_tiny_query_f64(const double *P, size_t n, int dim_eff, size_t ldp, const double *Q, size_t ldq, int nq,
                                 int kout, bool radius_mode, double radius, uint64_t index_base, uint64_t *idx_out,
                                 double *dist_out, size_t out_stride, hipStream_t s) {
    return launch_tiny_query<double>(P, n, dim_eff, ldp, Q, ldq, nq, kout, radius_mode, radius, index_base, idx_out,
                                     dist_out, out_stride, s);
}

// ---------------------------------------------------------------------------
// radius: exact check of the filter's survivors.  block = 64 threads = one query.
// ---------------------------------------------------------------------------
// COS (round 4: query_radius on a Cosine index behind the bf16 filter): the survivors are tested with Cosine::distance in
// the reference's arithmetic (exact_cosine_prefetched; cnorm = the rows' norms, qnorm = the queries'; len = the padded
// row length, zeros beyond dim add nothing)
template <typename T, bool COS>
__global__ __launch_bounds__(64) void radius_check_kernel(const uint32_t *__restrict__ rcnt,
                                                          const uint32_t *__restrict__ ridx, size_t nq_pad, int nseg,
                                                          uint32_t cap, const T *__restrict__ P, size_t ldp,
                                                          const T *__restrict__ Q, int dim, T r,
                                                          uint32_t *__restrict__ kept, uint32_t *__restrict__ nkept,
                                                          uint32_t *__restrict__ overflow, int ridx_stride,
                                                          uint32_t *__restrict__ over_q, const T *__restrict__ cnorm,
                                                          const T *__restrict__ qnorm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *srow = reinterpret_cast<uint32_t *>(smem);  // rows that pass the exact test
    __shared__ uint32_t n_pass;
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    if (lane == 0) n_pass = 0;
    __syncthreads();
    const T *qrow = Q + q * ldp;
    bool over = false;
    for (int s = 0; s < nseg; ++s) {
        uint32_t c = rcnt[(size_t)s * nq_pad + q];
        if (c > cap) { over = true; c = 0; }  // the call is re-run exactly; an overflowed list may hold unwritten slots
        const size_t base = ((size_t)s * nq_pad + q) * (size_t)cap;
        for (uint32_t e = lane; e < c; e += 64) {
            const uint32_t row = ridx[(base + e) * (size_t)ridx_stride];
            T d;
            if constexpr (COS) d = exact_cosine_prefetched<T>(qrow, P + (size_t)row * ldp, (int)ldp, qnorm[q], cnorm[row]);
            else d = exact_distance_seq<T>(qrow, P + (size_t)row * ldp, dim);
            if (d < r) srow[atomicAdd(&n_pass, 1u)] = row;  // strict '<' (src/ball_tree.rs:277); NaN never matches
        }
    }
    __syncthreads();
    const uint32_t n = n_pass;
    const size_t stride = (size_t)nseg * cap;
    for (uint32_t e = lane; e < n; e += 64) {  // ascending row order by rank counting (rows are unique)
        const uint32_t row = srow[e];
        uint32_t rk = 0;
        for (uint32_t j = 0; j < n; ++j) rk += srow[j] < row ? 1u : 0u;
        kept[q * stride + rk] = row;
    }
    if (lane == 0) {
        nkept[q] = n;
        if (over) atomicAdd(overflow, 1u);
        if (over_q) over_q[q] = over ? 1u : 0u;
    }
}

template <typename T>
hipError_t launch_radius_check(const uint32_t *rcnt, const uint32_t *ridx, size_t nq_pad, int nseg, uint32_t cap,
                               const T *P, size_t ldp, const T *Q, int nq, int dim, T r, uint32_t *kept,
                               uint32_t *nkept, uint32_t *overflow, int ridx_stride, uint32_t *over_q, hipStream_t s,
                               const T *cnorm, const T *qnorm) {
    const size_t sh = (size_t)nseg * cap * sizeof(uint32_t);
    if (sh > 64 * 1024) return hipErrorInvalidValue;
    if (cnorm || qnorm) {
        if (!cnorm || !qnorm || ldp % 8) return hipErrorInvalidValue;
        hipLaunchKernelGGL((radius_check_kernel<T, true>), dim3((unsigned)nq), dim3(64), sh, s, rcnt, ridx, nq_pad, nseg, cap,
                           P, ldp, Q, dim, r, kept, nkept, overflow, ridx_stride, over_q, cnorm, qnorm);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((radius_check_kernel<T, false>), dim3((unsigned)nq), dim3(64), sh, s, rcnt, ridx, nq_pad, nseg, cap, P,
                       ldp, Q, dim, r, kept, nkept, overflow, ridx_stride, over_q, (const T *)nullptr, (const T *)nullptr);
    return hipGetLastError();
}
template hipError_t launch_radius_check<float>(const uint32_t *, const uint32_t *, size_t, int, uint32_t, const float *, size_t,
                                               const float *, int, int, float, uint32_t *, uint32_t *, uint32_t *, int,
                                               uint32_t *, hipStream_t, const float *, const float *);
template hipError_t launch_radius_check<double>(const uint32_t *, const uint32_t *, size_t, int, uint32_t, const double *,
                                                size_t, const double *, int, int, double, uint32_t *, uint32_t *, uint32_t *,
                                                int, uint32_t *, hipStream_t, const double *, const double *);
hipError_t launch_radius_check_f32(const uint32_t *rcnt, const uint32_t *ridx, size_t nq_pad, int nseg, uint32_t cap,
                                   const float *P, size_t ldp, const float *Q, int nq, int dim, float r,
                                   uint32_t *kept, uint32_t *nkept, uint32_t *overflow, int ridx_stride,
                                   uint32_t *over_q, hipStream_t s) {
    return launch_radius_check<float>(rcnt, ridx, nq_pad, nseg, cap, P, ldp, Q, nq, dim, r, kept, nkept, overflow,
                                      ridx_stride, over_q, s, nullptr, nullptr);
}

__global__ void radius_gather_kernel(const uint32_t *__restrict__ kept, const uint32_t *__restrict__ nkept,
                                     const uint64_t *__restrict__ offsets, size_t kept_stride, uint64_t index_base,
                                     uint64_t *__restrict__ out) {
    const size_t q = blockIdx.x;
    const uint32_t n = nkept[q];
    const uint64_t o = offsets[q];
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) out[o + e] = index_base + kept[q * kept_stride + e];
}
hipError_t launch_radius_gather(const uint32_t *kept, const uint32_t *nkept, const uint64_t *offsets, int nq,
                                size_t kept_stride, uint64_t index_base, uint64_t *out, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(radius_gather_kernel, dim3((unsigned)nq), dim3(64), 0, s, kept, nkept, offsets, kept_stride,
                       index_base, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// shard merge: parts laid out [part][query][k_part]; (dist, idx) total order.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void merge_topk_kernel(const uint64_t *__restrict__ idx_parts,
                                                        const T *__restrict__ dist_parts, int n_parts,
                                                        size_t idx_part_stride, size_t dist_part_stride, int nq,
                                                        int k_part, int k_out, uint64_t *__restrict__ idx_out,
                                                        T *__restrict__ dist_out,
                                                        const uint32_t *__restrict__ nq_dev,
                                                        const uint32_t *__restrict__ osel, size_t out_stride,
                                                        uint32_t *__restrict__ host_count, int signed_keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    // the flagged-query count of this chunk goes to pinned host memory from here (a LATER call looks at it): no copy
    // command, whose completion signal cost the stream ~25 us
    if (host_count && q == 0 && lane == 0) *host_count = nq_dev ? *nq_dev : 0u;
    if (nq_dev && q >= (size_t)*nq_dev) return;  // device-driven query count (second tier, index.hip)
    const size_t qo = osel ? (size_t)osel[q] : q;  // output row
    uint32_t n = (uint32_t)n_parts * (uint32_t)k_part;
    using KeyT = typename KeyOf<T>::type;
    uint64_t *sidx = reinterpret_cast<uint64_t *>(smem);
    KeyT *skey = reinterpret_cast<KeyT *>(smem + sizeof(uint64_t) * (size_t)n);
    for (uint32_t e = lane; e < n; e += 64) {
        const uint32_t part = e / k_part, j = e % k_part;
        const size_t o = q * k_part + j;
        const uint64_t ix = idx_parts[(size_t)part * idx_part_stride + o];
        sidx[e] = ix;
        const T dv = dist_parts[(size_t)part * dist_part_stride + o];
        skey[e] = (ix == ~0ull) ? KeyOf<T>::kMax : signed_keys ? sel_key_signed(dv) : sel_key(dv);
    }
    __syncthreads();
    uint32_t n_valid = 0;
    for (uint32_t e = lane; e < n; e += 64) n_valid += skey[e] != KeyOf<T>::kMax;
    for (int d = 32; d > 0; d >>= 1) n_valid += __shfl_xor(n_valid, d);
    n = prune_to_topk<KeyT, uint64_t>(skey, sidx, n, (uint32_t)k_out, lane);
    __syncthreads();
    for (uint32_t e = lane; e < n; e += 64) {
        const KeyT k = skey[e];
        const uint64_t ix = sidx[e];
        if (k == KeyOf<T>::kMax) continue;
        const uint32_t r = rank_of<KeyT, uint64_t>(skey, sidx, n, k, ix);
        if (r < (uint32_t)k_out) {
            idx_out[qo * out_stride + r] = ix;
            dist_out[qo * out_stride + r] = signed_keys ? key_to_dist_signed(k) : key_to_dist(k);
        }
    }
    // absent tail (fewer than k_out valid entries over all parts)
    for (uint32_t r = n_valid + lane; r < (uint32_t)k_out; r += 64) {
        idx_out[qo * out_stride + r] = ~0ull;
        dist_out[qo * out_stride + r] = key_to_dist(KeyOf<T>::kNaN);
    }
}

// The same merge for ANY n_parts x k_part (no LDS residency): every part is ascending in (key, index) with its absent
// slots last -- what every query entry point and this merge itself produce -- so the rank of an entry in the union is
// its position in its own part plus, for every other part, the number of that part's entries below it: one binary
// search per (entry, other part), all reads served by L2.  One wave per query; only the first k_out entries of a part
// can reach the output.  (BallTree::query has no limit on k, src/ball_tree.rs:102-121: the sharded k-NN must not
// inherit one from a kernel's LDS budget -- 8 GPUs x k = 1000 is 96 KB of (key, index) pairs.)
template <typename T>
__global__ __launch_bounds__(64) void merge_sorted_topk_kernel(const uint64_t *__restrict__ idx_parts,
                                                               const T *__restrict__ dist_parts, int n_parts,
                                                               size_t idx_part_stride, size_t dist_part_stride,
                                                               int nq, int k_part, int k_out,
                                                               uint64_t *__restrict__ idx_out,
                                                               T *__restrict__ dist_out,
                                                               const uint32_t *__restrict__ nq_dev,
                                                               const uint32_t *__restrict__ osel, size_t out_stride,
                                                               uint32_t *__restrict__ host_count, int signed_keys) {
    using KeyT = typename KeyOf<T>::type;
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    if (host_count && q == 0 && lane == 0) *host_count = nq_dev ? *nq_dev : 0u;
    if (nq_dev && q >= (size_t)*nq_dev) return;
    const size_t qo = osel ? (size_t)osel[q] : q;
    const uint32_t kp = (uint32_t)k_part, ko = (uint32_t)k_out;
    const uint32_t lim = kp < ko ? kp : ko;  // positions of a part that can reach the output
    uint32_t n_valid = 0;
    for (int p = 0; p < n_parts; ++p) {
        const uint64_t *pi = idx_parts + (size_t)p * idx_part_stride + q * kp;
        const T *pd = dist_parts + (size_t)p * dist_part_stride + q * kp;
        // valid entries of this part (absent slots are a suffix): binary search for the first absent one
        uint32_t lo = 0, hi = kp;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pi[mid] != ~0ull) lo = mid + 1; else hi = mid;
        }
        const uint32_t nv = lo;
        n_valid += nv;
        for (uint32_t j = (uint32_t)lane; j < (nv < lim ? nv : lim); j += 64) {
            const uint64_t ix = pi[j];
            const KeyT key = signed_keys ? sel_key_signed(pd[j]) : sel_key(pd[j]);
            uint32_t rank = j;
            for (int o = 0; o < n_parts && rank < ko; ++o) {
                if (o == p) continue;
                const uint64_t *oi = idx_parts + (size_t)o * idx_part_stride + q * kp;
                const T *od = dist_parts + (size_t)o * dist_part_stride + q * kp;
                uint32_t a = 0, b = kp;  // first entry of part o that is absent or not below (key, ix)
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    const uint64_t mi = oi[mid];
                    const KeyT mk = signed_keys ? sel_key_signed(od[mid]) : sel_key(od[mid]);
                    const bool below = mi != ~0ull && (mk < key || (mk == key && mi < ix));
                    if (below) a = mid + 1; else b = mid;
                }
                rank += a;
            }
            if (rank < ko) {
                idx_out[qo * out_stride + rank] = ix;
                dist_out[qo * out_stride + rank] = signed_keys ? key_to_dist_signed(key) : key_to_dist(key);
            }
        }
    }
    for (uint32_t r = n_valid + (uint32_t)lane; r < ko; r += 64) {  // absent tail
        idx_out[qo * out_stride + r] = ~0ull;
        dist_out[qo * out_stride + r] = key_to_dist(KeyOf<T>::kNaN);
    }
}

template <typename T>
static hipError_t launch_merge_topk(const uint64_t *idx_parts, const T *dist_parts, int n_parts, size_t idx_part_stride,
                                    size_t dist_part_stride, int nq, int k_part, int k_out, uint64_t *idx_out,
                                    T *dist_out, hipStream_t s, const uint32_t *nq_dev, const uint32_t *osel,
                                    size_t out_stride, uint32_t *host_count, bool signed_keys) {
    const size_t sh = (size_t)n_parts * k_part * (8 + sizeof(typename KeyOf<T>::type));
    if (out_stride == 0) out_stride = (size_t)k_out;
    if (sh > 64 * 1024) {  // beyond the LDS-resident merge: the rank-by-binary-search merge of sorted parts
        hipLaunchKernelGGL(merge_sorted_topk_kernel<T>, dim3((unsigned)nq), dim3(64), 0, s, idx_parts, dist_parts, n_parts,
                           idx_part_stride, dist_part_stride, nq, k_part, k_out, idx_out, dist_out, nq_dev, osel,
                           out_stride, host_count, signed_keys ? 1 : 0);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(merge_topk_kernel<T>, dim3((unsigned)nq), dim3(64), sh, s, idx_parts, dist_parts, n_parts,
                       idx_part_stride, dist_part_stride, nq, k_part, k_out, idx_out, dist_out, nq_dev, osel, out_stride,
                       host_count, signed_keys ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_merge_topk_f32(const uint64_t *idx_parts, const float *dist_parts, int n_parts,
                                 size_t idx_part_stride, size_t dist_part_stride, int nq, int k_part, int k_out,
                                 uint64_t *idx_out, float *dist_out, hipStream_t s, const uint32_t *nq_dev,
                                 const uint32_t *osel, size_t out_stride, uint32_t *host_count, bool signed_keys) {
    return launch_merge_topk<float>(idx_parts, dist_parts, n_parts, idx_part_stride, dist_part_stride, nq, k_part, k_out,
                                    idx_out, dist_out, s, nq_dev, osel, out_stride, host_count, signed_keys);
}
hipError_t launch_merge_topk_f64(const uint64_t *idx_parts, const double *dist_parts, int n_parts,
                                 size_t idx_part_stride, size_t dist_part_stride, int nq, int k_part, int k_out,
                                 uint64_t *idx_out, double *dist_out, hipStream_t s, const uint32_t *nq_dev,
                                 const uint32_t *osel, size_t out_stride, uint32_t *host_count, bool signed_keys) {
    return launch_merge_topk<double>(idx_parts, dist_parts, n_parts, idx_part_stride, dist_part_stride, nq, k_part, k_out,
                                     idx_out, dist_out, s, nq_dev, osel, out_stride, host_count, signed_keys);
}

// ---------------------------------------------------------------------------
// fallback plumbing: list flagged queries, gather their rows, scatter results
// ---------------------------------------------------------------------------
__global__ void compact_flags_kernel(const uint32_t *__restrict__ flags, int nq, uint32_t *__restrict__ sel,
                                     uint32_t *__restrict__ nsel) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq && flags[q]) sel[atomicAdd(nsel, 1u)] = (uint32_t)q;
}
hipError_t launch_compact_flags(const uint32_t *flags, int nq, uint32_t *sel, uint32_t *nsel, hipStream_t s) {
    hipLaunchKernelGGL(compact_flags_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, flags, nq, sel,
                       nsel);
    return hipGetLastError();
}

template <typename T>
__global__ void gather_rows_kernel(const T *__restrict__ src, size_t ld, const uint32_t *__restrict__ sel,
                                   const uint32_t *__restrict__ nsel, uint32_t off, uint32_t max_rows,
                                   T *__restrict__ dst) {
    const uint32_t tot = *nsel;
    uint32_t cnt = tot > off ? tot - off : 0u;
    if (cnt > max_rows) cnt = max_rows;
    for (size_t i = blockIdx.x; i < (size_t)cnt; i += gridDim.x) {
        const T *s = src + (size_t)sel[off + i] * ld;
        T *d = dst + i * ld;
        for (size_t c = threadIdx.x; c < ld; c += blockDim.x) d[c] = s[c];
    }
}
template <typename T>
hipError_t launch_gather_rows(const T *src, size_t ld, const uint32_t *sel, const uint32_t *nsel, uint32_t off,
                              uint32_t max_rows, T *dst, hipStream_t s) {
    if (max_rows == 0) return hipSuccess;
    const unsigned grid = max_rows < 1024u ? max_rows : 1024u;
    hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(grid), dim3(64), 0, s, src, ld, sel, nsel, off, max_rows, dst);
    return hipGetLastError();
}
template hipError_t launch_gather_rows<float>(const float *, size_t, const uint32_t *, const uint32_t *, uint32_t, uint32_t,
                                              float *, hipStream_t);
template hipError_t launch_gather_rows<double>(const double *, size_t, const uint32_t *, const uint32_t *, uint32_t, uint32_t,
                                               double *, hipStream_t);
hipError_t launch_gather_rows_f32(const float *src, size_t ld, const uint32_t *sel, const uint32_t *nsel, uint32_t off,
                                  uint32_t max_rows, float *dst, hipStream_t s) {
    return launch_gather_rows<float>(src, ld, sel, nsel, off, max_rows, dst, s);
}

__global__ void scatter_results_kernel(const uint64_t *__restrict__ idx_in, const float *__restrict__ dist_in,
                                       const uint32_t *__restrict__ sel, const uint32_t *__restrict__ nsel,
                                       uint32_t off, uint32_t max_rows, int kout, uint64_t *__restrict__ idx_out,
                                       float *__restrict__ dist_out, size_t out_stride) {
    const uint32_t tot = *nsel;
    uint32_t cnt = tot > off ? tot - off : 0u;
    if (cnt > max_rows) cnt = max_rows;
    for (size_t i = blockIdx.x; i < (size_t)cnt; i += gridDim.x) {
        const size_t q = sel[off + i];
        for (int j = threadIdx.x; j < kout; j += blockDim.x) {
            idx_out[q * out_stride + j] = idx_in[i * kout + j];
            dist_out[q * out_stride + j] = dist_in[i * kout + j];
        }
    }
}
hipError_t launch_scatter_results_f32(const uint64_t *idx_in, const float *dist_in, const uint32_t *sel,
                                      const uint32_t *nsel, uint32_t off, uint32_t max_rows, int kout, uint64_t *idx_out,
                                      float *dist_out, size_t out_stride, hipStream_t s) {
    if (max_rows == 0) return hipSuccess;
    const unsigned grid = max_rows < 1024u ? max_rows : 1024u;
    hipLaunchKernelGGL(scatter_results_kernel, dim3(grid), dim3(64), 0, s, idx_in, dist_in, sel, nsel, off, max_rows,
                       kout, idx_out, dist_out, out_stride);
    return hipGetLastError();
}

}  // namespace pn
