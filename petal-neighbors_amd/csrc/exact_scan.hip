// exact_scan.hip -- exact Euclidean scan kernels for gfx950 (MI355X).
//
// Computes, for a tile of 64 queries x 64 corpus rows per workgroup step, the
// distance of every (query, row) pair in EXACTLY the reference's operation
// order -- petal-neighbors src/distance.rs:26-35:
//     sum = 0;  for k in 0..D { diff = q[k] - p[k]; sum += diff * diff; }  sqrt(sum)
// i.e. a strictly sequential fold, multiply and add rounded separately (never
// fused), followed by a correctly rounded sqrt.  The kernels therefore replace
// the leaf-scan inner loop of src/ball_tree.rs:217-226 / :275-282 with a
// brute-force scan whose distances are bit-identical by construction.
//
// Used as: (1) the always-correct engine (PN_ENGINE_EXACT, f32 and f64),
// (2) the per-query fallback when the MFMA filter cannot prove its result,
// (3) query_radius (two-pass CSR) and distance::pairwise.
//
// Work decomposition (VALU-bound: 3 VALU ops per coordinate per pair):
//   workgroup = 256 threads = 4 waves; tile = 64 queries x 64 rows;
//   wave w owns queries [16w, 16w+16) of the tile (so a query's top-k state is
//   touched by one wave only); lane (tq = lane&3, tp = lane>>2) accumulates a
//   4 x 4 register block: queries 16w+4tq.., rows 4tp..;
//   both operands are staged k-major in LDS in chunks of 32 coordinates, so any
//   D works and each LDS b128 read feeds 12..48 VALU ops.
//
// Top-k: per (segment, query) an append buffer in HBM with a running threshold:
//   a value enters if key < tau; when fewer than 64 free slots remain the owning
//   wave ranks the buffer by (key, index), keeps the kp smallest and lowers tau
//   to the kp-th.  Rows are scanned in ascending index order, so rejecting
//   key == tau keeps exactly the (key, index)-smallest set.
//
// This translation unit must never contract a*b+c: it is compiled with
// -ffp-contract=off AND every arithmetic function carries the pragma; the build
// greps the ISA for v_fma/v_fmac/v_pk_fma in these kernels (tests/test_build.py).
#include "pn_internal.h"
#include "topk_buffer.h"

namespace pn {

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

__device__ __forceinline__ uint32_t dist_key(float d) {
    return (d != d) ? KeyOf<float>::kNaN : __float_as_uint(d);
}
__device__ __forceinline__ uint64_t dist_key(double d) {
    return (d != d) ? KeyOf<double>::kNaN : (uint64_t)__double_as_longlong(d);
}
// Cosine distances can be negative (1 - dot / (|a||b|) a few ulp below zero): their keys use the order-preserving map
// of all floats (sign bit flipped for x >= +0, all bits for x < 0; -0 counts as +0, NaN above +inf as in ordered-float);
// select.hip maps them back (signed_keys)
__device__ __forceinline__ uint32_t dist_key_signed(float d) {
    if (d != d) return 0xFFC00000u;
    const uint32_t b = __float_as_uint(d == 0.0f ? 0.0f : d);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ uint64_t dist_key_signed(double d) {
    if (d != d) return 0xFFF8000000000000ull;
    const uint64_t b = (uint64_t)__double_as_longlong(d == 0.0 ? 0.0 : d);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ float pn_sqrt(float x) { return sqrtf(x); }   // correctly rounded (default hipcc)
__device__ __forceinline__ double pn_sqrt(double x) { return sqrt(x); }

// ---------------------------------------------------------------------------
// One 64 x 64 tile: acc[qi][pi] = fold over k of (q - p)^2, k ascending.
// Pt/Qt point at row 0 of the tile; rows are zero padded to a multiple of 8
// coordinates and row counts to a multiple of 64, so loads need no row guard.
// ---------------------------------------------------------------------------
// rowsel (nullable): the tile's rows are rows rowsel[0 .. n_valid) of src (a device-side selection list: the second
// tier's flagged queries, read in place instead of gathered by a launch of their own); rows beyond n_valid read row
// rowsel[0] -- they are masked by the caller.
template <typename T>
__device__ __forceinline__ void stage_chunk(const T *__restrict__ src, size_t ld, int kc0, int dim,
                                            T (*dst)[kTileQ], int tid, const uint32_t *__restrict__ rowsel = nullptr,
                                            int n_valid = 0) {
    using V = typename Vec4<T>::type;
    const int row = tid & 63;
    const int kq = (tid >> 6) * 8;
    const int k0 = kc0 + kq;
    T v[8];
    if (k0 < dim) {  // dim <= ld, groups of 8 never straddle the padded row end
        const size_t srow = rowsel ? (size_t)rowsel[row < n_valid ? row : 0] : (size_t)row;
        const T *p = src + srow * ld + k0;
        V a = *reinterpret_cast<const V *>(p);
        V b = *reinterpret_cast<const V *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (T)0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // coordinates at or beyond `dim` do not take part (zip truncation, src/distance.rs:27-28)
        dst[kq + j][row] = (k0 + j < dim) ? v[j] : (T)0;
    }
}

// COS: acc[qi][pi] = fold over k of q * p instead (the dot product of Cosine::distance, src/distance.rs:86-90:
// multiply and add separately rounded, ascending k).
template <typename T, bool COS = false>
__device__ __forceinline__ void compute_tile(const T *__restrict__ Pt, size_t ldp, const T *__restrict__ Qt,
                                             size_t ldq, int dim, bool stage_q, T (&acc)[4][4],
                                             T (*Qs)[kTileQ], T (*Ps)[kTileP], int tid, int qb, int pb,
                                             const uint32_t *__restrict__ qsel = nullptr, int q_valid = 0) {
#pragma clang fp contract(off)
    using V = typename Vec4<T>::type;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (T)0;

    for (int kc0 = 0; kc0 < dim; kc0 += kChunkK) {
        __syncthreads();  // previous chunk fully consumed
        if (stage_q || kc0 > 0 || dim > kChunkK) stage_chunk<T>(Qt, ldq, kc0, dim, Qs, tid, qsel, q_valid);
        stage_chunk<T>(Pt, ldp, kc0, dim, Ps, tid);
        __syncthreads();
        const int klen = (dim - kc0 < kChunkK) ? (dim - kc0) : kChunkK;
        for (int k = 0; k < klen; ++k) {
            const V qv = *reinterpret_cast<const V *>(&Qs[k][qb]);
            const V pv = *reinterpret_cast<const V *>(&Ps[k][pb]);
            const T q[4] = {qv.x, qv.y, qv.z, qv.w};
            const T p[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (COS) {
                        const T pr = q[a] * p[b];
                        acc[a][b] = acc[a][b] + pr;
                    } else {
                        const T diff = q[a] - p[b];   // v1 - v2 with v1 = query (src/ball_tree.rs:218)
                        const T sq = diff * diff;
                        acc[a][b] = acc[a][b] + sq;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------
// k-NN scan.  grid = (query tiles, segments), block = 256.
// ---------------------------------------------------------------------------
// COS: Cosine::distance = 1 - dot / (|q| |p|) (src/distance.rs:85-107) with the row norms pnorm[n_pad] and the query
// norms qnorm[nq_pad] computed by cosine_norms_kernel (each over its OWN vector's full length; the dot product over
// the shorter one -- the zip).
template <typename T, int M, bool COS>
__global__ __launch_bounds__(256) void exact_knn_kernel(
    const T *__restrict__ P, size_t n, int dim, size_t ldp, const T *__restrict__ Q, int nq, size_t ldq,
    uint32_t kp, size_t seg_len, typename KeyOf<T>::type *__restrict__ ckey, uint32_t *__restrict__ cidx,
    uint32_t *__restrict__ ccnt, typename KeyOf<T>::type *__restrict__ ctau, size_t nq_pad,
    const typename KeyOf<T>::type *__restrict__ lo_key, const uint32_t *__restrict__ lo_idx,
    const uint32_t *__restrict__ nq_dev, uint32_t nq_off, const T *__restrict__ pnorm, const T *__restrict__ qnorm,
    const uint32_t *__restrict__ qsel) {
    using KeyT = typename KeyOf<T>::type;
    constexpr uint32_t CAP = 64u * M;
    constexpr KeyT KMAX = KeyOf<T>::kMax;
    // device-driven query count (second tier behind a filter: the host never reads how many queries were flagged):
    // the grid is sized for the worst case, query tiles beyond the count leave at once
    if (nq_dev) {
        const uint32_t tot = *nq_dev;
        const uint32_t c = tot > nq_off ? tot - nq_off : 0u;
        nq = c < (uint32_t)nq ? (int)c : nq;
        if ((size_t)blockIdx.x * kTileQ >= (size_t)nq) return;
    }
    __shared__ __attribute__((aligned(32))) T Qs[kChunkK][kTileQ];
    __shared__ __attribute__((aligned(32))) T Ps[kChunkK][kTileP];
    __shared__ KeyT taus[kTileQ];
    __shared__ uint32_t cnts[kTileQ];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tq = lane & 3, tp = lane >> 2;
    const int qb = wave * 16 + tq * 4;  // first of this lane's 4 queries within the tile
    const int pb = tp * 4;              // first of this lane's 4 rows within the tile
    const size_t q0 = (size_t)blockIdx.x * kTileQ;
    const size_t seg = blockIdx.y;
    const size_t p_begin = seg * seg_len;
    const size_t p_end = (p_begin + seg_len < n) ? p_begin + seg_len : n;

    if (tid < kTileQ) { taus[tid] = KMAX; cnts[tid] = 0; }
    __syncthreads();

    // qsel: query r of this launch is row qsel[nq_off + r] of Q (the flagged queries of a filter tier, in place)
    const T *Qt = qsel ? Q : Q + q0 * ldq;
    const uint32_t *qsel_t = qsel ? qsel + nq_off + q0 : nullptr;
    const int q_valid = (int)((size_t)nq - q0 < (size_t)kTileQ ? (size_t)nq - q0 : (size_t)kTileQ);
    const size_t cbase0 = (seg * nq_pad + q0) * (size_t)CAP;
    // multi-round selection (k beyond one buffer): only entries strictly after (lo_key, lo_idx) in the
    // (key, row) order take part, i.e. the neighbours already returned by earlier rounds are skipped
    KeyT lok[4];
    uint32_t loi[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        lok[a] = lo_key ? lo_key[q0 + qb + a] : (KeyT)0;
        loi[a] = lo_key ? lo_idx[q0 + qb + a] : 0u;
    }
    const bool bounded = lo_key != nullptr;
    bool first = true;
    for (size_t p0 = p_begin; p0 < p_end; p0 += kTileP) {
        T acc[4][4];
        compute_tile<T, COS>(P + p0 * ldp, ldp, Qt, ldq, dim, first, acc, Qs, Ps, tid, qb, pb, qsel_t, q_valid);
        first = false;
        T qn4[4] = {(T)0, (T)0, (T)0, (T)0}, pn4[4] = {(T)0, (T)0, (T)0, (T)0};
        if (COS) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // (with a selection list the norms are indexed like the rows of Q: by the listed query number)
                const size_t qi = q0 + qb + a;
                qn4[a] = qsel_t ? (qi < (size_t)nq ? qnorm[qsel_t[qb + a]] : (T)1) : qnorm[qi];
                pn4[a] = pnorm[p0 + pb + a];
            }
        }

        KeyT key[4][4];
        KeyT tau_r[4];
        bool anyp = false;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            tau_r[a] = taus[qb + a];
            const bool qv = (q0 + qb + a) < (size_t)nq;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool v = qv && (p0 + pb + b) < p_end;
                KeyT kk = !v ? KMAX
                               : COS ? dist_key_signed((T)1 - acc[a][b] / (qn4[a] * pn4[b])) : dist_key(pn_sqrt(acc[a][b]));
                if (bounded && !(kk > lok[a] || (kk == lok[a] && (uint32_t)(p0 + pb + b) > loi[a]))) kk = KMAX;
                key[a][b] = kk;
                anyp |= kk < tau_r[a];
            }
        }
        if (__any(anyp)) {  // wave-uniform slow path: append, then compact where nearly full
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                uint32_t np = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) np += key[a][b] < tau_r[a] ? 1u : 0u;
                if (np) {  // one LDS atomic per (lane, query) reserves room for its survivors
                    size_t o = cbase0 + (size_t)(qb + a) * CAP + atomicAdd(&cnts[qb + a], np);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (key[a][b] < tau_r[a]) {
                            ckey[o] = key[a][b];
                            cidx[o] = (uint32_t)(p0 + pb + b);
                            ++o;
                        }
                }
            }
            wg_fence();
            for (int j = 0; j < 16; ++j) {
                const int qj = wave * 16 + j;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnts[qj]);
                if (c + 64u > CAP)  // fewer than 64 free slots: a full tile might not fit
                    compact_query<KeyT, M>(ckey, cidx, cbase0 + (size_t)qj * CAP, c, kp, lane, &taus[qj],
                                           &cnts[qj], KMAX);
            }
        }
    }
    // leave at most kp candidates per (segment, query) for the select kernel
    for (int j = 0; j < 16; ++j) {
        const int qj = wave * 16 + j;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnts[qj]);
        if (c > kp)
            compact_query<KeyT, M>(ckey, cidx, cbase0 + (size_t)qj * CAP, c, kp, lane, &taus[qj], &cnts[qj],
                                   KMAX);
    }
    if (lane < 16) {
        const int qj = wave * 16 + lane;
        ccnt[seg * nq_pad + q0 + qj] = cnts[qj];
        ctau[seg * nq_pad + q0 + qj] = taus[qj];
    }
}

template <typename T>
static hipError_t launch_exact_knn(const T *P, size_t n, int dim, size_t ldp, const T *Q, int nq, size_t ldq,
                                   int kp, size_t seg_len, const CandBuf &cb, const void *lo_key,
                                   const uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, const T *pnorm,
                                   const T *qnorm, hipStream_t s, const uint32_t *qsel = nullptr) {
    using KeyT = typename KeyOf<T>::type;
    dim3 grid((unsigned)(cb.nq_pad / kTileQ), (unsigned)cb.nseg), block(256);
    const bool cosm = pnorm != nullptr;
    auto *ck = static_cast<KeyT *>(cb.keys);
    auto *ct = static_cast<KeyT *>(cb.tau);
#define PN_LAUNCH1(MM, CC)                                                                               \
    hipLaunchKernelGGL((exact_knn_kernel<T, MM, CC>), grid, block, 0, s, P, n, dim, ldp, Q, nq, ldq,      \
                       (uint32_t)kp, seg_len, ck, cb.idx, cb.cnt, ct, cb.nq_pad, static_cast<const KeyT *>(lo_key),  \
                       lo_idx, nq_dev, nq_off, pnorm, qnorm, qsel)
#define PN_LAUNCH(MM)          \
    do {                       \
        if (cosm)              \
            PN_LAUNCH1(MM, true);  \
        else                   \
            PN_LAUNCH1(MM, false); \
    } while (0)
    switch (cb.cap / 64) {
        case 2: PN_LAUNCH(2); break;
        case 4: PN_LAUNCH(4); break;
        case 8: PN_LAUNCH(8); break;
        case 16: PN_LAUNCH(16); break;
        default: return hipErrorInvalidValue;
    }
#undef PN_LAUNCH
#undef PN_LAUNCH1
    return hipGetLastError();
}

hipError_t launch_exact_knn_f32(const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq,
                                size_t ldq, int kp, size_t seg_len, const CandBuf &cb, const void *lo_key,
                                const uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, const float *pnorm,
                                const float *qnorm, hipStream_t s, const uint32_t *qsel) {
    return launch_exact_knn<float>(P, n, dim, ldp, Q, nq, ldq, kp, seg_len, cb, lo_key, lo_idx, nq_dev, nq_off, pnorm,
                                   qnorm, s, qsel);
}
hipError_t launch_exact_knn_f64(const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq,
                                size_t ldq, int kp, size_t seg_len, const CandBuf &cb, const void *lo_key,
                                const uint32_t *lo_idx, const uint32_t *nq_dev, uint32_t nq_off, const double *pnorm,
                                const double *qnorm, hipStream_t s, const uint32_t *qsel) {
    return launch_exact_knn<double>(P, n, dim, ldp, Q, nq, ldq, kp, seg_len, cb, lo_key, lo_idx, nq_dev, nq_off, pnorm,
                                    qnorm, s, qsel);
}

// ---------------------------------------------------------------------------
// query_radius: { i : distance(q, p_i) < r } (strict, src/ball_tree.rs:277),
// ascending index.  Pass 1 counts per (query, segment); the host turns counts
// into offsets; pass 2 writes indices in row order: within a tile the 16 lanes
// that share a query scan their per-lane counts with stride-4 shuffles.
// ---------------------------------------------------------------------------
template <typename T, bool FILL, bool COS>
__global__ __launch_bounds__(256) void exact_radius_kernel(
    const T *__restrict__ P, size_t n, int dim, size_t ldp, const T *__restrict__ Q, int nq, size_t ldq, T r,
    size_t seg_len, int nseg, uint32_t *__restrict__ counts, const uint64_t *__restrict__ offsets,
    uint64_t *__restrict__ fill, uint64_t index_base, const T *__restrict__ pnorm, const T *__restrict__ qnorm,
    const uint32_t *__restrict__ qsel, const uint32_t *__restrict__ nq_dev, uint64_t capacity) {
    // qsel / nq_dev (nullable; the device-resident entry point, pn_query_radius_device_*): query r of the launch is row
    // qsel[r] of Q and only the first *nq_dev listed queries exist -- the grid is sized for nq and surplus tiles leave at
    // once; counts / offsets are indexed by r.  capacity: positions at or beyond it are counted, not written.
    if (nq_dev) {
        const uint32_t c = *nq_dev;
        nq = c < (uint32_t)nq ? (int)c : nq;
        if ((size_t)blockIdx.x * kTileQ >= (size_t)nq) return;
    }
    __shared__ __attribute__((aligned(32))) T Qs[kChunkK][kTileQ];
    __shared__ __attribute__((aligned(32))) T Ps[kChunkK][kTileP];
    __shared__ uint32_t cnts[kTileQ];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tq = lane & 3, tp = lane >> 2;
    const int qb = wave * 16 + tq * 4, pb = tp * 4;
    const size_t q0 = (size_t)blockIdx.x * kTileQ;
    const size_t seg = blockIdx.y;
    const size_t p_begin = seg * seg_len;
    const size_t p_end = (p_begin + seg_len < n) ? p_begin + seg_len : n;
    if (tid < kTileQ) cnts[tid] = 0;
    __syncthreads();

    const T *Qt = qsel ? Q : Q + q0 * ldq;
    const uint32_t *qsel_t = qsel ? qsel + q0 : nullptr;
    const int q_valid = (int)((size_t)nq - q0 < (size_t)kTileQ ? (size_t)nq - q0 : (size_t)kTileQ);
    uint64_t obase[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const size_t q = q0 + qb + a;
        obase[a] = (FILL && q < (size_t)nq) ? offsets[q * nseg + seg] : 0;
    }
    bool first = true;
    for (size_t p0 = p_begin; p0 < p_end; p0 += kTileP) {
        T acc[4][4];
        compute_tile<T, COS>(P + p0 * ldp, ldp, Qt, ldq, dim, first, acc, Qs, Ps, tid, qb, pb, qsel_t, q_valid);
        first = false;
        T qn4[4] = {(T)0, (T)0, (T)0, (T)0}, pn4[4] = {(T)0, (T)0, (T)0, (T)0};
        if (COS) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const size_t qi = q0 + qb + a;
                qn4[a] = qsel_t ? (qi < (size_t)nq ? qnorm[qsel_t[qb + a]] : (T)1) : qnorm[qi];
                pn4[a] = pnorm[p0 + pb + a];
            }
        }
        uint32_t mask[4];
        bool anyp = false;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bool qv = (q0 + qb + a) < (size_t)nq;
            uint32_t m = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool v = qv && (p0 + pb + b) < p_end;
                const T d = COS ? (T)1 - acc[a][b] / (qn4[a] * pn4[b]) : pn_sqrt(acc[a][b]);
                if (v && d < r) m |= 1u << b;  // NaN distances never match
            }
            mask[a] = m;
            anyp |= m != 0;
        }
        if (!__any(anyp)) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const uint32_t c = __popc(mask[a]);
            // inclusive scan over the 16 lanes (stride 4) that share this query
            uint32_t inc = c;
#pragma unroll
            for (int d = 4; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d);
                if (lane >= d) inc += t;
            }
            const uint32_t total = __shfl(inc, 60 + tq);
            if (total == 0) continue;
            const uint32_t before = cnts[qb + a];  // written only by this wave
            if (FILL) {
                uint64_t pos = obase[a] + before + (inc - c);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (mask[a] & (1u << b)) {
                        if (pos < capacity) fill[pos] = index_base + (uint64_t)(p0 + pb + b);
                        ++pos;
                    }
            }
            wg_fence();
            if (tp == 15) cnts[qb + a] = before + total;
            wg_fence();
        }
    }
    if (!FILL && lane < 16) {
        const size_t q = q0 + wave * 16 + lane;
        if (q < (size_t)nq) counts[q * nseg + seg] = cnts[wave * 16 + lane];
    }
}

template <typename T>
static hipError_t launch_exact_radius(const T *P, size_t n, int dim, size_t ldp, const T *Q, int nq, size_t ldq,
                                      T r, size_t seg_len, int nseg, uint32_t *counts, const uint64_t *offsets,
                                      uint64_t *fill, uint64_t index_base, const T *pnorm, const T *qnorm,
                                      hipStream_t s, const uint32_t *qsel = nullptr, const uint32_t *nq_dev = nullptr,
                                      uint64_t capacity = ~0ull) {
    dim3 grid((unsigned)(round_up((size_t)nq, kTileQ) / kTileQ), (unsigned)nseg), block(256);
#define PN_RAD(FF, CC)                                                                                          \
    hipLaunchKernelGGL((exact_radius_kernel<T, FF, CC>), grid, block, 0, s, P, n, dim, ldp, Q, nq, ldq, r, seg_len, \
                       nseg, counts, offsets, fill, index_base, pnorm, qnorm, qsel, nq_dev, capacity)
    if (fill) {
        if (pnorm) PN_RAD(true, true); else PN_RAD(true, false);
    } else {
        if (pnorm) PN_RAD(false, true); else PN_RAD(false, false);
    }
#undef PN_RAD
    return hipGetLastError();
}
hipError_t launch_exact_radius_f32(const float *P, size_t n, int dim, size_t ldp, const float *Q, int nq,
                                   size_t ldq, float r, size_t seg_len, int nseg, uint32_t *counts,
                                   const uint64_t *offsets, uint64_t *fill, uint64_t index_base, const float *pnorm,
                                   const float *qnorm, hipStream_t s, const uint32_t *qsel, const uint32_t *nq_dev,
                                   uint64_t capacity) {
    return launch_exact_radius<float>(P, n, dim, ldp, Q, nq, ldq, r, seg_len, nseg, counts, offsets, fill,
                                      index_base, pnorm, qnorm, s, qsel, nq_dev, capacity);
}
hipError_t launch_exact_radius_f64(const double *P, size_t n, int dim, size_t ldp, const double *Q, int nq,
                                   size_t ldq, double r, size_t seg_len, int nseg, uint32_t *counts,
                                   const uint64_t *offsets, uint64_t *fill, uint64_t index_base, const double *pnorm,
                                   const double *qnorm, hipStream_t s, const uint32_t *qsel, const uint32_t *nq_dev,
                                   uint64_t capacity) {
    return launch_exact_radius<double>(P, n, dim, ldp, Q, nq, ldq, r, seg_len, nseg, counts, offsets, fill,
                                       index_base, pnorm, qnorm, s, qsel, nq_dev, capacity);
}

// ---------------------------------------------------------------------------
// distance::pairwise (src/distance.rs:58-74): out[i][j] = distance(x_i, x_j), zero diagonal.  Like the reference,
// only the pairs i < j are evaluated and written twice ((a-b)^2 == (b-a)^2 exactly, so either triangle has the
// same bits): the grid is the upper triangle of 64 x 64 tiles (a tile pair (ti, tj), ti <= tj, from the linear block
// number), half the arithmetic of the full square.  VALU-bound: 3 packed vector operations per coordinate of a pair.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void exact_pairwise_kernel(const T *__restrict__ X, size_t n, int dim, size_t ld,
                                                             T *__restrict__ out, uint32_t tiles) {
    __shared__ __attribute__((aligned(32))) T Qs[kChunkK][kTileQ];
    __shared__ __attribute__((aligned(32))) T Ps[kChunkK][kTileP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tq = lane & 3, tp = lane >> 2;
    const int qb = wave * 16 + tq * 4, pb = tp * 4;
    // linear block number -> (ti, tj) with ti <= tj: row ti of the triangle starts at ti * tiles - ti (ti - 1) / 2
    const unsigned long long b = blockIdx.x;
    unsigned long long ti = (unsigned long long)(((2.0 * tiles + 1.0) - sqrt((2.0 * tiles + 1.0) * (2.0 * tiles + 1.0) - 8.0 * (double)b)) * 0.5);
    if (ti >= tiles) ti = tiles - 1;
    while (ti > 0 && ti * tiles - ti * (ti - 1) / 2 > b) --ti;
    while ((ti + 1) * tiles - (ti + 1) * ti / 2 <= b) ++ti;
    const unsigned long long tj = ti + (b - (ti * tiles - ti * (ti - 1) / 2));
    const size_t q0 = (size_t)ti * kTileQ, p0 = (size_t)tj * kTileP;
    T acc[4][4];
    compute_tile<T>(X + p0 * ld, ld, X + q0 * ld, ld, dim, true, acc, Qs, Ps, tid, qb, pb);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const size_t i = q0 + qb + a;
        if (i >= n) continue;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const size_t j = p0 + pb + bb;
            if (j >= n) continue;
            const T d = (i == j) ? (T)0 : pn_sqrt(acc[a][bb]);
            out[i * n + j] = d;
            if (ti != tj) out[j * n + i] = d;  // the mirrored write of src/distance.rs:69-70
        }
    }
}

template <typename T>
static hipError_t launch_exact_pairwise(const T *X, size_t n, int dim, size_t ld, T *out, hipStream_t s) {
    const unsigned long long t = (unsigned long long)(round_up(n, kTileQ) / kTileQ);
    const unsigned long long blocks = t * (t + 1) / 2;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((exact_pairwise_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, X, n, dim, ld, out, (uint32_t)t);
    return hipGetLastError();
}
hipError_t launch_exact_pairwise_f32(const float *X, size_t n, int dim, size_t ld, float *out, hipStream_t s) {
    return launch_exact_pairwise<float>(X, n, dim, ld, out, s);
}
hipError_t launch_exact_pairwise_f64(const double *X, size_t n, int dim, size_t ld, double *out, hipStream_t s) {
    return launch_exact_pairwise<double>(X, n, dim, ld, out, s);
}

// ---------------------------------------------------------------------------
// distance::pairwise(x, &Cosine) (src/distance.rs:58-74, 85-107): out[i][j] = 1 - dot(x_i, x_j) / (|x_i| |x_j|),
// zero diagonal.  Every sum is the reference's sequential fold in index order (no FMA: -ffp-contract=off); products
// and the denominator commute exactly, so both triangles equal the reference's mirrored fill.  Small matrices by
// nature (n^2 outputs): one thread per pair, norms first.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void cosine_norms_kernel(const T *__restrict__ X, size_t n, int dim, size_t ld, T *__restrict__ norms) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T *r = X + i * ld;
    T s = (T)0;
    for (int k = 0; k < dim; ++k) {
        const T pr = r[k] * r[k];
        s = s + pr;
    }
    norms[i] = pn_sqrt(s);
}
template <typename T>
__global__ void cosine_pairwise_kernel(const T *__restrict__ X, size_t n, int dim, size_t ld,
                                       const T *__restrict__ norms, T *__restrict__ out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    if (i == j) {
        out[i * n + j] = (T)0;
        return;
    }
    const T *a = X + i * ld, *b = X + j * ld;
    T dot = (T)0;
    for (int k = 0; k < dim; ++k) {
        const T pr = a[k] * b[k];
        dot = dot + pr;
    }
    const T den = norms[i] * norms[j];
    out[i * n + j] = (T)1 - dot / den;
}
hipError_t launch_cosine_norms_f32(const float *X, size_t n, int dim, size_t ld, float *norms, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((cosine_norms_kernel<float>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, X, n, dim, ld, norms);
    return hipGetLastError();
}
hipError_t launch_cosine_norms_f64(const double *X, size_t n, int dim, size_t ld, double *norms, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((cosine_norms_kernel<double>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, X, n, dim, ld, norms);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_cosine_pairwise(const T *X, size_t n, int dim, size_t ld, T *norms, T *out, hipStream_t s) {
    hipLaunchKernelGGL((cosine_norms_kernel<T>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, X, n, dim, ld, norms);
    hipLaunchKernelGGL((cosine_pairwise_kernel<T>), dim3((unsigned)((n + 127) / 128), (unsigned)n), dim3(128), 0, s, X, n,
                       dim, ld, norms, out);
    return hipGetLastError();
}
hipError_t launch_cosine_pairwise_f32(const float *X, size_t n, int dim, size_t ld, float *norms, float *out,
                                      hipStream_t s) {
    return launch_cosine_pairwise<float>(X, n, dim, ld, norms, out, s);
}
hipError_t launch_cosine_pairwise_f64(const double *X, size_t n, int dim, size_t ld, double *norms, double *out,
                                      hipStream_t s) {
    return launch_cosine_pairwise<double>(X, n, dim, ld, norms, out, s);
}

}  // namespace pn
