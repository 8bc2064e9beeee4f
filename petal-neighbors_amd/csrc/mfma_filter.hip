// mfma_filter.hip -- the hot kernel: batched pairwise lower bounds on the f32
// matrix cores of gfx950, with a fused running top-k' filter.  Replaces the
// per-query tree walk + leaf scan of petal-neighbors (src/ball_tree.rs:203-243)
// for f32 corpora with D <= 128; nothing Q x N is ever written to memory.
//
// WHAT IS COMPUTED.  For every (query q, corpus row p) one f32 value
//       L(q,p) = qn'(q) + pn'(p) + sum_k (-2 q_k) * p_k
// as ONE chain of D+2 fused multiply-adds on v_mfma_f32_32x32x2_f32 (exact
// f32, bit-for-bit an fmaf chain, no flush of subnormals).  qn', pn' are the
// squared norms accumulated in f64 and rounded DOWN after scaling by
// (1 - alpha), alpha = (2(D+2)+8) * 2^-24, minus 1e-36 (pack.hip: row_norms).
//
// WHY IT IS A LOWER BOUND.  Standard fma-chain analysis: the computed chain
// differs from the exact sum of its D+2 terms t_j by at most gamma_{D+2} *
// sum|t_j| (+ (D+2) * 2^-149 for gradual underflow), and
// sum|t_j| = qn' + pn' + 2 sum|q_k p_k| <= qn' + pn' + (|q|^2 + |p|^2).  With
// qn' <= |q|^2 (1 - alpha), pn' likewise:
//       L <= (|q|^2+|p|^2)[(1-alpha)(1+gamma) + gamma] - 2 q.p  <=  |q-p|^2
// because alpha >= 2 gamma_{D+2} / (1 + gamma_{D+2}).  So L <= d2(q,p) in REAL
// arithmetic for every pair; the order in which k is summed is irrelevant to
// the bound, which lets the kernel feed the MFMA 8 coordinates per 16-B LDS
// read.  select.hip turns "L >= tau for every dropped row" into a proof that
// the exact top-k is among the candidates, or flags the query for the exact
// engine.  Rows / queries beyond the real counts carry norm +inf: L = +inf.
//
// MAPPING (one process per GPU; grid = query tiles x corpus segments):
//   workgroup = 256 threads = 4 waves, 2 workgroups per CU (2 waves / SIMD so
//   one wave's filter epilogue overlaps the other's MFMAs);
//   wave w owns 32 queries; they are the MFMA **B** operand and stay in
//   VGPRs for the whole sweep (D/2 registers per lane, pre-scaled by -2);
//   corpus rows are the **A** operand: 64-row tiles, read from HBM once per
//   workgroup in 16-B coalesced loads, double buffered through LDS (register
//   staged: next tile's loads are issued before this tile's MFMAs, written
//   after), rows padded by one 16-B chunk so the b128 fragment reads are
//   bank-conflict free;
//   the 32x32 accumulator puts the QUERY on the lane (column) and 16 rows in
//   registers, so the filter is lane-local: min over 32 registers, one compare
//   against the lane's threshold, and a wave-uniform branch that is rarely
//   taken once the threshold has tightened.
//
// ROOFLINE: MFMA-bound.  Algorithmic work 2*D flop per pair; per 64x32 wave
// tile 2*(D/2+1) MFMAs of 64 cycles each.  HBM traffic is the corpus once per
// 128 queries (4*D*N*ceil(Q/128) bytes) -- about 2 TB/s at the MFMA roof for
// D = 128, far below the 8 TB/s peak.
#include "pn_internal.h"
#include "topk_buffer.h"

namespace pn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: HIP's float4 struct defeats SROA here

constexpr int kMfQ = 128;  // queries per workgroup
constexpr int kMfP = 64;   // corpus rows per LDS tile

template <int NKG, int M>
__global__ __launch_bounds__(256, 2) void mfma_filter_kernel(
    const float *__restrict__ P, const float *__restrict__ pnorm, size_t n, const float *__restrict__ Q,
    const float *__restrict__ qnorm, uint32_t kp, size_t seg_len, uint32_t *__restrict__ ckey,
    uint32_t *__restrict__ cidx, uint32_t *__restrict__ ccnt, uint32_t *__restrict__ ctau, size_t nq_pad) {
    constexpr int LD = 8 * NKG;            // floats per row in HBM
    constexpr int STR = LD + 4;            // floats per row in LDS (+ one 16-B chunk)
    constexpr uint32_t CAP = 64u * M;
    constexpr int CHUNKS = kMfP * LD / 4;  // 16-B chunks per tile
    constexpr int NLD = (CHUNKS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tiles = reinterpret_cast<float *>(smem_raw);                     // [2][64][STR]
    float *pnl = tiles + 2 * kMfP * STR;                                    // [2][64]
    uint32_t *taus = reinterpret_cast<uint32_t *>(pnl + 2 * kMfP);          // [128]
    uint32_t *cnts = taus + kMfQ;                                           // [128]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jq = lane & 31, h = lane >> 5;
    const size_t q0 = (size_t)blockIdx.x * kMfQ + (size_t)wave * 32;  // first query of this wave
    const size_t seg = blockIdx.y;
    const size_t p_begin = seg * seg_len;
    const size_t p_end = (p_begin + seg_len < n) ? p_begin + seg_len : n;
    uint32_t *taus_w = taus + wave * 32, *cnts_w = cnts + wave * 32;
    if (lane < 32) { taus_w[lane] = 0xFF800000u; cnts_w[lane] = 0; }  // f2s(+inf)

    // ---- B operand: this lane's query, coordinates {8kg + 4h + j}, scaled by -2 (exact)
    float b[4 * NKG];
    {
        const float *qrow = Q + (q0 + jq) * LD + 4 * h;
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(qrow + 8 * kg);
            b[4 * kg + 0] = -2.0f * v.x; b[4 * kg + 1] = -2.0f * v.y;
            b[4 * kg + 2] = -2.0f * v.z; b[4 * kg + 3] = -2.0f * v.w;
        }
    }
    const float bn = h ? qnorm[q0 + jq] : 1.0f;  // norm step: A = [pn', 1], B = [1, qn']
    const size_t cbase = (seg * nq_pad + q0) * (size_t)CAP;

    // register staging of the next tile.  Kept as plain, explicitly unrolled code at each use: a
    // lambda / helper / macro-wrapped loop is not unrolled early enough and sends st[] to scratch.
    f32x4 st[NLD];
    float stn = 0.0f;
    const uint32_t lane_off = (uint32_t)tid * 16u;

    float tau = __uint_as_float(0x7F800000u);
    if (p_begin < p_end) {
        {
            // uniform (SGPR) tile base + loop-invariant per-lane byte offset -> saddr-form loads whose
            // address VGPR is never recycled as an MFMA destination
            const char *src_ = reinterpret_cast<const char *>(P) + (p_begin) * (size_t)(LD * 4);
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_) {
                if (CHUNKS % 256 == 0 || tid + 256 * i_ < CHUNKS)
                    st[i_] = *reinterpret_cast<const f32x4 *>(src_ + (lane_off + 4096u * i_));
            }
            stn = pnorm[(p_begin) + (tid & 63)];  // unconditional: a branch here costs a vmcnt(0) in the MFMA chain
        }
        {
            float *dst_ = tiles + (0) * kMfP * STR;
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_) {
                const int c_ = tid + 256 * i_;
                if (CHUNKS % 256 == 0 || c_ < CHUNKS) {
                    const int row_ = c_ / (LD / 4), cc_ = c_ % (LD / 4);
                    *reinterpret_cast<f32x4 *>(dst_ + row_ * STR + 4 * cc_) = st[i_];
                }
            }
            if (tid < kMfP) pnl[(0) * kMfP + tid] = stn;
        }
    }
    __syncthreads();

    int cur = 0;
    for (size_t p0 = p_begin; p0 < p_end; p0 += kMfP, cur ^= 1) {
        const bool more = p0 + kMfP < p_end;
        if (more)
        {
            const char *src_ = reinterpret_cast<const char *>(P) + (p0 + kMfP) * (size_t)(LD * 4);
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_) {
                if (CHUNKS % 256 == 0 || tid + 256 * i_ < CHUNKS)
                    st[i_] = *reinterpret_cast<const f32x4 *>(src_ + (lane_off + 4096u * i_));
            }
            stn = pnorm[(p0 + kMfP) + (tid & 63)];
        }

        const float *tl = tiles + cur * kMfP * STR + jq * STR + 4 * h;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        {
            const float an0 = h ? 1.0f : pnl[cur * kMfP + jq];
            const float an1 = h ? 1.0f : pnl[cur * kMfP + 32 + jq];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(an0, bn, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(an1, bn, acc1, 0, 0, 0);
        }
        {
            // fragment reads run ONE k-group (8 MFMAs = 512 cycles) ahead of their use: a b128 LDS
            // read issued only two MFMAs early does not land in time when 8 waves share the LDS
            f32x4 a0 = *reinterpret_cast<const f32x4 *>(tl);
            f32x4 a1 = *reinterpret_cast<const f32x4 *>(tl + 32 * STR);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // group 0's reads
#pragma unroll
            for (int kg = 0; kg < NKG; ++kg) {
                f32x4 n0 = a0, n1 = a1;
                if (kg + 1 < NKG) {
                    n0 = *reinterpret_cast<const f32x4 *>(tl + 8 * (kg + 1));
                    n1 = *reinterpret_cast<const f32x4 *>(tl + 32 * STR + 8 * (kg + 1));
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the two DS reads first ...
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b[4 * kg + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b[4 * kg + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b[4 * kg + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b[4 * kg + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b[4 * kg + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b[4 * kg + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b[4 * kg + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b[4 * kg + 3], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // ... then this group's 8 MFMAs
                a0 = n0;
                a1 = n1;
            }
        }

        // ---- fused filter: the lane holds 32 lower bounds of ITS query
        float m = fminf(acc0[0], acc1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fminf(m, fminf(acc0[r], acc1[r]));
        // Staged tile -> LDS first: its wait retires the prefetch loads, so the (rare) slow path below
        // starts with an empty VMEM queue and can drain its own stores cheaply.
        if (more)
        {
            float *dst_ = tiles + (cur ^ 1) * kMfP * STR;
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_) {
                const int c_ = tid + 256 * i_;
                if (CHUNKS % 256 == 0 || c_ < CHUNKS) {
                    const int row_ = c_ / (LD / 4), cc_ = c_ % (LD / 4);
                    *reinterpret_cast<f32x4 *>(dst_ + row_ * STR + 4 * cc_) = st[i_];
                }
            }
            if (tid < kMfP) pnl[(cur ^ 1) * kMfP + tid] = stn;
        }
#ifdef PN_DIAG_NO_SLOWPATH  // timing-only diagnostic build: results are wrong, never shipped
        asm volatile("" ::"v"(m));
        if (false) {
#else
        if (__any(m < tau)) {
#endif
            // one LDS atomic per lane reserves room for all of its survivors
            uint32_t mask0 = 0, mask1 = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mask0 |= (acc0[r] < tau ? 1u : 0u) << r;
                mask1 |= (acc1[r] < tau ? 1u : 0u) << r;
            }
            const uint32_t npass = (uint32_t)__popc(mask0) + (uint32_t)__popc(mask1);
            if (npass) {
                size_t o = cbase + (size_t)jq * CAP + atomicAdd(&cnts_w[jq], npass);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = (uint32_t)p0 + (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D map of 32x32
                    if (mask0 & (1u << r)) {
                        ckey[o] = f2s(acc0[r]);
                        cidx[o] = row;
                        ++o;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = (uint32_t)p0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (mask1 & (1u << r)) {
                        ckey[o] = f2s(acc1[r]);
                        cidx[o] = row;
                        ++o;
                    }
                }
            }
            wg_fence();
            const uint32_t c = cnts_w[jq];
            unsigned long long need = __ballot(h == 0 && c + 64u > CAP);
            while (need) {
                const int j = __builtin_ctzll(need);
                need &= need - 1;
                const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
                compact_query<uint32_t, M>(ckey, cidx, cbase + (size_t)j * CAP, cj, kp, lane, &taus_w[j],
                                           &cnts_w[j], 0xFFFFFFFFu);
            }
            tau = s2f(taus_w[jq]);
            // Leave no candidate store pending: a mix of pending stores and the next tile's prefetch
            // loads makes hipcc's wait-count pass emit vmcnt(0) inside the next MFMA chain.
            __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        }

        __syncthreads();
    }

    // leave at most kp candidates per (segment, query); publish count and threshold
    {
        const uint32_t c = cnts_w[jq];
        unsigned long long need = __ballot(h == 0 && c > kp);
        while (need) {
            const int j = __builtin_ctzll(need);
            need &= need - 1;
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
            compact_query<uint32_t, M>(ckey, cidx, cbase + (size_t)j * CAP, cj, kp, lane, &taus_w[j], &cnts_w[j],
                                       0xFFFFFFFFu);
        }
        if (lane < 32) {
            ccnt[seg * nq_pad + q0 + lane] = cnts_w[lane];
            ctau[seg * nq_pad + q0 + lane] = taus_w[lane];
        }
    }
}

bool mfma_supported(int dim, size_t ld) {
    if (dim < 1) return false;
    switch (ld) {
        case 8: case 16: case 32: case 64: case 96: case 128: return true;
        default: return ld > 128 && ld % 128 == 0;  // wide rows: slab-accumulating kernel (mfma_filter_v2.hip)
    }
}
const char *mfma_kernel_name() { return "mfma_filter_kernel"; }

template <int NKG, int M>
static hipError_t launch_one(const float *P, const float *pnorm, size_t n, const float *Q, const float *qnorm,
                             const MfmaPlan &plan, const CandBuf &cb, hipStream_t s) {
    constexpr int LD = 8 * NKG;
    const size_t sh = (size_t)(2 * kMfP * (LD + 4) + 2 * kMfP) * sizeof(float) + 2 * kMfQ * sizeof(uint32_t);
    auto kern = mfma_filter_kernel<NKG, M>;
    static LdsAttrOnce lds_attr;  // per instantiation
    {
        const hipError_t e = lds_attr.ensure(reinterpret_cast<const void *>(kern), sh);
        if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)(cb.nq_pad / kMfQ), (unsigned)plan.nseg), block(256);
    hipLaunchKernelGGL(kern, grid, block, sh, s, P, pnorm, n, Q, qnorm, (uint32_t)plan.kp, plan.seg_len,
                       static_cast<uint32_t *>(cb.keys), cb.idx, cb.cnt, static_cast<uint32_t *>(cb.tau),
                       cb.nq_pad);
    return hipGetLastError();
}

hipError_t launch_mfma_filter_f32(const float *P, const float *pnorm, size_t n, size_t n_pad, int dim, size_t ldp,
                                  const float *Q, const float *qnorm, int nq, size_t ldq, const MfmaPlan &plan,
                                  const CandBuf &cb, hipStream_t s) {
    (void)n_pad; (void)dim; (void)nq;
    if (ldq != ldp || cb.nq_pad % kMfQ || plan.seg_len % kMfP) return hipErrorInvalidValue;
    const int m = cb.cap / 64;
#define PN_CASE(NKG)                                                                    \
    case 8 * NKG:                                                                       \
        if (m == 2) return launch_one<NKG, 2>(P, pnorm, n, Q, qnorm, plan, cb, s);      \
        if (m == 4) return launch_one<NKG, 4>(P, pnorm, n, Q, qnorm, plan, cb, s);      \
        return hipErrorInvalidValue;
    switch (ldp) {
        PN_CASE(1)
        PN_CASE(2)
        PN_CASE(4)
        PN_CASE(8)
        PN_CASE(12)
        PN_CASE(16)
        default: return hipErrorInvalidValue;
    }
#undef PN_CASE
}

}  // namespace pn
