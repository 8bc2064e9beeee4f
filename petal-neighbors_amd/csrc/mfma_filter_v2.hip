// mfma_filter_v2.hip -- second tier of the k-NN path (and the radius filter of the f32 tier): proven lower
// bounds of |q-p|^2 on the f32 matrix cores, persistent-partition structures.
//
// WHAT IS COMPUTED.  For every (query q, corpus row p) one f32 value
//       L(q,p) = qn'(q) + pn'(p) + sum_k (-2 q_k) * p_k
// as ONE chain of D+2 fused multiply-adds on v_mfma_f32_32x32x2_f32 (exact f32, bit-for-bit an fmaf chain, no flush of
// subnormals), query on the lane, 16 rows in registers.  qn', pn' are the squared norms accumulated in f64 and rounded
// DOWN after scaling by (1 - alpha), alpha = (2(D+2)+8) * 2^-24, minus 1e-36 (pack.hip: row_norms).
//
// WHY IT IS A LOWER BOUND.  Standard fma-chain analysis: the computed chain differs from the exact sum of its D+2 terms
// t_j by at most gamma_{D+2} * sum|t_j| (+ (D+2) * 2^-149 for gradual underflow), and
// sum|t_j| = qn' + pn' + 2 sum|q_k p_k| <= qn' + pn' + (|q|^2 + |p|^2).  With qn' <= |q|^2 (1 - alpha), pn' likewise:
//       L <= (|q|^2+|p|^2)[(1-alpha)(1+gamma) + gamma] - 2 q.p  <=  |q-p|^2
// because alpha >= 2 gamma_{D+2} / (1 + gamma_{D+2}).  So L <= d2(q,p) in REAL arithmetic for every pair; the order in
// which k is summed is irrelevant to the bound, which lets the kernel feed the MFMA 8 coordinates per 16-B LDS read.
// select.hip turns "L >= tau for every dropped row" into a proof that the exact top-k is among the candidates, or flags
// the query for the exact engine.  Rows / queries beyond the real counts carry norm +inf: L = +inf.
// (Round 1's first structure -- a static (query tile x segment) grid, mfma_filter.hip -- computed the same bound; it
// was retired in round 4 once nothing but PN_OPT_MFMA_STRUCTURE = 1 selected it.)
//
// How the chip is kept busy; every item below was measured (profiles/r01_*):
//
//  * BALANCED PERSISTENT PARTITION.  The work is the list of (query tile, 64-row tile) units in query-major
//    order; workgroup w of W owns the contiguous slice [w*U/W, (w+1)*U/W).  Every workgroup gets the same number
//    of MFMAs (+-1 tile) for any Q and N -- the static (query tile x segment) grid of round 1 left 6-7 %
//    of the workgroup slots empty at Q = 10^4.  A slice crosses at most a few query-tile boundaries; at each it
//    flushes its candidates and reloads the B operand.  "Segment" of a query tile = ordinal of the workgroup
//    among those that touch it (<= ceil(W/QT)+1 of them).
//  * SOFTWARE PIPELINE.  Two accumulator sets: the 65-MFMA chain of one 32-row block runs while the VALU
//    reduces the previous block (minimum of its 16 bounds, compare with the lane's threshold), so the matrix
//    pipe only drains at the one barrier per tile.
//  * TILES BY LDS-DMA (global_load_lds, 16 B per lane) into an XOR-swizzled image (D = 64, 128); register
//    staging for the other row lengths.
//  * CANDIDATE BUFFERS: template GC.  GC = false ("structure 2"): 64 (key,row) slots per query in LDS, one
//    workgroup per CU.  GC = true ("structure 3", the default): buffers in a per-workgroup HBM slab, LDS holds
//    only the tiles, TWO workgroups per CU -- a second wave per SIMD runs while the first sits in its rare path,
//    at a barrier or at the head of a chain (C2: 22.1 ms vs 24.0 ms), and k' up to 224 (64*M slots, M = 1, 2, 4).
//  * D > 128: mfma_filter_wide_kernel processes rows in 128-coordinate slabs.
//  * mfma_radius_kernel: the same chain against one fixed threshold, survivor lists instead of top-k' buffers.
//
// LDS (D = 128): 2 x 32 KB tiles + 0.5 KB norms (+ 64 KB candidates when GC = false).
#include "pn_internal.h"
#include "topk_buffer.h"

namespace pn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef PN_DIAG_COUNT  // diagnostic build only: event counters (slow blocks, compactions, appends)
__device__ unsigned long long g_dbg[8];
#define PN_COUNT(i, v) atomicAdd(&g_dbg[i], (unsigned long long)(v))
// shader-clock stamps around a region, summed per wave (lane 0) into g_dbg[i]
__device__ __forceinline__ unsigned long long pn_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define PN_T0() const unsigned long long t0_ = pn_stamp()
#define PN_T1(i) do { const unsigned long long t1_ = pn_stamp(); if (lane == 0) atomicAdd(&g_dbg[i], t1_ - t0_); } while (0)
#else
#define PN_COUNT(i, v) ((void)0)
#define PN_T0() ((void)0)
#define PN_T1(i) ((void)0)
#endif

constexpr int kV2Q = 128;      // queries per workgroup (4 waves x 32)
constexpr int kV2P = 64;       // rows per tile
constexpr uint32_t kV2Cap = 64;   // LDS slots per query
constexpr uint32_t kV2Keep = 32;  // slots per (segment, query) handed to the select kernel

// Ordering point for LDS traffic inside ONE wave (candidate buffers are wave-private): LDS
// executes a wave's operations in order, so only the compiler must be held back.  Unlike a
// workgroup-scope fence this does not wait for the in-flight prefetch of the next tile (vmcnt).
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// candidate buffers may instead live in HBM (2 workgroups per CU variant): then other lanes' stores
// must have left the wave before they are read back
__device__ __forceinline__ void cand_fence(bool hbm) {
    if (hbm) {
        wg_fence();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
        lds_fence();
    }
}

// radix-select compaction of ONE query's LDS buffer (<= 64 entries, one per lane); see topk_buffer.h
__device__ __forceinline__ void compact_lds(uint32_t *__restrict__ ck, uint32_t *__restrict__ ci, uint32_t n,
                                            uint32_t kp, int lane, uint32_t *tau_slot, uint32_t *cnt_slot,
                                            bool hbm = false) {
    if (lane == 0) PN_COUNT(1, 1);
    const bool valid = (uint32_t)lane < n;
    const uint32_t key = valid ? ck[lane] : 0xFFFFFFFFu;
    const uint32_t ix = valid ? ci[lane] : 0xFFFFFFFFu;
    uint32_t T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        if ((uint32_t)__popcll(__ballot(valid && key < cand)) < kp) T = cand;
    }
    const bool less = valid && key < T, eq = valid && key == T;
    const uint32_t n_less = (uint32_t)__popcll(__ballot(less));
    const uint32_t n_eq = (uint32_t)__popcll(__ballot(eq));
    const uint32_t need_eq = kp - n_less;
    uint32_t row_cut = 0xFFFFFFFFu;
    if (n_eq > need_eq) {
        uint32_t I = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = I | (1u << bit);
            if ((uint32_t)__popcll(__ballot(eq && ix < cand)) < need_eq) I = cand;
        }
        row_cut = I;
    }
    const bool keep = less || (eq && ix <= row_cut);
    const unsigned long long mask = __ballot(keep);
    cand_fence(hbm);  // all lanes have read their entry
    if (keep) {
        const uint32_t o = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        ck[o] = key;
        ci[o] = ix;
    }
    if (lane == 0) {
        *tau_slot = T;
        *cnt_slot = (uint32_t)__popcll(mask);
    }
    cand_fence(hbm);
}

// append the survivors of one 32-row block (16 registers per lane) to the lane's query buffer
__device__ __forceinline__ void append_block(const f32x16 &acc, float tau, uint32_t row0, int h, uint32_t *ckq,
                                             uint32_t *ciq, uint32_t *cnt_q) {
    uint32_t mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) mask |= (acc[r] < tau ? 1u : 0u) << r;
    const uint32_t npass = (uint32_t)__popc(mask);
    if (npass) {
        PN_COUNT(2, npass);
        uint32_t o = atomicAdd(cnt_q, npass);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (mask & (1u << r)) {
                ckq[o] = f2s(acc[r]);
                ciq[o] = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D map of the 32x32 MFMA
                ++o;
            }
        }
    }
}

// Everything one wave needs across tiles (kept in registers; passed by reference to the
// force-inlined step so the two accumulator sets can swap roles without copies).
template <int NKG>
struct V2Ctx {
    static constexpr int LD = 8 * NKG;
    // LDS-DMA staging (global_load_lds, 16 B per lane, 1 KiB per wave instruction, no VGPRs, no
    // ds_write) needs a lane-linear LDS image, i.e. unpadded rows; bank conflicts of the b128
    // fragment reads are then avoided by an XOR swizzle applied to the SOURCE address and to the
    // read address (both-sides-or-neither).  Instantiated for the row lengths where one wave
    // instruction covers whole rows and the swizzle is a plain 4-bit XOR: D = 64 and 128.
    static constexpr bool GLDS = (NKG == 16 || NKG == 8);
    static constexpr int STR = GLDS ? LD : LD + 4;
    static constexpr int CHUNKS = kV2P * LD / 4;
    static constexpr int NLD = (CHUNKS + 255) / 256;
    static constexpr int NDMA = CHUNKS / 256;  // wave instructions per wave per tile (GLDS)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// 65-MFMA chain of one 32-row block (rows at `tl`, norm operand `an`) into `w`, while the VALU
// scans the OTHER block `r` (16 bounds) in the matrix pipe's shadow: minimum, its register index
// and the number of bounds below tau.  ~80 VALU ops against ~900 free issue slots.
struct BlockScan {
    float m;        // smallest bound of the block for this lane's query
    int am;         // register holding it
    uint32_t npass; // bounds < tau
};
template <int NKG>
__device__ __forceinline__ BlockScan v2_chain(const float *trow, const int (&foff)[NKG], f32x4 a_first, float an,
                                              float bn,
                                              const float (&b)[4 * NKG], f32x16 &w, const f32x16 &r, float tau) {
    // trow = this lane's row in the LDS tile; foff[kg] = float offset of k-group kg's fragment in
    // that row: chunk (2kg+h), stored at chunk (2kg+h)^(row&15) when the tile came by LDS-DMA.
    // The offsets are loop invariant and live in registers (computing the XOR per read put ~2 ms
    // of address arithmetic into the chain).
    auto frag = [&](int kg) -> const f32x4 * { return reinterpret_cast<const f32x4 *>(trow + foff[kg]); };
    f32x4 a = a_first;  // k-group 0, read by the caller ahead of time
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = 0.0f;
    BlockScan sc{r[0], 0, r[0] < tau ? 1u : 0u};
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
        f32x4 nx = a;
        if (kg + 1 < NKG) nx = *frag(kg + 1);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * kg + 0], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * kg + 1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * kg + 2], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * kg + 3], w, 0, 0, 0);
        a = nx;
        if (kg == 0) {  // scan of the other block: issued behind the first MFMAs
#ifdef PN_LIGHT_SCAN
            // minimum only (8 v_min3); the rare path recomputes which register and how many
#pragma unroll
            for (int i = 1; i < 16; ++i) sc.m = fminf(sc.m, r[i]);
            sc.npass = sc.m < tau ? 1u : 0u;
#else
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                const bool lt = r[i] < sc.m;
                sc.m = lt ? r[i] : sc.m;
                sc.am = lt ? i : sc.am;
                sc.npass += r[i] < tau ? 1u : 0u;
            }
#endif
        }
    }
    // norm step last: its LDS operand has had the whole chain to arrive
    w = __builtin_amdgcn_mfma_f32_32x32x2f32(an, bn, w, 0, 0, 0);
    return sc;
}

// LDS-DMA issue of one 64-row tile (+ its 64 norms) into LDS buffer `buf`: wave w, instruction i
// writes the 1 KiB run g = w*NDMA + i; lane l of it is 16-B chunk (l mod CPR) of row
// g*RPI + l/CPR and fetches the source chunk with the low four index bits XORed by (row & 15).
template <int NKG>
__device__ __forceinline__ void v2_dma_tile(const float *__restrict__ P, const float *__restrict__ pnorm, float *tiles,
                                            float *pnl, uint32_t rt, int buf, int wave, int lane) {
    constexpr int LD = V2Ctx<NKG>::LD, NDMA = V2Ctx<NKG>::NDMA;
    constexpr int CPR = LD / 4;      // chunks per row
    constexpr int RPI = 64 / CPR;    // rows per wave instruction
    const char *src = reinterpret_cast<const char *>(P) + (size_t)rt * (size_t)(kV2P * LD * 4);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int g = wv * NDMA + i;
        const int row = g * RPI + lane / CPR;
        const int pos = lane % CPR;
        const int chunk = pos ^ (row & 15);
        __builtin_amdgcn_global_load_lds((glb_void *)(src + (size_t)row * (LD * 4) + chunk * 16),
                                         (lds_void *)(tiles + buf * kV2P * LD + g * 256), 16, 0, 0);
    }
    if (wv == 0)
        __builtin_amdgcn_global_load_lds((glb_void *)(pnorm + (size_t)rt * kV2P + lane),
                                         (lds_void *)(pnl + buf * kV2P), 4, 0, 0);
}

// rare path: append the survivors of one 32-row block, compact where more than 32 slots are used.
// Steady state: a lane has at most ONE survivor, which is its minimum -- appended without looking
// at the other 15 registers; the general 16-way descent runs only when some lane has several.
template <int M>
__device__ __forceinline__ void v2_slow(const f32x16 &acc, const BlockScan &sc, float &tau, uint32_t row0, int h,
                                        int wave, int lane, uint32_t kp, uint32_t *cand_k, uint32_t *cand_i,
                                        uint32_t *taus_w, uint32_t *cnts_w, uint32_t *ckq, uint32_t *ciq,
                                        uint32_t *cnt_q, uint32_t *tau_q, bool hbm) {
    if (lane == 0) PN_COUNT(0, 1);
#ifdef PN_LIGHT_SCAN
    if (true) {
#else
    if (__any(sc.npass > 1)) {
#endif
        append_block(acc, tau, row0, h, ckq, ciq, cnt_q);
    } else if (sc.npass) {
        PN_COUNT(2, 1);
        const uint32_t o = atomicAdd(cnt_q, 1u);
        ckq[o] = f2s(sc.m);
        ciq[o] = row0 + (sc.am & 3) + 8 * (sc.am >> 2) + 4 * h;  // C/D map of the 32x32 MFMA
    }
    cand_fence(hbm);
    constexpr uint32_t CAP = 64u * M;
    const uint32_t c = *cnt_q;
    unsigned long long need = __ballot(h == 0 && c > CAP - 32);
    if (need) {
        do {
            const int j = __builtin_ctzll(need);
            need &= need - 1;
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
            if constexpr (M == 1)
                compact_lds(cand_k + (wave * 32 + j) * CAP, cand_i + (wave * 32 + j) * CAP, cj, kp, lane, &taus_w[j],
                            &cnts_w[j], hbm);
            else  // larger buffers live in HBM only: the generic radix-select compaction (topk_buffer.h)
                compact_query<uint32_t, M>(cand_k, cand_i, (size_t)(wave * 32 + j) * CAP, cj, kp, lane, &taus_w[j],
                                           &cnts_w[j], 0xFFFFFFFFu);
        } while (need);
        tau = s2f(*tau_q);
    }
}

// One pipeline step over tile `rt` (64 rows = two 32-row blocks).  The two accumulators swap roles
// inside the step, so nothing is copied and the matrix pipe is fed back to back:
//   chain(block 0 of rt) -> acc0   while filtering acc1 = block 1 of tile rt-1
//   chain(block 1 of rt) -> acc1   while filtering acc0
// then tile rt+1 is written to the other LDS buffer and the workgroup meets at ONE barrier.
template <int NKG, bool GC, int M>
__device__ __forceinline__ void v2_step(const float *__restrict__ P, const float *__restrict__ pnorm, float *tiles,
                                        float *pnl, uint32_t *cand_k, uint32_t *cand_i, uint32_t *taus_w,
                                        uint32_t *cnts_w, uint32_t *ckq, uint32_t *ciq, uint32_t *cnt_q,
                                        uint32_t *tau_q, const float (&b)[4 * NKG], const int (&foff)[NKG], float bn,
                                        float &tau, f32x4 (&st)[V2Ctx<NKG>::NLD], float &stn, uint32_t rt, uint32_t rt1, int cur,
                                        f32x16 &acc0, f32x16 &acc1, uint32_t kp, int tid, int wave, int lane, int jq,
                                        int h, uint32_t lane_off) {
    constexpr int LD = V2Ctx<NKG>::LD, STR = V2Ctx<NKG>::STR, CHUNKS = V2Ctx<NKG>::CHUNKS, NLD = V2Ctx<NKG>::NLD;
    constexpr bool GLDS = V2Ctx<NKG>::GLDS;
    const bool more = rt + 1 < rt1;
    const float *trow = tiles + cur * kV2P * STR + jq * STR;
    // first fragments of BOTH blocks and the norm operands are read now, right behind the barrier:
    // block 1's then arrive during block 0's chain instead of stalling its own
    const f32x4 fa0 = *reinterpret_cast<const f32x4 *>(trow + foff[0]);
    const f32x4 fa1 = *reinterpret_cast<const f32x4 *>(trow + 32 * STR + foff[0]);
    const float an0 = h ? 1.0f : pnl[cur * kV2P + jq];
    const float an1 = h ? 1.0f : pnl[cur * kV2P + 32 + jq];
#ifdef PN_DIAG_NO_STAGE
    if (false) {
#else
    if (more) {
#endif
        if constexpr (GLDS) {
            v2_dma_tile<NKG>(P, pnorm, tiles, pnl, rt + 1, cur ^ 1, wave, lane);
        } else {
            const char *src_ = reinterpret_cast<const char *>(P) + (size_t)(rt + 1) * (size_t)(kV2P * LD * 4);
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_)
                if (CHUNKS % 256 == 0 || tid + 256 * i_ < CHUNKS)
                    st[i_] = *reinterpret_cast<const f32x4 *>(src_ + (lane_off + 4096u * i_));
            stn = pnorm[(size_t)(rt + 1) * kV2P + (tid & 63)];
        }
    }

    const BlockScan s1 = v2_chain<NKG>(trow, foff, fa0, an0, bn, b, acc0, acc1, tau);
#ifdef PN_DIAG_NO_SLOWPATH
    asm volatile("" ::"v"(s1.m), "v"(s1.am), "v"(s1.npass));
#else
    if (__any(s1.npass != 0)) {
        PN_T0();
        v2_slow<M>(acc1, s1, tau, (rt - 1) * kV2P + 32, h, wave, lane, kp, cand_k, cand_i, taus_w, cnts_w, ckq, ciq,
                cnt_q, tau_q, GC);
        PN_T1(3);
    }
#endif
    const BlockScan s0 = v2_chain<NKG>(trow + 32 * STR, foff, fa1, an1, bn, b, acc1, acc0, tau);
#ifdef PN_DIAG_NO_SLOWPATH
    asm volatile("" ::"v"(s0.m), "v"(s0.am), "v"(s0.npass));
#else
    if (__any(s0.npass != 0)) {
        PN_T0();
        v2_slow<M>(acc0, s0, tau, rt * kV2P, h, wave, lane, kp, cand_k, cand_i, taus_w, cnts_w, ckq, ciq, cnt_q, tau_q, GC);
        PN_T1(3);
    }
#endif

    // register-staged tile rt+1 -> LDS[cur^1] (its loads had both chains to land); with LDS-DMA the
    // data is already on its way into LDS and the barrier below carries the vmcnt(0)
#ifdef PN_DIAG_NO_STAGE
    if (false) {
#else
    if (!GLDS && more) {
#endif
        float *dst_ = tiles + (cur ^ 1) * kV2P * STR;
#pragma unroll
        for (int i_ = 0; i_ < NLD; ++i_) {
            const int c_ = tid + 256 * i_;
            if (CHUNKS % 256 == 0 || c_ < CHUNKS)
                *reinterpret_cast<f32x4 *>(dst_ + (c_ / (LD / 4)) * STR + 4 * (c_ % (LD / 4))) = st[i_];
        }
        if (tid < kV2P) pnl[(cur ^ 1) * kV2P + tid] = stn;
    }
#ifndef PN_DIAG_NO_BARRIER
    {
        PN_T0();
        __syncthreads();
        PN_T1(4);
    }
#endif
}

// GC = false: candidate buffers in LDS, one workgroup per CU.  GC = true: candidate buffers in HBM
// (one 64 KB slab per workgroup in `gcand`), LDS holds only the tiles -> two workgroups per CU, so
// each SIMD has a second wave to run while the first sits in its rare path, at a barrier or at the
// head of a chain.
template <int NKG, bool GC, int M>
__global__ __launch_bounds__(256, GC ? 2 : 1) void mfma_filter_v2_kernel(
    const float *__restrict__ P, const float *__restrict__ pnorm, uint32_t n_tiles, const float *__restrict__ Q,
    const float *__restrict__ qnorm, uint32_t q_tiles, uint32_t kp, uint32_t *__restrict__ ckey,
    uint32_t *__restrict__ cidx, uint32_t *__restrict__ ccnt, uint32_t *__restrict__ ctau, size_t nq_pad,
    uint32_t *__restrict__ gcand, uint32_t keep) {
    static_assert(GC || M == 1, "LDS candidate buffers hold 64 slots per query");
    constexpr uint32_t CAP = 64u * M;  // candidate slots per query (HBM buffers may be larger than 64)
    constexpr int LD = V2Ctx<NKG>::LD, STR = V2Ctx<NKG>::STR, CHUNKS = V2Ctx<NKG>::CHUNKS, NLD = V2Ctx<NKG>::NLD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tiles = reinterpret_cast<float *>(smem_raw);                   // [2][64][STR]
    float *pnl = tiles + 2 * kV2P * STR;                                  // [2][64]
    uint32_t *lds_tail = reinterpret_cast<uint32_t *>(pnl + 2 * kV2P);
    uint32_t *cand_k = GC ? gcand + (size_t)blockIdx.x * (2 * kV2Q * CAP) : lds_tail;  // [128][CAP]
    uint32_t *cand_i = cand_k + kV2Q * CAP;                                             // [128][CAP]
    uint32_t *taus = GC ? lds_tail : cand_i + kV2Q * CAP;                               // [128]
    uint32_t *cnts = taus + kV2Q;                                                          // [128]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jq = lane & 31, h = lane >> 5;
    const int ql = wave * 32 + jq;  // this lane's query within the workgroup tile
    uint32_t *ckq = cand_k + ql * CAP, *ciq = cand_i + ql * CAP;
    uint32_t *cnt_q = cnts + ql, *tau_q = taus + ql;
    uint32_t *taus_w = taus + wave * 32, *cnts_w = cnts + wave * 32;
    const uint32_t lane_off = (uint32_t)tid * 16u;

    const unsigned long long U = (unsigned long long)q_tiles * n_tiles;
    const unsigned long long W = gridDim.x, w = blockIdx.x;
    unsigned long long u0 = w * U / W;
    const unsigned long long u1 = (w + 1) * U / W;
#ifdef PN_DIAG_COUNT
    const unsigned long long tk0_ = pn_stamp();
#endif

    while (u0 < u1) {
        const uint32_t qt = (uint32_t)(u0 / n_tiles);
        const uint32_t rt0 = (uint32_t)(u0 % n_tiles);
        const unsigned long long q_end = (unsigned long long)(qt + 1) * n_tiles;
        const unsigned long long run_end = u1 < q_end ? u1 : q_end;
        const uint32_t rt1 = rt0 + (uint32_t)(run_end - u0);
        // ordinal of this workgroup among those touching query tile qt
        unsigned long long wf = ((unsigned long long)qt * n_tiles) * W / U;
        while ((wf + 1) * U / W <= (unsigned long long)qt * n_tiles) ++wf;
        while (wf > 0 && wf * U / W > (unsigned long long)qt * n_tiles) --wf;
        const uint32_t seg = (uint32_t)(w - wf);
        const size_t q0 = (size_t)qt * kV2Q + (size_t)wave * 32;

        // ---- per-run state
        if (h == 0) { *tau_q = 0xFF800000u; *cnt_q = 0; }  // sortable(+inf)
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
        const float bn = h ? qnorm[q0 + jq] : 1.0f;
        float tau = __uint_as_float(0x7F800000u);
        int foff[NKG];
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg)
            foff[kg] = 4 * (V2Ctx<NKG>::GLDS ? ((2 * kg + h) ^ (jq & 15)) : (2 * kg + h));

        f32x4 st[NLD];
        float stn = 0.0f;
        // ---- prologue: first tile -> LDS[0]
        __syncthreads();  // previous run's readers are done with both buffers
        if constexpr (V2Ctx<NKG>::GLDS) {
            v2_dma_tile<NKG>(P, pnorm, tiles, pnl, rt0, 0, wave, lane);
        } else {
            const char *src_ = reinterpret_cast<const char *>(P) + (size_t)rt0 * (size_t)(kV2P * LD * 4);
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_)
                if (CHUNKS % 256 == 0 || tid + 256 * i_ < CHUNKS)
                    st[i_] = *reinterpret_cast<const f32x4 *>(src_ + (lane_off + 4096u * i_));
            stn = pnorm[(size_t)rt0 * kV2P + (tid & 63)];
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_) {
                const int c_ = tid + 256 * i_;
                if (CHUNKS % 256 == 0 || c_ < CHUNKS)
                    *reinterpret_cast<f32x4 *>(tiles + (c_ / (LD / 4)) * STR + 4 * (c_ % (LD / 4))) = st[i_];
            }
            if (tid < kV2P) pnl[tid] = stn;
        }
        __syncthreads();

        // acc1 enters every step holding the not-yet-filtered second block of the previous tile
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = __uint_as_float(0x7F800000u); }
        int cur = 0;
        for (uint32_t rt = rt0; rt < rt1; ++rt, cur ^= 1)
            v2_step<NKG, GC, M>(P, pnorm, tiles, pnl, cand_k, cand_i, taus_w, cnts_w, ckq, ciq, cnt_q, tau_q, b, foff, bn, tau,
                         st, stn, rt, rt1, cur, acc0, acc1, kp, tid, wave, lane, jq, h, lane_off);
        {  // drain: second block of the last tile
            BlockScan sc{acc1[0], 0, acc1[0] < tau ? 1u : 0u};
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                const bool lt = acc1[i] < sc.m;
                sc.m = lt ? acc1[i] : sc.m;
                sc.am = lt ? i : sc.am;
                sc.npass += acc1[i] < tau ? 1u : 0u;
            }
#ifndef PN_DIAG_NO_SLOWPATH
            if (__any(sc.npass != 0))
                v2_slow<M>(acc1, sc, tau, (rt1 - 1) * kV2P + 32, h, wave, lane, kp, cand_k, cand_i, taus_w, cnts_w, ckq,
                        ciq, cnt_q, tau_q, GC);
#else
            asm volatile("" ::"v"(sc.m));
#endif
        }

        // ---- flush this run: <= kp candidates per query, count and threshold, to HBM
        {
            const uint32_t c = *cnt_q;
            unsigned long long need = __ballot(h == 0 && c > kp);
            while (need) {
                const int j = __builtin_ctzll(need);
                need &= need - 1;
                const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
                if constexpr (M == 1)
                    compact_lds(cand_k + (wave * 32 + j) * CAP, cand_i + (wave * 32 + j) * CAP, cj, kp, lane,
                                &taus_w[j], &cnts_w[j], GC);
                else
                    compact_query<uint32_t, M>(cand_k, cand_i, (size_t)(wave * 32 + j) * CAP, cj, kp, lane, &taus_w[j],
                                               &cnts_w[j], 0xFFFFFFFFu);
            }
            cand_fence(GC);
            const size_t gq = (size_t)seg * nq_pad + q0;  // first query of this wave in [seg][query]
            for (int j = 0; j < 32; ++j) {
                const uint32_t cj = cnts_w[j];
                for (uint32_t e = lane; e < cj; e += 64) {
                    ckey[(gq + j) * keep + e] = cand_k[(wave * 32 + j) * CAP + e];
                    cidx[(gq + j) * keep + e] = cand_i[(wave * 32 + j) * CAP + e];
                }
            }
            if (lane < 32) {
                ccnt[gq + lane] = cnts_w[lane];
                ctau[gq + lane] = taus_w[lane];
            }
        }
        u0 = run_end;
    }
#ifdef PN_DIAG_COUNT
    { const unsigned long long tk1_ = pn_stamp(); if (lane == 0) atomicAdd(&g_dbg[6], tk1_ - tk0_); }
#endif
}

// ===========================================================================
// WIDE rows (D > 128, row length a multiple of 128): same partition, same LDS candidate buffers,
// but a tile's bound is accumulated over D/128 "slabs" of 64 rows x 128 coordinates.  Slabs form
// one linear LDS-DMA pipeline across tiles (slab s+1 lands while slab s feeds the matrix pipe);
// the B operand (this wave's 32 queries, 128 coordinates = 64 VGPRs) is re-read from L2 once per
// slab into the other of two register sets while the current one is in use.  The filter of a tile
// (~2 x 80 VALU ops) runs unoverlapped after its last slab: < 2 % of a tile's D/128 x 8320 MFMA
// cycles.  Algorithmic work is unchanged: 2*D flop per pair, f32 MFMA roof.
// ===========================================================================
__device__ __forceinline__ void wide_load_b(const float *__restrict__ qrow, float (&b)[64]) {
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(qrow + 8 * kg);
        b[4 * kg + 0] = -2.0f * v.x; b[4 * kg + 1] = -2.0f * v.y;
        b[4 * kg + 2] = -2.0f * v.z; b[4 * kg + 3] = -2.0f * v.w;
    }
}

__device__ __forceinline__ void wide_chain(const float *trow, const int (&foff)[16], const float (&b)[64], f32x16 &w) {
    f32x4 a = *reinterpret_cast<const f32x4 *>(trow + foff[0]);
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
        f32x4 nx = a;
        if (kg + 1 < 16) nx = *reinterpret_cast<const f32x4 *>(trow + foff[kg + 1]);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * kg + 0], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * kg + 1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * kg + 2], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * kg + 3], w, 0, 0, 0);
        a = nx;
    }
}

// LDS-DMA of slab (tile rt, chunk c): 64 rows x 128 floats taken from rows of length ldp
__device__ __forceinline__ void wide_dma_slab(const float *__restrict__ P, size_t ldp, float *tiles, uint32_t rt, int c,
                                              int buf, int wave, int lane) {
    const char *src = reinterpret_cast<const char *>(P) + ((size_t)rt * kV2P * ldp + (size_t)c * 128) * 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = wv * 8 + i;
        const int row = g * 2 + lane / 32;
        const int pos = lane % 32;
        const int chunk = pos ^ (row & 15);
        __builtin_amdgcn_global_load_lds((glb_void *)(src + (size_t)row * ldp * 4 + chunk * 16),
                                         (lds_void *)(tiles + buf * kV2P * 128 + g * 256), 16, 0, 0);
    }
}

// one slab: DMA + B prefetch for the next slab, two 64-MFMA chains, (last chunk) norm + filter, barrier
template <bool GC>
__device__ __forceinline__ void wide_step(const float *__restrict__ P, const float *__restrict__ pnorm, size_t ldp,
                                          const float *__restrict__ qbase, float *tiles, float *pnl, uint32_t *cand_k,
                                          uint32_t *cand_i, uint32_t *taus_w, uint32_t *cnts_w, uint32_t *ckq,
                                          uint32_t *ciq, uint32_t *cnt_q, uint32_t *tau_q, const int (&foff)[16],
                                          float bn, float &tau, float (&bcur)[64], float (&bnext)[64],
                                          uint32_t s, uint32_t total, uint32_t rt0, int nc, f32x16 &acc0, f32x16 &acc1,
                                          uint32_t kp, int wave, int lane, int jq, int h) {
    const uint32_t rt = rt0 + s / nc;
    const int c = (int)(s % nc);
    const int cur = (int)(s & 1);
    if (s + 1 < total) {
        const uint32_t rtn = rt0 + (s + 1) / nc;
        const int cn = (int)((s + 1) % nc);
        wide_dma_slab(P, ldp, tiles, rtn, cn, cur ^ 1, wave, lane);
        if (cn == 0 && __builtin_amdgcn_readfirstlane(wave) == 0)
            __builtin_amdgcn_global_load_lds((glb_void *)(pnorm + (size_t)rtn * kV2P + lane),
                                             (lds_void *)(pnl + (rtn & 1) * kV2P), 4, 0, 0);
        if (!GC) wide_load_b(qbase + (size_t)cn * 128, bnext);
    }
    // two workgroups per CU: one register set, loaded for THIS slab; the other workgroup's wave on the
    // SIMD covers the L2 latency
    if (GC) wide_load_b(qbase + (size_t)c * 128, bcur);
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    }
    const float *trow = tiles + cur * kV2P * 128 + jq * 128;
    wide_chain(trow, foff, bcur, acc0);
    wide_chain(trow + 32 * 128, foff, bcur, acc1);
    if (c == nc - 1) {
        const float an0 = h ? 1.0f : pnl[(rt & 1) * kV2P + jq];
        const float an1 = h ? 1.0f : pnl[(rt & 1) * kV2P + 32 + jq];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(an0, bn, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(an1, bn, acc1, 0, 0, 0);
        float m0 = acc0[0], m1 = acc1[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) { m0 = fminf(m0, acc0[r]); m1 = fminf(m1, acc1[r]); }
        if (__any(m0 < tau)) {
            append_block(acc0, tau, rt * kV2P, h, ckq, ciq, cnt_q);
            cand_fence(GC);
            const uint32_t cq = *cnt_q;
            unsigned long long need = __ballot(h == 0 && cq > kV2Cap - 32);
            if (need) {
                do {
                    const int j = __builtin_ctzll(need);
                    need &= need - 1;
                    compact_lds(cand_k + (wave * 32 + j) * kV2Cap, cand_i + (wave * 32 + j) * kV2Cap,
                                (uint32_t)__builtin_amdgcn_readlane((int)cq, j), kp, lane, &taus_w[j], &cnts_w[j], GC);
                } while (need);
                tau = s2f(*tau_q);
            }
        }
        if (__any(m1 < tau)) {
            append_block(acc1, tau, rt * kV2P + 32, h, ckq, ciq, cnt_q);
            cand_fence(GC);
            const uint32_t cq = *cnt_q;
            unsigned long long need = __ballot(h == 0 && cq > kV2Cap - 32);
            if (need) {
                do {
                    const int j = __builtin_ctzll(need);
                    need &= need - 1;
                    compact_lds(cand_k + (wave * 32 + j) * kV2Cap, cand_i + (wave * 32 + j) * kV2Cap,
                                (uint32_t)__builtin_amdgcn_readlane((int)cq, j), kp, lane, &taus_w[j], &cnts_w[j], GC);
                } while (need);
                tau = s2f(*tau_q);
            }
        }
    }
    __syncthreads();
}

template <bool GC>
__global__ __launch_bounds__(256, GC ? 2 : 1) void mfma_filter_wide_kernel(
    const float *__restrict__ P, const float *__restrict__ pnorm, uint32_t n_tiles, const float *__restrict__ Q,
    const float *__restrict__ qnorm, uint32_t q_tiles, uint32_t kp, int nc, size_t ldp, uint32_t *__restrict__ ckey,
    uint32_t *__restrict__ cidx, uint32_t *__restrict__ ccnt, uint32_t *__restrict__ ctau, size_t nq_pad,
    uint32_t *__restrict__ gcand) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tiles = reinterpret_cast<float *>(smem_raw);                   // [2][64][128]
    float *pnl = tiles + 2 * kV2P * 128;                                  // [2][64]
    uint32_t *lds_tail = reinterpret_cast<uint32_t *>(pnl + 2 * kV2P);
    uint32_t *cand_k = GC ? gcand + (size_t)blockIdx.x * (2 * kV2Q * kV2Cap) : lds_tail;  // [128][64]
    uint32_t *cand_i = cand_k + kV2Q * kV2Cap;
    uint32_t *taus = GC ? lds_tail : cand_i + kV2Q * kV2Cap;
    uint32_t *cnts = taus + kV2Q;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jq = lane & 31, h = lane >> 5;
    const int ql = wave * 32 + jq;
    uint32_t *ckq = cand_k + ql * kV2Cap, *ciq = cand_i + ql * kV2Cap;
    uint32_t *cnt_q = cnts + ql, *tau_q = taus + ql;
    uint32_t *taus_w = taus + wave * 32, *cnts_w = cnts + wave * 32;
    int foff[16];
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) foff[kg] = 4 * ((2 * kg + h) ^ (jq & 15));

    const unsigned long long U = (unsigned long long)q_tiles * n_tiles;
    const unsigned long long W = gridDim.x, w = blockIdx.x;
    unsigned long long u0 = w * U / W;
    const unsigned long long u1 = (w + 1) * U / W;
    while (u0 < u1) {
        const uint32_t qt = (uint32_t)(u0 / n_tiles);
        const uint32_t rt0 = (uint32_t)(u0 % n_tiles);
        const unsigned long long q_end = (unsigned long long)(qt + 1) * n_tiles;
        const unsigned long long run_end = u1 < q_end ? u1 : q_end;
        const uint32_t rt1 = rt0 + (uint32_t)(run_end - u0);
        unsigned long long wf = ((unsigned long long)qt * n_tiles) * W / U;
        while ((wf + 1) * U / W <= (unsigned long long)qt * n_tiles) ++wf;
        while (wf > 0 && wf * U / W > (unsigned long long)qt * n_tiles) --wf;
        const uint32_t seg = (uint32_t)(w - wf);
        const size_t q0 = (size_t)qt * kV2Q + (size_t)wave * 32;

        if (h == 0) { *tau_q = 0xFF800000u; *cnt_q = 0; }
        const float *qbase = Q + (q0 + jq) * ldp + 4 * h;
        const float bn = h ? qnorm[q0 + jq] : 1.0f;
        float tau = __uint_as_float(0x7F800000u);
        float bA[64], bB[64];
        const uint32_t total = (rt1 - rt0) * (uint32_t)nc;

        __syncthreads();
        wide_dma_slab(P, ldp, tiles, rt0, 0, 0, wave, lane);
        if (__builtin_amdgcn_readfirstlane(wave) == 0)
            __builtin_amdgcn_global_load_lds((glb_void *)(pnorm + (size_t)rt0 * kV2P + lane),
                                             (lds_void *)(pnl + (rt0 & 1) * kV2P), 4, 0, 0);
        if (!GC) wide_load_b(qbase, bA);
        __syncthreads();

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        if constexpr (GC) {
            for (uint32_t s = 0; s < total; ++s)
                wide_step<true>(P, pnorm, ldp, qbase, tiles, pnl, cand_k, cand_i, taus_w, cnts_w, ckq, ciq, cnt_q, tau_q,
                                foff, bn, tau, bA, bA, s, total, rt0, nc, acc0, acc1, kp, wave, lane, jq, h);
        } else {
            for (uint32_t s = 0; s < total; s += 2) {
                wide_step<false>(P, pnorm, ldp, qbase, tiles, pnl, cand_k, cand_i, taus_w, cnts_w, ckq, ciq, cnt_q, tau_q,
                                 foff, bn, tau, bA, bB, s, total, rt0, nc, acc0, acc1, kp, wave, lane, jq, h);
                if (s + 1 < total)
                    wide_step<false>(P, pnorm, ldp, qbase, tiles, pnl, cand_k, cand_i, taus_w, cnts_w, ckq, ciq, cnt_q,
                                     tau_q, foff, bn, tau, bB, bA, s + 1, total, rt0, nc, acc0, acc1, kp, wave, lane, jq,
                                     h);
            }
        }

        {  // flush this run
            const uint32_t c = *cnt_q;
            unsigned long long need = __ballot(h == 0 && c > kp);
            while (need) {
                const int j = __builtin_ctzll(need);
                need &= need - 1;
                compact_lds(cand_k + (wave * 32 + j) * kV2Cap, cand_i + (wave * 32 + j) * kV2Cap,
                            (uint32_t)__builtin_amdgcn_readlane((int)c, j), kp, lane, &taus_w[j], &cnts_w[j], GC);
            }
            cand_fence(GC);
            const size_t gq = (size_t)seg * nq_pad + q0;
            for (int j = 0; j < 32; ++j) {
                const uint32_t cj = cnts_w[j];
                if ((uint32_t)lane < cj) {
                    ckey[(gq + j) * kV2Keep + lane] = cand_k[(wave * 32 + j) * kV2Cap + lane];
                    cidx[(gq + j) * kV2Keep + lane] = cand_i[(wave * 32 + j) * kV2Cap + lane];
                }
            }
            if (lane < 32) {
                ccnt[gq + lane] = cnts_w[lane];
                ctau[gq + lane] = taus_w[lane];
            }
        }
        u0 = run_end;
    }
}

// ===========================================================================
// query_radius on the matrix cores (D <= 128): the same lower bound L against ONE fixed threshold
//   tau_r = (r^2 + 1e-37) / (1 - (D+4) 2^-24), rounded up.
// L > tau_r  =>  |q-p|^2 > tau_r  =>  the reference's f32 fold s_ref >= r^2  =>  sqrt(s_ref) >= r,
// so a row is dropped only when the reference's strict test `dist < r` (src/ball_tree.rs:277)
// would drop it too.  Survivors (rare for sparse results) are appended to per-(segment, query)
// lists in HBM with a global atomic; a list that overflows its capacity sends the batch to the
// exact two-pass engine.  No threshold state, so no LDS buffers: 2 workgroups per CU.
// ===========================================================================
template <int NKG>
__device__ __forceinline__ void plain_chain(const float *trow, const int (&foff)[NKG], float an, float bn,
                                            const float (&b)[4 * NKG], f32x16 &w) {
    f32x4 a = *reinterpret_cast<const f32x4 *>(trow + foff[0]);
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = 0.0f;
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
        f32x4 nx = a;
        if (kg + 1 < NKG) nx = *reinterpret_cast<const f32x4 *>(trow + foff[kg + 1]);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * kg + 0], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * kg + 1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * kg + 2], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * kg + 3], w, 0, 0, 0);
        a = nx;
    }
    w = __builtin_amdgcn_mfma_f32_32x32x2f32(an, bn, w, 0, 0, 0);
}

__device__ __forceinline__ void radius_append(const f32x16 &acc, float tau_excl, uint32_t row0, int h,
                                              uint32_t *__restrict__ rcnt, uint32_t *__restrict__ ridx, size_t cell,
                                              uint32_t cap) {
    uint32_t mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) mask |= (acc[r] < tau_excl ? 1u : 0u) << r;
    const uint32_t npass = (uint32_t)__popc(mask);
    if (npass) {
        uint32_t o = atomicAdd(&rcnt[cell], npass);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (mask & (1u << r)) {
                if (o < cap) ridx[cell * cap + o] = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                ++o;
            }
        }
    }
}

template <int NKG>
__global__ __launch_bounds__(256, 2) void mfma_radius_kernel(
    const float *__restrict__ P, const float *__restrict__ pnorm, uint32_t n_tiles, const float *__restrict__ Q,
    const float *__restrict__ qnorm, uint32_t q_tiles, float tau_excl, uint32_t cap, uint32_t *__restrict__ rcnt,
    uint32_t *__restrict__ ridx, size_t nq_pad) {
    constexpr int LD = V2Ctx<NKG>::LD, STR = V2Ctx<NKG>::STR, CHUNKS = V2Ctx<NKG>::CHUNKS, NLD = V2Ctx<NKG>::NLD;
    constexpr bool GLDS = V2Ctx<NKG>::GLDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tiles = reinterpret_cast<float *>(smem_raw);  // [2][64][STR]
    float *pnl = tiles + 2 * kV2P * STR;                 // [2][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jq = lane & 31, h = lane >> 5;
    const uint32_t lane_off = (uint32_t)tid * 16u;
    int foff[NKG];
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) foff[kg] = 4 * (GLDS ? ((2 * kg + h) ^ (jq & 15)) : (2 * kg + h));

    const unsigned long long U = (unsigned long long)q_tiles * n_tiles;
    const unsigned long long W = gridDim.x, w = blockIdx.x;
    unsigned long long u0 = w * U / W;
    const unsigned long long u1 = (w + 1) * U / W;
    while (u0 < u1) {
        const uint32_t qt = (uint32_t)(u0 / n_tiles);
        const uint32_t rt0 = (uint32_t)(u0 % n_tiles);
        const unsigned long long q_end = (unsigned long long)(qt + 1) * n_tiles;
        const unsigned long long run_end = u1 < q_end ? u1 : q_end;
        const uint32_t rt1 = rt0 + (uint32_t)(run_end - u0);
        unsigned long long wf = ((unsigned long long)qt * n_tiles) * W / U;
        while ((wf + 1) * U / W <= (unsigned long long)qt * n_tiles) ++wf;
        while (wf > 0 && wf * U / W > (unsigned long long)qt * n_tiles) --wf;
        const uint32_t seg = (uint32_t)(w - wf);
        const size_t q0 = (size_t)qt * kV2Q + (size_t)wave * 32;
        const size_t cell = (size_t)seg * nq_pad + q0 + jq;

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
        const float bn = h ? qnorm[q0 + jq] : 1.0f;
        f32x4 st[NLD];
        float stn = 0.0f;
        __syncthreads();
        if constexpr (GLDS) {
            v2_dma_tile<NKG>(P, pnorm, tiles, pnl, rt0, 0, wave, lane);
        } else {
            const char *src_ = reinterpret_cast<const char *>(P) + (size_t)rt0 * (size_t)(kV2P * LD * 4);
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_)
                if (CHUNKS % 256 == 0 || tid + 256 * i_ < CHUNKS)
                    st[i_] = *reinterpret_cast<const f32x4 *>(src_ + (lane_off + 4096u * i_));
            stn = pnorm[(size_t)rt0 * kV2P + (tid & 63)];
#pragma unroll
            for (int i_ = 0; i_ < NLD; ++i_) {
                const int c_ = tid + 256 * i_;
                if (CHUNKS % 256 == 0 || c_ < CHUNKS)
                    *reinterpret_cast<f32x4 *>(tiles + (c_ / (LD / 4)) * STR + 4 * (c_ % (LD / 4))) = st[i_];
            }
            if (tid < kV2P) pnl[tid] = stn;
        }
        __syncthreads();

        int cur = 0;
        for (uint32_t rt = rt0; rt < rt1; ++rt, cur ^= 1) {
            const bool more = rt + 1 < rt1;
            if (more) {
                if constexpr (GLDS) {
                    v2_dma_tile<NKG>(P, pnorm, tiles, pnl, rt + 1, cur ^ 1, wave, lane);
                } else {
                    const char *src_ = reinterpret_cast<const char *>(P) + (size_t)(rt + 1) * (size_t)(kV2P * LD * 4);
#pragma unroll
                    for (int i_ = 0; i_ < NLD; ++i_)
                        if (CHUNKS % 256 == 0 || tid + 256 * i_ < CHUNKS)
                            st[i_] = *reinterpret_cast<const f32x4 *>(src_ + (lane_off + 4096u * i_));
                    stn = pnorm[(size_t)(rt + 1) * kV2P + (tid & 63)];
                }
            }
            const float *trow = tiles + cur * kV2P * STR + jq * STR;
            const float an0 = h ? 1.0f : pnl[cur * kV2P + jq];
            const float an1 = h ? 1.0f : pnl[cur * kV2P + 32 + jq];
            f32x16 acc0, acc1;
            plain_chain<NKG>(trow, foff, an0, bn, b, acc0);
            plain_chain<NKG>(trow + 32 * STR, foff, an1, bn, b, acc1);
            float m = fminf(acc0[0], acc1[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fminf(m, fminf(acc0[r], acc1[r]));
            if (!GLDS && more) {
                float *dst_ = tiles + (cur ^ 1) * kV2P * STR;
#pragma unroll
                for (int i_ = 0; i_ < NLD; ++i_) {
                    const int c_ = tid + 256 * i_;
                    if (CHUNKS % 256 == 0 || c_ < CHUNKS)
                        *reinterpret_cast<f32x4 *>(dst_ + (c_ / (LD / 4)) * STR + 4 * (c_ % (LD / 4))) = st[i_];
                }
                if (tid < kV2P) pnl[(cur ^ 1) * kV2P + tid] = stn;
            }
            if (__any(m < tau_excl)) {
                radius_append(acc0, tau_excl, rt * kV2P, h, rcnt, ridx, cell, cap);
                radius_append(acc1, tau_excl, rt * kV2P + 32, h, rcnt, ridx, cell, cap);
                __builtin_amdgcn_s_waitcnt(0x0070);  // drain the list stores before the counters
            }
            __syncthreads();
        }
        u0 = run_end;
    }
}

template <int NKG>
static hipError_t launch_radius_t(const float *P, const float *pnorm, uint32_t n_tiles, const float *Q,
                                  const float *qnorm, uint32_t q_tiles, float tau_excl, uint32_t cap, uint32_t *rcnt,
                                  uint32_t *ridx, size_t nq_pad, int n_wg, hipStream_t s) {
    const size_t sh = (size_t)(2 * kV2P * V2Ctx<NKG>::STR + 2 * kV2P) * sizeof(float);
    auto kern = mfma_radius_kernel<NKG>;
    static LdsAttrOnce lds_attr;  // per instantiation
    {
        const hipError_t e = lds_attr.ensure(reinterpret_cast<const void *>(kern), sh);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(256), sh, s, P, pnorm, n_tiles, Q, qnorm, q_tiles, tau_excl, cap,
                       rcnt, ridx, nq_pad);
    return hipGetLastError();
}

// rcnt [nseg][nq_pad] must be zeroed; ridx [nseg][nq_pad][cap]; nseg >= mfma_v2_max_segments(q_tiles, n_wg)
hipError_t launch_mfma_radius_f32(const float *P, const float *pnorm, size_t n, size_t ldp, const float *Q,
                                  const float *qnorm, size_t nq_pad, float tau_excl, uint32_t cap, uint32_t *rcnt,
                                  uint32_t *ridx, int n_wg, hipStream_t s) {
    const uint32_t n_tiles = (uint32_t)((n + kV2P - 1) / kV2P);
    const uint32_t q_tiles = (uint32_t)(nq_pad / kV2Q);
    switch (ldp) {
        case 8: return launch_radius_t<1>(P, pnorm, n_tiles, Q, qnorm, q_tiles, tau_excl, cap, rcnt, ridx, nq_pad, n_wg, s);
        case 16: return launch_radius_t<2>(P, pnorm, n_tiles, Q, qnorm, q_tiles, tau_excl, cap, rcnt, ridx, nq_pad, n_wg, s);
        case 32: return launch_radius_t<4>(P, pnorm, n_tiles, Q, qnorm, q_tiles, tau_excl, cap, rcnt, ridx, nq_pad, n_wg, s);
        case 64: return launch_radius_t<8>(P, pnorm, n_tiles, Q, qnorm, q_tiles, tau_excl, cap, rcnt, ridx, nq_pad, n_wg, s);
        case 96: return launch_radius_t<12>(P, pnorm, n_tiles, Q, qnorm, q_tiles, tau_excl, cap, rcnt, ridx, nq_pad, n_wg, s);
        case 128: return launch_radius_t<16>(P, pnorm, n_tiles, Q, qnorm, q_tiles, tau_excl, cap, rcnt, ridx, nq_pad, n_wg, s);
        default: return hipErrorInvalidValue;
    }
}

#ifdef PN_DIAG_COUNT
extern "C" int pn_debug_read(unsigned long long *out, int reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(z)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif

int mfma_v2_max_segments(size_t q_tiles, int n_wg) {
    return (int)((n_wg + q_tiles - 1) / q_tiles) + 1;
}

template <int NKG, bool GC, int M>
static hipError_t launch_v2(const float *P, const float *pnorm, uint32_t n_tiles, const float *Q, const float *qnorm,
                            uint32_t q_tiles, uint32_t kp, const CandBuf &cb, int n_wg, uint32_t *gcand, hipStream_t s) {
    const size_t sh = (size_t)(2 * kV2P * V2Ctx<NKG>::STR + 2 * kV2P) * sizeof(float) +
                      (size_t)((GC ? 0 : 2 * kV2Q * 64) + 2 * kV2Q) * sizeof(uint32_t);
    auto kern = mfma_filter_v2_kernel<NKG, GC, M>;
    static LdsAttrOnce lds_attr;  // per instantiation
    {
        const hipError_t e = lds_attr.ensure(reinterpret_cast<const void *>(kern), sh);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(256), sh, s, P, pnorm, n_tiles, Q, qnorm, q_tiles, kp,
                       static_cast<uint32_t *>(cb.keys), cb.idx, cb.cnt, static_cast<uint32_t *>(cb.tau), cb.nq_pad,
                       gcand, (uint32_t)cb.cap);
    return hipGetLastError();
}

template <int NKG>
static hipError_t launch_v2_pick(const float *P, const float *pnorm, uint32_t n_tiles, const float *Q,
                                 const float *qnorm, uint32_t q_tiles, uint32_t kp, const CandBuf &cb, int n_wg,
                                 uint32_t *gcand, hipStream_t s) {
    if (!gcand) return launch_v2<NKG, false, 1>(P, pnorm, n_tiles, Q, qnorm, q_tiles, kp, cb, n_wg, nullptr, s);
    if (kp <= 30) return launch_v2<NKG, true, 1>(P, pnorm, n_tiles, Q, qnorm, q_tiles, kp, cb, n_wg, gcand, s);
    if (kp <= 94) return launch_v2<NKG, true, 2>(P, pnorm, n_tiles, Q, qnorm, q_tiles, kp, cb, n_wg, gcand, s);
    return launch_v2<NKG, true, 4>(P, pnorm, n_tiles, Q, qnorm, q_tiles, kp, cb, n_wg, gcand, s);
}

// slots per query in the HBM candidate buffers for a given k'
int mfma_v2_cap_for(int kp) { return kp <= 30 ? 64 : kp <= 94 ? 128 : 256; }
size_t mfma_v2_gcand_bytes(int n_wg, int kp) {
    return (size_t)n_wg * 2 * kV2Q * (size_t)mfma_v2_cap_for(kp) * sizeof(uint32_t);
}

// cb: keys/idx [nseg][nq_pad][32], cnt/tau [nseg][nq_pad] PRE-INITIALISED to 0 / sortable(+inf)
hipError_t launch_mfma_filter_v2_f32(const float *P, const float *pnorm, size_t n, size_t ldp, const float *Q,
                                     const float *qnorm, size_t ldq, int kp, const CandBuf &cb, int n_wg,
                                     uint32_t *gcand, hipStream_t s) {
    if (ldq != ldp || cb.nq_pad % kV2Q || kp < 1 || kp > cb.cap || kp > (gcand && ldp <= 128 ? 224 : 30))
        return hipErrorInvalidValue;
    const uint32_t n_tiles = (uint32_t)((n + kV2P - 1) / kV2P);
    const uint32_t q_tiles = (uint32_t)(cb.nq_pad / kV2Q);
    if (cb.nseg < mfma_v2_max_segments(q_tiles, n_wg)) return hipErrorInvalidValue;
    if (ldp > 128) {
        if (ldp % 128 || kp > 30) return hipErrorInvalidValue;
        const size_t sh = (size_t)(2 * kV2P * 128 + 2 * kV2P) * sizeof(float) +
                          (size_t)((gcand ? 0 : 2 * kV2Q * kV2Cap) + 2 * kV2Q) * sizeof(uint32_t);
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(256), sh, s, P, pnorm, n_tiles, Q, qnorm, q_tiles,
                               (uint32_t)kp, (int)(ldp / 128), ldp, static_cast<uint32_t *>(cb.keys), cb.idx, cb.cnt,
                               static_cast<uint32_t *>(cb.tau), cb.nq_pad, gcand);
            return hipGetLastError();
        };
        return gcand ? launch(mfma_filter_wide_kernel<true>) : launch(mfma_filter_wide_kernel<false>);
    }
    switch (ldp) {
        case 8: return launch_v2_pick<1>(P, pnorm, n_tiles, Q, qnorm, q_tiles, (uint32_t)kp, cb, n_wg, gcand, s);
        case 16: return launch_v2_pick<2>(P, pnorm, n_tiles, Q, qnorm, q_tiles, (uint32_t)kp, cb, n_wg, gcand, s);
        case 32: return launch_v2_pick<4>(P, pnorm, n_tiles, Q, qnorm, q_tiles, (uint32_t)kp, cb, n_wg, gcand, s);
        case 64: return launch_v2_pick<8>(P, pnorm, n_tiles, Q, qnorm, q_tiles, (uint32_t)kp, cb, n_wg, gcand, s);
        case 96: return launch_v2_pick<12>(P, pnorm, n_tiles, Q, qnorm, q_tiles, (uint32_t)kp, cb, n_wg, gcand, s);
        case 128: return launch_v2_pick<16>(P, pnorm, n_tiles, Q, qnorm, q_tiles, (uint32_t)kp, cb, n_wg, gcand, s);
        default: return hipErrorInvalidValue;
    }
}

// rows the f32 MFMA tier serves: padded row lengths its K loop is instantiated for
bool mfma_supported(int dim, size_t ld) {
    if (dim < 1) return false;
    switch (ld) {
        case 8: case 16: case 32: case 64: case 96: case 128: return true;
        default: return ld > 128 && ld % 128 == 0;  // wide rows: the slab-accumulating kernel
    }
}

}  // namespace pn
