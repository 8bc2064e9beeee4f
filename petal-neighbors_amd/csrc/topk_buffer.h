// topk_buffer.h -- per-(segment, query) candidate buffers shared by the scan
// kernels (exact_scan.hip, mfma_filter_v2.hip).
//
// A buffer lives in HBM: `cap` (key, row) slots, an entry count and a running
// threshold tau.  A value enters when key < tau.  The wave that owns the query
// appends through an LDS counter; when fewer than 64 free slots remain it ranks
// the buffer by the total order (key, row), keeps the kp smallest and lowers
// tau to the kp-th.  Invariant: every row of the segment that is NOT in the
// buffer has key >= tau.
#pragma once
#include "pn_internal.h"

namespace pn {

__device__ __forceinline__ uint32_t bcast_lane(uint32_t v, int ln) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, ln);
}
__device__ __forceinline__ uint64_t bcast_lane(uint64_t v, int ln) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, ln);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), ln);
    return ((uint64_t)hi << 32) | lo;
}

// wave-local ordering point for LDS / HBM traffic that other lanes of the SAME
// workgroup read back (candidate buffers, cnt/tau words)
__device__ __forceinline__ void wg_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---------------------------------------------------------------------------
// Compaction of one query's buffer by its owning wave: keep the kp smallest
// entries under the total order (key, row) and lower tau to the kp-th key.
//
// Radix select instead of sorting: the kp-th smallest key T is found one bit at
// a time, each step a ballot + popcount over the wave (32 or 64 steps of a few
// scalar instructions, independent of the buffer size).  Entries with key < T
// are kept; of the entries with key == T the (kp - #less) smallest rows are
// kept, found by the same bit-wise select on the row index when there are more
// ties than room.  Survivors are written back compactly (unordered: the select
// kernel ranks them at the end).  Requires n > kp.
// ---------------------------------------------------------------------------
template <typename KeyT, int M>
__device__ __forceinline__ void compact_query(KeyT *__restrict__ ckey, uint32_t *__restrict__ cidx, size_t base,
                                              uint32_t n, uint32_t kp, int lane, KeyT *tau_slot,
                                              uint32_t *cnt_slot, KeyT key_max) {
    KeyT key[M];
    uint32_t ix[M];
    bool valid[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint32_t slot = m * 64 + lane;
        valid[m] = slot < n;
        key[m] = valid[m] ? ckey[base + slot] : key_max;
        ix[m] = valid[m] ? cidx[base + slot] : 0xFFFFFFFFu;
    }
    // T = kp-th smallest key: the largest T with #{key < T} < kp
    KeyT T = 0;
    for (int bit = (int)sizeof(KeyT) * 8 - 1; bit >= 0; --bit) {
        const KeyT cand = T | ((KeyT)1 << bit);
        uint32_t cnt = 0;
#pragma unroll
        for (int m = 0; m < M; ++m) cnt += (uint32_t)__popcll(__ballot(valid[m] && key[m] < cand));
        if (cnt < kp) T = cand;
    }
    uint32_t n_less = 0, n_eq = 0;
    bool sel[M], eq[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        sel[m] = valid[m] && key[m] < T;
        eq[m] = valid[m] && key[m] == T;
        n_less += (uint32_t)__popcll(__ballot(sel[m]));
        n_eq += (uint32_t)__popcll(__ballot(eq[m]));
    }
    const uint32_t need_eq = kp - n_less;  // >= 1
    uint32_t row_cut = 0xFFFFFFFFu;         // keep ties with row <= row_cut
    if (n_eq > need_eq) {                   // more ties than room: the need_eq smallest rows win
        uint32_t I = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = I | (1u << bit);
            uint32_t cnt = 0;
#pragma unroll
            for (int m = 0; m < M; ++m) cnt += (uint32_t)__popcll(__ballot(eq[m] && ix[m] < cand));
            if (cnt < need_eq) I = cand;
        }
        row_cut = I;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t pos = 0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const bool keep = sel[m] || (eq[m] && ix[m] <= row_cut);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t o = pos + (uint32_t)__popcll(mask & lt);
            ckey[base + o] = key[m];
            cidx[base + o] = ix[m];
        }
        pos += (uint32_t)__popcll(mask);
    }
    if (lane == 0) {
        *tau_slot = T;
        *cnt_slot = pos;  // == kp
    }
    wg_fence();
}

// order-preserving float <-> uint32 map (negative lower bounds occur)
__device__ __forceinline__ uint32_t f2s(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float s2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

}  // namespace pn
