// topk_buffer.h -- per-(segment, query) candidate buffers shared by the scan
// kernels (exact_scan.hip, mfma_filter.hip).
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
// Rank-and-keep compaction of one query's candidate buffer by its owning wave.
// ---------------------------------------------------------------------------
template <typename KeyT, int M>
__device__ __forceinline__ void compact_query(KeyT *__restrict__ ckey, uint32_t *__restrict__ cidx, size_t base,
                                              uint32_t n, uint32_t kp, int lane, KeyT *tau_slot,
                                              uint32_t *cnt_slot, KeyT key_max) {
    KeyT key[M];
    uint32_t ix[M], rank[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint32_t slot = m * 64 + lane;
        const bool v = slot < n;
        key[m] = v ? ckey[base + slot] : key_max;
        ix[m] = v ? cidx[base + slot] : 0xFFFFFFFFu;
        rank[m] = 0;
    }
#pragma unroll
    for (int m2 = 0; m2 < M; ++m2) {
        int lim = (int)n - m2 * 64;
        lim = lim > 64 ? 64 : lim;
        for (int ln = 0; ln < lim; ++ln) {
            const KeyT bk = bcast_lane(key[m2], ln);
            const uint32_t bi = bcast_lane(ix[m2], ln);
#pragma unroll
            for (int m = 0; m < M; ++m)
                rank[m] += ((bk < key[m]) || (bk == key[m] && bi < ix[m])) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if ((uint32_t)(m * 64 + lane) < n && rank[m] < kp) {
            ckey[base + rank[m]] = key[m];
            cidx[base + rank[m]] = ix[m];
            if (rank[m] == kp - 1) *tau_slot = key[m];
        }
    }
    if (lane == 0) *cnt_slot = n < kp ? n : kp;
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
