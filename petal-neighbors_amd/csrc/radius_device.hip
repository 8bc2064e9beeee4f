// radius_device.hip -- device-side plumbing of pn_query_radius_device_{f32,f64} (round 4).
//
// BallTree::query_radius (src/ball_tree.rs:137-142, 250-294) returns a list whose length is data-dependent.  The host
// entry points read counts back, scan them on the host and size the output there (index.hip, radius_finish /
// radius_exact): two stream synchronisations inside every call.  The device entry point keeps everything in HBM: the
// caller supplies the output's capacity, the per-query counts are scanned on the device, entries beyond the capacity
// are counted but not written, and the total (which tells the caller whether to call again with a larger buffer) is
// one more word in HBM.  These kernels are that glue: listing the queries a filter tier could not serve, combining
// the two sources of counts, an exclusive scan, per-(query, segment) offsets for the exact fill pass, and the gather
// of the filter tier's kept lists.  All of it is HBM-bound integer work over nq (or nq x nseg) words.
#include "pn_internal.h"

namespace pn {

// The list is built in query order by ONE block of 1024 threads walking the queries (10^5 queries: ~100 trips of a ballot
// and a 16-entry prefix -- tens of microseconds; the lists themselves are usually a handful of queries long)
__global__ __launch_bounds__(1024) void rad_list_kernel(const uint32_t *__restrict__ over, const uint32_t *__restrict__ bad,
                                                        int nq, uint32_t *__restrict__ nkept, uint32_t *__restrict__ sel,
                                                        uint32_t *__restrict__ pos, uint32_t *__restrict__ nsel) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {
        const int q = q0 + tid;
        const bool need = q < nq && ((over && over[q]) || (bad && bad[q]));
        const unsigned long long m = __ballot(need);
        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = base, all = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) off += wsum[w];
            all += wsum[w];
        }
        if (q < nq) {
            if (need) {
                sel[off + before] = (uint32_t)q;
                pos[q] = off + before;
                if (nkept) nkept[q] = 0;
            } else {
                pos[q] = 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        if (tid == 0) base += all;
        __syncthreads();
    }
    if (tid == 0) *nsel = base;
}
hipError_t launch_rad_list(const uint32_t *over, const uint32_t *bad, int nq, uint32_t *nkept, uint32_t *sel,
                           uint32_t *pos, uint32_t *nsel, hipStream_t s) {
    hipLaunchKernelGGL(rad_list_kernel, dim3(1), dim3(1024), 0, s, over, bad, nq, nkept, sel, pos, nsel);
    return hipGetLastError();
}

__global__ void rad_counts_kernel(const uint32_t *__restrict__ nkept, const uint32_t *__restrict__ pos,
                                  const uint32_t *__restrict__ counts_x, int nseg, int nq, uint32_t *__restrict__ fin) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t r = pos ? pos[q] : (uint32_t)q;
    uint32_t c = 0;
    if (r != 0xFFFFFFFFu) {
        for (int sg = 0; sg < nseg; ++sg) c += counts_x[(size_t)r * nseg + sg];
    } else if (nkept) {
        c = nkept[q];
    }
    fin[q] = c;
}
hipError_t launch_rad_counts(const uint32_t *nkept, const uint32_t *pos, const uint32_t *counts_x, int nseg, int nq,
                             uint32_t *fin, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(rad_counts_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, nkept, pos, counts_x, nseg,
                       nq, fin);
    return hipGetLastError();
}

// exclusive scan in three passes over chunks of 4096: chunk sums, a one-block scan of the sums, chunk scans + base
constexpr int kScanChunk = 4096;
__global__ __launch_bounds__(256) void scan_sums_kernel(const uint32_t *__restrict__ in, size_t n, uint64_t *__restrict__ sums) {
    __shared__ uint64_t ws[4];
    const size_t c0 = (size_t)blockIdx.x * kScanChunk;
    uint64_t t = 0;
    for (int i = threadIdx.x; i < kScanChunk; i += 256)
        if (c0 + i < n) t += in[c0 + i];
    for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(64) void scan_bases_kernel(uint64_t *__restrict__ sums, size_t n_chunks, uint64_t *__restrict__ total,
                                                        uint64_t *__restrict__ last_out) {
    // one wave: sums[i] <- exclusive prefix; sums[n_chunks] <- total
    const int lane = threadIdx.x;
    uint64_t run = 0;
    for (size_t i0 = 0; i0 < n_chunks; i0 += 64) {
        const size_t i = i0 + lane;
        const uint64_t v = i < n_chunks ? sums[i] : 0;
        uint64_t inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (i < n_chunks) sums[i] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) {
        sums[n_chunks] = run;
        if (total) *total = run;
        if (last_out) *last_out = run;
    }
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *__restrict__ in, size_t n, const uint64_t *__restrict__ bases,
                                                         uint64_t *__restrict__ out) {
    __shared__ uint64_t ws[4];
    const size_t c0 = (size_t)blockIdx.x * kScanChunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // thread t owns the 16 consecutive elements [c0 + 16 t, c0 + 16 t + 16)
    uint32_t v[16];
    uint64_t mine = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const size_t e = c0 + (size_t)tid * 16 + i;
        v[i] = e < n ? in[e] : 0u;
        mine += v[i];
    }
    uint64_t inc = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    uint64_t run = bases[blockIdx.x] + inc - mine;
    for (int w = 0; w < wave; ++w) run += ws[w];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const size_t e = c0 + (size_t)tid * 16 + i;
        if (e < n) out[e] = run;
        run += v[i];
    }
}
hipError_t launch_exclusive_scan_u32(const uint32_t *in, size_t n, uint64_t *offsets, uint64_t *scratch, uint64_t *total,
                                     hipStream_t s) {
    const size_t n_chunks = (n + kScanChunk - 1) / kScanChunk;
    if (n_chunks)
        hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, in, n, scratch);
    hipLaunchKernelGGL(scan_bases_kernel, dim3(1), dim3(64), 0, s, scratch, n_chunks, total, offsets + n);
    if (n_chunks)
        hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, in, n, scratch, offsets);
    return hipGetLastError();
}

__global__ void rad_seg_offsets_kernel(const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ sel,
                                       const uint32_t *__restrict__ nsel, int nq, const uint32_t *__restrict__ counts_x,
                                       int nseg, uint64_t *__restrict__ offs_x) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nr = nsel ? (*nsel < (uint32_t)nq ? *nsel : (uint32_t)nq) : (uint32_t)nq;
    if (r >= nr) return;
    uint64_t run = offsets[sel ? sel[r] : r];
    for (int sg = 0; sg < nseg; ++sg) {
        offs_x[(size_t)r * nseg + sg] = run;
        run += counts_x[(size_t)r * nseg + sg];
    }
}
hipError_t launch_rad_seg_offsets(const uint64_t *offsets, const uint32_t *sel, const uint32_t *nsel, int nq,
                                  const uint32_t *counts_x, int nseg, uint64_t *offs_x, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(rad_seg_offsets_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, offsets, sel, nsel, nq,
                       counts_x, nseg, offs_x);
    return hipGetLastError();
}

__global__ void radius_gather_cap_kernel(const uint32_t *__restrict__ kept, const uint32_t *__restrict__ nkept,
                                         const uint64_t *__restrict__ offsets, size_t kept_stride, uint64_t index_base,
                                         uint64_t *__restrict__ out, uint64_t capacity) {
    const size_t q = blockIdx.x;
    const uint32_t n = nkept[q];
    const uint64_t o = offsets[q];
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x)
        if (o + e < capacity) out[o + e] = index_base + kept[q * kept_stride + e];
}
hipError_t launch_radius_gather_cap(const uint32_t *kept, const uint32_t *nkept, const uint64_t *offsets, int nq,
                                    size_t kept_stride, uint64_t index_base, uint64_t *out, uint64_t capacity,
                                    hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(radius_gather_cap_kernel, dim3((unsigned)nq), dim3(64), 0, s, kept, nkept, offsets, kept_stride,
                       index_base, out, capacity);
    return hipGetLastError();
}

}  // namespace pn
