// pack.hip -- HBM layout kernels (gfx950): row packing / zero padding, the
// scaled row norms the MFMA lower bound needs, and the synthetic generator.
//
// Device layout of a corpus (and of a packed query batch): row-major, row
// length `ld` = dim rounded up to 8 elements, rows zero-padded on the right,
// row COUNT rounded up (zero rows below).  Zero padding never changes a
// distance: the reference fold (src/distance.rs:26-35) adds (0-0)*(0-0) = +0.
#include "pn_internal.h"

namespace pn {

template <typename T>
__global__ void pack_rows_kernel(const T *__restrict__ src, size_t n, size_t cols, size_t row_stride,
                                 T *__restrict__ dst, size_t n_pad, size_t ld) {
    const size_t total = n_pad * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / ld, c = i % ld;
        dst[i] = (r < n && c < cols) ? src[r * row_stride + c] : (T)0;
    }
}

template <typename T>
static hipError_t launch_pack_rows(const T *src, size_t n, size_t cols, size_t row_stride, T *dst, size_t n_pad,
                                   size_t ld, hipStream_t s) {
    const size_t total = n_pad * ld;
    if (total == 0) return hipSuccess;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((pack_rows_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, src, n, cols, row_stride, dst,
                       n_pad, ld);
    return hipGetLastError();
}
hipError_t launch_pack_rows_f32(const float *src, size_t n, size_t cols, size_t row_stride, float *dst, size_t n_pad,
                                size_t ld, hipStream_t s) {
    return launch_pack_rows<float>(src, n, cols, row_stride, dst, n_pad, ld, s);
}
hipError_t launch_pack_rows_f64(const double *src, size_t n, size_t cols, size_t row_stride, double *dst,
                                size_t n_pad, size_t ld, hipStream_t s) {
    return launch_pack_rows<double>(src, n, cols, row_stride, dst, n_pad, ld, s);
}

// ---------------------------------------------------------------------------
// counter-based generator shared (by restatement) with oracle_fill_uniform_f32
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + ctr;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
__global__ void fill_uniform_kernel(float *__restrict__ out, uint64_t count, uint64_t seed, uint64_t first) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (float)(mix32(seed, first + i) >> 8) * (1.0f / 16777216.0f);
}
hipError_t launch_fill_uniform_f32(float *out, uint64_t count, uint64_t seed, uint64_t first, hipStream_t s) {
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, count, seed, first);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// row_norms: norm_out[i] = RD( (sum_k x[i][k]^2 in f64) * (1 - alpha) ) - 1e-36
// for i < n, +inf for padding rows (their lower bound is then +inf and they can
// never become candidates).  HBM-bound single pass: half a wave per row, 16-B
// loads, f64 accumulation (so the only error left in the norm is the final
// round-DOWN), shuffle reduction.  Flags non-finite norms: such an index (or
// query) is served by the exact engine instead.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ X, size_t n_pad, size_t n,
                                                        int dim, size_t ld, double one_minus_alpha,
                                                        float *__restrict__ norm_out,
                                                        uint32_t *__restrict__ nonfinite_flag) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r0 = wave_global * 2; r0 < n_pad; r0 += n_waves * 2) {
        const size_t r = r0 + half;
        double acc = 0.0;
        if (r < n) {
            const float *row = X + r * ld;
            for (int c = l32 * 4; c < dim; c += 128) {  // ld is a multiple of 8 and rows are zero padded
                const float4 v = *reinterpret_cast<const float4 *>(row + c);
                acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            }
        }
        for (int d = 16; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        if (l32 == 0 && r < n_pad) {
            float o;
            if (r < n) {
                o = __double2float_rd(acc * one_minus_alpha) - 1e-36f;
                if (!(acc < 3.0e38)) {  // inf or NaN (or about to overflow f32)
                    o = __uint_as_float(0x7F800000u);
                    atomicOr(nonfinite_flag, 1u);
                }
            } else {
                o = __uint_as_float(0x7F800000u);
            }
            norm_out[r] = o;
        }
    }
}

hipError_t launch_row_norms_f32(const float *X, size_t n_pad, size_t n, int dim, size_t ld, float alpha,
                                float *norm_out, uint32_t *nonfinite_flag, hipStream_t s) {
    if (n_pad == 0) return hipSuccess;
    size_t blocks = (n_pad / 2 + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, n_pad, n, dim, ld,
                       1.0 - (double)alpha, norm_out, nonfinite_flag);
    return hipGetLastError();
}

}  // namespace pn
