// tools/mfma_rate2.hip -- micro-probe 2: the filter kernel's inner structure in isolation.
//  V0: 65-MFMA chains, operands from registers
//  V1: A operand streamed from LDS by ds_read_b128, one k-group ahead
//  V2: V1 + min-reduce of the other accumulator in the chain's shadow (ping-pong)
//  V3: V2 + one __syncthreads per 130 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NKG = 16, STR = 132;

template <int V>
__device__ __forceinline__ float chain(const float *tl, float an, float bn, const float (&b)[64], f32x16 &w, const f32x16 &r) {
    f32x4 a = V >= 1 ? *reinterpret_cast<const f32x4 *>(tl) : f32x4{an, bn, an, bn};
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = 0.f;
    w = __builtin_amdgcn_mfma_f32_32x32x2f32(an, bn, w, 0, 0, 0);
    float m = r[0];
    if (V >= 2) {
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fminf(m, r[i]);
    }
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
        f32x4 nx = a;
        if (V >= 1 && kg + 1 < NKG) nx = *reinterpret_cast<const f32x4 *>(tl + 8 * (kg + 1));
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * kg + 0], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * kg + 1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * kg + 2], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * kg + 3], w, 0, 0, 0);
        a = nx;
    }
    return m;
}

template <int V>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters, float seed) {
    __shared__ float tiles[2 * 64 * STR];
    const int tid = threadIdx.x, lane = tid & 63, jq = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * 64 * STR; i += 256) tiles[i] = seed + i * 1e-6f;
    float b[64];
    for (int i = 0; i < 64; ++i) b[i] = seed * (i + 1) + lane * 1e-3f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 1e30f; }
    float tau = -1e30f, msum = 0.f;
    int cur = 0;
    for (int it = 0; it < iters; ++it, cur ^= 1) {
        const float *tl = tiles + cur * 64 * STR + jq * STR + 4 * h;
        float m1 = chain<V>(tl, 1.f, seed, b, acc0, acc1);
        if (__any(m1 < tau)) msum += m1;
        float m0 = chain<V>(tl + 32 * STR, 1.f, seed, b, acc1, acc0);
        if (__any(m0 < tau)) msum += m0;
        if (V >= 3) __syncthreads();
    }
    float s = msum;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(float *d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<V><<<256, 256>>>(d, 10, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<V><<<256, 256>>>(d, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)iters * 130;
    printf("V%d: %.3f ms, %.1f cycles/MFMA @2.4GHz, per 130-MFMA tile %.0f cycles (ideal 8320)\n", V, ms,
           ms * 1e6 / mf * 2.4, ms * 1e6 / iters * 2.4);
}
int main() {
    float *d; (void)hipMalloc(&d, 256 * 256 * 4);
    run<0>(d, 5000); run<1>(d, 5000); run<2>(d, 5000); run<3>(d, 5000);
    return 0;
}
