// tools/mfma_rate.hip -- micro-probe: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 as
// (a) ONE dependent accumulator chain, (b) two / (c) four interleaved chains; one wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters, float a0, float b0) {
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 128 / NCH; ++i)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < NCH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH>
void run(float *d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NCH><<<256, 256>>>(d, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NCH><<<256, 256>>>(d, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)iters * 128;
    double flops = mf * 32 * 32 * 2 * 2 * 4 * 256;  // per MFMA 4096 flop, 4 waves/WG, 256 WGs
    printf("chains=%d: %.3f ms, %.1f ns/MFMA (%.1f cycles @2.4GHz), %.1f TFLOP/s\n", NCH, ms, ms * 1e6 / mf,
           ms * 1e6 / mf * 2.4, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float *d; hipMalloc(&d, 256 * 256 * 4);
    run<1>(d, 20000); run<2>(d, 20000); run<4>(d, 20000);
    return 0;
}
