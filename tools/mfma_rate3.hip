// tools/mfma_rate3.hip -- micro-probe 3: does a SIMD lose matrix throughput when TWO waves interleave
// dependent v_mfma_f32_32x32x2_f32 chains?  (a) 1 wave/SIMD, (b) 2 waves/SIMD as one 512-thread
// workgroup, (c) 2 waves/SIMD as two 256-thread workgroups per CU; plus the LDS-fed variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void probe(float *out, int iters, float a0, float b0, int use_lds) {
    __shared__ float tile[64 * 132];
    for (int i = threadIdx.x; i < 64 * 132; i += THREADS) tile[i] = a0 + i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float *trow = tile + (lane & 31) * 132 + 4 * (lane >> 5);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float b = b0 + lane * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        f32x4 a = use_lds ? *reinterpret_cast<const f32x4 *>(trow) : f32x4{a0, b0, a0, b0};
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) {
            f32x4 nx = a;
            if (use_lds && kg + 1 < 16) nx = *reinterpret_cast<const f32x4 *>(trow + 8 * (kg + 1));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc, 0, 0, 0);
            a = nx;
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int THREADS, int MINW>
void run(const char *name, float *d, int blocks, int iters, int use_lds) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<THREADS, MINW><<<blocks, THREADS>>>(d, 10, 1.f, 1.f, use_lds);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<THREADS, MINW><<<blocks, THREADS>>>(d, iters, 1.f, 1.f, use_lds);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * THREADS / 64;
    const double flops = (double)iters * 64 * 4096.0 * waves;
    printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float *d; (void)hipMalloc(&d, 1024 * 512 * 4);
    run<256, 1>("1 wave/SIMD, regs", d, 256, 20000, 0);
    run<512, 1>("2 waves/SIMD (one 512-thread WG), regs", d, 256, 20000, 0);
    run<256, 2>("2 waves/SIMD (two 256-thread WGs), regs", d, 512, 20000, 0);
    run<256, 1>("1 wave/SIMD, A from LDS", d, 256, 20000, 1);
    run<256, 2>("2 waves/SIMD (two WGs), A from LDS", d, 512, 20000, 1);
    return 0;
}
