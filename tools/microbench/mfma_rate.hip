// Bare issue rate of the two f32-input MFMA forms on this box (operands in registers, independent accumulators):
// what a projection kernel can at best get out of the matrix pipe.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/microbench/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SAMEOP>
__global__ void __launch_bounds__(64) k32(float *out, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x + i; b[i] = threadIdx.x * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(SAMEOP ? a[0] : a[(i + j) & 3], SAMEOP ? b[0] : b[(i * 2 + j) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(64) k16(float *out, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x + i; b[i] = threadIdx.x * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + j) & 3], b[(i * 2 + j) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename F>
void run(const char *name, F launch, double flops_per_wave_iter, int waves, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch(iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch(iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %2d waves/SIMD: %7.1f TF/s\n", name, waves / 1024, flops_per_wave_iter * iters * waves * 5 / (ms * 1e-3) / 1e12);
    fflush(stdout);
}
int main()
{
    float *out;
    (void)hipMalloc(&out, 4096 * 64 * 4);
    const int iters = 2000;
    for (int w : {1024, 2048, 4096}) {
        run("32x32x2, 6 accumulators", [&](int it) { hipLaunchKernelGGL((k32<6, false>), dim3(w), dim3(64), 0, 0, out, it); }, 6 * 4 * 4096.0, w, iters);
        run("32x32x2, 6 accumulators, same operands", [&](int it) { hipLaunchKernelGGL((k32<6, true>), dim3(w), dim3(64), 0, 0, out, it); }, 6 * 4 * 4096.0, w, iters);
        run("32x32x2, 3 accumulators", [&](int it) { hipLaunchKernelGGL((k32<3, false>), dim3(w), dim3(64), 0, 0, out, it); }, 3 * 4 * 4096.0, w, iters);
        run("32x32x2, 12 accumulators", [&](int it) { hipLaunchKernelGGL((k32<12, false>), dim3(w), dim3(64), 0, 0, out, it); }, 12 * 4 * 4096.0, w, iters);
        run("16x16x4, 24 accumulators", [&](int it) { hipLaunchKernelGGL((k16<24>), dim3(w), dim3(64), 0, 0, out, it); }, 24 * 4 * 2048.0, w, iters);
        run("16x16x4, 12 accumulators", [&](int it) { hipLaunchKernelGGL((k16<12>), dim3(w), dim3(64), 0, 0, out, it); }, 12 * 4 * 2048.0, w, iters);
    }
    return 0;
}
