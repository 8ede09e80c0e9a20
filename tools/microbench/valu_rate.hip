// Issue-rate microbenchmark: wave64 v_fma_f32 vs v_pk_fma_f32 vs v_fmac with DPP, at 1 / 2 / 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0001f, c = 0.5f;
    const float2v mm = {m, m}, cc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                             "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char *name, float *out, int waves_per_simd)
{
    const int iters = 20000, blocks = 256 * 4 * waves_per_simd;   // one 64-thread block per wave slot
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, 100);
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * 64;                        // wave instructions per wave
    const double cyc = ms * 1e-3 * 2.4e9;                          // nominal clock
    printf("%-14s %d waves/SIMD: %.3f ms, %.2f cycles (at 2.4 GHz) per wave-instruction per SIMD\n", name, waves_per_simd, ms,
           cyc / (inst * waves_per_simd));
}

int main()
{
    float *out; hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", out, w);
        run<1>("v_pk_fma_f32", out, w);
        run<2>("v_fmac_f32", out, w);
        run<3>("v_exp_f32", out, w);
    }
    return 0;
}
