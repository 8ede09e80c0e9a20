// K7 -- the MLLA block's gate: out = concat(a0, a1) * SiLU(act), forward and backward in one pass each.
//
// Replaces three eager kernels forward (SiLU of act_proj's output, torch.cat of the two attention
// branches, the product; reference nnUNetTrainer_MLAgg_2D_dt_MS.py:888, 899, 902) and four backward.
// Pure streaming op: 16-byte accesses, one token row segment per lane.  Algorithmic bytes per element:
// 12 forward (a, act in; out), 24 backward (dout, a, act in; da, dact out).
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// a0, a1: (rows, h) contiguous; act: (rows, 2h) with row stride act_stride; out: (rows, 2h) contiguous
__global__ void __launch_bounds__(256)
gate_fwd_kernel(const float *__restrict__ a0, const float *__restrict__ a1, const float *__restrict__ act,
                int act_stride, float *__restrict__ out, long rows, int h)
{
    const int q = h >> 2;                       // float4 per half row
    const long n = rows * 2 * q;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (2 * q);
        const int c4 = (int)(i - r * 2 * q);
        const float4 a = c4 < q ? *reinterpret_cast<const float4 *>(a0 + r * h + 4 * c4)
                                : *reinterpret_cast<const float4 *>(a1 + r * h + 4 * (c4 - q));
        const float4 g = *reinterpret_cast<const float4 *>(act + r * act_stride + 4 * c4);
        float4 o;
        o.x = a.x * g.x * sigmoid_f(g.x); o.y = a.y * g.y * sigmoid_f(g.y);
        o.z = a.z * g.z * sigmoid_f(g.z); o.w = a.w * g.w * sigmoid_f(g.w);
        *reinterpret_cast<float4 *>(out + r * 2 * h + 4 * c4) = o;
    }
}

__global__ void __launch_bounds__(256)
gate_bwd_kernel(const float *__restrict__ dout, int dout_stride, const float *__restrict__ a0,
                const float *__restrict__ a1, const float *__restrict__ act, int act_stride, float *__restrict__ da0,
                float *__restrict__ da1, float *__restrict__ dact, int dact_stride, long rows, int h)
{
    const int q = h >> 2;
    const long n = rows * 2 * q;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (2 * q);
        const int c4 = (int)(i - r * 2 * q);
        const bool lo = c4 < q;
        const float4 a = lo ? *reinterpret_cast<const float4 *>(a0 + r * h + 4 * c4)
                            : *reinterpret_cast<const float4 *>(a1 + r * h + 4 * (c4 - q));
        const float4 g = *reinterpret_cast<const float4 *>(act + r * act_stride + 4 * c4);
        const float4 d = *reinterpret_cast<const float4 *>(dout + r * dout_stride + 4 * c4);
        float4 da, dg;
        const float gs[4] = {g.x, g.y, g.z, g.w}, as[4] = {a.x, a.y, a.z, a.w}, ds[4] = {d.x, d.y, d.z, d.w};
        float ra[4], rg[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = sigmoid_f(gs[j]);
            ra[j] = ds[j] * gs[j] * s;                                   // d a   = dout * silu(act)
            rg[j] = ds[j] * as[j] * s * (1.f + gs[j] * (1.f - s));       // d act = dout * a * silu'(act)
        }
        da = make_float4(ra[0], ra[1], ra[2], ra[3]);
        dg = make_float4(rg[0], rg[1], rg[2], rg[3]);
        if (lo) *reinterpret_cast<float4 *>(da0 + r * h + 4 * c4) = da;
        else *reinterpret_cast<float4 *>(da1 + r * h + 4 * (c4 - q)) = da;
        *reinterpret_cast<float4 *>(dact + r * dact_stride + 4 * c4) = dg;
    }
}

inline int grid_for(long n)
{
    long b = (n + 255) / 256;
    return (int)(b < 4096 ? b : 4096);
}

}  // namespace

extern "C" int mlagg_gate_fwd(const float *a0, const float *a1, const float *act, int act_stride, float *out, long rows,
                              int h, void *stream)
{
    if (!a0 || !a1 || !act || !out) return MLAGG_E_NULLPTR;
    if (rows <= 0 || h <= 0 || (h & 3) || act_stride < 2 * h || (act_stride & 3)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_GATE_FWD, st);
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(rows * (h / 2))), dim3(256), 0, st, a0, a1, act, act_stride, out, rows,
                       h);
    return (int)hipGetLastError();
}

extern "C" int mlagg_gate_bwd(const float *dout, int dout_stride, const float *a0, const float *a1, const float *act,
                              int act_stride, float *da0, float *da1, float *dact, int dact_stride, long rows, int h, void *stream)
{
    if (!dout || !a0 || !a1 || !act || !da0 || !da1 || !dact) return MLAGG_E_NULLPTR;
    if (rows <= 0 || h <= 0 || (h & 3) || act_stride < 2 * h || (act_stride & 3) || dout_stride < 2 * h ||
        (dout_stride & 3) || dact_stride < 2 * h || (dact_stride & 3))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_GATE_BWD, st);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(rows * (h / 2))), dim3(256), 0, st, dout, dout_stride, a0, a1, act,
                       act_stride, da0, da1, dact, dact_stride, rows, h);
    return (int)hipGetLastError();
}
