// K13 -- epilogue of the library convolutions on NCHW maps: bias, residual and GELU in ONE pass.
//
// The convolutions of the stem, the MedNeXt down / decoder blocks and PatchExpand run in MIOpen without their bias
// (model.Conv2d); what follows them in the reference is elementwise (nnUNetTrainer_MLAgg_2D_dt_MS.py:307-324 MedNeXtBlock:
// conv2 -> GELU, conv3 -> + x; :358-366 down block: conv3 -> + res_conv(x); :984-1001 project: conv -> GELU; :498-546
// PatchExpand: conv -> + res_conv).  Round 1 spent one ATen launch per step of that chain (broadcast bias add, GELU, add:
// 41 + 12 + 9 launches and ~2 reads + 2 writes of the map per step where 1 + 1 suffice).
//   forward   y = act(x + bias[c] + res)      act: 0 none, 1 GELU (erf form, torch's default)
//             GELU: `x` is overwritten with the pre-activation x + bias (+ res) -- saved for backward -- and y is a new
//             map (1 read + 2 writes instead of 2 + 2); no activation: in place, one read + one write (+ res read)
//   backward  GELU: dpre = dy * gelu'(pre) in one pass that also leaves the per-plane sums of dpre for the bias gradient
//             (d(bias) = sum over batch and pixels, finished by the column-sum kernel); no activation: dy passes through
//             and d(bias) is mlagg_channel_sum(dy).
// HBM-bound: 8 (12 with a residual) bytes per element forward, 12 backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "internal.h"
#include "mlagg_hip.h"
#include "prof.h"

namespace {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x)
{
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// one workgroup per (batch, channel) plane
template <bool GELU>
__global__ void __launch_bounds__(256)
channel_epilogue_fwd_kernel(float *__restrict__ x, const float *__restrict__ bias, const float *__restrict__ res,
                            float *__restrict__ y, int C, long HW)
{
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t off = ((size_t)b * C + c) * HW;
    const float bv = bias ? bias[c] : 0.f;
    float *xp = x + off;
    const float *rp = res ? res + off : nullptr;
    float *yp = GELU ? y + off : xp;
    const long n4 = ((HW & 3) == 0) ? HW >> 2 : 0;
    for (long i = threadIdx.x; i < n4; i += 256) {
        float4 v = reinterpret_cast<float4 *>(xp)[i];
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        if (rp) {
            const float4 r = reinterpret_cast<const float4 *>(rp)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (GELU) {
            reinterpret_cast<float4 *>(xp)[i] = v;                                  // pre-activation, kept for backward
            v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
        }
        reinterpret_cast<float4 *>(yp)[i] = v;
    }
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256) {
        float v = xp[i] + bv + (rp ? rp[i] : 0.f);
        if (GELU) { xp[i] = v; v = gelu_f(v); }
        yp[i] = v;
    }
}

// dpre = dy * gelu'(pre); part[b][c] = sum over the plane of dpre
__global__ void __launch_bounds__(256)
channel_gelu_bwd_kernel(const float *__restrict__ pre, const float *__restrict__ dy, float *__restrict__ dpre,
                        float *__restrict__ part, int C, long HW)
{
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t off = ((size_t)b * C + c) * HW;
    const float *pp = pre + off, *gp = dy + off;
    float *dp = dpre + off;
    float s = 0.f;
    const long n4 = ((HW & 3) == 0) ? HW >> 2 : 0;
    for (long i = threadIdx.x; i < n4; i += 256) {
        const float4 p = reinterpret_cast<const float4 *>(pp)[i], g = reinterpret_cast<const float4 *>(gp)[i];
        const float4 d = make_float4(g.x * gelu_grad_f(p.x), g.y * gelu_grad_f(p.y), g.z * gelu_grad_f(p.z), g.w * gelu_grad_f(p.w));
        reinterpret_cast<float4 *>(dp)[i] = d;
        s += (d.x + d.y) + (d.z + d.w);
    }
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256) {
        const float d = gp[i] * gelu_grad_f(pp[i]);
        dp[i] = d;
        s += d;
    }
    if (part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) part[(size_t)b * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

}  // namespace

extern "C" int mlagg_channel_epilogue_fwd(float *x, const float *bias, const float *res, float *y, int B, int C, long HW, int act,
                                          void *stream)
{
    if (!x || (act == 1 && !y)) return MLAGG_E_NULLPTR;
    if (B <= 0 || C <= 0 || HW <= 0 || B > 65535 || (act != 0 && act != 1)) return MLAGG_E_UNSUPPORTED;
    if ((((uintptr_t)x) & 15) || (res && (((uintptr_t)res) & 15)) || (y && (((uintptr_t)y) & 15))) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CHANNEL_EPI, st);
    if (act == 1) hipLaunchKernelGGL(channel_epilogue_fwd_kernel<true>, dim3(C, B), dim3(256), 0, st, x, bias, res, y, C, HW);
    else hipLaunchKernelGGL(channel_epilogue_fwd_kernel<false>, dim3(C, B), dim3(256), 0, st, x, bias, res, y, C, HW);
    return (int)hipGetLastError();
}

// dbias may be NULL (convolution without bias); workspace: mlagg_channel_sum_workspace_floats(B, C) floats
extern "C" int mlagg_channel_gelu_bwd(const float *pre, const float *dy, float *dpre, float *dbias, float *workspace, int B, int C,
                                      long HW, void *stream)
{
    if (!pre || !dy || !dpre || (dbias && !workspace)) return MLAGG_E_NULLPTR;
    if (B <= 0 || C <= 0 || HW <= 0 || B > 65535) return MLAGG_E_UNSUPPORTED;
    if ((((uintptr_t)pre) & 15) || (((uintptr_t)dy) & 15) || (((uintptr_t)dpre) & 15)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CHANNEL_EPI, st);
    hipLaunchKernelGGL(channel_gelu_bwd_kernel, dim3(C, B), dim3(256), 0, st, pre, dy, dpre, dbias ? workspace : nullptr, C, HW);
    if (dbias)
        hipLaunchKernelGGL(mlagg_internal::column_sum_kernel<false>, dim3((C + 63) / 64), dim3(1024), 0, st, workspace, B, C, C, dbias);
    return (int)hipGetLastError();
}
