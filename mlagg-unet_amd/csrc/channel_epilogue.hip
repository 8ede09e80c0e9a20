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
#include "lpio.h"
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

// ---- 16-bit modes: the convolution output x arrives in bf16 / fp16 (what the 16-bit library convolution wrote) and is NOT modified;
// y = act(x + bias[c] + res) leaves in the element type its consumer wants (16-bit when only the next convolution reads it, fp32 when
// it joins the residual stream).  Backward recomputes the pre-activation from x (+ bias + res): dx = dy * act'(pre) in x's type (the
// convolution's gradient operand), per-plane sums for d(bias) in the same pass.  vec: HW % 4 == 0.
using namespace mlagg_lpio;

template <bool GELU>
__global__ void __launch_bounds__(256)
channel_epilogue_lp_fwd_kernel(const void *__restrict__ x, int xdt, const float *__restrict__ bias, const void *__restrict__ res, int rdt,
                               void *__restrict__ y, int ydt, int C, long HW)
{
    const int c = blockIdx.x, b = blockIdx.y;
    const long plane = (long)b * C + c;
    const float bv = bias ? bias[c] : 0.f;
    const void *xp = plane_ptr(x, plane, HW, xdt);
    const void *rp = res ? plane_ptr(res, plane, HW, rdt) : nullptr;
    void *yp = plane_ptr(y, plane, HW, ydt);
    const long n4 = HW >> 2;
    constexpr int UN = 4;
    for (long i0 = threadIdx.x; i0 < n4; i0 += 256 * UN) {
        float4 v[UN], r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = min(i0 + 256 * u, n4 - 1);
            v[u] = ld4(xp, i, xdt);
            r[u] = rp ? ld4(rp, i, rdt) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + 256 * u;
            if (i < n4) {
                float4 o = make_float4(v[u].x + bv + r[u].x, v[u].y + bv + r[u].y, v[u].z + bv + r[u].z, v[u].w + bv + r[u].w);
                if (GELU) o = make_float4(gelu_f(o.x), gelu_f(o.y), gelu_f(o.z), gelu_f(o.w));
                st4(yp, i, ydt, o);
            }
        }
    }
}

template <bool GELU>
__global__ void __launch_bounds__(256)
channel_epilogue_lp_bwd_kernel(const void *__restrict__ x, int xdt, const float *__restrict__ bias, const void *__restrict__ res, int rdt,
                               const void *__restrict__ dy, int gdt, void *__restrict__ dx, int dxdt, float *__restrict__ part, int C,
                               long HW)
{
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const long plane = (long)b * C + c;
    const float bv = (GELU && bias) ? bias[c] : 0.f;
    const void *xp = GELU ? plane_ptr(x, plane, HW, xdt) : nullptr;
    const void *rp = (GELU && res) ? plane_ptr(res, plane, HW, rdt) : nullptr;
    const void *gp = plane_ptr(dy, plane, HW, gdt);
    void *dp = plane_ptr(dx, plane, HW, dxdt);
    const long n4 = HW >> 2;
    float s = 0.f;
    constexpr int UN = 4;
    for (long i0 = threadIdx.x; i0 < n4; i0 += 256 * UN) {
        float4 v[UN], r[UN], g[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = min(i0 + 256 * u, n4 - 1);
            g[u] = ld4(gp, i, gdt);
            v[u] = GELU ? ld4(xp, i, xdt) : make_float4(0.f, 0.f, 0.f, 0.f);
            r[u] = rp ? ld4(rp, i, rdt) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + 256 * u;
            if (i < n4) {
                float4 d = g[u];
                if (GELU)
                    d = make_float4(d.x * gelu_grad_f(v[u].x + bv + r[u].x), d.y * gelu_grad_f(v[u].y + bv + r[u].y),
                                    d.z * gelu_grad_f(v[u].z + bv + r[u].z), d.w * gelu_grad_f(v[u].w + bv + r[u].w));
                st4(dp, i, dxdt, d);
                s += (d.x + d.y) + (d.z + d.w);
            }
        }
    }
    if (part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) part[plane] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

inline bool lp_ok(int dt) { return dt == MLAGG_DTYPE_F32 || dt == MLAGG_DTYPE_BF16 || dt == MLAGG_DTYPE_F16; }
inline bool lp_aligned(const void *p, int dt) { return (((uintptr_t)p) & (dt == 0 ? 15 : 7)) == 0; }

}  // namespace

extern "C" int mlagg_channel_epilogue_lp_fwd(const void *x, int x_dtype, const float *bias, const void *res, int res_dtype, void *y,
                                             int y_dtype, int B, int C, long HW, int act, void *stream)
{
    if (!x || !y) return MLAGG_E_NULLPTR;
    if (B <= 0 || C <= 0 || HW <= 0 || B > 65535 || (HW & 3) || (act != 0 && act != 1)) return MLAGG_E_UNSUPPORTED;
    if (!lp_ok(x_dtype) || !lp_ok(y_dtype) || (res && !lp_ok(res_dtype))) return MLAGG_E_UNSUPPORTED;
    if (!lp_aligned(x, x_dtype) || !lp_aligned(y, y_dtype) || (res && !lp_aligned(res, res_dtype))) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CHANNEL_EPI, st);
    if (act == 1) hipLaunchKernelGGL(channel_epilogue_lp_fwd_kernel<true>, dim3(C, B), dim3(256), 0, st, x, x_dtype, bias, res, res_dtype, y, y_dtype, C, HW);
    else hipLaunchKernelGGL(channel_epilogue_lp_fwd_kernel<false>, dim3(C, B), dim3(256), 0, st, x, x_dtype, bias, res, res_dtype, y, y_dtype, C, HW);
    return (int)hipGetLastError();
}

// dx = dy * act'(x + bias + res) in dx_dtype; dbias (C, may be NULL) = its sum over batch and pixels (workspace:
// mlagg_channel_sum_workspace_floats(B, C) floats).  act 0: x, bias, res are not read (may be NULL): a typed copy of dy + the bias sums.
extern "C" int mlagg_channel_epilogue_lp_bwd(const void *x, int x_dtype, const float *bias, const void *res, int res_dtype,
                                             const void *dy, int dy_dtype, void *dx, int dx_dtype, float *dbias, float *workspace,
                                             int B, int C, long HW, int act, void *stream)
{
    if (!dy || !dx || (act == 1 && !x) || (dbias && !workspace)) return MLAGG_E_NULLPTR;
    if (B <= 0 || C <= 0 || HW <= 0 || B > 65535 || (HW & 3) || (act != 0 && act != 1)) return MLAGG_E_UNSUPPORTED;
    if (!lp_ok(dy_dtype) || !lp_ok(dx_dtype) || (act == 1 && !lp_ok(x_dtype)) || (act == 1 && res && !lp_ok(res_dtype))) return MLAGG_E_UNSUPPORTED;
    if (!lp_aligned(dy, dy_dtype) || !lp_aligned(dx, dx_dtype) || (act == 1 && !lp_aligned(x, x_dtype)) ||
        (act == 1 && res && !lp_aligned(res, res_dtype)))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CHANNEL_EPI, st);
    float *part = dbias ? workspace : nullptr;
    if (act == 1) hipLaunchKernelGGL(channel_epilogue_lp_bwd_kernel<true>, dim3(C, B), dim3(256), 0, st, x, x_dtype, bias, res, res_dtype, dy, dy_dtype, dx, dx_dtype, part, C, HW);
    else hipLaunchKernelGGL(channel_epilogue_lp_bwd_kernel<false>, dim3(C, B), dim3(256), 0, st, x, x_dtype, bias, res, res_dtype, dy, dy_dtype, dx, dx_dtype, part, C, HW);
    if (dbias)
        hipLaunchKernelGGL(mlagg_internal::column_sum_kernel<false>, dim3((C + 63) / 64), dim3(1024), 0, st, workspace, B, C, C, dbias);
    return (int)hipGetLastError();
}

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
