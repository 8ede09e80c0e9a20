// K10 -- per-plane normalisation of NCHW maps with the activation that follows it, forward and backward:
//   y = act( (x - mean_bc) * rstd_bc * gamma_c + beta_c ),   mean / var over the H*W pixels of one (batch, channel) plane.
// This one kernel pair is
//   * nn.GroupNorm(C, C)                      of MedNeXtBlock / MedNeXtDownBlock / PatchExpand (reference
//                                             nnUNetTrainer_MLAgg_2D_dt_MS.py:268-270, 357, 500-502), act = none;
//   * nn.InstanceNorm2d + LeakyReLU(0.01)     of the MONAI UnetResBlock in encoder0 / decoder0 (structure vendored at
//                                             MambaSkip.py:581-667; T:1339-1357), gamma = beta = NULL;
//   * nn.InstanceNorm2d(affine) + SiLU        of the MSMM convolution branch (MambaSkip.py:700-706).
// ATen runs them as native_group_norm / native_batch_norm (+ a separate activation kernel each way): 2.2 ms per step at
// config 2.  A workgroup owns one plane: pass 1 sums it from HBM, passes 2 (variance about the mean) and 3 (write) re-read
// it from L2 (a plane is at most 256 KB); backward the same with two reductions.  d(gamma), d(beta) leave as per-plane
// partials and are summed over the batch by the shared column-sum kernel.
// HBM-bound: 8 bytes per element forward, 12 backward (fp32 maps).
//
// Round 3: every map (x, res, y; dy, dx, dres) carries an element-type code (lpio.h): in the bf16 / fp16 modes a map that only a
// 16-bit library convolution reads or wrote stays 16-bit in memory (the reference's autocast tensors), statistics and arithmetic fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlagg_hip.h"
#include "prof.h"
#include "internal.h"
#include "lpio.h"

namespace {

enum Act { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SILU = 2 };

__device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();                                   // red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope)
{
    if (act == ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == ACT_SILU) return v / (1.f + __expf(-v));
    return v;
}

__device__ __forceinline__ float act_bwd(float v, int act, float slope)      // d act / d v at pre-activation v
{
    if (act == ACT_LEAKY) return v > 0.f ? 1.f : slope;
    if (act == ACT_SILU) {
        const float s = 1.f / (1.f + __expf(-v));
        return s * (1.f + v * (1.f - s));
    }
    return 1.f;
}


using namespace mlagg_lpio;

struct PNF {                 // forward operands
    const void *x, *res;
    void *y;
    const float *gamma, *beta;
    float *stats;
    int xdt, rdt, ydt, C, act;
    long HW;
    float eps, slope;
};

struct PNB {                 // backward operands
    const void *x, *dy, *res;
    void *dx, *dres;
    const float *gamma, *beta, *stats;
    float *part;
    int xdt, gdt, rdt, C, act;
    long HW;
    float slope;
    long dy_skip;            // planes between the last channel of a sample of dy and the first of the next (dy = a channel slice of a wider map)
};

// streaming forward: three passes over the plane (the second and third from L2); VEC: HW % 4 == 0 and aligned planes
template <bool VEC>
__global__ void __launch_bounds__(256)
plane_norm_fwd_kernel(PNF a)
{
    __shared__ float red[4];
    const long plane = blockIdx.x, HW = a.HW;
    const int c = (int)(plane % a.C);
    const void *xp = plane_ptr(a.x, plane, HW, a.xdt);
    void *yp = plane_ptr(a.y, plane, HW, a.ydt);
    const void *rp = a.res ? plane_ptr(a.res, plane, HW, a.rdt) : nullptr;
    const long n4 = VEC ? HW >> 2 : 0;
    float s = 0.f;
    for (long i = threadIdx.x; i < n4; i += 256) {
        const float4 v = ld4(xp, i, a.xdt);
        s += (v.x + v.y) + (v.z + v.w);
    }
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256) s += ld1(xp, i, a.xdt);
    const float mean = block_sum(s, red) / (float)HW;
    float q = 0.f;
    for (long i = threadIdx.x; i < n4; i += 256) {
        const float4 v = ld4(xp, i, a.xdt);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256) { const float d = ld1(xp, i, a.xdt) - mean; q += d * d; }
    const float rstd = rsqrtf(block_sum(q, red) / (float)HW + a.eps);
    const float ga = (a.gamma ? a.gamma[c] : 1.f) * rstd, be = (a.beta ? a.beta[c] : 0.f) - mean * ga;
    const int act = a.act;
    const float slope = a.slope;
    for (long i = threadIdx.x; i < n4; i += 256) {
        const float4 v = ld4(xp, i, a.xdt);
        const float4 r = rp ? ld4(rp, i, a.rdt) : make_float4(0.f, 0.f, 0.f, 0.f);
        st4(yp, i, a.ydt, make_float4(act_fwd(v.x * ga + be + r.x, act, slope), act_fwd(v.y * ga + be + r.y, act, slope),
                                      act_fwd(v.z * ga + be + r.z, act, slope), act_fwd(v.w * ga + be + r.w, act, slope)));
    }
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256)
        st1(yp, i, a.ydt, act_fwd(ld1(xp, i, a.xdt) * ga + be + (rp ? ld1(rp, i, a.rdt) : 0.f), act, slope));
    if (threadIdx.x == 0) { a.stats[2 * plane] = mean; a.stats[2 * plane + 1] = rstd; }
}

// g = dy * act'(pre);  dx = rstd * gamma * (g - mean(g) - xhat * mean(g * xhat));  partials: sum g * xhat, sum g
// RES as a template flag and 4 groups per loop trip: planes are few (B * C workgroups), so the loop itself has to provide the
// memory-level parallelism.
template <bool VEC, bool RES>
__global__ void __launch_bounds__(256)
plane_norm_bwd_kernel(PNB a)
{
    __shared__ float red[4];
    const long plane = blockIdx.x, HW = a.HW;
    const int c = (int)(plane % a.C), act = a.act;
    const float slope = a.slope;
    const void *xp = plane_ptr(a.x, plane, HW, a.xdt), *gp = plane_ptr(a.dy, plane + (plane / a.C) * a.dy_skip, HW, a.gdt);
    void *dp = plane_ptr(a.dx, plane, HW, a.xdt);
    const void *rp = RES ? plane_ptr(a.res, plane, HW, a.rdt) : nullptr;
    void *drp = (RES && a.dres) ? plane_ptr(a.dres, plane, HW, a.rdt) : nullptr;
    const float mean = a.stats[2 * plane], rstd = a.stats[2 * plane + 1];
    const float ga = a.gamma ? a.gamma[c] : 1.f, be = a.beta ? a.beta[c] : 0.f;
    const long n4 = VEC ? HW >> 2 : 0;
    float s1 = 0.f, s2 = 0.f;
    auto term = [&](float xv, float gv, float rv, float &xh) {
        xh = (xv - mean) * rstd;
        return act == ACT_NONE ? gv : gv * act_bwd(xh * ga + be + rv, act, slope);
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int UN = 4;
    long i0 = threadIdx.x;
    for (; i0 + 256 * (UN - 1) < n4; i0 += 256 * UN) {
        float4 x4[UN], g4[UN], r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            x4[u] = ld4(xp, i0 + 256 * u, a.xdt);
            g4[u] = ld4(gp, i0 + 256 * u, a.gdt);
            r[u] = RES ? ld4(rp, i0 + 256 * u, a.rdt) : z4;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float xh;
            float t = term(x4[u].x, g4[u].x, r[u].x, xh); s1 += t; s2 += t * xh;
            t = term(x4[u].y, g4[u].y, r[u].y, xh); s1 += t; s2 += t * xh;
            t = term(x4[u].z, g4[u].z, r[u].z, xh); s1 += t; s2 += t * xh;
            t = term(x4[u].w, g4[u].w, r[u].w, xh); s1 += t; s2 += t * xh;
        }
    }
    for (long i = i0; i < n4; i += 256) {
        const float4 x4 = ld4(xp, i, a.xdt), g4 = ld4(gp, i, a.gdt);
        const float4 r = RES ? ld4(rp, i, a.rdt) : z4;
        float xh;
        float t = term(x4.x, g4.x, r.x, xh); s1 += t; s2 += t * xh;
        t = term(x4.y, g4.y, r.y, xh); s1 += t; s2 += t * xh;
        t = term(x4.z, g4.z, r.z, xh); s1 += t; s2 += t * xh;
        t = term(x4.w, g4.w, r.w, xh); s1 += t; s2 += t * xh;
    }
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256) {
        float xh;
        const float t = term(ld1(xp, i, a.xdt), ld1(gp, i, a.gdt), RES ? ld1(rp, i, a.rdt) : 0.f, xh);
        s1 += t; s2 += t * xh;
    }
    const float S1 = block_sum(s1, red), S2 = block_sum(s2, red);
    const float m1 = S1 / (float)HW, m2 = S2 / (float)HW, k = rstd * ga;
    auto finish = [&](long i, const float4 &x4, const float4 &g4, const float4 &r) {
        float xh;
        float4 o, tr;
        float t = term(x4.x, g4.x, r.x, xh); o.x = k * (t - m1 - xh * m2); tr.x = t;
        t = term(x4.y, g4.y, r.y, xh); o.y = k * (t - m1 - xh * m2); tr.y = t;
        t = term(x4.z, g4.z, r.z, xh); o.z = k * (t - m1 - xh * m2); tr.z = t;
        t = term(x4.w, g4.w, r.w, xh); o.w = k * (t - m1 - xh * m2); tr.w = t;
        st4(dp, i, a.xdt, o);
        if (RES && drp) st4(drp, i, a.rdt, tr);              // gradient of the residual input (optional)
    };
    i0 = threadIdx.x;
    for (; i0 + 256 * (UN - 1) < n4; i0 += 256 * UN) {
        float4 x4[UN], g4[UN], r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            x4[u] = ld4(xp, i0 + 256 * u, a.xdt);
            g4[u] = ld4(gp, i0 + 256 * u, a.gdt);
            r[u] = RES ? ld4(rp, i0 + 256 * u, a.rdt) : z4;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) finish(i0 + 256 * u, x4[u], g4[u], r[u]);
    }
    for (long i = i0; i < n4; i += 256) finish(i, ld4(xp, i, a.xdt), ld4(gp, i, a.gdt), RES ? ld4(rp, i, a.rdt) : z4);
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256) {
        float xh;
        const float t = term(ld1(xp, i, a.xdt), ld1(gp, i, a.gdt), RES ? ld1(rp, i, a.rdt) : 0.f, xh);
        st1(dp, i, a.xdt, k * (t - m1 - xh * m2));
        if (RES && drp) st1(drp, i, a.rdt, t);
    }
    if (threadIdx.x == 0 && a.part) { a.part[2 * plane] = S2; a.part[2 * plane + 1] = S1; }     // d(gamma), d(beta) of this plane
}

// Register-resident forward for planes of at most NT * 64 elements (HW % 4 == 0): every thread keeps its 16 groups of 4 of the
// plane in VGPRs, so the plane is read from HBM exactly once (the streaming kernel above re-reads it twice, and 256
// workgroups x 256 KB do not stay in L2).  NT = 256 covers 128 x 128 maps, NT = 1024 256 x 256: 0.84 -> 0.61 ms per step.
// (The same for the backward needs 2 x 64 data registers: measured slower at 155 VGPRs, spills at 1024 threads; it streams.)
template <int NT>
__device__ __forceinline__ float block_sum_nt(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}

constexpr int NV = 16;

template <int NT>
__global__ void __launch_bounds__(NT)
plane_norm_fwd_reg_kernel(PNF a)
{
    __shared__ float red[NT / 64];
    const long plane = blockIdx.x, HW = a.HW;
    const int c = (int)(plane % a.C);
    const void *xp = plane_ptr(a.x, plane, HW, a.xdt);
    void *yp = plane_ptr(a.y, plane, HW, a.ydt);
    const void *rp = a.res ? plane_ptr(a.res, plane, HW, a.rdt) : nullptr;
    const int n4 = (int)(HW >> 2);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * NT;
        v[i] = idx < n4 ? ld4(xp, idx, a.xdt) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = block_sum_nt<NT>(s, red) / (float)HW;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (threadIdx.x + i * NT < n4) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float rstd = rsqrtf(block_sum_nt<NT>(q, red) / (float)HW + a.eps);
    const float ga = (a.gamma ? a.gamma[c] : 1.f) * rstd, be = (a.beta ? a.beta[c] : 0.f) - mean * ga;
    const int act = a.act;
    const float slope = a.slope;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * NT;
        if (idx < n4) {
            const float4 r = rp ? ld4(rp, idx, a.rdt) : make_float4(0.f, 0.f, 0.f, 0.f);
            st4(yp, idx, a.ydt, make_float4(act_fwd(v[i].x * ga + be + r.x, act, slope), act_fwd(v[i].y * ga + be + r.y, act, slope),
                                            act_fwd(v[i].z * ga + be + r.z, act, slope), act_fwd(v[i].w * ga + be + r.w, act, slope)));
        }
    }
    if (threadIdx.x == 0) { a.stats[2 * plane] = mean; a.stats[2 * plane + 1] = rstd; }
}

// ------------------------------------------------------------------------------------------------------------------
// Large planes (the 3-D network: InstanceNorm3d over 96 x 160 x 160 = 2.46 M voxels, 64 planes per map): one workgroup per plane
// leaves 3/4 of the CUs idle and streams 10 MB through a single workgroup three times (2.5-3.9 ms per launch).  A plane is cut
// into SEG-float segments, one workgroup each:
//   forward 1: the segment stays in registers: its mean and its sum of squared deviations about THAT mean;
//   forward 2: every workgroup combines the S partials (Chan's update: exact pooled mean / variance, no sum-of-squares
//              cancellation), then normalises its segment: 2 reads + 1 write of the map instead of 3 + 1 through 64 workgroups;
//   backward 1 / 2: the two per-plane sums as segment partials, then the gradient; d(gamma) / d(beta) partials per plane as before.
// ------------------------------------------------------------------------------------------------------------------
constexpr int SEG4 = 4096;                       // float4 per segment: 256 threads x 16

__global__ void __launch_bounds__(256)
plane_split_stats_kernel(const void *__restrict__ x, int xdt, float *__restrict__ partial, long HW, int S)
{
    __shared__ float red[4];
    const long plane = blockIdx.y;
    const int seg = blockIdx.x;
    const void *xp = plane_ptr(x, plane, HW, xdt);
    const long n4 = HW >> 2, lo = (long)seg * SEG4, cnt = min((long)SEG4, n4 - lo);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const long idx = threadIdx.x + i * 256;
        v[i] = idx < cnt ? ld4(xp, lo + idx, xdt) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float n = 4.f * (float)cnt, mean = block_sum(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (threadIdx.x + i * 256 < cnt) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float m2 = block_sum(q, red);
    if (threadIdx.x == 0) {
        float *p = partial + (plane * S + seg) * 3;
        p[0] = mean; p[1] = m2; p[2] = n;
    }
}

// pooled mean / rstd of a plane from its S segment partials, by every thread of the workgroup (S <= a few hundred)
__device__ __forceinline__ void pooled_stats(const float *__restrict__ partial, long plane, int S, long HW, float eps, float *red,
                                             float &mean, float &rstd)
{
    const float *p = partial + plane * S * 3;
    float a = 0.f;
    for (int i = threadIdx.x; i < S; i += 256) a += p[3 * i] * p[3 * i + 2];
    mean = block_sum(a, red) / (float)HW;
    float b = 0.f;
    for (int i = threadIdx.x; i < S; i += 256) { const float d = p[3 * i] - mean; b += p[3 * i + 1] + p[3 * i + 2] * d * d; }
    rstd = rsqrtf(block_sum(b, red) / (float)HW + eps);
}

__global__ void __launch_bounds__(256)
plane_split_apply_kernel(PNF a, const float *__restrict__ partial, int S)
{
    __shared__ float red[4];
    const long plane = blockIdx.y, HW = a.HW;
    const int seg = blockIdx.x, c = (int)(plane % a.C), act = a.act;
    const float slope = a.slope;
    float mean, rstd;
    pooled_stats(partial, plane, S, HW, a.eps, red, mean, rstd);
    const float ga = (a.gamma ? a.gamma[c] : 1.f) * rstd, be = (a.beta ? a.beta[c] : 0.f) - mean * ga;
    const void *xp = plane_ptr(a.x, plane, HW, a.xdt);
    void *yp = plane_ptr(a.y, plane, HW, a.ydt);
    const bool has_res = a.res != nullptr;
    const void *rp = has_res ? plane_ptr(a.res, plane, HW, a.rdt) : xp;
    const int rdt = has_res ? a.rdt : a.xdt;
    float *stats = a.stats;
    const long n4 = HW >> 2, lo = (long)seg * SEG4, cnt = min((long)SEG4, n4 - lo);
    float4 v[NV], r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const long idx = min((long)(threadIdx.x + i * 256), cnt - 1);          // unconditional loads (clamped)
        v[i] = ld4(xp, lo + idx, a.xdt);
        r[i] = ld4(rp, lo + idx, rdt);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const long idx = threadIdx.x + i * 256;
        if (idx < cnt) {
            const float r0 = has_res ? r[i].x : 0.f, r1 = has_res ? r[i].y : 0.f, r2 = has_res ? r[i].z : 0.f, r3 = has_res ? r[i].w : 0.f;
            st4(yp, lo + idx, a.ydt, make_float4(act_fwd(v[i].x * ga + be + r0, act, slope), act_fwd(v[i].y * ga + be + r1, act, slope),
                                                 act_fwd(v[i].z * ga + be + r2, act, slope), act_fwd(v[i].w * ga + be + r3, act, slope)));
        }
    }
    if (seg == 0 && threadIdx.x == 0) { stats[2 * plane] = mean; stats[2 * plane + 1] = rstd; }
}

template <bool RES, bool APPLY>
__global__ void __launch_bounds__(256)
plane_split_bwd_kernel(PNB a, float *__restrict__ partial, int S)
{
    __shared__ float red[4];
    const long plane = blockIdx.y, HW = a.HW;
    const int seg = blockIdx.x, c = (int)(plane % a.C), act = a.act;
    const float slope = a.slope;
    const float mean = a.stats[2 * plane], rstd = a.stats[2 * plane + 1];
    const float ga = a.gamma ? a.gamma[c] : 1.f, be = a.beta ? a.beta[c] : 0.f;
    const void *xp = plane_ptr(a.x, plane, HW, a.xdt), *gp = plane_ptr(a.dy, plane + (plane / a.C) * a.dy_skip, HW, a.gdt);
    const void *rp = RES ? plane_ptr(a.res, plane, HW, a.rdt) : xp;
    void *dxp = plane_ptr(a.dx, plane, HW, a.xdt);
    void *drp = (RES && a.dres) ? plane_ptr(a.dres, plane, HW, a.rdt) : nullptr;
    float *part = a.part;
    const long n4 = HW >> 2, lo = (long)seg * SEG4, cnt = min((long)SEG4, n4 - lo);
    auto term = [&](float xv, float gv, float rv, float &xh) {
        xh = (xv - mean) * rstd;
        return act == ACT_NONE ? gv : gv * act_bwd(xh * ga + be + rv, act, slope);
    };
    float m1 = 0.f, m2 = 0.f;
    if (APPLY) {
        const float *p = partial + plane * S * 2;
        float a = 0.f, b = 0.f;
        for (int i = threadIdx.x; i < S; i += 256) { a += p[2 * i]; b += p[2 * i + 1]; }
        const float S1 = block_sum(a, red), S2 = block_sum(b, red);
        m1 = S1 / (float)HW; m2 = S2 / (float)HW;
        if (seg == 0 && threadIdx.x == 0 && part) { part[2 * plane] = S2; part[2 * plane + 1] = S1; }
    }
    const float k = rstd * ga;
    float s1 = 0.f, s2 = 0.f;
    constexpr int UN = 4;
#pragma unroll
    for (int i0 = 0; i0 < NV; i0 += UN) {
        float4 x4[UN], g4[UN], r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long idx = min((long)(threadIdx.x + (i0 + u) * 256), cnt - 1);
            x4[u] = ld4(xp, lo + idx, a.xdt);
            g4[u] = ld4(gp, lo + idx, a.gdt);
            r[u] = RES ? ld4(rp, lo + idx, a.rdt) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long idx = threadIdx.x + (i0 + u) * 256;
            if (idx >= cnt) continue;
            float xh;
            float4 o, tr;
            float t = term(x4[u].x, g4[u].x, r[u].x, xh); s1 += t; s2 += t * xh; o.x = k * (t - m1 - xh * m2); tr.x = t;
            t = term(x4[u].y, g4[u].y, r[u].y, xh); s1 += t; s2 += t * xh; o.y = k * (t - m1 - xh * m2); tr.y = t;
            t = term(x4[u].z, g4[u].z, r[u].z, xh); s1 += t; s2 += t * xh; o.z = k * (t - m1 - xh * m2); tr.z = t;
            t = term(x4[u].w, g4[u].w, r[u].w, xh); s1 += t; s2 += t * xh; o.w = k * (t - m1 - xh * m2); tr.w = t;
            if (APPLY) {
                st4(dxp, lo + idx, a.xdt, o);
                if (RES && drp) st4(drp, lo + idx, a.rdt, tr);
            }
        }
    }
    if (!APPLY) {
        const float S1 = block_sum(s1, red), S2 = block_sum(s2, red);
        if (threadIdx.x == 0) { partial[(plane * S + seg) * 2] = S1; partial[(plane * S + seg) * 2 + 1] = S2; }
    }
}

inline int split_segments(long HW)          // 0: the one-workgroup-per-plane kernels
{
    if ((HW & 3) || HW < 64L * 4096) return 0;
    return (int)(((HW >> 2) + SEG4 - 1) / SEG4);
}

int check(int B, int C, long HW, int act)
{
    if (B <= 0 || C <= 0 || HW <= 0 || (long)B * C > 2147483647L || act < 0 || act > 2) return MLAGG_E_UNSUPPORTED;
    return 0;
}

inline bool dtype_ok(int dt) { return dt == MLAGG_DTYPE_F32 || dt == MLAGG_DTYPE_BF16 || dt == MLAGG_DTYPE_F16; }
inline bool aligned(const void *p, int dt) { return (((uintptr_t)p) & (dt == 0 ? 15 : 7)) == 0; }

}  // namespace

extern "C" size_t mlagg_plane_norm_fwd_workspace_floats(int B, int C, long HW)
{
    return (size_t)(B > 0 ? B : 0) * (C > 0 ? C : 0) * split_segments(HW) * 3;
}

extern "C" size_t mlagg_plane_norm_bwd_workspace_floats(int B, int C, long HW)
{
    return (size_t)(B > 0 ? B : 0) * (C > 0 ? C : 0) * (2 + 2 * (size_t)split_segments(HW));
}

extern "C" int mlagg_plane_norm_fwd(const void *x, const float *gamma, const float *beta, const void *res, void *y, float *stats,
                                    float *workspace, int B, int C, long HW, float eps, int act, float slope, int x_dtype,
                                    int res_dtype, int y_dtype, void *stream)
{
    if (!x || !y || !stats) return MLAGG_E_NULLPTR;
    if (int rc = check(B, C, HW, act)) return rc;
    if (!dtype_ok(x_dtype) || !dtype_ok(y_dtype) || (res && !dtype_ok(res_dtype))) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = (HW & 3) == 0 && aligned(x, x_dtype) && aligned(y, y_dtype) && (!res || aligned(res, res_dtype));
    PNF a{x, res, y, gamma, beta, stats, x_dtype, res_dtype, y_dtype, C, act, HW, eps, slope};
    MLAGG_TIMED(K_PLANE_NORM_FWD, st);
    const int S = vec ? split_segments(HW) : 0;
    if (S > 0) {
        if (!workspace) return MLAGG_E_WORKSPACE;
        if (S > 65535 || (long)B * C > 65535) return MLAGG_E_UNSUPPORTED;
        hipLaunchKernelGGL(plane_split_stats_kernel, dim3(S, B * C), dim3(256), 0, st, x, x_dtype, workspace, HW, S);
        hipLaunchKernelGGL(plane_split_apply_kernel, dim3(S, B * C), dim3(256), 0, st, a, workspace, S);
        return (int)hipGetLastError();
    }
    if (vec && HW <= 256 * 4 * NV) hipLaunchKernelGGL(plane_norm_fwd_reg_kernel<256>, dim3(B * C), dim3(256), 0, st, a);
    else if (vec && HW <= 1024 * 4 * NV) hipLaunchKernelGGL(plane_norm_fwd_reg_kernel<1024>, dim3(B * C), dim3(1024), 0, st, a);
    else if (vec) hipLaunchKernelGGL(plane_norm_fwd_kernel<true>, dim3(B * C), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(plane_norm_fwd_kernel<false>, dim3(B * C), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int mlagg_plane_norm_bwd(const void *x, const void *dy, const float *gamma, const float *beta, const void *res,
                                    const float *stats, void *dx, void *dres, float *dgamma, float *dbeta, float *workspace,
                                    int B, int C, long HW, int act, float slope, int x_dtype, int dy_dtype, int res_dtype, void *stream)
{
    return mlagg_plane_norm_bwd_strided(x, dy, 0, gamma, beta, res, stats, dx, dres, dgamma, dbeta, workspace, B, C, HW, act, slope, x_dtype,
                                        dy_dtype, res_dtype, stream);
}

// the same with dy a channel slice of a wider map: dy_batch elements between samples (0 or C * HW: contiguous), a multiple of HW
extern "C" int mlagg_plane_norm_bwd_strided(const void *x, const void *dy, long dy_batch, const float *gamma, const float *beta,
                                            const void *res, const float *stats, void *dx, void *dres, float *dgamma, float *dbeta,
                                            float *workspace, int B, int C, long HW, int act, float slope, int x_dtype, int dy_dtype,
                                            int res_dtype, void *stream)
{
    if (!x || !dy || !stats || !dx || ((dgamma || dbeta) && !workspace)) return MLAGG_E_NULLPTR;
    if (int rc = check(B, C, HW, act)) return rc;
    if (dy_batch == 0) dy_batch = (long)C * HW;
    if (dy_batch < (long)C * HW || dy_batch % HW) return MLAGG_E_UNSUPPORTED;
    if (!dtype_ok(x_dtype) || !dtype_ok(dy_dtype) || (res && !dtype_ok(res_dtype))) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = (HW & 3) == 0 && aligned(x, x_dtype) && aligned(dy, dy_dtype) && aligned(dx, x_dtype) &&
                     (!res || aligned(res, res_dtype)) && (!dres || aligned(dres, res_dtype));
    float *part = (dgamma || dbeta) ? workspace : nullptr;
    PNB a{x, dy, res, dx, dres, gamma, beta, stats, part, x_dtype, dy_dtype, res_dtype, C, act, HW, slope, dy_batch / HW - C};
    const int S = vec ? split_segments(HW) : 0;
    if (S > 0) {
        if (!workspace) return MLAGG_E_WORKSPACE;
        if (S > 65535 || (long)B * C > 65535) return MLAGG_E_UNSUPPORTED;
        float *partial = workspace + (size_t)B * C * 2;
        MLAGG_TIMED(K_PLANE_NORM_BWD, st);
        const dim3 grid(S, B * C), block(256);
        if (res) {
            hipLaunchKernelGGL((plane_split_bwd_kernel<true, false>), grid, block, 0, st, a, partial, S);
            hipLaunchKernelGGL((plane_split_bwd_kernel<true, true>), grid, block, 0, st, a, partial, S);
        } else {
            hipLaunchKernelGGL((plane_split_bwd_kernel<false, false>), grid, block, 0, st, a, partial, S);
            hipLaunchKernelGGL((plane_split_bwd_kernel<false, true>), grid, block, 0, st, a, partial, S);
        }
    } else {
        MLAGG_TIMED(K_PLANE_NORM_BWD, st);
#define MLAGG_PN_BWD(VEC, RES) hipLaunchKernelGGL((plane_norm_bwd_kernel<VEC, RES>), dim3(B * C), dim3(256), 0, st, a)
        const bool with_res = res != nullptr;
        if (vec && with_res) MLAGG_PN_BWD(true, true);
        else if (vec) MLAGG_PN_BWD(true, false);
        else if (with_res) MLAGG_PN_BWD(false, true);
        else MLAGG_PN_BWD(false, false);
#undef MLAGG_PN_BWD
    }
    if (part)           // partials are a (B) x (2C) matrix [c][dgamma | dbeta] interleaved: columns 2c, 2c+1
        hipLaunchKernelGGL(mlagg_internal::column_sum_interleaved_kernel<0>, dim3((2 * C + 63) / 64), dim3(1024), 0, st, part, B,
                           2 * C, dgamma, dbeta);
    return (int)hipGetLastError();
}
