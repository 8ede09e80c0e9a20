// K2n -- depthwise 3x3 convolution on NCHW maps, stride 1 or 2, zero padding 1, + bias.
//
// Replaces nn.Conv2d(groups=C) in MedNeXtBlock.conv1 (reference nnUNetTrainer_MLAgg_2D_dt_MS.py:256-263, 310) and
// MedNeXtDownBlock.conv1 (T:349-356), which stay NCHW because full MIOpen convolutions surround them.
// MIOpen has no tuned fp32 depthwise solver for these shapes on gfx950 and falls back to its naive
// reference kernels (naive_conv_ab_nonpacked_{fwd,bwd}_nchw: 1.8 ms per step in the round-1 profile).
// Lane = output pixel along W (coalesced rows), one (batch, channel) plane slice per workgroup, the 9
// weights are workgroup-uniform (scalar loads).  Weight gradients: LDS block reduction, one partial
// row per workgroup, then the shared column-sum kernel.
//
// HBM-bound: 4 * (H*W + Ho*Wo) bytes per plane forward.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"
#include "internal.h"

namespace {

struct G {
    int B, C, H, W, Ho, Wo, stride;
};

template <int S>                      // S: compile-time stride (divisions and remainders fold), 0 = runtime g.stride
__global__ void __launch_bounds__(256)
dwconv_nchw_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       float *__restrict__ y, G g)
{
    const int stride = S ? S : g.stride;
    const int c = blockIdx.y, b = blockIdx.z;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.Ho * g.Wo) return;
    const int oy = idx / g.Wo, ox = idx - oy * g.Wo;
    const float *xp = x + ((size_t)b * g.C + c) * g.H * g.W;
    const float *wp = w + c * 9;
    float acc = bias ? bias[c] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride + ky - 1;
        if (iy < 0 || iy >= g.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride + kx - 1;
            if (ix >= 0 && ix < g.W) acc += wp[ky * 3 + kx] * xp[(size_t)iy * g.W + ix];
        }
    }
    y[((size_t)b * g.C + c) * g.Ho * g.Wo + idx] = acc;
}

// dx[iy][ix] = sum_{ky,kx} w[ky][kx] * dy[oy][ox] with oy*s + ky - 1 = iy, ox*s + kx - 1 = ix
template <int S>
__global__ void __launch_bounds__(256)
dwconv_nchw_bwd_data_kernel(const float *__restrict__ dy, const float *__restrict__ w, float *__restrict__ dx, G g)
{
    const int stride = S ? S : g.stride;
    const int c = blockIdx.y, b = blockIdx.z;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.H * g.W) return;
    const int iy = idx / g.W, ix = idx - iy * g.W;
    const float *dp = dy + ((size_t)b * g.C + c) * g.Ho * g.Wo;
    const float *wp = w + c * 9;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + 1 - ky;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= g.Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + 1 - kx;
            if (tx < 0 || tx % stride) continue;
            const int ox = tx / stride;
            if (ox < g.Wo) acc += wp[ky * 3 + kx] * dp[(size_t)oy * g.Wo + ox];
        }
    }
    dx[((size_t)b * g.C + c) * g.H * g.W + idx] = acc;
}

// Stride-1 fast path (W % 4 == 0; every MedNeXtBlock.conv1 of the decoder): a thread owns 4 consecutive pixels of one row,
// reads each of the 3 input rows as one aligned float4 plus its two neighbours (9 loads for 4 outputs instead of 36
// guarded dword loads) and stores a float4.  FLIP = true applies the kernel rotated by 180 degrees: the data gradient.
template <bool FLIP>
__global__ void __launch_bounds__(256)
dwconv_nchw_s1_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                      float *__restrict__ y, G g, const float *__restrict__ add = nullptr)
{
    const int c = blockIdx.y, b = blockIdx.z;
    const int W4 = g.W >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.H * W4) return;
    const int oy = idx / W4, x0 = 4 * (idx - oy * W4);
    const float *xp = x + ((size_t)b * g.C + c) * g.H * g.W;
    float wk[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) wk[j] = w[c * 9 + (FLIP ? 8 - j : j)];
    const float bv = (!FLIP && bias) ? bias[c] : 0.f;
    float acc[4] = {bv, bv, bv, bv};
    const size_t out = ((size_t)b * g.C + c) * g.H * g.W + (size_t)oy * g.W + x0;
    if (add) {                                                   // a second gradient of the same map (the block's residual): summed here
        const float4 a = *reinterpret_cast<const float4 *>(add + out);
        acc[0] = a.x; acc[1] = a.y; acc[2] = a.z; acc[3] = a.w;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        if (iy < 0 || iy >= g.H) continue;
        const float *row = xp + (size_t)iy * g.W + x0;
        const float4 m = *reinterpret_cast<const float4 *>(row);
        const float l = x0 > 0 ? row[-1] : 0.f;
        const float r = x0 + 4 < g.W ? row[4] : 0.f;
        const float v[6] = {l, m.x, m.y, m.z, m.w, r};
#pragma unroll
        for (int o = 0; o < 4; ++o)
            acc[o] += wk[3 * ky] * v[o] + wk[3 * ky + 1] * v[o + 1] + wk[3 * ky + 2] * v[o + 2];
    }
    *reinterpret_cast<float4 *>(y + out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// Stride-1 weight gradient, same 4-pixel strips: dw[ky][kx] += dy[p] * x[p + (ky-1, kx-1)], dbias += dy[p]
__global__ void __launch_bounds__(256)
dwconv_nchw_s1_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part, G g)
{
    __shared__ float red[10][4];
    const int c = blockIdx.y, b = blockIdx.z;
    const float *xp = x + ((size_t)b * g.C + c) * g.H * g.W;
    const float *dp = dy + ((size_t)b * g.C + c) * g.H * g.W;
    float gw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gb = 0.f;
    const int W4 = g.W >> 2, n = g.H * W4;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(lo + per, n);
    for (int idx = lo + threadIdx.x; idx < hi; idx += blockDim.x) {
        const int oy = idx / W4, x0 = 4 * (idx - oy * W4);
        const float4 gv = *reinterpret_cast<const float4 *>(dp + (size_t)oy * g.W + x0);
        const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
        gb += (gv.x + gv.y) + (gv.z + gv.w);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            if (iy < 0 || iy >= g.H) continue;
            const float *row = xp + (size_t)iy * g.W + x0;
            const float4 m = *reinterpret_cast<const float4 *>(row);
            const float l = x0 > 0 ? row[-1] : 0.f;
            const float r = x0 + 4 < g.W ? row[4] : 0.f;
            const float v[6] = {l, m.x, m.y, m.z, m.w, r};
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                gw[3 * ky] += gq[o] * v[o];
                gw[3 * ky + 1] += gq[o] * v[o + 1];
                gw[3 * ky + 2] += gq[o] * v[o + 2];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        float v = j < 9 ? gw[j] : gb;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[j][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        part[(((size_t)b * gridDim.x + blockIdx.x) * g.C + c) * 10 + threadIdx.x] = s;
    }
}

// one workgroup per (slice of a plane, channel, batch): part[(b * nslices + slice)][c][10]
template <int S>
__global__ void __launch_bounds__(256)
dwconv_nchw_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part, G g)
{
    const int stride = S ? S : g.stride;
    __shared__ float red[10][4];
    const int c = blockIdx.y, b = blockIdx.z;
    const float *xp = x + ((size_t)b * g.C + c) * g.H * g.W;
    const float *dp = dy + ((size_t)b * g.C + c) * g.Ho * g.Wo;
    float gw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gb = 0.f;
    const int n = g.Ho * g.Wo;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(lo + per, n);
    for (int idx = lo + threadIdx.x; idx < hi; idx += blockDim.x) {
        const int oy = idx / g.Wo, ox = idx - oy * g.Wo;
        const float gv = dp[idx];
        gb += gv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride + ky - 1;
            if (iy < 0 || iy >= g.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride + kx - 1;
                if (ix >= 0 && ix < g.W) gw[ky * 3 + kx] += gv * xp[(size_t)iy * g.W + ix];
            }
        }
    }
    // wave reduction (64 lanes), then across the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        float v = j < 9 ? gw[j] : gb;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[j][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        part[(((size_t)b * gridDim.x + blockIdx.x) * g.C + c) * 10 + threadIdx.x] = s;
    }
}

// part[rows][C][10] -> dw[C][9], dbias[C] (overwrite)
__global__ void dwconv_nchw_wgrad_reduce_kernel(const float *__restrict__ part, int rows, int C,
                                                float *__restrict__ dw, float *__restrict__ dbias)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 10) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += part[(size_t)r * C * 10 + i];
    const int c = i / 10, j = i - c * 10;
    if (j < 9) dw[c * 9 + j] = s;
    else if (dbias) dbias[c] = s;
}

int make_g(G &g, int B, int C, int H, int W, int stride)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || B > 65535 || C > 65535)
        return MLAGG_E_UNSUPPORTED;
    g.B = B; g.C = C; g.H = H; g.W = W; g.stride = stride;
    g.Ho = (H + 2 - 3) / stride + 1;
    g.Wo = (W + 2 - 3) / stride + 1;
    return 0;
}

inline int wgrad_slices(const G &g)
{
    const int n = g.Ho * g.Wo;
    int s = (n + 4095) / 4096;          // ~16 pixels per thread and workgroup
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int mlagg_dwconv3x3_nchw_fwd(const float *x, const float *w, const float *bias, float *y, int B, int C,
                                        int H, int W, int stride, void *stream)
{
    if (!x || !w || !y) return MLAGG_E_NULLPTR;
    G g;
    if (int rc = make_g(g, B, C, H, W, stride)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_DWCONV_NCHW_FWD, st);
    if (stride == 1 && (W & 3) == 0)
        hipLaunchKernelGGL(dwconv_nchw_s1_kernel<false>, dim3((H * (W >> 2) + 255) / 256, C, B), dim3(256), 0, st, x, w, bias, y, g);
    else if (stride == 2)
        hipLaunchKernelGGL(dwconv_nchw_fwd_kernel<2>, dim3((g.Ho * g.Wo + 255) / 256, C, B), dim3(256), 0, st, x, w, bias, y, g);
    else
        hipLaunchKernelGGL(dwconv_nchw_fwd_kernel<0>, dim3((g.Ho * g.Wo + 255) / 256, C, B), dim3(256), 0, st, x, w, bias, y, g);
    return (int)hipGetLastError();
}

extern "C" size_t mlagg_dwconv3x3_nchw_bwd_workspace_floats(int B, int C, int H, int W, int stride)
{
    G g;
    if (make_g(g, B, C, H, W, stride)) return 0;
    return (size_t)B * wgrad_slices(g) * C * 10;
}

extern "C" int mlagg_dwconv3x3_nchw_bwd(const float *x, const float *w, const float *dy, float *dx, float *dw,
                                        float *dbias, float *workspace, int B, int C, int H, int W, int stride,
                                        void *stream)
{
    return mlagg_dwconv3x3_nchw_bwd_res(x, w, dy, nullptr, dx, dw, dbias, workspace, B, C, H, W, stride, stream);
}

// ... with dx = data gradient + dres: the block's residual path (`x1 = x + conv3(...)`, T:256-300) hands its gradient of the same map
// to this kernel instead of to an add_ kernel behind it.  dres != NULL needs the stride-1 strip kernel (W % 4 == 0).
extern "C" int mlagg_dwconv3x3_nchw_bwd_res(const float *x, const float *w, const float *dy, const float *dres, float *dx, float *dw,
                                            float *dbias, float *workspace, int B, int C, int H, int W, int stride, void *stream)
{
    if (!x || !w || !dy || !dx || !dw || !workspace) return MLAGG_E_NULLPTR;
    G g;
    if (int rc = make_g(g, B, C, H, W, stride)) return rc;
    if (dres && !(stride == 1 && (W & 3) == 0)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        MLAGG_TIMED(K_DWCONV_NCHW_BWD, st);
        const bool strips = stride == 1 && (W & 3) == 0;
        if (strips)
            hipLaunchKernelGGL(dwconv_nchw_s1_kernel<true>, dim3((H * (W >> 2) + 255) / 256, C, B), dim3(256), 0, st, dy, w,
                               nullptr, dx, g, dres);
        else if (stride == 2)
            hipLaunchKernelGGL(dwconv_nchw_bwd_data_kernel<2>, dim3((H * W + 255) / 256, C, B), dim3(256), 0, st, dy, w, dx, g);
        else
            hipLaunchKernelGGL(dwconv_nchw_bwd_data_kernel<0>, dim3((H * W + 255) / 256, C, B), dim3(256), 0, st, dy, w, dx, g);
        const int ns = wgrad_slices(g);
        if (strips)
            hipLaunchKernelGGL(dwconv_nchw_s1_bwd_weight_kernel, dim3(ns, C, B), dim3(256), 0, st, x, dy, workspace, g);
        else if (stride == 2)
            hipLaunchKernelGGL(dwconv_nchw_bwd_weight_kernel<2>, dim3(ns, C, B), dim3(256), 0, st, x, dy, workspace, g);
        else
            hipLaunchKernelGGL(dwconv_nchw_bwd_weight_kernel<0>, dim3(ns, C, B), dim3(256), 0, st, x, dy, workspace, g);
        hipLaunchKernelGGL(dwconv_nchw_wgrad_reduce_kernel, dim3((C * 10 + 255) / 256), dim3(256), 0, st, workspace,
                           B * ns, C, dw, dbias);
    }
    return (int)hipGetLastError();
}
