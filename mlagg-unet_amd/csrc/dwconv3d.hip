// K2v -- depthwise 3x3x3 convolution (+ bias, optional fused SiLU) on token-major volumes (B, D*H*W, C).
//
// Replaces the depthwise nn.Conv3d + SiLU of SS3D (reference variants/mamba/UMambaEnc_SS3D.py:166-174, 331-333), which the
// library runs with its naive reference kernels (2.9 ms per forward + backward at 2 x 96 x 24 x 40 x 40) and which needs
// the volume channel-major, i.e. a transpose each way around it.  Lane = channel quad, so every access is a contiguous run
// of channels of one token; the 27-tap halo is re-read through L1 / L2, never through HBM.
//   forward   y = act(bias + sum_j w[c][j] x[t + off_j])       act: SiLU, pre-activation saved
//   backward  g = dy * act'(pre);  dx = sum_j w[c][26 - j] g[t + off_j];  dw[c][j] = sum_{b,t} g[t] x[t + off_j];  db = sum g
// Every load is unconditional (clamped neighbour, tap weight zeroed outside the volume): guarded loads would each be
// waited for before the next is issued ("Guarded memory operations").
// HBM-bound: 8 C bytes per token forward, 20 C backward.
#include <hip/hip_runtime.h>

#include "internal.h"
#include "mlagg_hip.h"
#include "prof.h"

namespace {

struct VGeom {
    int B, D, H, W, C, x_stride, y_stride;
};

__device__ __forceinline__ float silu3_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu3_f(float x)
{
    const float s = 1.f / (1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}

constexpr int TOK = 64;            // tokens per workgroup of the gather kernel

// FLIP = false: forward; FLIP = true: data gradient (taps mirrored, no bias, no activation)
template <bool SILU, bool FLIP>
__global__ void __launch_bounds__(256)
dwconv3d_gather_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       float *__restrict__ y, float *__restrict__ pre, VGeom g)
{
    extern __shared__ float sw[];                       // [27][C]: tap-major weights of all channels
    const int C = g.C, CQ = C >> 2;
    for (int i = threadIdx.x; i < 27 * C; i += 256) {
        const int j = i / C, c = i - j * C;
        sw[i] = w[c * 27 + (FLIP ? 26 - j : j)];
    }
    __syncthreads();
    const int L = g.D * g.H * g.W, HW = g.H * g.W;
    const int b = blockIdx.y;
    const size_t base = (size_t)b * L;
    for (int i = threadIdx.x; i < TOK * CQ; i += 256) {
        const int q = i % CQ, t = blockIdx.x * TOK + i / CQ;
        if (t >= L) break;
        const int d = t / HW, r = t - d * HW, h = r / g.W, wq = r - h * g.W;
        const int c = 4 * q;
        float4 acc = (!FLIP && bias) ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int dd = d + kd - 1;
            const bool okd = dd >= 0 && dd < g.D;
            const int dc = okd ? dd : d;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hh = h + kh - 1;
                const bool okh = okd && hh >= 0 && hh < g.H;
                const int hc = okh ? hh : h;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ww = wq + kw - 1;
                    const bool ok = okh && ww >= 0 && ww < g.W;
                    const int tn = ok ? (dc * g.H + hc) * g.W + ww : t;
                    const float4 v = *reinterpret_cast<const float4 *>(x + (base + tn) * g.x_stride + c);
                    float4 wv = *reinterpret_cast<const float4 *>(sw + (kd * 9 + kh * 3 + kw) * C + c);
                    if (!ok) wv = make_float4(0.f, 0.f, 0.f, 0.f);
                    acc.x += wv.x * v.x; acc.y += wv.y * v.y; acc.z += wv.z * v.z; acc.w += wv.w * v.w;
                }
            }
        }
        if (SILU) {
            if (pre) *reinterpret_cast<float4 *>(pre + (base + t) * C + c) = acc;
            acc = make_float4(silu3_f(acc.x), silu3_f(acc.y), silu3_f(acc.z), silu3_f(acc.w));
        }
        *reinterpret_cast<float4 *>(y + (base + t) * g.y_stride + c) = acc;
    }
}

constexpr int WTOK = 256;          // tokens per workgroup of the weight-gradient kernel
constexpr int WLANES = 4;          // token lanes per channel (64 channels x 4 lanes = 256 threads)

// part[(b * chunks + chunk)][C][28]: 27 tap sums + the bias sum of one token chunk; gbuf = dy * act'(pre) for the data gradient
template <bool SILU>
__global__ void __launch_bounds__(256)
dwconv3d_wgrad_kernel(const float *__restrict__ x, int x_stride, const float *__restrict__ dy, int dy_stride,
                      const float *__restrict__ pre, float *__restrict__ part, float *__restrict__ gbuf, VGeom g)
{
    __shared__ float red[WLANES][28][64];
    const int cx = threadIdx.x & 63, ln = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx, C = g.C;
    const int L = g.D * g.H * g.W, HW = g.H * g.W;
    const int b = blockIdx.z;
    const size_t base = (size_t)b * L;
    const bool cok = c < C;
    const int cc = cok ? c : 0;
    float acc[28];
#pragma unroll
    for (int j = 0; j < 28; ++j) acc[j] = 0.f;
    const int t_end = min((int)(blockIdx.x + 1) * WTOK, L);
    for (int t = blockIdx.x * WTOK + ln; t < t_end; t += WLANES) {
        float gv = dy[(base + t) * dy_stride + cc];
        if (SILU) {
            gv *= dsilu3_f(pre[(base + t) * C + cc]);
            if (cok) gbuf[(base + t) * C + c] = gv;
        }
        const int d = t / HW, r = t - d * HW, h = r / g.W, wq = r - h * g.W;
        acc[27] += gv;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int dd = d + kd - 1;
            const bool okd = dd >= 0 && dd < g.D;
            const int dc = okd ? dd : d;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hh = h + kh - 1;
                const bool okh = okd && hh >= 0 && hh < g.H;
                const int hc = okh ? hh : h;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ww = wq + kw - 1;
                    const bool ok = okh && ww >= 0 && ww < g.W;
                    const int tn = ok ? (dc * g.H + hc) * g.W + ww : t;
                    const float v = x[(base + tn) * x_stride + cc];
                    acc[kd * 9 + kh * 3 + kw] += ok ? gv * v : 0.f;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 28; ++j) red[ln][j][cx] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 28 * 64; i += 256) {
        const int j = i >> 6, k = i & 63;
        const int ck = blockIdx.y * 64 + k;
        if (ck < C) {
            const float s = (red[0][j][k] + red[1][j][k]) + (red[2][j][k] + red[3][j][k]);
            part[(((size_t)b * gridDim.x + blockIdx.x) * C + ck) * 28 + j] = s;
        }
    }
}

// part[rows][C][28] -> dw[C][27], dbias[C]; workgroup = 64 columns x 16 row-groups, every output written once
__global__ void __launch_bounds__(1024)
dwconv3d_wgrad_reduce_kernel(const float *__restrict__ part, int rows, int C, float *__restrict__ dw, float *__restrict__ dbias)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cx, cols = C * 28;
    float s0 = 0.f, s1 = 0.f;
    if (i < cols) {
        int r = rg;
        for (; r + 16 < rows; r += 32) {
            s0 += part[(size_t)r * cols + i];
            s1 += part[(size_t)(r + 16) * cols + i];
        }
        if (r < rows) s0 += part[(size_t)r * cols + i];
    }
    red[rg][cx] = s0 + s1;
    __syncthreads();
    if (rg == 0 && i < cols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cx];
        const int c = i / 28, j = i - c * 28;
        if (j < 27) dw[c * 27 + j] = s;
        else if (dbias) dbias[c] = s;
    }
}

int make_vgeom(VGeom &g, int B, int D, int H, int W, int C, int xs, int ys)
{
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || B > 65535) return MLAGG_E_UNSUPPORTED;
    if ((C & 3) || (xs & 3) || (ys & 3) || xs < C || ys < C) return MLAGG_E_UNSUPPORTED;
    if ((long long)D * H * W > 2147483647LL / 4 || 27 * C * 4 > 64 * 1024) return MLAGG_E_UNSUPPORTED;
    g.B = B; g.D = D; g.H = H; g.W = W; g.C = C; g.x_stride = xs; g.y_stride = ys;
    return 0;
}

size_t wgrad_rows(int B, int D, int H, int W) { return (size_t)B * (((size_t)D * H * W + WTOK - 1) / WTOK); }

}  // namespace

extern "C" int mlagg_dwconv3d_fwd(const float *x, int x_stride, const float *w, const float *bias, float *y, int y_stride,
                                  float *pre, int B, int D, int H, int W, int C, int silu, void *stream)
{
    if (!x || !w || !y) return MLAGG_E_NULLPTR;
    VGeom g;
    if (int rc = make_vgeom(g, B, D, H, W, C, x_stride, y_stride)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((D * H * W + TOK - 1) / TOK, B);
    const size_t lds = (size_t)27 * C * sizeof(float);
    MLAGG_TIMED(K_DWCONV_FWD, st);
    if (silu) hipLaunchKernelGGL((dwconv3d_gather_kernel<true, false>), grid, dim3(256), lds, st, x, w, bias, y, pre, g);
    else hipLaunchKernelGGL((dwconv3d_gather_kernel<false, false>), grid, dim3(256), lds, st, x, w, bias, y, pre, g);
    return (int)hipGetLastError();
}

extern "C" size_t mlagg_dwconv3d_bwd_workspace_floats(int B, int D, int H, int W, int C)
{
    // partial rows of the weight gradient + the (B, D*H*W, C) buffer of dy * silu'(pre)
    return wgrad_rows(B, D, H, W) * (size_t)C * 28 + (size_t)B * D * H * W * C;
}

extern "C" int mlagg_dwconv3d_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride,
                                  const float *pre, float *dx, int dx_stride, float *dw, float *dbias, float *workspace,
                                  int B, int D, int H, int W, int C, int silu, void *stream)
{
    if (!x || !w || !dy || !dx || !dw || !workspace || (silu && !pre)) return MLAGG_E_NULLPTR;
    VGeom g;
    if (int rc = make_vgeom(g, B, D, H, W, C, x_stride, dx_stride)) return rc;
    if (dy_stride < C || (dy_stride & 3)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int L = D * H * W, chunks = (L + WTOK - 1) / WTOK;
    float *part = workspace;
    float *gbuf = silu ? workspace + wgrad_rows(B, D, H, W) * (size_t)C * 28 : nullptr;
    {
        MLAGG_TIMED(K_DWCONV_BWD_WEIGHT, st);
        const dim3 gridw(chunks, (C + 63) / 64, B);
        if (silu) hipLaunchKernelGGL(dwconv3d_wgrad_kernel<true>, gridw, dim3(256), 0, st, x, x_stride, dy, dy_stride, pre, part, gbuf, g);
        else hipLaunchKernelGGL(dwconv3d_wgrad_kernel<false>, gridw, dim3(256), 0, st, x, x_stride, dy, dy_stride, pre, part, gbuf, g);
        hipLaunchKernelGGL(dwconv3d_wgrad_reduce_kernel, dim3((C * 28 + 63) / 64), dim3(1024), 0, st, part, B * chunks, C, dw, dbias);
    }
    {
        MLAGG_TIMED(K_DWCONV_BWD_DATA, st);
        VGeom gd = g;                       // source = g (or dy when no SiLU), destination = dx
        gd.x_stride = silu ? C : dy_stride;
        gd.y_stride = dx_stride;
        hipLaunchKernelGGL((dwconv3d_gather_kernel<false, true>), dim3((L + TOK - 1) / TOK, B), dim3(256), (size_t)27 * C * sizeof(float),
                           st, silu ? gbuf : dy, w, nullptr, dx, nullptr, gd);
    }
    return (int)hipGetLastError();
}
