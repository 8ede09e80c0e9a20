// K2 -- depthwise 3x3 convolution (+ bias, optional fused SiLU) on token-major (B, H*W, C) maps.
//
// Replaces the NCHW depthwise Conv2d calls on the path, each of which the reference wraps in
// permute/contiguous pairs because its token mixers live in (B, N, C): MLLABlock.dwc + SiLU
// (nnUNetTrainer_MLAgg_2D_dt_MS.py:890), the pooled branch's LePE (T:781-782), SS2D_skip.conv2d + SiLU
// (MambaSkip.py:521-523) and ConvolutionalGLU.dwconv (M:553).  Lane = channel, so every load and
// store is a contiguous run of channels of one token (coalesced for any C that is a multiple of 4);
// the 3x3 halo is re-read through L1/L2, never through HBM.
//
// HBM-bound: algorithmic bytes 8 * C * N per image forward (read x, write y).
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"
#include "internal.h"

namespace {

struct Geom {
    int batch, H, W, C, x_stride, y_stride;
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x)
{
    const float s = 1.f / (1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}

// LDS-tiled 3x3 gather, shared by the forward (FLIP = false: y = bias + sum_j w[j] x[t + off_j], optional
// SiLU with the pre-activation saved) and the data gradient (FLIP = true: dx = sum_j w[8 - j] g[t + off_j]).
// A workgroup owns an 8 x 16 token tile and 32 channels: the 10 x 18 halo is read from HBM once as
// 128-byte channel runs into LDS, each output float4 is then 9 conflict-free ds_read_b128 (two tokens of
// 8 lanes cover one 256-byte bank row).  Without the tile the 9 taps are 9 L2 round trips per output.
constexpr int TY = 8, TX = 16, CCH = 32;
constexpr int HY = TY + 2, HX = TX + 2;

template <bool SILU, bool FLIP>
__global__ void __launch_bounds__(256)
dwconv_tiled_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                    const float *__restrict__ res, float *__restrict__ y, float *__restrict__ pre, Geom g,
                    const float *__restrict__ gate = nullptr, int gate_stride = 0)
{
    __shared__ float4 tile[HY * HX * (CCH / 4)];
    const int tiles_x = (g.W + TX - 1) / TX;
    const int y0 = (blockIdx.x / tiles_x) * TY, x0 = (blockIdx.x % tiles_x) * TX;
    const int c0 = blockIdx.y * CCH, b = blockIdx.z;
    const int N = g.H * g.W;
    const int c4 = threadIdx.x & 7, c = c0 + 4 * c4;
    const bool cok = c < g.C;
    // halo: 180 tokens x 8 float4 = 1440 float4, 6 per thread, ALL requested before the first LDS write and unconditional
    // (clamped token / channel, value dropped): a guarded load per trip is waited for before the next one is issued
    constexpr int NH = (HY * HX * 8 + 255) / 256;
    float4 hv[NH];
    const int cs = cok ? c : 0;
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int hl = min(i >> 3, HY * HX - 1), hy = hl / HX, hx = hl - hy * HX;
        const int yy = min(max(y0 + hy - 1, 0), g.H - 1), xx = min(max(x0 + hx - 1, 0), g.W - 1);
        hv[u] = *reinterpret_cast<const float4 *>(x + ((size_t)b * N + (size_t)yy * g.W + xx) * g.x_stride + cs);
    }
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < HY * HX * 8) {
            const int hl = i >> 3, hy = hl / HX, hx = hl - hy * HX;
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
            const bool ok = cok && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            tile[hl * 8 + c4] = ok ? hv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float wr[4][9];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 9; ++j) wr[e][j] = cok ? w[(c + e) * 9 + (FLIP ? 8 - j : j)] : 0.f;
    const float4 bv = (!FLIP && bias && cok) ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (TY * TX * 8) / 256; ++it) {
        const int tl = (threadIdx.x >> 3) + 32 * it, ty = tl / TX, tx = tl - ty * TX;
        const int yy = y0 + ty, xx = x0 + tx;
        if (!cok || yy >= g.H || xx >= g.W) continue;
        float4 acc = bv;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float4 v = tile[((ty + j / 3) * HX + tx + j % 3) * 8 + c4];
            acc.x += wr[0][j] * v.x; acc.y += wr[1][j] * v.y; acc.z += wr[2][j] * v.z; acc.w += wr[3][j] * v.w;
        }
        const size_t t = (size_t)b * N + (size_t)yy * g.W + xx;
        if (!SILU && !FLIP && res) {          // y = conv(x) + bias + res: the sum with the attention output (T:782)
            const float4 rv = *reinterpret_cast<const float4 *>(res + t * g.C + c);
            acc.x += rv.x; acc.y += rv.y; acc.z += rv.z; acc.w += rv.w;
        }
        if (SILU) {
            if (pre) *reinterpret_cast<float4 *>(pre + t * g.C + c) = acc;
            acc = make_float4(silu_f(acc.x), silu_f(acc.y), silu_f(acc.z), silu_f(acc.w));
            if (gate) {                        // y = SiLU(conv(x)) * gate: the product of ConvolutionalGLU (MambaSkip.py:575) in the same pass
                const float4 q = *reinterpret_cast<const float4 *>(gate + t * gate_stride + c);
                acc.x *= q.x; acc.y *= q.y; acc.z *= q.z; acc.w *= q.w;
            }
        }
        *reinterpret_cast<float4 *>(y + t * g.y_stride + c) = acc;
    }
}

// dw[c][j] = sum_{b,t} g[t] x[t + off_j], dbias[c] = sum g[t].  Workgroup = (batch b, image row y,
// 64 channels); lane = channel (coalesced), the 4 waves split the row into 4 column segments.  A lane
// walks its segment with a sliding 3x3 register window, so each step loads 3 new x values and 1 g
// instead of 10.  LDS-reduced over the 4 segments, one partial row per workgroup, summed by
// dwconv_wgrad_reduce_kernel (deterministic, no atomics).
template <bool SILU>
__global__ void __launch_bounds__(256)
dwconv_bwd_weight_kernel(const float *__restrict__ x, int x_stride, const float *__restrict__ dy, int dy_stride,
                         const float *__restrict__ pre, float *__restrict__ part, float *__restrict__ gbuf, int batch,
                         int H, int W, int C, const float *__restrict__ gate = nullptr, int gate_stride = 0,
                         float *__restrict__ dgate = nullptr, int dgate_stride = 0)
{
    __shared__ float red[4][10][64];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    const int y = blockIdx.x, b = blockIdx.z;
    const size_t N = (size_t)H * W;
    const int seg = (W + 3) / 4;
    const int x_begin = ph * seg, x_end = min(x_begin + seg, W);
    float gw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gb = 0.f;
    if (c < C && x_begin < x_end) {
        const float *xb = x + (size_t)b * N * x_stride + c;
        const bool up = y > 0, dn = y + 1 < H;
        // window columns: l (x-1), m (x), r (x+1) for rows y-1, y, y+1
        float l[3], m[3], r[3];
        auto col = [&](int xx, float (&v)[3]) {
            const bool in = xx >= 0 && xx < W;
            v[0] = (in && up) ? xb[((size_t)(y - 1) * W + xx) * x_stride] : 0.f;
            v[1] = in ? xb[((size_t)y * W + xx) * x_stride] : 0.f;
            v[2] = (in && dn) ? xb[((size_t)(y + 1) * W + xx) * x_stride] : 0.f;
        };
        col(x_begin - 1, l);
        col(x_begin, m);
        // 4 columns per iteration: the 12 x loads and 4 (or 8) gradient loads of the group are issued together before
        // any arithmetic (the walk is latency-bound: one channel per lane, 32 dependent steps per segment)
        constexpr int UN = 4;
        int xx = x_begin;
        for (; xx + UN <= x_end; xx += UN) {
            float rr[UN][3], gv[UN], pv[UN], qv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                col(xx + 1 + u, rr[u]);
                const size_t t = (size_t)b * N + (size_t)y * W + xx + u;
                gv[u] = dy[t * dy_stride + c];
                pv[u] = SILU ? pre[t * C + c] : 0.f;
                qv[u] = (SILU && gate) ? gate[t * gate_stride + c] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                float g1 = gv[u];
                if (SILU) {
                    // gated form: y = SiLU(pre) * gate, so d(gate) = dy * SiLU(pre) and the gradient entering the SiLU is dy * gate
                    if (dgate) dgate[((size_t)b * N + (size_t)y * W + xx + u) * dgate_stride + c] = g1 * silu_f(pv[u]);
                    g1 *= qv[u] * dsilu_f(pv[u]);
                    if (gbuf) gbuf[((size_t)b * N + (size_t)y * W + xx + u) * C + c] = g1;
                }
                gb += g1;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    gw[3 * ky + 0] += g1 * l[ky];
                    gw[3 * ky + 1] += g1 * m[ky];
                    gw[3 * ky + 2] += g1 * rr[u][ky];
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { l[ky] = m[ky]; m[ky] = rr[u][ky]; }
            }
        }
        for (; xx < x_end; ++xx) {
            col(xx + 1, r);
            const size_t t = (size_t)b * N + (size_t)y * W + xx;
            float gv = dy[t * dy_stride + c];
            if (SILU) {
                const float pv1 = pre[t * C + c];
                if (dgate) dgate[t * dgate_stride + c] = gv * silu_f(pv1);
                gv *= (gate ? gate[t * gate_stride + c] : 1.f) * dsilu_f(pv1);
                if (gbuf) gbuf[t * C + c] = gv;      // g = dy * (gate) * silu'(pre), reused by the data-gradient kernel
            }
            gb += gv;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                gw[3 * ky + 0] += gv * l[ky];
                gw[3 * ky + 1] += gv * m[ky];
                gw[3 * ky + 2] += gv * r[ky];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { l[ky] = m[ky]; m[ky] = r[ky]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) red[ph][j][cx] = gw[j];
    red[ph][9][cx] = gb;
    __syncthreads();
    for (int i = threadIdx.x; i < 640; i += 256) {
        const int j = i >> 6, cc = i & 63;
        if (blockIdx.y * 64 + cc < C) {
            const float s = red[0][j][cc] + red[1][j][cc] + red[2][j][cc] + red[3][j][cc];
            const size_t row = (size_t)b * gridDim.x + blockIdx.x;
            part[(row * C + blockIdx.y * 64 + cc) * 10 + j] = s;
        }
    }
}

// part[rows][C][10] -> dw[C][9], dbias[C] (ACC: += into what is there); workgroup = 64 columns x 16 row-groups
template <bool ACC>
__global__ void __launch_bounds__(1024)
dwconv_wgrad_reduce_kernel(const float *__restrict__ part, int rows, int C, float *__restrict__ dw,
                           float *__restrict__ dbias)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cx, cols = C * 10;
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
        const int c = i / 10, j = i - c * 10;
        if (j < 9) dw[c * 9 + j] = ACC ? dw[c * 9 + j] + s : s;
        else if (dbias) dbias[c] = ACC ? dbias[c] + s : s;
    }
}

int make_geom(Geom &g, int batch, int H, int W, int C, int xs, int ys)
{
    if (batch <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || batch > 65535) return MLAGG_E_UNSUPPORTED;
    if (xs < C || ys < C || (xs & 3) || (ys & 3)) return MLAGG_E_UNSUPPORTED;
    g.batch = batch; g.H = H; g.W = W; g.C = C; g.x_stride = xs; g.y_stride = ys;
    return 0;
}

}  // namespace

namespace mlagg_internal {
size_t dwconv_wgrad_workspace_floats(int batch, int H, int W, int C)
{
    (void)W;
    return (size_t)batch * H * C * 10;      // one partial row per (batch, image row)
}

void dwconv_wgrad_launch(const float *x, int x_stride, const float *dy, int dy_stride, const float *pre, float *dw,
                         float *dbias, float *part, int batch, int H, int W, int C, int silu, hipStream_t st,
                         float *gbuf, bool accumulate, const float *gate, int gate_stride, float *dgate, int dgate_stride)
{
    const int chunks = H;
    const dim3 grid(chunks, (C + 63) / 64, batch);
    {
        MLAGG_TIMED(K_DWCONV_BWD_WEIGHT, st);
        if (silu)
            hipLaunchKernelGGL(dwconv_bwd_weight_kernel<true>, grid, dim3(256), 0, st, x, x_stride, dy, dy_stride, pre,
                               part, gbuf, batch, H, W, C, gate, gate_stride, dgate, dgate_stride);
        else
            hipLaunchKernelGGL(dwconv_bwd_weight_kernel<false>, grid, dim3(256), 0, st, x, x_stride, dy, dy_stride, pre,
                               part, gbuf, batch, H, W, C, (const float *)nullptr, 0, (float *)nullptr, 0);
        if (accumulate)
            hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel<true>, dim3((C * 10 + 63) / 64), dim3(1024), 0, st, part,
                               batch * chunks, C, dw, dbias);
        else
            hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel<false>, dim3((C * 10 + 63) / 64), dim3(1024), 0, st, part,
                               batch * chunks, C, dw, dbias);
    }
}
}  // namespace mlagg_internal

extern "C" size_t mlagg_dwconv3x3_bwd_workspace_floats(int batch, int H, int W, int C)
{
    // partial rows of the weight gradient + the (batch, H*W, C) buffer of dy * silu'(pre)
    return mlagg_internal::dwconv_wgrad_workspace_floats(batch, H, W, C) + (size_t)batch * H * W * C;
}

namespace {
int dwconv_fwd(const float *x, int x_stride, const float *w, const float *bias, const float *res, float *y, int y_stride, float *pre,
               int batch, int H, int W, int C, int silu, const float *gate, int gate_stride, void *stream);
int dwconv_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride, const float *pre, float *dx, int dx_stride,
               float *dw, float *dbias, float *workspace, int batch, int H, int W, int C, int silu, const float *gate, int gate_stride,
               float *dgate, int dgate_stride, void *stream);
}  // namespace

extern "C" int mlagg_dwconv3x3_fwd(const float *x, int x_stride, const float *w, const float *bias, const float *res, float *y,
                                   int y_stride, float *pre, int batch, int H, int W, int C, int silu,
                                   void *stream)
{
    return dwconv_fwd(x, x_stride, w, bias, res, y, y_stride, pre, batch, H, W, C, silu, nullptr, 0, stream);
}

// y = SiLU(conv(x) + bias) * gate: the depthwise conv of ConvolutionalGLU with the GLU product in its epilogue (MambaSkip.py:559-577)
extern "C" int mlagg_dwconv3x3_gated_fwd(const float *x, int x_stride, const float *w, const float *bias, const float *gate,
                                         int gate_stride, float *y, int y_stride, float *pre, int batch, int H, int W, int C,
                                         void *stream)
{
    if (!gate || !pre) return MLAGG_E_NULLPTR;
    if (gate_stride < C || (gate_stride & 3)) return MLAGG_E_UNSUPPORTED;
    return dwconv_fwd(x, x_stride, w, bias, nullptr, y, y_stride, pre, batch, H, W, C, 1, gate, gate_stride, stream);
}

extern "C" int mlagg_dwconv3x3_gated_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride, const float *pre,
                                         const float *gate, int gate_stride, float *dx, int dx_stride, float *dgate, int dgate_stride,
                                         float *dw, float *dbias, float *workspace, int batch, int H, int W, int C, void *stream)
{
    if (!gate || !dgate || !pre) return MLAGG_E_NULLPTR;
    if (gate_stride < C || (gate_stride & 3) || dgate_stride < C) return MLAGG_E_UNSUPPORTED;
    return dwconv_bwd(x, x_stride, w, dy, dy_stride, pre, dx, dx_stride, dw, dbias, workspace, batch, H, W, C, 1, gate, gate_stride,
                      dgate, dgate_stride, stream);
}

namespace {
int dwconv_fwd(const float *x, int x_stride, const float *w, const float *bias, const float *res, float *y, int y_stride, float *pre,
               int batch, int H, int W, int C, int silu, const float *gate, int gate_stride, void *stream)
{
    if (!x || !w || !y) return MLAGG_E_NULLPTR;
    Geom g;
    if (int rc = make_geom(g, batch, H, W, C, x_stride, y_stride)) return rc;
    if (res && silu) return MLAGG_E_UNSUPPORTED;
    const dim3 grid(((W + TX - 1) / TX) * ((H + TY - 1) / TY), (C + CCH - 1) / CCH, batch), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        MLAGG_TIMED(K_DWCONV_FWD, st);
        if (silu)
            hipLaunchKernelGGL((dwconv_tiled_kernel<true, false>), grid, block, 0, st, x, w, bias, (const float *)nullptr, y, pre, g, gate,
                               gate_stride);
        else
            hipLaunchKernelGGL((dwconv_tiled_kernel<false, false>), grid, block, 0, st, x, w, bias, res, y, pre, g, (const float *)nullptr, 0);
    }
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mlagg_dwconv3x3_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride,
                                   const float *pre, float *dx, int dx_stride, float *dw, float *dbias,
                                   float *workspace, int batch, int H, int W, int C, int silu, void *stream)
{
    return dwconv_bwd(x, x_stride, w, dy, dy_stride, pre, dx, dx_stride, dw, dbias, workspace, batch, H, W, C, silu, nullptr, 0, nullptr, 0,
                      stream);
}

namespace {
int dwconv_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride, const float *pre, float *dx, int dx_stride,
               float *dw, float *dbias, float *workspace, int batch, int H, int W, int C, int silu, const float *gate, int gate_stride,
               float *dgate, int dgate_stride, void *stream)
{
    if (!x || !w || !dy || !dx || !dw || !workspace || (silu && !pre)) return MLAGG_E_NULLPTR;
    Geom g;
    if (int rc = make_geom(g, batch, H, W, C, x_stride, dx_stride)) return rc;
    if (dy_stride < C || (dy_stride & 3)) return MLAGG_E_UNSUPPORTED;
    const dim3 grid(((W + TX - 1) / TX) * ((H + TY - 1) / TY), (C + CCH - 1) / CCH, batch), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // weight gradient first: with SiLU it also emits g = dy * silu'(pre) once per element, so the data
    // gradient below is a plain 9-tap gather of g instead of 9 x (load dy, load pre, evaluate silu')
    float *gbuf = silu ? workspace + mlagg_internal::dwconv_wgrad_workspace_floats(batch, H, W, C) : nullptr;
    mlagg_internal::dwconv_wgrad_launch(x, x_stride, dy, dy_stride, pre, dw, dbias, workspace, batch, H, W, C, silu, st,
                                        gbuf, false, gate, gate_stride, dgate, dgate_stride);
    {
        MLAGG_TIMED(K_DWCONV_BWD_DATA, st);
        Geom gd = g;                       // source = g (or dy when no SiLU), destination = dx
        gd.x_stride = silu ? C : dy_stride;
        gd.y_stride = dx_stride;
        hipLaunchKernelGGL((dwconv_tiled_kernel<false, true>), grid, block, 0, st, silu ? gbuf : dy, w, (const float *)nullptr,
                           (const float *)nullptr, dx, (float *)nullptr, gd, (const float *)nullptr, 0);
    }
    return (int)hipGetLastError();
}
}  // namespace
