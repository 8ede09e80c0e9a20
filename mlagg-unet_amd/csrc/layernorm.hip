// K6 -- LayerNorm over the channel dimension of token-major rows, forward and backward.
//
// Replaces ATen's layer_norm on the path's short rows (C = 48 .. 768): nn.LayerNorm in MLLABlock
// (nnUNetTrainer_MLAgg_2D_dt_MS.py:887, 907), the pooled branch (T:723), PatchEmbed's project (T:984-1001) and the
// MSMM block (MambaSkip.py:536, 741-742).  ATen launches one workgroup per row; at C = 96 and 163 840 rows
// that is 160 us forward and 310 us backward per call in the round-1 profile against ~25 / ~40 us of HBM
// time.  Here a row is owned by C/12 lanes (3 float4 per lane, so every supported C fills whole
// power-of-two lane groups: 4, 8, 16, 32 or 64 lanes), statistics are cross-lane butterflies, each wave
// streams several rows per iteration with 16-byte accesses, and the weight/bias gradients are
// accumulated per lane over the grid-stride loop, reduced through LDS to one partial row per workgroup
// and summed by a second tiny kernel (deterministic, no atomics).
//
// HBM-bound: algorithmic bytes 8*C per row forward (x in, y out), 12*C backward (x, dy in; dx out).
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"
#include "internal.h"

namespace {

constexpr int BLOCK = 256;

template <int LPR>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// RES: the residual junction in front of the norm rides in the same pass -- the row normalised is
//   xsum = x + branch * scale[row / rows_per_sample]      (scale: per-sample stochastic-depth factor, NULL = 1)
// and xsum is written next to y (x = skip; T:905-907: x = x + drop_path(...); x = x + drop_path(mlp(norm2(x)))).
template <int LPR, int F4, bool RES>
__global__ void __launch_bounds__(BLOCK)
layernorm_fwd_kernel(const float *__restrict__ x, int x_stride, const float *__restrict__ gamma,
                     const float *__restrict__ beta, float *__restrict__ y, float *__restrict__ stats, int rows,
                     float eps, const float *__restrict__ branch, const float *__restrict__ scale, float *__restrict__ xsum,
                     int rows_per_sample)
{
    constexpr int C = LPR * F4 * 4;
    constexpr int RPB = BLOCK / LPR;                   // rows per workgroup iteration
    const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    float4 g[F4], b[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        g[i] = *reinterpret_cast<const float4 *>(gamma + 4 * (sub + LPR * i));
        b[i] = beta ? *reinterpret_cast<const float4 *>(beta + 4 * (sub + LPR * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long row = (long)blockIdx.x * RPB + rl; row < rows; row += (long)gridDim.x * RPB) {
        float4 v[F4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            v[i] = *reinterpret_cast<const float4 *>(x + row * x_stride + 4 * (sub + LPR * i));
            if (RES) {
                const float sc = scale ? scale[row / rows_per_sample] : 1.f;
                const float4 br = *reinterpret_cast<const float4 *>(branch + row * C + 4 * (sub + LPR * i));
                v[i].x += br.x * sc; v[i].y += br.y * sc; v[i].z += br.z * sc; v[i].w += br.w * sc;
                *reinterpret_cast<float4 *>(xsum + row * C + 4 * (sub + LPR * i)) = v[i];
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = group_sum<LPR>(s) * (1.f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = rsqrtf(group_sum<LPR>(q) * (1.f / C) + eps);
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            float4 o;
            o.x = v[i].x * rstd * g[i].x + b[i].x; o.y = v[i].y * rstd * g[i].y + b[i].y;
            o.z = v[i].z * rstd * g[i].z + b[i].z; o.w = v[i].w * rstd * g[i].w + b[i].w;
            *reinterpret_cast<float4 *>(y + row * C + 4 * (sub + LPR * i)) = o;
        }
        if (stats && sub == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }
}

// RES: dx = LayerNorm'(dy) + dres (the gradient reaching xsum from its other consumers, NULL = none) is the gradient of the skip
// input; dbranch = dx * scale[sample] (NULL when scale is: the two gradients are then the same tensor).
template <int LPR, int F4, bool RES>
__global__ void __launch_bounds__(BLOCK)
layernorm_bwd_kernel(const float *__restrict__ x, int x_stride, const float *__restrict__ dy, int dy_stride,
                     const float *__restrict__ gamma, const float *__restrict__ stats, float *__restrict__ dx,
                     float *__restrict__ part, int rows, const float *__restrict__ dres, const float *__restrict__ scale,
                     float *__restrict__ dbranch, int rows_per_sample)
{
    constexpr int C = LPR * F4 * 4;
    constexpr int RPB = BLOCK / LPR;
    __shared__ float red[2][RPB][C + 4];
    const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    float4 g[F4], dg[F4], db[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        g[i] = *reinterpret_cast<const float4 *>(gamma + 4 * (sub + LPR * i));
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long row = (long)blockIdx.x * RPB + rl; row < rows; row += (long)gridDim.x * RPB) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        float4 xh[F4], gy[F4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + row * x_stride + 4 * (sub + LPR * i));
            const float4 d = *reinterpret_cast<const float4 *>(dy + row * dy_stride + 4 * (sub + LPR * i));
            xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            gy[i] = make_float4(d.x * g[i].x, d.y * g[i].y, d.z * g[i].z, d.w * g[i].w);
            s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
            s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
            dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
            db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        }
        const float m1 = group_sum<LPR>(s1) * (1.f / C), m2 = group_sum<LPR>(s2) * (1.f / C);
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            float4 o;
            o.x = rstd * (gy[i].x - m1 - xh[i].x * m2); o.y = rstd * (gy[i].y - m1 - xh[i].y * m2);
            o.z = rstd * (gy[i].z - m1 - xh[i].z * m2); o.w = rstd * (gy[i].w - m1 - xh[i].w * m2);
            if (RES) {
                if (dres) {
                    const float4 dr = *reinterpret_cast<const float4 *>(dres + row * C + 4 * (sub + LPR * i));
                    o.x += dr.x; o.y += dr.y; o.z += dr.z; o.w += dr.w;
                }
                if (dbranch) {
                    const float sc = scale[row / rows_per_sample];
                    *reinterpret_cast<float4 *>(dbranch + row * C + 4 * (sub + LPR * i)) = make_float4(o.x * sc, o.y * sc, o.z * sc, o.w * sc);
                }
            }
            *reinterpret_cast<float4 *>(dx + row * C + 4 * (sub + LPR * i)) = o;
        }
    }
    // per-workgroup partial of d(gamma), d(beta): sum the RPB row-lanes through LDS
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        *reinterpret_cast<float4 *>(&red[0][rl][4 * (sub + LPR * i)]) = dg[i];
        *reinterpret_cast<float4 *>(&red[1][rl][4 * (sub + LPR * i)]) = db[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += BLOCK) {
        const int which = i / C, c = i - which * C;
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < RPB; ++r) s += red[which][r][c];
        part[((size_t)blockIdx.x * 2 + which) * C + c] = s;
    }
}

// C -> (lanes per row, float4 per lane): a row is owned by a power-of-two lane group.  C = 12 * lanes (48 .. 768: the 2-D MLAgg
// network), the power-of-two and 5 * 64-multiples of the 3-D network (32 .. 640: UMambaEnc_SS3D.py features and 2x expansions)
struct RowShape { int lpr, f4; };
inline RowShape row_shape(int C)
{
    switch (C) {
    case 48: return {4, 3};
    case 96: return {8, 3};
    case 192: return {16, 3};
    case 384: return {32, 3};
    case 768: return {64, 3};
    case 32: return {8, 1};
    case 64: return {16, 1};
    case 128: return {32, 1};
    case 256: return {64, 1};
    case 512: return {64, 2};
    case 320: return {16, 5};
    case 640: return {32, 5};
    default: return {0, 0};
    }
}
inline int lanes_per_row(int C) { return row_shape(C).lpr; }

inline int grid_blocks(long rows, int lpr)
{
    const int rpb = BLOCK / lpr;
    long need = (rows + rpb - 1) / rpb;
    return (int)(need < 1024 ? need : 1024);       // grid-stride beyond 4 workgroups per CU
}

#define MLAGG_LN_DISPATCH(C, MACRO)                                                                                   \
    switch (C) {                                                                                                      \
    case 48: MACRO(4, 3); break;                                                                                      \
    case 96: MACRO(8, 3); break;                                                                                      \
    case 192: MACRO(16, 3); break;                                                                                    \
    case 384: MACRO(32, 3); break;                                                                                    \
    case 768: MACRO(64, 3); break;                                                                                    \
    case 32: MACRO(8, 1); break;                                                                                      \
    case 64: MACRO(16, 1); break;                                                                                     \
    case 128: MACRO(32, 1); break;                                                                                    \
    case 256: MACRO(64, 1); break;                                                                                    \
    case 512: MACRO(64, 2); break;                                                                                    \
    case 320: MACRO(16, 5); break;                                                                                    \
    case 640: MACRO(32, 5); break;                                                                                    \
    default: return MLAGG_E_UNSUPPORTED;                                                                              \
    }

}  // namespace

extern "C" int mlagg_layernorm_supported(int C) { return lanes_per_row(C) != 0; }

extern "C" size_t mlagg_layernorm_bwd_workspace_floats(int rows, int C)
{
    const int lpr = lanes_per_row(C);
    return lpr ? (size_t)grid_blocks(rows, lpr) * 2 * C : 0;
}

extern "C" int mlagg_layernorm_fwd(const float *x, int x_stride, const float *gamma, const float *beta, float *y,
                                   float *stats, int rows, int C, float eps, void *stream)
{
    if (!x || !gamma || !y) return MLAGG_E_NULLPTR;
    const int lpr = lanes_per_row(C);
    if (!lpr || rows <= 0 || x_stride < C || (x_stride & 3)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_blocks(rows, lpr)), block(BLOCK);
    MLAGG_TIMED(K_LAYERNORM_FWD, st);
#define MLAGG_LN_FWD(LPR, F4) hipLaunchKernelGGL((layernorm_fwd_kernel<LPR, F4, false>), grid, block, 0, st, x, x_stride, gamma, beta, y, stats, rows, eps, nullptr, nullptr, nullptr, 1)
    MLAGG_LN_DISPATCH(C, MLAGG_LN_FWD)
#undef MLAGG_LN_FWD
    return (int)hipGetLastError();
}

extern "C" int mlagg_layernorm_bwd(const float *x, int x_stride, const float *dy, int dy_stride, const float *gamma,
                                   const float *stats, float *dx, float *dgamma, float *dbeta, float *workspace,
                                   int rows, int C, void *stream)
{
    if (!x || !dy || !gamma || !stats || !dx || !dgamma || !workspace) return MLAGG_E_NULLPTR;
    const int lpr = lanes_per_row(C);
    if (!lpr || rows <= 0 || x_stride < C || (x_stride & 3) || dy_stride < C || (dy_stride & 3))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = grid_blocks(rows, lpr);
    const dim3 grid(nb), block(BLOCK);
    {
        MLAGG_TIMED(K_LAYERNORM_BWD, st);
#define MLAGG_LN_BWD(LPR, F4) hipLaunchKernelGGL((layernorm_bwd_kernel<LPR, F4, false>), grid, block, 0, st, x, x_stride, dy, dy_stride, gamma, stats, dx, workspace, rows, nullptr, nullptr, nullptr, 1)
        MLAGG_LN_DISPATCH(C, MLAGG_LN_BWD)
#undef MLAGG_LN_BWD
    }
    // partial rows are [d(gamma) | d(beta)]: column sums of an (nb x 2C) matrix
    hipLaunchKernelGGL(mlagg_internal::column_sum_split_kernel<0>, dim3((2 * C + 63) / 64), dim3(1024), 0, st, workspace, nb,
                       2 * C, dbeta ? 2 * C : C, C, dgamma, dbeta);
    return (int)hipGetLastError();
}

extern "C" int mlagg_residual_layernorm_fwd(const float *skip, const float *branch, const float *scale, const float *gamma,
                                            const float *beta, float *xsum, float *y, float *stats, int rows, int rows_per_sample,
                                            int C, float eps, void *stream)
{
    if (!skip || !branch || !gamma || !xsum || !y) return MLAGG_E_NULLPTR;
    const int lpr = lanes_per_row(C);
    if (!lpr || rows <= 0 || rows_per_sample <= 0 || rows % rows_per_sample) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_blocks(rows, lpr)), block(BLOCK);
    MLAGG_TIMED(K_LAYERNORM_FWD, st);
#define MLAGG_LN_FWD(LPR, F4) hipLaunchKernelGGL((layernorm_fwd_kernel<LPR, F4, true>), grid, block, 0, st, skip, C, gamma, beta, y, stats, rows, eps, branch, scale, xsum, rows_per_sample)
    MLAGG_LN_DISPATCH(C, MLAGG_LN_FWD)
#undef MLAGG_LN_FWD
    return (int)hipGetLastError();
}

extern "C" int mlagg_residual_layernorm_bwd(const float *xsum, const float *dy, int dy_stride, const float *dres, const float *scale,
                                            const float *gamma, const float *stats, float *dskip, float *dbranch, float *dgamma,
                                            float *dbeta, float *workspace, int rows, int rows_per_sample, int C, void *stream)
{
    if (!xsum || !dy || !gamma || !stats || !dskip || !dgamma || !workspace) return MLAGG_E_NULLPTR;
    if ((scale == nullptr) != (dbranch == nullptr)) return MLAGG_E_NULLPTR;
    const int lpr = lanes_per_row(C);
    if (!lpr || rows <= 0 || rows_per_sample <= 0 || rows % rows_per_sample || dy_stride < C || (dy_stride & 3))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = grid_blocks(rows, lpr);
    const dim3 grid(nb), block(BLOCK);
    {
        MLAGG_TIMED(K_LAYERNORM_BWD, st);
#define MLAGG_LN_BWD(LPR, F4) hipLaunchKernelGGL((layernorm_bwd_kernel<LPR, F4, true>), grid, block, 0, st, xsum, C, dy, dy_stride, gamma, stats, dskip, workspace, rows, dres, scale, dbranch, rows_per_sample)
        MLAGG_LN_DISPATCH(C, MLAGG_LN_BWD)
#undef MLAGG_LN_BWD
    }
    hipLaunchKernelGGL(mlagg_internal::column_sum_split_kernel<0>, dim3((2 * C + 63) / 64), dim3(1024), 0, st, workspace, nb,
                       2 * C, dbeta ? 2 * C : C, C, dgamma, dbeta);
    return (int)hipGetLastError();
}
