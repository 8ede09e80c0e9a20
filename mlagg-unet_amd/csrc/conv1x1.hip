// K18 -- 1 x 1 convolutions on channel-major (NCHW) maps as GEMMs on the 16-bit matrix instructions with fp32 accuracy (bf16x3.h):
//   forward / data gradient   y[b][o][p] = sum_i w[o][i] x[b][i][p]            (data gradient: the same call on w^T and dy)
//   weight gradient           dW[o][i]   = sum_b sum_p dy[b][o][p] x[b][i][p]
// The expand / compress convolutions of the MedNeXt blocks, the down / up projections between the stages and the skip
// projections of the decoder (nnUNetTrainer_MLAgg_2D_dt_MS.py:279-316 MedNeXtBlock conv2 / conv3, :319-367 down / up blocks,
// :972-1001 Project): 36 convolution calls of the 256 x 256 step that MIOpen runs as fp32 GEMM kernels (1.4 ms forward, 2.9 ms
// backward per step at 50-70 TFLOP/s).
//
// No layout change and no LDS.  Forward: D rows = output channels, D columns = pixels, contraction = input channel; a lane's B
// operand is 8 consecutive input channels of ITS pixels -- eight loads it issues itself, each a T-float vector of T consecutive
// pixels (tile j of the T pixel tiles takes the pixels base + T * lane + j, as in K5w: one load per channel row feeds all T tiles,
// a half-wave reads 32 T consecutive floats of the row), its A operand 8 consecutive k of a weight row (two 16-byte loads, L2).
// The accumulator rows leave as T-float stores of consecutive pixels.  Weight gradient: the pixel is the contraction, contiguous
// in BOTH operands: a lane reads 8 consecutive pixels of its channel row (two 16-byte loads) per operand tile; split over
// (sample, pixel slab), partial blocks summed in a fixed order by a second launch (no atomics).
// One wave per workgroup; every load of the loops is unconditional (clamped addresses, duplicates dropped at the store).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mlagg_hip.h"
#include "prof.h"
#include "bf16x3.h"
#include "opmode.h"
#include "internal.h"

namespace {

using bf16x3::f32x16;

template <int T>
struct __attribute__((packed, aligned(4))) FVec {
    float v[T];
};

struct C1Geom {
    int B, O, I, P;
    long x_batch, y_batch;          // floats between consecutive samples
    int I_valid;                    // channels x really has (<= I): the weight's columns beyond are zero, the rows beyond are not read
    int accumulate;                 // y += result (a second gradient of the same map: the other convolution of a two-branch block)
};

using opmode::split8;

// y[b][o][p] (+ bias[o]); grid (pixel groups of 32 TP, channel groups of 32 TO, batch); I % 16 == 0.  DT: operand form (opmode.h)
template <int TO, int TP, int DT>
__global__ void __launch_bounds__(64)
conv1x1_fwd_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
                   C1Geom g)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    const int p0 = blockIdx.x * (32 * TP), o0 = blockIdx.y * (32 * TO), b = blockIdx.z;
    f32x16 acc[TO][TP];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;
    const int pnat = p0 + TP * col, pc = min(pnat, g.P - TP);
    const float *xb = X + (size_t)b * g.x_batch + pc;                                // + row * P
    const int lastrow = g.I_valid - 1;                                               // ragged contraction: rows past it are the last row again
    const float *wb[TO];
#pragma unroll
    for (int a = 0; a < TO; ++a) wb[a] = W + (size_t)min(o0 + 32 * a + col, g.O - 1) * g.I + 8 * kh;      // + 16 blk
    const int nblk = g.I / 16;
    float4 wa[2][TO][2];
    FVec<TP> xv[2][8];
    auto fetch = [&](float4 (&A)[TO][2], FVec<TP> (&Xv)[8], int blk) {
        const int kb = 16 * min(blk, nblk - 1);
#pragma unroll
        for (int t = 0; t < 8; ++t) Xv[t] = *reinterpret_cast<const FVec<TP> *>(xb + (size_t)min(kb + 8 * kh + t, lastrow) * g.P);
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            A[a][0] = *reinterpret_cast<const float4 *>(wb[a] + kb);
            A[a][1] = *reinterpret_cast<const float4 *>(wb[a] + kb + 4);
        }
    };
    auto consume = [&](const float4 (&A)[TO][2], const FVec<TP> (&Xv)[8]) {
        uint4 bq[TP][3];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const float f[8] = {Xv[0].v[j], Xv[1].v[j], Xv[2].v[j], Xv[3].v[j], Xv[4].v[j], Xv[5].v[j], Xv[6].v[j], Xv[7].v[j]};
            split8<DT>(f, bq[j]);
        }
        uint4 aq[TO][3];
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            const float f[8] = {A[a][0].x, A[a][0].y, A[a][0].z, A[a][0].w, A[a][1].x, A[a][1].y, A[a][1].z, A[a][1].w};
            split8<DT>(f, aq[a]);
        }
        opmode::mfma_tiles<DT, TO, TP>(aq, bq, acc);
    };
    fetch(wa[0], xv[0], 0);
    int blk = 0;
    for (; blk + 2 <= nblk; blk += 2) {                    // buffer 0 holds block blk
        fetch(wa[1], xv[1], blk + 1);
        consume(wa[0], xv[0]);
        fetch(wa[0], xv[0], blk + 2);
        consume(wa[1], xv[1]);
    }
    if (blk < nblk) consume(wa[0], xv[0]);
    // D of tile (a, j): row R = (r & 3) + 8 (r >> 2) + 4 kh -> channel o0 + 32 a + R, column = lane -> pixels pc + j
    float *yb = Y + (size_t)b * g.y_batch + pc;
#pragma unroll
    for (int a = 0; a < TO; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (o >= g.O) continue;
            const float bv = bias ? bias[o] : 0.f;
            float *dst = yb + (size_t)o * g.P;
            if (pc == pnat) {
                FVec<TP> v;
#pragma unroll
                for (int j = 0; j < TP; ++j) v.v[j] = acc[a][j][r] + bv;
                if (g.accumulate) {
                    const FVec<TP> old = *reinterpret_cast<const FVec<TP> *>(dst);
#pragma unroll
                    for (int j = 0; j < TP; ++j) v.v[j] += old.v[j];
                }
                *reinterpret_cast<FVec<TP> *>(dst) = v;
            } else {
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    if (pc + j >= pnat) dst[j] = acc[a][j][r] + bv + (g.accumulate ? dst[j] : 0.f);
            }
        }
    }
}

struct W1Geom {
    int B, O, I, P;
    long dy_batch, x_batch;
    int slab, nslabs;               // pixels per slab (multiple of 16), slabs per sample
};

// part[(b * nslabs + s)][O * I] = sum over the slab's pixels of dy[b][o][p] x[b][i][p]; grid (B * nslabs, o groups, i groups)
template <int TO, int TI, int DT>
__global__ void __launch_bounds__(64)
conv1x1_wgrad_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, W1Geom g)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    const int b = blockIdx.x / g.nslabs, s = blockIdx.x % g.nslabs;
    const int o0 = blockIdx.y * (32 * TO), i0 = blockIdx.z * (32 * TI);
    f32x16 acc[TO][TI];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;
    const int pb = s * g.slab, pe = min(pb + g.slab, g.P);
    const int nblk = (pe - pb) / 16;                       // P % 16 == 0 and slab % 16 == 0: whole blocks only
    const float *ap[TO], *bp[TI];
#pragma unroll
    for (int a = 0; a < TO; ++a) ap[a] = dY + (size_t)b * g.dy_batch + (size_t)min(o0 + 32 * a + col, g.O - 1) * g.P + pb + 8 * kh;
#pragma unroll
    for (int j = 0; j < TI; ++j) bp[j] = X + (size_t)b * g.x_batch + (size_t)min(i0 + 32 * j + col, g.I - 1) * g.P + pb + 8 * kh;
    float4 av[2][TO][2], bv[2][TI][2];
    auto fetch = [&](float4 (&A)[TO][2], float4 (&Bv)[TI][2], int blk) {
        const int off = 16 * min(blk, nblk - 1);
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            A[a][0] = *reinterpret_cast<const float4 *>(ap[a] + off);
            A[a][1] = *reinterpret_cast<const float4 *>(ap[a] + off + 4);
        }
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            Bv[j][0] = *reinterpret_cast<const float4 *>(bp[j] + off);
            Bv[j][1] = *reinterpret_cast<const float4 *>(bp[j] + off + 4);
        }
    };
    auto consume = [&](const float4 (&A)[TO][2], const float4 (&Bv)[TI][2]) {
        uint4 bq[TI][3];
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const float f[8] = {Bv[j][0].x, Bv[j][0].y, Bv[j][0].z, Bv[j][0].w, Bv[j][1].x, Bv[j][1].y, Bv[j][1].z, Bv[j][1].w};
            split8<DT>(f, bq[j]);
        }
        uint4 aq[TO][3];
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            const float f[8] = {A[a][0].x, A[a][0].y, A[a][0].z, A[a][0].w, A[a][1].x, A[a][1].y, A[a][1].z, A[a][1].w};
            split8<DT>(f, aq[a]);
        }
        opmode::mfma_tiles<DT, TO, TI>(aq, bq, acc);
    };
    if (nblk > 0) {
        fetch(av[0], bv[0], 0);
        int blk = 0;
        for (; blk + 2 <= nblk; blk += 2) {
            fetch(av[1], bv[1], blk + 1);
            consume(av[0], bv[0]);
            fetch(av[0], bv[0], blk + 2);
            consume(av[1], bv[1]);
        }
        if (blk < nblk) consume(av[0], bv[0]);
    }
    float *prow = part + (size_t)blockIdx.x * ((size_t)g.O * g.I);
#pragma unroll
    for (int a = 0; a < TO; ++a) {
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int i = i0 + 32 * j + col;
            if (i >= g.I) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (o < g.O) prow[(size_t)o * g.I + i] = acc[a][j][r];
            }
        }
    }
}

int make_wgeom(W1Geom &g, int B, int O, int I, int P, long dy_batch, long x_batch)
{
    if (B <= 0 || O <= 0 || I <= 0 || P <= 0 || (P & 15)) return MLAGG_E_UNSUPPORTED;
    if (dy_batch < (long)O * P || x_batch < (long)I * P || ((dy_batch | x_batch) & 3)) return MLAGG_E_UNSUPPORTED;
    g = W1Geom{B, O, I, P, dy_batch, x_batch, 0, 0};
    const int og = (O + 95) / 96, ig = (I + 95) / 96;
    // 1.5 waves per SIMD (1536: best of 1024 / 1536 / 2048 / 3072 / 4096 on the step's shapes) over (sample, slab, channel groups); at least 64 pixels per slab
    static const int target = [] { const char *e = getenv("MLAGG_K18_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1536; }();
    int per_sample = (target + B * og * ig - 1) / (B * og * ig);
    if (per_sample < 1) per_sample = 1;
    int slab = (P + per_sample - 1) / per_sample;
    slab = ((slab + 15) / 16) * 16;
    if (slab < 64) slab = 64;
    g.slab = slab;
    g.nslabs = (P + slab - 1) / slab;
    if ((long)B * g.nslabs > 2147483647L || og > 65535 || ig > 65535) return MLAGG_E_UNSUPPORTED;
    return 0;
}

void pick_fwd_tile(int B, int O, int I, int P, int &to, int &tp)
{
    static const char *env = getenv("MLAGG_K18_TILE");
    if (env && env[0] >= '1' && env[0] <= '3' && env[1] == ',' && env[2] >= '1' && env[2] <= '3') {
        to = env[0] - '0';
        tp = env[2] - '0';
        if (32 * (to - 1) >= O) to = (O + 31) / 32;
    }
}

template <int TO, int TP>
void launch_fwd(const float *x, const float *w, const float *bias, float *y, const C1Geom &g, int dt, hipStream_t st)
{
    const dim3 grid((g.P + 32 * TP - 1) / (32 * TP), (g.O + 32 * TO - 1) / (32 * TO), g.B);
    if (dt == MLAGG_DTYPE_BF16)
        hipLaunchKernelGGL((conv1x1_fwd_kernel<TO, TP, MLAGG_DTYPE_BF16>), grid, dim3(64), 0, st, x, w, bias, y, g);
    else if (dt == MLAGG_DTYPE_F16)
        hipLaunchKernelGGL((conv1x1_fwd_kernel<TO, TP, MLAGG_DTYPE_F16>), grid, dim3(64), 0, st, x, w, bias, y, g);
    else
        hipLaunchKernelGGL((conv1x1_fwd_kernel<TO, TP, MLAGG_DTYPE_BF16X3>), grid, dim3(64), 0, st, x, w, bias, y, g);
}

template <int TO, int TI>
void launch_wgrad(const float *dy, const float *x, float *part, const W1Geom &g, int dt, hipStream_t st)
{
    const dim3 grid(g.B * g.nslabs, (g.O + 32 * TO - 1) / (32 * TO), (g.I + 32 * TI - 1) / (32 * TI));
    if (dt == MLAGG_DTYPE_BF16)
        hipLaunchKernelGGL((conv1x1_wgrad_kernel<TO, TI, MLAGG_DTYPE_BF16>), grid, dim3(64), 0, st, dy, x, part, g);
    else if (dt == MLAGG_DTYPE_F16)
        hipLaunchKernelGGL((conv1x1_wgrad_kernel<TO, TI, MLAGG_DTYPE_F16>), grid, dim3(64), 0, st, dy, x, part, g);
    else
        hipLaunchKernelGGL((conv1x1_wgrad_kernel<TO, TI, MLAGG_DTYPE_BF16X3>), grid, dim3(64), 0, st, dy, x, part, g);
}

}  // namespace

extern "C" int mlagg_conv1x1_supported(int O, int I, long P)
{
    return O > 0 && I > 0 && (I % 16) == 0 && P >= 96 && (P % 16) == 0 && P < (1L << 30);
}

// y = w . x in the operand form `dtype` (MLAGG_DTYPE_BF16X3: the fp32 layers; MLAGG_DTYPE_BF16 / _F16: the 16-bit modes -- operands rounded
// once, one product, fp32 sums; x, w, y stay fp32 in memory)
extern "C" int mlagg_conv1x1_fwd_ragged(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B,
                                        int O, int I, int I_valid, long P, int dtype, void *stream)
{
    return mlagg_conv1x1_fwd_acc(x, x_batch, w, bias, y, y_batch, B, O, I, I_valid, P, dtype, 0, stream);
}

// ... accumulate != 0: y += w . x (+ bias) -- the data gradient of the SECOND convolution that reads a map is added to the first one's
extern "C" int mlagg_conv1x1_fwd_acc(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B,
                                     int O, int I, int I_valid, long P, int dtype, int accumulate, void *stream)
{
    if (!x || !w || !y) return MLAGG_E_NULLPTR;
    if (!opmode::valid(dtype) || I_valid <= 0 || I_valid > I) return MLAGG_E_UNSUPPORTED;
    if (B <= 0 || B > 65535 || !mlagg_conv1x1_supported(O, I, P)) return MLAGG_E_UNSUPPORTED;
    if (x_batch < (long)I_valid * P || y_batch < (long)O * P || (reinterpret_cast<uintptr_t>(w) & 15)) return MLAGG_E_UNSUPPORTED;
    C1Geom g{B, O, I, (int)P, x_batch, y_batch, I_valid, accumulate ? 1 : 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CONV1X1, st);
    // tile = (32 TO output channels) x (32 TP pixels) per wave.  MLAGG_K18_TILE="TO,TP" overrides (tuning only).
    // measured (tools/bench_conv1x1.py, MLAGG_K18_TILE sweeps): 3 x 2 tiles for wide outputs (two waves per SIMD cover the short
    // contraction loops' load latency), 3 pixel tiles when the output has at most 64 channels
    int to = O <= 32 ? 1 : (O <= 64 ? 2 : 3), tp = O <= 64 ? 3 : 2;
    pick_fwd_tile(B, O, I, (int)P, to, tp);
    switch (to * 4 + tp) {
    case 1 * 4 + 1: launch_fwd<1, 1>(x, w, bias, y, g, dtype, st); break;
    case 1 * 4 + 2: launch_fwd<1, 2>(x, w, bias, y, g, dtype, st); break;
    case 1 * 4 + 3: launch_fwd<1, 3>(x, w, bias, y, g, dtype, st); break;
    case 2 * 4 + 1: launch_fwd<2, 1>(x, w, bias, y, g, dtype, st); break;
    case 2 * 4 + 2: launch_fwd<2, 2>(x, w, bias, y, g, dtype, st); break;
    case 2 * 4 + 3: launch_fwd<2, 3>(x, w, bias, y, g, dtype, st); break;
    case 3 * 4 + 1: launch_fwd<3, 1>(x, w, bias, y, g, dtype, st); break;
    case 3 * 4 + 2: launch_fwd<3, 2>(x, w, bias, y, g, dtype, st); break;
    default: launch_fwd<3, 3>(x, w, bias, y, g, dtype, st); break;
    }
    return (int)hipGetLastError();
}

extern "C" int mlagg_conv1x1_fwd_lp(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B,
                                    int O, int I, long P, int dtype, void *stream)
{
    return mlagg_conv1x1_fwd_ragged(x, x_batch, w, bias, y, y_batch, B, O, I, I, P, dtype, stream);
}

extern "C" int mlagg_conv1x1_fwd(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B,
                                 int O, int I, long P, void *stream)
{
    return mlagg_conv1x1_fwd_lp(x, x_batch, w, bias, y, y_batch, B, O, I, P, MLAGG_DTYPE_BF16X3, stream);
}

extern "C" size_t mlagg_conv1x1_wgrad_workspace_floats(int B, int O, int I, long P)
{
    W1Geom g;
    if (P >= (1L << 30) || make_wgeom(g, B, O, I, (int)P, (long)O * P, (long)I * P)) return 0;
    return (size_t)B * g.nslabs * O * I;
}

extern "C" int mlagg_conv1x1_wgrad_lp(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B,
                                      int O, int I, long P, int dtype, void *stream)
{
    if (!dy || !x || !dW || !workspace) return MLAGG_E_NULLPTR;
    if (!opmode::valid(dtype)) return MLAGG_E_UNSUPPORTED;
    if (P >= (1L << 30)) return MLAGG_E_UNSUPPORTED;
    W1Geom g;
    if (int rc = make_wgeom(g, B, O, I, (int)P, dy_batch, x_batch)) return rc;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CONV1X1, st);
    const int to = O > 64 ? 3 : (O + 31) / 32, ti = I > 64 ? 3 : (I + 31) / 32;
    switch (to * 4 + ti) {
    case 1 * 4 + 1: launch_wgrad<1, 1>(dy, x, workspace, g, dtype, st); break;
    case 1 * 4 + 2: launch_wgrad<1, 2>(dy, x, workspace, g, dtype, st); break;
    case 1 * 4 + 3: launch_wgrad<1, 3>(dy, x, workspace, g, dtype, st); break;
    case 2 * 4 + 1: launch_wgrad<2, 1>(dy, x, workspace, g, dtype, st); break;
    case 2 * 4 + 2: launch_wgrad<2, 2>(dy, x, workspace, g, dtype, st); break;
    case 2 * 4 + 3: launch_wgrad<2, 3>(dy, x, workspace, g, dtype, st); break;
    case 3 * 4 + 1: launch_wgrad<3, 1>(dy, x, workspace, g, dtype, st); break;
    case 3 * 4 + 2: launch_wgrad<3, 2>(dy, x, workspace, g, dtype, st); break;
    default: launch_wgrad<3, 3>(dy, x, workspace, g, dtype, st); break;
    }
    const int n = O * I;
    hipLaunchKernelGGL(mlagg_internal::column_sum_kernel<false>, dim3((n + 63) / 64), dim3(1024), 0, st, workspace, B * g.nslabs, n, n,
                       dW);
    return (int)hipGetLastError();
}

extern "C" int mlagg_conv1x1_wgrad(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B,
                                   int O, int I, long P, void *stream)
{
    return mlagg_conv1x1_wgrad_lp(dy, dy_batch, x, x_batch, dW, workspace, B, O, I, P, MLAGG_DTYPE_BF16X3, stream);
}
