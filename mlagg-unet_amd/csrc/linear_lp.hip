// K5 in mixed precision -- the same projections as linear.hip (forward y = x W^T + b, backward dx = dy W), for the
// reference's DEFAULT train step: nnUNetTrainer.train_step runs the network under autocast (nnUNetTrainer.py:848),
// where every nn.Linear multiplies fp16 (or, BASELINE configs[2], bf16) operands and accumulates in fp32.
//
// Here: activations and weights STAY fp32 in HBM (every other kernel of the path consumes fp32); the tile loads
// round them to bf16 / fp16 on their way into LDS and the products run on v_mfma_f32_32x32x16_{bf16,f16} (16x the rate
// of the f32 MFMA), accumulating in fp32.  The result equals autocast's for these layers up to its own rounding of the
// OUTPUT to 16 bits, which this path does not do.  With the matrix pipe 16x faster the kernel is a pure stream of
// x and y rows: HBM-bound by 4 (K + N) bytes per token.
//
// MODE 2 (MLAGG_DTYPE_BF16X3) is the fp32 layers' form of the same kernel: every fp32 operand is split on its way into LDS into three
// bf16 pieces x = hi + mid + lo (8 + 8 + 8 significand bits: the split is exact up to 2^-24 |x|), and the product is the six
// partial products whose weight is >= 2^-16: hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi -- each bf16 x bf16 product is exact in
// fp32, the matrix core accumulates in fp32.  The dropped terms (mid.lo, lo.mid, lo.lo) are 2^-24 of the product, below the
// rounding of an fp32 FMA chain: against float64 the result is as close as rocBLAS's / K5's fp32 GEMM (tests/test_blocks_gpu.py
// holds both to the same bound).  Six 32x32x16 bf16 MFMAs replace the eight 32x32x2 f32 MFMAs of a 16-deep k block at 1/16 of
// their cost each: the matrix-pipe time of K5 drops 2.7x, and the fp32 projections stop being matrix-bound.
//
// Tiling (wave64): workgroup = 4 waves, output tile 128 rows x 96 columns, one wave per 32 rows, 3 MFMA column tiles
// per wave (48 accumulator VGPRs); K in chunks of 32: A tile [128][32] and B tile [96][32] as 16-bit, k contiguous,
// row pitch 80 bytes (the 16-byte operand fetch of 16 consecutive rows then covers 64 distinct banks).  A lane's MFMA
// operand is 8 consecutive k of one row: one ds_read_b128.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "mlagg_hip.h"
#include "prof.h"
#include "bf16x3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 96, KC = 32;
constexpr int PITCH = KC + 8;        // 16-bit elements per tile row: 80 bytes

struct LGeom {
    int M, N, K, x_stride, w_stride, y_stride;
};

// two fp32 -> one dword of two 16-bit values (round to nearest even), low half = first value
template <bool BF16>
__device__ __forceinline__ unsigned pack2(float a, float b)
{
    if (BF16) {
        const __hip_bfloat162 v = __float22bfloat162_rn(make_float2(a, b));
        return *reinterpret_cast<const unsigned *>(&v);
    }
    const __half2 v = __floats2half2_rn(a, b);
    return *reinterpret_cast<const unsigned *>(&v);
}

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(const uint4 &a, const uint4 &b, f32x16 c)
{
    if (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a), *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
}

using bf16x3::split3;

// MODE: 0 fp16, 1 bf16 (operands rounded once: the 16-bit modes), 2 bf16 x 3 (fp32-accurate, see the header)
template <bool W_NK, int MODE>
__global__ void __launch_bounds__(256)
linear_lp_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                 float *__restrict__ Y, LGeom g)
{
    constexpr bool BF16 = MODE != 0;
    constexpr int NP = MODE == 2 ? 3 : 1;                   // operand images per tile
    __shared__ unsigned short sA[NP * BM * PITCH];
    __shared__ unsigned short sB[NP * BN * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 ra[4], rb[3];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // A: 128 rows x 8 float4
            const int row = (tid >> 3) + 32 * i, c4 = tid & 7;
            const int m = m0 + row, k = kc + 4 * c4;
            ra[i] = (m < g.M && k < g.K) ? *reinterpret_cast<const float4 *>(X + (size_t)m * g.x_stride + k)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i;
            if (W_NK) {                                     // W[n][k]: 96 rows x 8 float4 along k
                const int n = n0 + (idx >> 3), k = kc + 4 * (idx & 7);
                rb[i] = (n < g.N && k < g.K) ? *reinterpret_cast<const float4 *>(W + (size_t)n * g.w_stride + k)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                // W[k][n]: the image wanted in LDS is [n][k] with k contiguous, so a lane takes FOUR k of ONE column n (4 scalar
                // loads, each coalesced across the lanes' consecutive n) and writes them as one 8-byte LDS store, like the W_NK
                // case -- a float4 along n would have to be scattered as four 2-byte stores per operand image
                const int n = n0 + idx % BN, k = kc + 4 * (idx / BN);
                const int nn = min(n, g.N - 1);             // clamped, unconditional loads; the value is dropped below
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = W[(size_t)min(k + j, g.K - 1) * g.w_stride + nn];
                const bool nok = n < g.N;
                rb[i] = make_float4(nok && k < g.K ? e[0] : 0.f, nok && k + 1 < g.K ? e[1] : 0.f, nok && k + 2 < g.K ? e[2] : 0.f,
                                    nok && k + 3 < g.K ? e[3] : 0.f);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned short *d = sA + ((tid >> 3) + 32 * i) * PITCH + 4 * (tid & 7);
            if (MODE == 2) {
                unsigned h0, m0_, l0, h1, m1, l1;
                split3(ra[i].x, ra[i].y, h0, m0_, l0);
                split3(ra[i].z, ra[i].w, h1, m1, l1);
                *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(d + BM * PITCH) = make_uint2(m0_, m1);
                *reinterpret_cast<uint2 *>(d + 2 * BM * PITCH) = make_uint2(l0, l1);
            } else {
                *reinterpret_cast<uint2 *>(d) = make_uint2(pack2<BF16>(ra[i].x, ra[i].y), pack2<BF16>(ra[i].z, ra[i].w));
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i;
            unsigned pl[NP], ph[NP];                        // per image: values (x, y) and (z, w) of the float4
            if (MODE == 2) {
                split3(rb[i].x, rb[i].y, pl[0], pl[NP > 1 ? 1 : 0], pl[NP > 2 ? 2 : 0]);
                split3(rb[i].z, rb[i].w, ph[0], ph[NP > 1 ? 1 : 0], ph[NP > 2 ? 2 : 0]);
            } else {
                pl[0] = pack2<BF16>(rb[i].x, rb[i].y);
                ph[0] = pack2<BF16>(rb[i].z, rb[i].w);
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                // rows n, k contiguous (W_NK: as loaded; otherwise the lane's four k of column idx % BN)
                unsigned short *d = W_NK ? sB + q * BN * PITCH + (idx >> 3) * PITCH + 4 * (idx & 7)
                                         : sB + q * BN * PITCH + (idx % BN) * PITCH + 4 * (idx / BN);
                *reinterpret_cast<uint2 *>(d) = make_uint2(pl[q], ph[q]);
            }
        }
    };

    fetch(0);
    for (int kc = 0; kc < g.K; kc += KC) {
        __syncthreads();                     // previous chunk's operand reads are done
        stage();
        __syncthreads();
        if (kc + KC < g.K) fetch(kc + KC);   // in flight during the MFMAs below
#pragma unroll
        for (int p = 0; p < KC / 16; ++p) {
            // lane (col, kh): row 32 wave + col of A / column 32 t + col of B, k = 16 p + 8 kh .. + 8
            uint4 a[NP], b[3][NP];
#pragma unroll
            for (int q = 0; q < NP; ++q)
                a[q] = *reinterpret_cast<const uint4 *>(sA + q * BM * PITCH + (32 * wave + col) * PITCH + 16 * p + 8 * kh);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    b[t][q] = *reinterpret_cast<const uint4 *>(sB + q * BN * PITCH + (32 * t + col) * PITCH + 16 * p + 8 * kh);
            if (MODE == 2) {
                // the six partial products, smallest first, term-major over the three column tiles: consecutive MFMAs are independent
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        acc[t] = mfma16<true>(a[bf16x3::kTermA[term] % NP], b[t][bf16x3::kTermB[term] % NP], acc[t]);
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = mfma16<BF16>(a[0], b[t][0], acc[t]);
            }
        }
    }
    // epilogue: D[row = (r & 3) + 8 * (r >> 2) + 4 * kh][col]; 128-byte row runs per store instruction
    // Stores must not sit in per-row `if (m < M)` blocks: each exec-masked block gets an `s_waitcnt vmcnt(0)` in front of its
    // store from the compiler, and the 48 stores of a tile then complete ONE AFTER THE OTHER (a quarter of the kernel's time
    // at stage 0).  Rows: a uniform test (every shape of the model has M % 128 == 0); columns: one lane mask around the 16
    // stores of a column block.
    const bool rows_full = m0 + BM <= g.M;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = n0 + 32 * t + col;
        if (n0 + 32 * t >= g.N) break;
        const float bv = bias ? bias[min(n, g.N - 1)] : 0.f;
        float *yp = Y + (size_t)(m0 + 32 * wave + 4 * kh) * g.y_stride + n;
        if (rows_full) {
            if (n < g.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * g.y_stride] = acc[t][r] + bv;
            }
        } else if (n < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < g.M) Y[(size_t)m * g.y_stride + n] = acc[t][r] + bv;
            }
        }
    }
}

int check(const LGeom &g)
{
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return MLAGG_E_UNSUPPORTED;
    if ((g.K & 3) || (g.x_stride & 3) || g.x_stride < g.K || g.y_stride < g.N) return MLAGG_E_UNSUPPORTED;
    if ((g.N + BN - 1) / BN > 65535) return MLAGG_E_UNSUPPORTED;
    return 0;
}

template <bool W_NK>
int launch(const float *x, const float *w, const float *bias, float *y, const LGeom &g, int dtype, hipStream_t st)
{
    const dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN), block(256);
    if (dtype == MLAGG_DTYPE_BF16)
        hipLaunchKernelGGL((linear_lp_kernel<W_NK, 1>), grid, block, 0, st, x, w, bias, y, g);
    else if (dtype == MLAGG_DTYPE_F16)
        hipLaunchKernelGGL((linear_lp_kernel<W_NK, 0>), grid, block, 0, st, x, w, bias, y, g);
    else if (dtype == MLAGG_DTYPE_BF16X3)
        hipLaunchKernelGGL((linear_lp_kernel<W_NK, 2>), grid, block, 0, st, x, w, bias, y, g);
    else
        return MLAGG_E_UNSUPPORTED;
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mlagg_linear_lp_fwd(const float *x, int x_stride, const float *w, const float *bias, float *y, int y_stride,
                                   int M, int N, int K, int dtype, void *stream)
{
    if (!x || !w || !y) return MLAGG_E_NULLPTR;
    LGeom g{M, N, K, x_stride, K, y_stride};
    if (int rc = check(g)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_LINEAR_FWD, st);
    return launch<true>(x, w, bias, y, g, dtype, st);
}

// dx (M, I) = dy (M, O) . W (O, I): K = O, N = I
extern "C" int mlagg_linear_lp_dgrad(const float *dy, int dy_stride, const float *w, float *dx, int dx_stride, int M, int O,
                                     int I, int dtype, void *stream)
{
    if (!dy || !w || !dx) return MLAGG_E_NULLPTR;
    LGeom g{M, I, O, dy_stride, I, dx_stride};
    if (int rc = check(g)) return rc;
    if (I & 3) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_LINEAR_DGRAD, st);
    return launch<false>(dy, w, nullptr, dx, g, dtype, st);
}
