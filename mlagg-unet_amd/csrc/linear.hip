// K5 -- the dense in/out projections of the token-major path on fp32 MFMA:
//   forward   y[M, N] = x[M, K] . W[N, K]^T + bias[N]          (W_NK = true : weight rows are output features)
//   backward  dx[M, N] = dy[M, K] . W[K, N]                    (W_NK = false: same weight, no transpose)
// for M = batch * tokens in the 10^4..10^5 range and N, K <= 1536: every nn.Linear of MLLABlock, Mlp,
// AggregatedAttention and SS2D_skip (reference nnUNetTrainer_MLAgg_2D_dt_MS.py:687-690, 887-907,
// MambaSkip.py:518, 538, 572-575).  (dW/db of the same layers: linear_wgrad.hip.)
//
// Shape of the problem: tall-skinny, HBM-bound on the activations (read M*K, write M*N floats; the weight
// is a few hundred KB and stays in L2).  Tiling for wave64 / v_mfma_f32_32x32x2_f32 (exact fp32):
//   * workgroup = 4 waves, output tile 128 rows x 96 columns (3 MFMA tiles per wave: 48 accumulator
//     registers), K consumed in chunks of 32 staged through LDS: A tile [128][33], B tile [32][97]
//     (odd pitches: the per-lane operand fetches A[m = lane & 31][k = lane >> 5] and
//     B[k = lane >> 5][n = lane & 31] are conflict-free ds_read_b32);
//   * the next chunk's global loads (7 float4 per lane) are issued before the 48 MFMAs of the current
//     chunk and written to LDS after them, so HBM latency hides behind the matrix pipe;
//   * rows of x / y are touched as 128-byte runs (coalesced), bias is fused into the store.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 96, KC = 32;
constexpr int AP = KC + 1;      // A tile pitch
constexpr int BPI = BN + 1;     // B tile pitch

struct LGeom {
    int M, N, K, x_stride, w_stride, y_stride;
};

template <bool W_NK>
__global__ void __launch_bounds__(256)
linear_mfma_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                   float *__restrict__ Y, LGeom g)
{
    __shared__ float sA[BM * AP];
    __shared__ float sB[KC * BPI];
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
            } else {                                        // W[k][n]: 32 rows x 24 float4 along n
                const int k = kc + idx / 24, n = n0 + 4 * (idx % 24);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K) {
                    const float *p = W + (size_t)k * g.w_stride + n;
                    if (n + 3 < g.N) v = *reinterpret_cast<const float4 *>(p);
                    else {
                        if (n < g.N) v.x = p[0];
                        if (n + 1 < g.N) v.y = p[1];
                        if (n + 2 < g.N) v.z = p[2];
                    }
                }
                rb[i] = v;
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = sA + ((tid >> 3) + 32 * i) * AP + 4 * (tid & 7);
            d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i;
            if (W_NK) {
                float *d = sB + (4 * (idx & 7)) * BPI + (idx >> 3);
                d[0] = rb[i].x; d[BPI] = rb[i].y; d[2 * BPI] = rb[i].z; d[3 * BPI] = rb[i].w;
            } else {
                float *d = sB + (idx / 24) * BPI + 4 * (idx % 24);
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
            }
        }
    };

    fetch(0);
    for (int kc = 0; kc < g.K; kc += KC) {
        __syncthreads();                     // previous chunk's operand reads are done
        stage();
        __syncthreads();
        if (kc + KC < g.K) fetch(kc + KC);   // in flight during the MFMAs below
        const float *ap = sA + (32 * wave + col) * AP + kh;
        const float *bp = sB + kh * BPI + col;
#pragma unroll
        for (int p = 0; p < KC / 2; ++p) {
            const float a = ap[2 * p];
            const float b0 = bp[(2 * p) * BPI], b1 = bp[(2 * p) * BPI + 32], b2 = bp[(2 * p) * BPI + 64];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
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
    if ((g.M + BM - 1) / BM > 2147483647 / 1 || (g.N + BN - 1) / BN > 65535) return MLAGG_E_UNSUPPORTED;
    return 0;
}

}  // namespace

extern "C" int mlagg_linear_fwd(const float *x, int x_stride, const float *w, const float *bias, float *y, int y_stride,
                                int M, int N, int K, void *stream)
{
    if (!x || !w || !y) return MLAGG_E_NULLPTR;
    LGeom g{M, N, K, x_stride, K, y_stride};
    if (int rc = check(g)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_LINEAR_FWD, st);
    hipLaunchKernelGGL(linear_mfma_kernel<true>, dim3((M + BM - 1) / BM, (N + BN - 1) / BN), dim3(256), 0, st, x, w, bias,
                       y, g);
    return (int)hipGetLastError();
}

// dx (M, I) = dy (M, O) . W (O, I): K = O, N = I
extern "C" int mlagg_linear_dgrad(const float *dy, int dy_stride, const float *w, float *dx, int dx_stride, int M, int O,
                                  int I, void *stream)
{
    if (!dy || !w || !dx) return MLAGG_E_NULLPTR;
    LGeom g{M, I, O, dy_stride, I, dx_stride};
    if (int rc = check(g)) return rc;
    if (I & 3) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_LINEAR_DGRAD, st);
    hipLaunchKernelGGL(linear_mfma_kernel<false>, dim3((M + BM - 1) / BM, (I + BN - 1) / BN), dim3(256), 0, st, dy, w,
                       nullptr, dx, g);
    return (int)hipGetLastError();
}
