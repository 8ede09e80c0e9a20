// K5 (round 4 form) -- y = x W^T (+ bias, + epilogue) for token-major activations, fp32 in / fp32 out, products on
// v_mfma_f32_32x32x16_bf16 with every fp32 operand as three exact bf16 pieces and six partial products (csrc/bf16x3.h): the
// accuracy of the fp32 matrix instruction at 6/16 of its matrix-pipe time.
//
// What it replaces: forward and data gradient of every token-major nn.Linear of the path (MLLABlock in / act / out projections and
// Mlp, reference nnUNetTrainer_MLAgg_2D_dt_MS.py:176-192, 887-907; the q / kv / sr projections, T:687-690, 719-723; SS2D_skip's
// in / x / out projections and the gated MLP, MambaSkip.py:518, 538, 559-577) -- at EVERY token count, including the stage-2 / 3
// shapes (10 240 and 2 560 tokens) that rounds 1-3 left to rocBLAS / hipBLASLt.
//
// Differences to the round-3 kernel (linear_lp.hip MODE 2), which spent its VALU time splitting BOTH operands on the way into LDS
// behind two barriers per 32-deep k chunk (matrix pipe 48 % busy, profiles/round3_j_conv3x3_pmc_final_form.md):
//   * the WEIGHT is split once per step, not once per workgroup: `mlagg_weight_image` lays its three bf16 pieces out as
//     [piece][n][k] (k contiguous, rows padded with zeros to a multiple of 32), and -- for the data gradient -- the same for W^T.
//     The kernel copies image rows global -> LDS with 16-byte moves (no VALU), double-buffered: ONE barrier per k chunk;
//   * the ACTIVATION never touches LDS: a lane's MFMA operand is 8 consecutive k of its own row -- 32 bytes it loads itself
//     (the wave reads its 32 rows as whole 128-byte lines over the two k halves), splits in registers (44 VALU instructions per
//     16-deep step against 18 MFMAs of 32 cycles) and feeds to the matrix core.  Every element of x is split exactly once;
//   * epilogues: bias; bias + GELU writing the pre-activation AND the activation (Mlp fc1, T:188-190); multiplication by GELU'(pre)
//     for the data gradient that flows back through that GELU (the fc2 data gradient) -- the two ATen GELU passes are gone.
// Tiling: workgroup = NW waves (4 or 2), output tile 32 NW rows x 96 columns, a wave owns 32 rows x 3 column tiles (48
// accumulator VGPRs); the six terms are issued term-major over the tiles (consecutive MFMAs never depend on each other).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mlagg_hip.h"
#include "prof.h"
#include "bf16x3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // a native vector: arrays of HIP's uint4 struct land in scratch

constexpr int KC = 32;
constexpr int PITCH = KC + 8;            // 16-bit elements per LDS row: 80 bytes (16 consecutive rows cover all 64 banks)

struct X3Geom {
    int M, N, K, Kp, x_stride, y_stride, pre_stride;
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x)
{
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// EPI: 0 y = acc + bias; 1 y = acc + bias (pre-activation), y2 = GELU(y); 2 y = acc * GELU'(pre)
constexpr int APITCH = KC + 4;           // floats per LDS row of the raw activation tile: 144 bytes (16 consecutive rows: 64 banks)

// NW waves x 32 rows, TN column tiles of 32 per wave (3: N a multiple of 96 -- the stacked projections of the long token counts; 2: a
// third more workgroups per CU and no half-empty third tile for the other widths).  One register set of prefetched operands: a second
// set (chunk c + 2 in flight) was measured and gained nothing -- its 40 VGPRs cost a wave per SIMD
// (profiles/round4_d_linear_x3_variants.log)
template <int NW, int TN, int EPI>
__global__ void __launch_bounds__(64 * NW)
linear_x3_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Wimg, const float *__restrict__ bias,
                 float *__restrict__ Y, float *__restrict__ Y2, const float *__restrict__ PRE, X3Geom g)
{
    constexpr int NT = 64 * NW;
    constexpr int BN = 32 * TN;
    constexpr int NITEM = 3 * BN * 4;       // 16-byte moves of one k chunk of the weight image: 3 pieces x BN rows x 4 segments
    constexpr int NLD = (NITEM + NT - 1) / NT;
    constexpr int BM = 32 * NW;
    constexpr int NLA = BM * 8 / NT;                        // float4 of the activation tile per thread: 4
    __shared__ float sA[BM * APITCH];
    __shared__ unsigned short sB[3 * BN * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const size_t piece_elems = (size_t)g.N * g.Kp;
    // the 16-byte moves of this thread: item -> (piece, row, segment); rows past N are clamped copies (their columns are never stored)
    unsigned src_off[NLD], dst_off[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = min(tid + NT * j, NITEM - 1);
        const int piece = i / (BN * 4), rem = i - piece * (BN * 4), row = rem >> 2, seg = rem & 3;
        src_off[j] = (unsigned)(piece * piece_elems + (size_t)min(n0 + row, g.N - 1) * g.Kp + 8 * seg);       // elements
        dst_off[j] = (unsigned)(piece * BN * PITCH + row * PITCH + 8 * seg);
    }
    // activation tile: thread -> (row tid >> 3 + 32 i, float4 tid & 7): a row's 128 bytes are one coalesced run of 8 lanes
    const float *xa[NLA];
#pragma unroll
    for (int i = 0; i < NLA; ++i) xa[i] = X + (size_t)min(m0 + (tid >> 3) + (NT / 8) * i, g.M - 1) * g.x_stride + 4 * (tid & 7);
    struct Regs {
        u32x4 rb[NLD];
        float4 ra[NLA];
        bool ok;
    };
    auto fetch = [&](Regs &R, int kc) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) R.rb[j] = *reinterpret_cast<const u32x4 *>(Wimg + (size_t)src_off[j] + kc);
        // K % 4 == 0: a float4 lies inside the row or (zero-padded tail of the last chunk) outside it as a whole; outside: a valid
        // address is read and the value dropped WHEN IT IS STAGED (a select here would make the wave wait for the load at once)
        R.ok = kc + 4 * (tid & 7) < g.K;
#pragma unroll
        for (int i = 0; i < NLA; ++i) R.ra[i] = *reinterpret_cast<const float4 *>(xa[i] + (R.ok ? kc : 0));
    };
    auto stage = [&](const Regs &R) {
        // unconditional stores: the threads past the last item repeat it (same address, same value)
#pragma unroll
        for (int j = 0; j < NLD; ++j) *reinterpret_cast<u32x4 *>(&sB[dst_off[j]]) = R.rb[j];
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            *reinterpret_cast<float4 *>(sA + ((tid >> 3) + (NT / 8) * i) * APITCH + 4 * (tid & 7)) =
                make_float4(R.ok ? R.ra[i].x : 0.f, R.ok ? R.ra[i].y : 0.f, R.ok ? R.ra[i].z : 0.f, R.ok ? R.ra[i].w : 0.f);
    };
    auto multiply = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // the lane's operand: 8 consecutive k of its own row, split here (every element of x is split once, by one lane)
            const float *ap = sA + (32 * wave + col) * APITCH + 16 * p + 8 * kh;
            const float4 lo4 = *reinterpret_cast<const float4 *>(ap), hi4 = *reinterpret_cast<const float4 *>(ap + 4);
            uint4 aq[3];
            bf16x3::split3(lo4.x, lo4.y, aq[0].x, aq[1].x, aq[2].x);
            bf16x3::split3(lo4.z, lo4.w, aq[0].y, aq[1].y, aq[2].y);
            bf16x3::split3(hi4.x, hi4.y, aq[0].z, aq[1].z, aq[2].z);
            bf16x3::split3(hi4.z, hi4.w, aq[0].w, aq[1].w, aq[2].w);
            uint4 b[TN][3];
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    b[t][q] = *reinterpret_cast<const uint4 *>(sB + q * BN * PITCH + (32 * t + col) * PITCH + 16 * p + 8 * kh);
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc[t] = bf16x3::mfma(aq[bf16x3::kTermA[term]], b[t][bf16x3::kTermB[term]], acc[t]);
        }
    };

    const int nchunks = g.Kp / KC;
    Regs R;
    fetch(R, 0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                                     // the previous chunk's operand reads are done
        stage(R);
        __syncthreads();
        if (c + 1 < nchunks) fetch(R, (c + 1) * KC);         // in flight during the MFMAs below
        multiply();
    }

    const bool rows_full = m0 + 32 * NW <= g.M;
    const int mrow = m0 + 32 * wave + 4 * kh;                   // first row of this lane's accumulator registers
    float *yb = Y + (size_t)mrow * g.y_stride;
    float *y2b = EPI == 1 ? Y2 + (size_t)mrow * g.y_stride : nullptr;
    const float *pb = EPI == 2 ? PRE + (size_t)mrow * g.pre_stride : nullptr;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n = n0 + 32 * t + col;
        if (n0 + 32 * t >= g.N) break;
        const float bv = (EPI != 2 && bias) ? bias[min(n, g.N - 1)] : 0.f;
        if (n < g.N) {
            if (rows_full) {
                float pre[16];
                if (EPI == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pre[r] = pb[((r & 3) + 8 * (r >> 2)) * g.pre_stride + n];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = ((r & 3) + 8 * (r >> 2)) * g.y_stride + n;
                    const float v = acc[t][r] + bv;
                    if (EPI == 2) {
                        yb[ro] = acc[t][r] * gelu_grad_f(pre[r]);
                    } else {
                        yb[ro] = v;
                        if (EPI == 1) y2b[ro] = gelu_f(v);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (mrow + dr < g.M) {
                        const float v = acc[t][r] + bv;
                        if (EPI == 2) {
                            yb[dr * g.y_stride + n] = acc[t][r] * gelu_grad_f(pb[dr * g.pre_stride + n]);
                        } else {
                            yb[dr * g.y_stride + n] = v;
                            if (EPI == 1) y2b[dr * g.y_stride + n] = gelu_f(v);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight images.  One workgroup per 32 x 32 tile of W (N, K): img[piece][n][k] (pitch Kp) and, when asked for, the image of W^T,
// imgT[piece][k][n] (pitch Np) through an LDS transpose; padding columns are written as zeros.
// ------------------------------------------------------------------------------------------
struct ImgJob {
    const float *w;
    unsigned short *img;       // (3, N, Kp) or NULL
    unsigned short *imgT;      // (3, K, Np) or NULL
    int N, K, w_stride, pad;
};

__device__ __forceinline__ void image_tile(const ImgJob &jb, int tile)
{
    __shared__ unsigned short tl[3][32][34];
    const int N = jb.N, K = jb.K;
    const int Kp = (K + 31) & ~31, Np = (N + 31) & ~31;
    const int tk = Kp / 32;
    const int n0 = (tile / tk) * 32, k0 = (tile % tk) * 32;
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const int n = n0 + r, k = k0 + c4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < K) v[j] = jb.w[(size_t)n * jb.w_stride + k + j];
    }
    unsigned h0, m0, l0, h1, m1, l1;
    bf16x3::split3(v[0], v[1], h0, m0, l0);
    bf16x3::split3(v[2], v[3], h1, m1, l1);
    const unsigned pc[3][2] = {{h0, h1}, {m0, m1}, {l0, l1}};
    if (jb.img && n < N) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            *reinterpret_cast<uint2 *>(jb.img + ((size_t)q * N + n) * Kp + k) = make_uint2(pc[q][0], pc[q][1]);
    }
    if (jb.imgT) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            tl[q][r][c4] = (unsigned short)(pc[q][0] & 0xffffu);
            tl[q][r][c4 + 1] = (unsigned short)(pc[q][0] >> 16);
            tl[q][r][c4 + 2] = (unsigned short)(pc[q][1] & 0xffffu);
            tl[q][r][c4 + 3] = (unsigned short)(pc[q][1] >> 16);
        }
        __syncthreads();
        // thread (r, c4) now owns row k0 + r of W^T, columns n0 + c4 .. + 3 (zeros beyond N: the tile rows n >= N hold splits of 0)
        const int kk = k0 + r;
        if (kk < K) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned a = tl[q][c4][r] | ((unsigned)tl[q][c4 + 1][r] << 16);
                const unsigned b = tl[q][c4 + 2][r] | ((unsigned)tl[q][c4 + 3][r] << 16);
                *reinterpret_cast<uint2 *>(jb.imgT + ((size_t)q * K + kk) * Np + n0 + c4) = make_uint2(a, b);
            }
        }
    }
}

__global__ void __launch_bounds__(256) weight_image_kernel(ImgJob jb)
{
    image_tile(jb, blockIdx.x);
}

// every weight of a network in one launch: blockIdx.y = job, blockIdx.x = tile (jobs with fewer tiles leave early)
__global__ void __launch_bounds__(256) weight_images_kernel(const ImgJob *__restrict__ jobs)
{
    const ImgJob jb = jobs[blockIdx.y];
    const int tiles = (((jb.K + 31) & ~31) / 32) * ((jb.N + 31) / 32);
    if ((int)blockIdx.x >= tiles) return;               // uniform per workgroup
    image_tile(jb, blockIdx.x);
}

template <int NW, int TN>
int launch_x3(const float *x, const unsigned short *img, const float *bias, float *y, float *y2, const float *pre, const X3Geom &g,
              int epilogue, hipStream_t st)
{
    const dim3 grid((g.M + 32 * NW - 1) / (32 * NW), (g.N + 32 * TN - 1) / (32 * TN)), block(64 * NW);
    if (grid.y > 65535) return MLAGG_E_UNSUPPORTED;
    switch (epilogue) {
    case 0: hipLaunchKernelGGL((linear_x3_kernel<NW, TN, 0>), grid, block, 0, st, x, img, bias, y, y2, pre, g); break;
    case 1: hipLaunchKernelGGL((linear_x3_kernel<NW, TN, 1>), grid, block, 0, st, x, img, bias, y, y2, pre, g); break;
    case 2: hipLaunchKernelGGL((linear_x3_kernel<NW, TN, 2>), grid, block, 0, st, x, img, bias, y, y2, pre, g); break;
    default: return MLAGG_E_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t mlagg_weight_image_bytes(int rows, int cols)
{
    return (size_t)3 * rows * ((cols + 31) & ~31) * sizeof(unsigned short);
}

extern "C" int mlagg_weight_image(const float *w, int w_stride, void *img, void *img_t, int N, int K, void *stream)
{
    if (!w || (!img && !img_t)) return MLAGG_E_NULLPTR;
    if (N <= 0 || K <= 0 || w_stride < K) return MLAGG_E_UNSUPPORTED;
    ImgJob jb{w, static_cast<unsigned short *>(img), static_cast<unsigned short *>(img_t), N, K, w_stride, 0};
    const int tiles = (((K + 31) & ~31) / 32) * ((N + 31) / 32);
    hipLaunchKernelGGL(weight_image_kernel, dim3(tiles), dim3(256), 0, static_cast<hipStream_t>(stream), jb);
    return (int)hipGetLastError();
}

extern "C" int mlagg_weight_images(const void *jobs, int n_jobs, int max_tiles, void *stream)
{
    if (!jobs) return MLAGG_E_NULLPTR;
    if (n_jobs <= 0 || n_jobs > 65535 || max_tiles <= 0) return MLAGG_E_UNSUPPORTED;
    hipLaunchKernelGGL(weight_images_kernel, dim3(max_tiles, n_jobs), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const ImgJob *>(jobs));
    return (int)hipGetLastError();
}

extern "C" int mlagg_linear_x3_supported(int M, int N, int K)
{
    return M > 0 && N > 0 && K > 0 && (K & 7) == 0 && (size_t)3 * N * ((K + 31) & ~31) < (1ull << 32);
}

extern "C" int mlagg_linear_x3(const float *x, int x_stride, const void *w_image, const float *bias, float *y, int y_stride,
                               float *y_act, const float *pre, int pre_stride, int M, int N, int K, int epilogue, void *stream)
{
    if (!x || !w_image || !y) return MLAGG_E_NULLPTR;
    if (!mlagg_linear_x3_supported(M, N, K) || (x_stride & 3) || x_stride < K || y_stride < N) return MLAGG_E_UNSUPPORTED;
    if ((epilogue == 1 && !y_act) || (epilogue == 2 && (!pre || pre_stride < N))) return MLAGG_E_NULLPTR;
    X3Geom g{M, N, K, (K + 31) & ~31, x_stride, y_stride, pre_stride};
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_LINEAR_FWD, st);
    const unsigned short *img = static_cast<const unsigned short *>(w_image);
    // tile shape (profiles/round4_d_linear_x3_variants.log): 96-column tiles only where N is a multiple of 96 AND the grid is long
    // (stage 0 / 1 projections: >= 1024 workgroups); everywhere else 64-column tiles -- a third more workgroups per CU, 10-15 % faster
    // on the 10 240- and 2 560-token shapes; 64-row workgroups for the pooled branch's few hundred rows.
    // MLAGG_X3_TILE=<NW><TN> forces one variant (benchmarks)
    static const int forced = [] { const char *e = getenv("MLAGG_X3_TILE"); return e ? atoi(e) : 0; }();
    int nw = M <= 1024 ? 2 : 4, tn = 2;
    if (N % 96 == 0 && (long)((M + 127) / 128) * (N / 96) >= 1024) tn = 3;
    if (forced) { nw = forced / 10; tn = forced % 10; }
    if (nw == 4 && tn == 3) return launch_x3<4, 3>(x, img, bias, y, y_act, pre, g, epilogue, st);
    if (nw == 4 && tn == 2) return launch_x3<4, 2>(x, img, bias, y, y_act, pre, g, epilogue, st);
    if (nw == 2 && tn == 3) return launch_x3<2, 3>(x, img, bias, y, y_act, pre, g, epilogue, st);
    if (nw == 2 && tn == 2) return launch_x3<2, 2>(x, img, bias, y, y_act, pre, g, epilogue, st);
    return MLAGG_E_UNSUPPORTED;
}
