// Experiment for K5 ("K5 projections"): y[M, N] = x[M, K] . W[N, K]^T + bias with NO LDS and no barrier.
// Both operands of v_mfma_f32_32x32x2_f32 are "row index = lane % 32, contraction index along the row", and the contraction
// order is free as long as A and B agree, so every lane loads float4s along k straight from global memory:
// lanes 0..31 take k0 .. k0+3, lanes 32..63 take k0+4 .. k0+7, and MFMA j of a k8-step uses component j of both.
// One wave owns a (32 TM) x (32 TN) tile; waves are independent.
//   hipcc --offload-arch=gfx950 -O3 -o linear_direct tools/microbench/linear_direct.hip && ./linear_direct
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int TM, int TN, int WAVES, int OCC>
__global__ void __launch_bounds__(64 * WAVES, OCC)
linear_direct_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                     float *__restrict__ Y, int M, int N, int K, int xs, int ws, int ys)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int m0 = (blockIdx.x * WAVES + wave) * 32 * TM, n0 = blockIdx.y * 32 * TN;
    if (m0 >= M) return;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *ap[TM], *bp[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) ap[a] = X + (size_t)min(m0 + 32 * a + col, M - 1) * xs + 4 * kh;
#pragma unroll
    for (int b = 0; b < TN; ++b) bp[b] = W + (size_t)min(n0 + 32 * b + col, N - 1) * ws + 4 * kh;

    float4 a0[TM], b0[TN], a1[TM], b1[TN];
    auto load = [&](float4 (&ra)[TM], float4 (&rb)[TN], int k) {
#pragma unroll
        for (int a = 0; a < TM; ++a) ra[a] = *reinterpret_cast<const float4 *>(ap[a] + k);
#pragma unroll
        for (int b = 0; b < TN; ++b) rb[b] = *reinterpret_cast<const float4 *>(bp[b] + k);
    };
    auto mma = [&](const float4 (&ra)[TM], const float4 (&rb)[TN]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const float av = j == 0 ? ra[a].x : (j == 1 ? ra[a].y : (j == 2 ? ra[a].z : ra[a].w));
                    const float bv = j == 0 ? rb[b].x : (j == 1 ? rb[b].y : (j == 2 ? rb[b].z : rb[b].w));
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                }
    };
    load(a0, b0, 0);
    int k = 0;
    for (; k + 16 <= K; k += 16) {          // K % 8 == 0
        load(a1, b1, k + 8);
        mma(a0, b0);
        if (k + 16 < K) load(a0, b0, k + 16);
        mma(a1, b1);
    }
    if (k < K) mma(a0, b0);                 // odd number of k8 steps
    const bool full = m0 + 32 * TM <= M && n0 + 32 * TN <= N;      // uniform: unconditional stores (see K5's epilogue)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + 32 * b + col;
        if (n0 + 32 * b >= N) break;
        const float bv = bias ? bias[min(n, N - 1)] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Y[(size_t)(m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh) * ys + n] = acc[a][b][r] + bv;
            } else if (n < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (m < M) Y[(size_t)m * ys + n] = acc[a][b][r] + bv;
                }
            }
        }
    }
}

// Persistent form: a wave walks tiles t = wave id, wave id + #waves, ... and requests the first operands of the NEXT tile
// before the last MFMAs of the current one, so the start-up latency of a tile hides behind its predecessor.
// MFMA_ONLY: no loads in the loop (register operands): the matrix pipe's ceiling for this instruction mix.
template <int TM, int TN, int WAVES, int OCC, bool MFMA_ONLY>
__global__ void __launch_bounds__(64 * WAVES, OCC)
linear_persistent_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                         float *__restrict__ Y, int M, int N, int K, int xs, int ws, int ys)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int nct = (N + 32 * TN - 1) / (32 * TN), nrt = (M + 32 * TM - 1) / (32 * TM);
    const int ntiles = nct * nrt, nw = gridDim.x * WAVES;
    float4 a0[TM], b0[TN], a1[TM], b1[TN];
    const float *ap[TM], *bp[TN];
    auto point = [&](int t) {
        const int m0 = (t / nct) * 32 * TM, n0 = (t % nct) * 32 * TN;
#pragma unroll
        for (int a = 0; a < TM; ++a) ap[a] = X + (size_t)min(m0 + 32 * a + col, M - 1) * xs + 4 * kh;
#pragma unroll
        for (int b = 0; b < TN; ++b) bp[b] = W + (size_t)min(n0 + 32 * b + col, N - 1) * ws + 4 * kh;
    };
    auto load = [&](float4 (&ra)[TM], float4 (&rb)[TN], int k) {
        if (MFMA_ONLY) return;
#pragma unroll
        for (int a = 0; a < TM; ++a) ra[a] = *reinterpret_cast<const float4 *>(ap[a] + k);
#pragma unroll
        for (int b = 0; b < TN; ++b) rb[b] = *reinterpret_cast<const float4 *>(bp[b] + k);
    };
    int t = blockIdx.x * WAVES + wave;
    if (t >= ntiles) return;
    point(t);
    if (MFMA_ONLY) {
#pragma unroll
        for (int a = 0; a < TM; ++a) a0[a] = a1[a] = make_float4(lane, 1.f, 2.f, 3.f);
#pragma unroll
        for (int b = 0; b < TN; ++b) b0[b] = b1[b] = make_float4(1.f, lane, 2.f, 3.f);
    }
    load(a0, b0, 0);
    for (; t < ntiles; t += nw) {
        const int m0 = (t / nct) * 32 * TM, n0 = (t % nct) * 32 * TN;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        auto mma = [&](const float4 (&ra)[TM], const float4 (&rb)[TN]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const float av = j == 0 ? ra[a].x : (j == 1 ? ra[a].y : (j == 2 ? ra[a].z : ra[a].w));
                        const float bv = j == 0 ? rb[b].x : (j == 1 ? rb[b].y : (j == 2 ? rb[b].z : rb[b].w));
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                    }
        };
        // K % 16 == 0 in this experiment: pairs of k8 steps; the last pair prefetches the next tile's first step
        for (int k = 0; k < K; k += 16) {
            load(a1, b1, k + 8);
            mma(a0, b0);
            if (k + 16 < K) load(a0, b0, k + 16);
            else if (t + nw < ntiles) { point(t + nw); load(a0, b0, 0); }
            mma(a1, b1);
        }
        const bool full = m0 + 32 * TM <= M && n0 + 32 * TN <= N;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + 32 * b + col;
            if (n0 + 32 * b >= N) break;
            const float bv = bias ? bias[min(n, N - 1)] : 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Y[(size_t)(m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh) * ys + n] = acc[a][b][r] + bv;
                } else if (n < N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (m < M) Y[(size_t)m * ys + n] = acc[a][b][r] + bv;
                    }
                }
            }
        }
    }
}

template <int TM, int TN, int WAVES, int OCC, bool MFMA_ONLY>
float run_persistent(const float *X, const float *W, const float *b, float *Y, int M, int N, int K, int iters)
{
    const dim3 grid(256 * 4 * OCC / WAVES);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((linear_persistent_kernel<TM, TN, WAVES, OCC, MFMA_ONLY>), grid, dim3(64 * WAVES), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((linear_persistent_kernel<TM, TN, WAVES, OCC, MFMA_ONLY>), grid, dim3(64 * WAVES), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

// Column blocks one after the other: the wave owns (32 TM) rows x (32 TN) columns but computes ONE 32-column block at a
// time (TM accumulators), so the stores of block b drain while the MFMAs of block b + 1 run -- the store burst of a tile is
// TN times shorter and overlaps compute inside the wave (x is re-read TN times, from L1 / L2).
template <int TM, int TN, int OCC>
__global__ void __launch_bounds__(64, OCC)
linear_seq_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                  float *__restrict__ Y, int M, int N, int K, int xs, int ws, int ys)
{
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 32 * TM, n0 = blockIdx.y * 32 * TN;
    if (m0 >= M) return;
    const float *ap[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) ap[a] = X + (size_t)min(m0 + 32 * a + col, M - 1) * xs + 4 * kh;
    const bool full = m0 + 32 * TM <= M && n0 + 32 * TN <= N;
#pragma unroll 1
    for (int b = 0; b < TN; ++b) {
        if (n0 + 32 * b >= N) break;
        const int n = n0 + 32 * b + col;
        const float *bp = W + (size_t)min(n, N - 1) * ws + 4 * kh;
        f32x16 acc[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        float4 a0[TM], a1[TM], b0, b1;
        auto load = [&](float4 (&ra)[TM], float4 &rb, int k) {
#pragma unroll
            for (int a = 0; a < TM; ++a) ra[a] = *reinterpret_cast<const float4 *>(ap[a] + k);
            rb = *reinterpret_cast<const float4 *>(bp + k);
        };
        auto mma = [&](const float4 (&ra)[TM], const float4 &rb) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const float av = j == 0 ? ra[a].x : (j == 1 ? ra[a].y : (j == 2 ? ra[a].z : ra[a].w));
                    const float bv = j == 0 ? rb.x : (j == 1 ? rb.y : (j == 2 ? rb.z : rb.w));
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
                }
        };
        load(a0, b0, 0);
        int k = 0;
        for (; k + 16 <= K; k += 16) {
            load(a1, b1, k + 8);
            mma(a0, b0);
            if (k + 16 < K) load(a0, b0, k + 16);
            mma(a1, b1);
        }
        if (k < K) mma(a0, b0);
        const float bv = bias ? bias[min(n, N - 1)] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Y[(size_t)(m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh) * ys + n] = acc[a][r] + bv;
            } else if (n < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (m < M) Y[(size_t)m * ys + n] = acc[a][r] + bv;
                }
            }
        }
    }
}

template <int TM, int TN, int OCC>
float run_seq(const float *X, const float *W, const float *b, float *Y, int M, int N, int K, int iters)
{
    const dim3 grid((M + 32 * TM - 1) / (32 * TM), (N + 32 * TN - 1) / (32 * TN));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((linear_seq_kernel<TM, TN, OCC>), grid, dim3(64), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((linear_seq_kernel<TM, TN, OCC>), grid, dim3(64), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

// Epilogue software-pipelined into the NEXT tile's k-loop: a persistent wave keeps two accumulator sets; while the MFMAs of
// tile i + 1 run, the 96 stores of tile i are issued 16 at a time behind the first NP k16-steps.  One wave per SIMD
// (~250 registers), operands double-buffered at a distance of one k16-step (48 MFMAs).
template <int TM, int TN, int NP>
__global__ void __launch_bounds__(64, 1)
linear_overlap_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                      float *__restrict__ Y, int M, int N, int K, int xs, int ws, int ys)
{
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, kh = lane >> 5;
    const int nct = N / (32 * TN), nrt = M / (32 * TM);              // full tiles only in this experiment
    const int ntiles = nct * nrt, nw = gridDim.x;
    constexpr int NST = TM * TN * 16, SPS = (NST + NP - 1) / NP;      // stores per pipelined step
    float4 a0[2][TM], b0[2][TN], a1[2][TM], b1[2][TN];
    const float *ap[TM], *bp[TN];
    auto point = [&](int t) {
        const int m0 = (t / nct) * 32 * TM, n0 = (t % nct) * 32 * TN;
#pragma unroll
        for (int a = 0; a < TM; ++a) ap[a] = X + (size_t)(m0 + 32 * a + col) * xs + 4 * kh;
#pragma unroll
        for (int b = 0; b < TN; ++b) bp[b] = W + (size_t)(n0 + 32 * b + col) * ws + 4 * kh;
    };
    auto load = [&](float4 (&ra)[2][TM], float4 (&rb)[2][TN], int k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int a = 0; a < TM; ++a) ra[h][a] = *reinterpret_cast<const float4 *>(ap[a] + k + 8 * h);
#pragma unroll
            for (int b = 0; b < TN; ++b) rb[h][b] = *reinterpret_cast<const float4 *>(bp[b] + k + 8 * h);
        }
    };
    auto mma = [&](f32x16 (&acc)[TM][TN], const float4 (&ra)[2][TM], const float4 (&rb)[2][TN]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const float av = j == 0 ? ra[h][a].x : (j == 1 ? ra[h][a].y : (j == 2 ? ra[h][a].z : ra[h][a].w));
                        const float bv = j == 0 ? rb[h][b].x : (j == 1 ? rb[h][b].y : (j == 2 ? rb[h][b].z : rb[h][b].w));
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                    }
    };
    // stores [lo, hi) of a finished tile (flattened index (a * TN + b) * 16 + r), all compile-time
    auto store_range = [&](const f32x16 (&acc)[TM][TN], int t, const float (&bv)[TN], auto lo_c, auto hi_c) {
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        const int m0 = (t / nct) * 32 * TM, n0 = (t % nct) * 32 * TN;
#pragma unroll
        for (int i = lo; i < hi; ++i) {
            if (i >= NST) break;
            const int a = i / (TN * 16), b = (i / 16) % TN, r = i % 16;
            Y[(size_t)(m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh) * ys + n0 + 32 * b + col] = acc[a][b][r] + bv[b];
        }
    };
    f32x16 accA[TM][TN], accB[TM][TN];
    float bvA[TN], bvB[TN];
    // one tile: zero `cur`, run its k-loop; behind the first NP steps issue SPS stores of `prev` (tile tp) each
    auto tile = [&](f32x16 (&cur)[TM][TN], float (&bcur)[TN], int t, const f32x16 (&prev)[TM][TN], const float (&bprev)[TN], int tp,
                    bool have_prev, int tnext) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[a][b][r] = 0.f;
        const int n0 = (t % nct) * 32 * TN;
#pragma unroll
        for (int b = 0; b < TN; ++b) bcur[b] = bias ? bias[n0 + 32 * b + col] : 0.f;
        // operands of step 0 are already in a0 / b0 (requested by the previous tile or the prologue)
        int k = 0;
#define MLAGG_STEP(S, CURBUF_A, CURBUF_B, NXT_A, NXT_B)                                                        \
        {                                                                                                      \
            if (k + 16 < K) load(NXT_A, NXT_B, k + 16);                                                        \
            else if (tnext < ntiles) { point(tnext); load(NXT_A, NXT_B, 0); }                                  \
            mma(cur, CURBUF_A, CURBUF_B);                                                                      \
            if (have_prev) store_range(prev, tp, bprev, std::integral_constant<int, (S) * SPS>(),              \
                                       std::integral_constant<int, ((S) + 1) * SPS>());                       \
            k += 16;                                                                                           \
        }
        // NP (even) pipelined steps, buffers alternate 0 / 1
        if (NP >= 2) { MLAGG_STEP(0, a0, b0, a1, b1) MLAGG_STEP(1, a1, b1, a0, b0) }
        if (NP >= 4) { MLAGG_STEP(2, a0, b0, a1, b1) MLAGG_STEP(3, a1, b1, a0, b0) }
        if (NP >= 6) { MLAGG_STEP(4, a0, b0, a1, b1) MLAGG_STEP(5, a1, b1, a0, b0) }
#undef MLAGG_STEP
        for (; k < K; k += 32) {                    // the rest of K (multiples of 32 beyond 16 NP), no stores left
            if (k + 16 < K) load(a1, b1, k + 16);
            mma(cur, a0, b0);
            if (k + 32 < K) load(a0, b0, k + 32);
            else if (tnext < ntiles) { point(tnext); load(a0, b0, 0); }
            mma(cur, a1, b1);
        }
    };
    int t = blockIdx.x;
    if (t >= ntiles) return;
    point(t);
    load(a0, b0, 0);
    tile(accA, bvA, t, accB, bvB, 0, false, t + nw);
    while (true) {
        const int t2 = t + nw;
        if (t2 >= ntiles) { store_range(accA, t, bvA, std::integral_constant<int, 0>(), std::integral_constant<int, NST>()); break; }
        tile(accB, bvB, t2, accA, bvA, t, true, t2 + nw);
        const int t3 = t2 + nw;
        if (t3 >= ntiles) { store_range(accB, t2, bvB, std::integral_constant<int, 0>(), std::integral_constant<int, NST>()); break; }
        tile(accA, bvA, t3, accB, bvB, t2, true, t3 + nw);
        t = t3;
    }
}

template <int TM, int TN, int NP>
float run_overlap(const float *X, const float *W, const float *b, float *Y, int M, int N, int K, int iters)
{
    const dim3 grid(1024);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((linear_overlap_kernel<TM, TN, NP>), grid, dim3(64), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((linear_overlap_kernel<TM, TN, NP>), grid, dim3(64), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

__global__ void reference_kernel(const float *X, const float *W, const float *bias, float *Y, int M, int N, int K)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = bias[n];
    for (int k = 0; k < K; ++k) s += X[(size_t)m * K + k] * W[(size_t)n * K + k];
    Y[(size_t)m * N + n] = s;
}

template <int TM, int TN, int WAVES, int OCC>
float run(const float *X, const float *W, const float *b, float *Y, int M, int N, int K, int iters)
{
    const dim3 grid((M + 32 * TM * WAVES - 1) / (32 * TM * WAVES), (N + 32 * TN - 1) / (32 * TN));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((linear_direct_kernel<TM, TN, WAVES, OCC>), grid, dim3(64 * WAVES), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((linear_direct_kernel<TM, TN, WAVES, OCC>), grid, dim3(64 * WAVES), 0, 0, X, W, b, Y, M, N, K, K, K, N);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

int main()
{
    const int shapes[][3] = {{163840, 96, 192}, {163840, 96, 96}, {163840, 192, 96}, {163840, 48, 144}, {40960, 192, 384},
                             {40960, 384, 192}, {10240, 384, 768}, {10240, 768, 384}, {2560, 768, 1536}, {2560, 1536, 768},
                             {217600, 48, 96}, {163840, 48, 256}, {163840, 128, 48}};
    for (auto &s : shapes) {
        const int M = s[0], K = s[1], N = s[2];
        std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
        unsigned r = 12345u;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 9) & 0xffff) / 65536.f - 0.5f; };
        for (auto &v : hx) v = rnd();
        for (auto &v : hw) v = rnd();
        for (auto &v : hb) v = rnd();
        float *X, *W, *b, *Y, *Yr;
        CHECK(hipMalloc(&X, hx.size() * 4)); CHECK(hipMalloc(&W, hw.size() * 4)); CHECK(hipMalloc(&b, N * 4));
        CHECK(hipMalloc(&Y, (size_t)M * N * 4)); CHECK(hipMalloc(&Yr, (size_t)M * N * 4));
        CHECK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(reference_kernel, dim3((N + 63) / 64, M), dim3(64), 0, 0, X, W, b, Yr, M, N, K);
        const double fl = 2.0 * M * K * N;
        float t[6];
        t[0] = run<2, 3, 1, 2>(X, W, b, Y, M, N, K, 20);
        // check the first variant
        std::vector<float> y((size_t)M * N), yr((size_t)M * N);
        CHECK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(yr.data(), Yr, y.size() * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (size_t i = 0; i < y.size(); i += 97) md = fmax(md, fabs((double)y[i] - yr[i]));
        t[1] = run<2, 3, 4, 2>(X, W, b, Y, M, N, K, 20);
        t[2] = run<2, 3, 4, 3>(X, W, b, Y, M, N, K, 20);
        t[3] = run<1, 3, 4, 4>(X, W, b, Y, M, N, K, 20);
        t[4] = run<2, 2, 4, 3>(X, W, b, Y, M, N, K, 20);
        t[5] = run<4, 3, 2, 1>(X, W, b, Y, M, N, K, 20);
        if (K >= 96 && (K - 96) % 32 == 0 && N % 96 == 0 && M % 64 == 0) {
            const float o0 = run_overlap<2, 3, 6>(X, W, b, Y, M, N, K, 20);
            CHECK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost));
            double md4 = 0;
            for (size_t i = 0; i < y.size(); i += 97) md4 = fmax(md4, fabs((double)y[i] - yr[i]));
            printf("   epilogue overlapped into the next tile, 2x3, 1 wave / SIMD: %6.1f us  %5.1f TF (maxdiff %.2e)\n", o0, fl / o0 / 1e6, md4);
        }
        {
            const float q0 = run_seq<4, 3, 2>(X, W, b, Y, M, N, K, 20);
            CHECK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost));
            double md3 = 0;
            for (size_t i = 0; i < y.size(); i += 97) md3 = fmax(md3, fabs((double)y[i] - yr[i]));
            const float q1 = run_seq<4, 3, 3>(X, W, b, Y, M, N, K, 20);
            const float q2 = run_seq<2, 3, 4>(X, W, b, Y, M, N, K, 20);
            const float q3 = run_seq<4, 6, 2>(X, W, b, Y, M, N, K, 20);
            const float q4 = run_seq<2, 6, 4>(X, W, b, Y, M, N, K, 20);
            printf("   column-sequential 4x3 occ2 %6.1f us (maxdiff %.2e) | 4x3 occ3 %6.1f | 2x3 occ4 %6.1f | 4x6 occ2 %6.1f | 2x6 occ4 %6.1f\n", q0, md3, q1, q2, q3, q4);
        }
        if (K % 16 == 0) {
            const float p0 = run_persistent<2, 3, 1, 2, false>(X, W, b, Y, M, N, K, 20);
            CHECK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost));
            double md2 = 0;
            for (size_t i = 0; i < y.size(); i += 97) md2 = fmax(md2, fabs((double)y[i] - yr[i]));
            const float p1 = run_persistent<2, 3, 1, 3, false>(X, W, b, Y, M, N, K, 20);
            const float p2 = run_persistent<2, 2, 1, 3, false>(X, W, b, Y, M, N, K, 20);
            const float p3 = run_persistent<2, 3, 1, 2, true>(X, W, b, Y, M, N, K, 20);
            const float p4 = run_persistent<2, 3, 1, 1, true>(X, W, b, Y, M, N, K, 20);
            printf("   persistent 2x3 occ2 %6.1f us (maxdiff %.2e) | occ3 %6.1f | 2x2 occ3 %6.1f | MFMA+stores only occ2 %6.1f  occ1 %6.1f (%.1f TF)\n",
                   p0, md2, p1, p2, p3, p4, fl / p3 / 1e6);
        }
        printf("(%6d,%4d,%4d) maxdiff %.2e | 2x3 w1 occ2 %6.1f us %5.1f TF | 2x3 w4 occ2 %6.1f | 2x3 w4 occ3 %6.1f | 1x3 w4 occ4 %6.1f | 2x2 w4 occ3 %6.1f | 4x3 w2 occ1 %6.1f\n",
               M, K, N, md, t[0], fl / t[0] / 1e6, t[1], t[2], t[3], t[4], t[5]);
        fflush(stdout);
        hipFree(X); hipFree(W); hipFree(b); hipFree(Y); hipFree(Yr);
    }
    return 0;
}
