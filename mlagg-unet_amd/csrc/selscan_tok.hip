// K1f -- the four-direction, multi-scale selective scan of the MSMM skip module on TOKEN-MAJOR tensors.
//
// What it replaces: SS2D_skip.forward_corev0 behind x_proj (reference MambaSkip.py:405-473): the stack / transpose / flip /
// cat chains that build the four scan sequences (M:414-422), the dt einsum (M:430-436), selective_scan_fn (M:445-451), the
// inverse re-orderings (M:455-471) and the four-way sum (M:534).  Rounds 1-3 ran this as K1' (cross_scan / cross_merge: ten
// data-movement launches, 0.64 ms, a 4x-expanded (B, 4*96, L) copy of x and of every gradient) around the (B, D, L) scan of
// csrc/selscan.hip.  Here the scan kernels read and write the token-major tensors directly:
//   xc    (B, L, 96)    conv outputs of all scales, natural token order -- u, shared by the four directions
//   xdbl  (B, L, 4*36)  x_proj output, per direction [dt0 dt1 dt2 pad | B(16) | C(16)] (the pad column keeps B / C 16-byte aligned)
//   idx   (4, L) int32  token visited by direction k at scan position t (closed form of M:419-422, built once per shape by the host)
// and write  yk (B, L, 4*96): direction k's output at the token's natural position (summed by mlagg_block_sum: no permutation),
// backward   duk (B, L, 4*96) likewise and dxdbl (B, L, 4*36): every direction owns its columns -- no atomics, no zero-fill.
//
// Why this layout is the better one for the scan itself (not only because K1' disappears): an 8-step tile of a direction is 8
// tokens, and a token row of u / dy is 384 contiguous bytes (three whole 128-byte lines).  In the (B, D, L) form a tile touched
// 32 bytes of each of 96 row lines, and each line came back four times (tiles apart in time): 1.6e7 of the 2.9e7 lines the
// round-3 backward pulled through L2 were such re-fetches (profiles/round3_j_pmc_selscan.md).
//
// Decomposition, arithmetic and lane mapping are those of csrc/selscan.hip (chunks of 64 steps, three passes forward,
// local + prefix + group-per-wave backward); only the memory paths differ.  Fixed shape: 96 channels per direction, 16 states,
// rank 3, 4 directions (every MLAgg-UNet configuration: d_model 48); other shapes take the (B, D, L) kernels.
#include "selscan_common.h"

namespace {

constexpr int HC = 96;            // channels per direction (d_inner)
constexpr int KD = 4;             // directions
constexpr int DIMT = KD * HC;     // 384 scan channels
constexpr int RK = 3;             // rank of the dt projection
constexpr int XB = 36;            // floats per direction in a projection row
constexpr int XW = KD * XB;       // 144
constexpr int CBT = 32;           // channels per workgroup of the chunk passes (see SCAN_CB in selscan.hip)
constexpr int NBLK = HC / CBT;    // 3

struct TokGeom {
    int batch, L, nchunks;
};

// the lane's 4 steps of delta' from the staged rank rows sR[t] = (dt0, dt1, dt2, pad)
__device__ __forceinline__ float4 tok_delta(const float4 *__restrict__ sR, int s, float w0, float w1, float w2, float bias, int t, int L)
{
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 v = sR[4 * s + j];
        const float x = softplus_f(w0 * v.x + w1 * v.y + w2 * v.z + bias);
        r[j] = (t + j < L) ? x : 0.f;              // beyond the sequence: a = 1, b = 0 (identity step)
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

__device__ __forceinline__ void pin4t(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ float f4at(const float4 &v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// ------------------------------------------------------------------------------------------
// forward pass 1 (FINAL = false): chunk from zero state -> (end state, sum delta')
// forward pass 3 (FINAL = true) : chunk from its entry state -> yk, tile-entry states for backward
// Workgroup = (chunk, direction k, 32-channel block, batch); lane = (channel cl, state quad s) for the recurrence,
// (step tid >> 3, float4 tid & 7) for the loads: a step's 32 channels of u are ONE 128-byte line, its projection
// row 9 float4.  All loads of the chunk are issued up front (indices first, then the rows they address).
// ------------------------------------------------------------------------------------------
template <bool FINAL>
__global__ void __launch_bounds__(128, 4)
tok_fwd_kernel(const float *__restrict__ xc, const float *__restrict__ xdbl, const int *__restrict__ idx,
               const float *__restrict__ Wdt, const float *__restrict__ A, const float *__restrict__ Dv,
               const float *__restrict__ dbias, float *__restrict__ yk, float *__restrict__ cstate, float *__restrict__ cdsum,
               float *__restrict__ csub, TokGeom gm)
{
    __shared__ float su[CBT * UP], sd[CBT * UP], sB[ST * BP], sC[ST * BP];
    __shared__ float4 sR[2][ST];
    __shared__ int sI[ST];

    const int tid = threadIdx.x, L = gm.L;
    const int chunk = blockIdx.x, k = blockIdx.y / NBLK, blk = blockIdx.y - k * NBLK, b = blockIdx.z;
    const int cl = tid >> 2, s = tid & 3, c0 = blk * CBT;
    const int d = k * HC + c0 + cl;
    const int lstep = tid >> 3, lq = tid & 7;
    const int tc0 = chunk * TC;
    const bool whole = tc0 + TC <= L;                       // uniform: every step of the chunk lies inside the sequence

    const int *idk = idx + (size_t)k * L;
    const float *xcb = xc + (size_t)b * L * HC + c0;
    const float *xdb = xdbl + (size_t)b * L * XW + k * XB;
    const float w0 = Wdt[d * RK], w1 = Wdt[d * RK + 1], w2 = Wdt[d * RK + 2];
    const float bias = dbias ? dbias[d] : 0.f;
    const float Dd = (FINAL && Dv) ? Dv[d] : 0.f;
    const size_t srow = ((size_t)b * gm.nchunks + chunk) * DIMT + d;

    float A2[4], h[4];
    {
        const float4 a4 = *reinterpret_cast<const float4 *>(A + d * NS + 4 * s);
        A2[0] = a4.x * LOG2E; A2[1] = a4.y * LOG2E; A2[2] = a4.z * LOG2E; A2[3] = a4.w * LOG2E;
        h[0] = h[1] = h[2] = h[3] = 0.f;
    }
    if (FINAL) {
        const float4 h0 = *reinterpret_cast<const float4 *>(cstate + srow * NS + 4 * s);
        h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w;
    }
    float dsum = 0.f;

    int tokv[NSUB];
    float4 pu[NSUB], px[NSUB], px2[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t = tc0 + sub * ST + lstep;
        tokv[sub] = idk[whole ? t : min(t, L - 1)];         // clamped: rows beyond the sequence are read and dropped
    }
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const unsigned ro = (unsigned)tokv[sub];
        pu[sub] = ldg_at<float4>(xcb, ro * HC + 4 * lq);
        px[sub] = ldg_at<float4>(xdb, ro * XW + 4 * lq);    // float4 0: rank row, 1..4: B, 5..7: C[0..11]
        px2[sub] = FINAL ? ldg_at<float4>(xdb, ro * XW + 32) : make_float4(0.f, 0.f, 0.f, 0.f);      // C[12..15]
    }
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) { pin4t(pu[sub]); pin4t(px[sub]); if (FINAL) pin4t(px2[sub]); }
    if (lq == 0) sR[0][lstep] = px[0];                      // rank rows run one sub-tile ahead of the barriers below

#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t0 = tc0 + sub * ST;
        if (FINAL && sub > 0)         // state entering 8-step tile 2 * sub: saved so that backward does not re-sweep the chunk
            *reinterpret_cast<float4 *>(csub + ((srow / DIMT * NT8 + (2 * sub - 1)) * DIMT + d) * NS + 4 * s) =
                make_float4(h[0], h[1], h[2], h[3]);
        lds_barrier();
        {
            float *dst = su + (4 * lq) * UP + lstep;        // transposed: [channel][step]
            dst[0] = pu[sub].x; dst[UP] = pu[sub].y; dst[2 * UP] = pu[sub].z; dst[3 * UP] = pu[sub].w;
        }
        *reinterpret_cast<float4 *>(sd + cl * UP + 4 * s) = tok_delta(sR[sub & 1], s, w0, w1, w2, bias, whole ? 0 : t0 + 4 * s, whole ? 4 : L);
        if (sub + 1 < NSUB && lq == 0) sR[(sub + 1) & 1][lstep] = px[sub + 1];
        if (lq >= 1 && lq <= 4) *reinterpret_cast<float4 *>(sB + lstep * BP + 4 * (lq - 1)) = px[sub];
        if (FINAL) {
            if (lq >= 5) *reinterpret_cast<float4 *>(sC + lstep * BP + 4 * (lq - 5)) = px[sub];
            if (lq == 0) {
                *reinterpret_cast<float4 *>(sC + lstep * BP + 12) = px2[sub];
                sI[lstep] = tokv[sub];
            }
        }
        lds_barrier();
        float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of a 4-step group with the group (register pressure)
            if (FINAL && q == 2)                    // ... and the state entering the odd 8-step tile 2 * sub + 1
                *reinterpret_cast<float4 *>(csub + ((srow / DIMT * NT8 + 2 * sub) * DIMT + d) * NS + 4 * s) =
                    make_float4(h[0], h[1], h[2], h[3]);
            const float4 dv = *reinterpret_cast<const float4 *>(sd + cl * UP + 4 * q);
            const float4 uv = *reinterpret_cast<const float4 *>(su + cl * UP + 4 * q);
            float yq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * q + j;
                const float dl = f4at(dv, j), uu = f4at(uv, j);
                const float dlu = dl * uu;
                const float4 Bq = *reinterpret_cast<const float4 *>(sB + t * BP + 4 * s);
                h[0] = fast_exp2(dl * A2[0]) * h[0] + dlu * Bq.x;
                h[1] = fast_exp2(dl * A2[1]) * h[1] + dlu * Bq.y;
                h[2] = fast_exp2(dl * A2[2]) * h[2] + dlu * Bq.z;
                h[3] = fast_exp2(dl * A2[3]) * h[3] + dlu * Bq.w;
                if (FINAL) {
                    const float4 Cq = *reinterpret_cast<const float4 *>(sC + t * BP + 4 * s);
                    const float yp = Cq.x * h[0] + Cq.y * h[1] + Cq.z * h[2] + Cq.w * h[3];
                    yq[j] = quad_sum(yp) + Dd * uu;
                } else {
                    dsum += dl;
                }
            }
            if (FINAL && s == q) yv = make_float4(yq[0], yq[1], yq[2], yq[3]);
        }
        if (FINAL) {
            // lane (cl, s) leaves with steps 4s .. 4s+3 of its channel: per store instruction the 16 channels of a wave are 64
            // contiguous bytes of a token row, the other wave of the workgroup writes the other half of the line
            const int4 tk = *reinterpret_cast<const int4 *>(sI + 4 * s);
            float *ykb = yk + (size_t)b * L * DIMT + k * HC + c0;
            if (whole) {
                stg_at<float>(ykb, (unsigned)tk.x * DIMT + cl, yv.x);
                stg_at<float>(ykb, (unsigned)tk.y * DIMT + cl, yv.y);
                stg_at<float>(ykb, (unsigned)tk.z * DIMT + cl, yv.z);
                stg_at<float>(ykb, (unsigned)tk.w * DIMT + cl, yv.w);
            } else {
                const int t = t0 + 4 * s;
                if (t < L) ykb[(size_t)tk.x * DIMT + cl] = yv.x;
                if (t + 1 < L) ykb[(size_t)tk.y * DIMT + cl] = yv.y;
                if (t + 2 < L) ykb[(size_t)tk.z * DIMT + cl] = yv.z;
                if (t + 3 < L) ykb[(size_t)tk.w * DIMT + cl] = yv.w;
            }
        }
        // the sub-tile's recurrence is FINISHED here (see selscan_fwd_kernel: without the anchor the branch-free pass-1 body is
        // scheduled with the LDS reads of all four sub-tiles parked in scratch)
        asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(dsum));
    }
    if (!FINAL) {
        *reinterpret_cast<float4 *>(cstate + srow * NS + 4 * s) = make_float4(h[0], h[1], h[2], h[3]);
        if (s == 0) cdsum[srow] = dsum;
    }
}

// ------------------------------------------------------------------------------------------
// backward pass 1: reverse-local summaries q_l = a_l (q_{l+1} + dy_l C_l) from q = 0 at the chunk end.  dy is the
// token-major gradient of the merged output: direction k's dy at position t is dy[idx[k][t]] (the four-way sum's adjoint).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128, 4)
tok_bwd_local_kernel(const float *__restrict__ xdbl, const int *__restrict__ idx, const float *__restrict__ Wdt,
                     const float *__restrict__ A, const float *__restrict__ dbias, const float *__restrict__ dy,
                     float *__restrict__ cq, TokGeom gm)
{
    __shared__ float sg[CBT * UP], sd[CBT * UP], sC[ST * BP];
    __shared__ float4 sR[2][ST];

    const int tid = threadIdx.x, L = gm.L;
    const int chunk = blockIdx.x, k = blockIdx.y / NBLK, blk = blockIdx.y - k * NBLK, b = blockIdx.z;
    const int cl = tid >> 2, s = tid & 3, c0 = blk * CBT;
    const int d = k * HC + c0 + cl;
    const int lstep = tid >> 3, lq = tid & 7;
    const int tc0 = chunk * TC;
    const bool whole = tc0 + TC <= L;

    const int *idk = idx + (size_t)k * L;
    const float *dyb = dy + (size_t)b * L * HC + c0;
    const float *xdb = xdbl + (size_t)b * L * XW + k * XB;
    const float w0 = Wdt[d * RK], w1 = Wdt[d * RK + 1], w2 = Wdt[d * RK + 2];
    const float bias = dbias ? dbias[d] : 0.f;

    float A2[4], q[4];
    {
        const float4 a4 = *reinterpret_cast<const float4 *>(A + d * NS + 4 * s);
        A2[0] = a4.x * LOG2E; A2[1] = a4.y * LOG2E; A2[2] = a4.z * LOG2E; A2[3] = a4.w * LOG2E;
        q[0] = q[1] = q[2] = q[3] = 0.f;
    }
    int tokv[NSUB];
    float4 pg[NSUB], px[NSUB];
    const unsigned xoff = lq == 0 ? 0u : (lq <= 4 ? 16u + 4u * lq : 0u);     // float4 0: rank row; 5..8 (floats 20..35): C
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t = tc0 + sub * ST + lstep;
        tokv[sub] = idk[whole ? t : min(t, L - 1)];
    }
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const unsigned ro = (unsigned)tokv[sub];
        pg[sub] = ldg_at<float4>(dyb, ro * HC + 4 * lq);
        px[sub] = ldg_at<float4>(xdb, ro * XW + xoff);
    }
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) { pin4t(pg[sub]); pin4t(px[sub]); }
    if (lq == 0) sR[(NSUB - 1) & 1][lstep] = px[NSUB - 1];
#pragma unroll
    for (int sub = NSUB - 1; sub >= 0; --sub) {
        const int t0 = tc0 + sub * ST;
        lds_barrier();
        {
            // beyond the sequence dy must read as zero (its rows are clamped copies of real tokens)
            const bool inr = whole || t0 + lstep < L;
            float *dst = sg + (4 * lq) * UP + lstep;
            dst[0] = inr ? pg[sub].x : 0.f; dst[UP] = inr ? pg[sub].y : 0.f; dst[2 * UP] = inr ? pg[sub].z : 0.f;
            dst[3 * UP] = inr ? pg[sub].w : 0.f;
        }
        *reinterpret_cast<float4 *>(sd + cl * UP + 4 * s) = tok_delta(sR[sub & 1], s, w0, w1, w2, bias, whole ? 0 : t0 + 4 * s, whole ? 4 : L);
        if (sub > 0 && lq == 0) sR[(sub - 1) & 1][lstep] = px[sub - 1];
        if (lq >= 1 && lq <= 4) *reinterpret_cast<float4 *>(sC + lstep * BP + 4 * (lq - 1)) = px[sub];
        lds_barrier();
#pragma unroll
        for (int qq = 3; qq >= 0; --qq) {
            __builtin_amdgcn_sched_barrier(0);
            const float4 dv = *reinterpret_cast<const float4 *>(sd + cl * UP + 4 * qq);
            const float4 gv = *reinterpret_cast<const float4 *>(sg + cl * UP + 4 * qq);
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const int t = 4 * qq + j;
                const float dl = f4at(dv, j), gy = f4at(gv, j);
                const float4 Cq = *reinterpret_cast<const float4 *>(sC + t * BP + 4 * s);
                q[0] = fast_exp2(dl * A2[0]) * (q[0] + gy * Cq.x);
                q[1] = fast_exp2(dl * A2[1]) * (q[1] + gy * Cq.y);
                q[2] = fast_exp2(dl * A2[2]) * (q[2] + gy * Cq.z);
                q[3] = fast_exp2(dl * A2[3]) * (q[3] + gy * Cq.w);
            }
        }
        asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    }
    const size_t srow = ((size_t)b * gm.nchunks + chunk) * DIMT + d;
    *reinterpret_cast<float4 *>(cq + srow * NS + 4 * s) = make_float4(q[0], q[1], q[2], q[3]);
}

// ------------------------------------------------------------------------------------------
// backward pass 3: ONE wave owns (batch b, direction k, chunk) and all 96 channels -- the group-per-wave kernel of
// csrc/selscan.hip (lane = 4 * channel lane + state quad, J = 6 channel slots walked per 8-step tile, dB / dC / d(dtr) summed in
// the lane and crossed once per tile) on the token-major tensors.  Per (tile, slot) a lane reads u and dy of its channel for
// steps 2s, 2s+1: two token rows, 64 contiguous bytes per row and wave, the neighbouring slot takes the other half of the line
// one iteration later.  The 8 token indices of a tile are fetched a tile ahead (tk_prev), the rows they address an iteration
// ahead (nx), so neither dependent load is exposed.
// WHOLE: L is a multiple of the 64-step chunk (the 256 x 256 shape: 21760 = 340 * 64): no clamps or range selects.
// ------------------------------------------------------------------------------------------
template <bool WHOLE>
__global__ void __launch_bounds__(64, 2)
tok_bwd_group_kernel(const float *__restrict__ xc, const float *__restrict__ xdbl, const int *__restrict__ idx,
                     const float *__restrict__ Wdt, const float *__restrict__ A, const float *__restrict__ Dv,
                     const float *__restrict__ dbias, const float *__restrict__ dy, const float *__restrict__ cstate,
                     const float *__restrict__ csub, const float *__restrict__ cq, float *__restrict__ duk,
                     float *__restrict__ dxdbl, float *__restrict__ part, TokGeom gm)
{
    __shared__ float sB[NS * BP], sC[NS * BP];          // [n][16 steps] of the current 16-step tile
    __shared__ float sR[RMAX * ST];                     // rank rows [r][16 steps]
    __shared__ float sD[16 * SP8], sX[16 * SP8];        // per channel lane: delta', delta' u of the current 8 steps
    __shared__ float sY[16 * SP8];                      // dy of the current 8 steps
    __shared__ float4 sQ[JMAX * 64], sA[JMAX * 64];     // per (channel slot, lane): reverse carry q[4], dA[4]
    __shared__ float2 sE[JMAX * 16 * 3];                // per channel: {dD, d(bias)}, {dW0, dW1}, {dW2, -}

    constexpr int J = HC / 16;
    const int chunk = blockIdx.x, k = blockIdx.y, b = blockIdx.z;
    const int L = gm.L;
    const int tc0 = chunk * TC;
    const size_t crow = (size_t)b * gm.nchunks + chunk;            // row of the per-chunk tensors

    {
        const int lane = threadIdx.x, cl = lane >> 2, s = lane & 3;
        for (int j = 0; j < J; ++j) {
            sQ[j * 64 + lane] = *reinterpret_cast<const float4 *>(cq + (crow * DIMT + k * HC + cl + 16 * j) * NS + 4 * s);
            sA[j * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int i = lane; i < J * 48; i += 64) sE[i] = make_float2(0.f, 0.f);
    }
    __syncthreads();

    const int m_first = WHOLE ? TC / T8 - 1 : min(TC / T8 - 1, (L - 1 - tc0) / T8);     // last tile that starts inside the sequence
    const int n_it = (m_first + 1) * J;

    const int *idk = idx + (size_t)k * L;
    const float *xcb = xc + (size_t)b * L * HC, *dyb = dy + (size_t)b * L * HC;
    const float *xdb = xdbl + (size_t)b * L * XW + k * XB;
    float *dukb = duk + (size_t)b * L * DIMT + k * HC;
    float *dxb = dxdbl + (size_t)b * L * XW + k * XB;
    const float *Ak = A + (size_t)k * HC * NS, *Wk = Wdt + (size_t)k * HC * RK;
    const float *bk_ = (dbias ? dbias : A) + k * HC, *Dk = (Dv ? Dv : A) + k * HC;

    struct Stream { float2 uv; float2 gy; float4 hv; };
    // tokens of steps 2s, 2s+1 of tile m (L is a multiple of 4: the pair is inside the sequence or outside it as a whole)
    auto tile_tok = [&](int m, int ln) -> int2 {
        const int t = tc0 + m * T8 + 2 * (ln & 3);
        return *reinterpret_cast<const int2 *>(idk + ((WHOLE || t < L) ? t : 0));
    };
    auto fetch = [&](int m, int j, int2 tk, int ln) -> Stream {
        Stream f;
        const int cl = ln >> 2, s = ln & 3;
        const unsigned c = cl + 16 * j;
        const float *sbase = m == 0 ? cstate + (crow * DIMT + k * HC) * NS : csub + ((crow * NT8 + (m - 1)) * DIMT + k * HC) * NS;
        f.hv = ldg_at<float4>(sbase, c * NS + 4 * s);
        f.gy.x = ldg_at<float>(dyb, (unsigned)tk.x * HC + c);
        f.gy.y = ldg_at<float>(dyb, (unsigned)tk.y * HC + c);
        f.uv.x = ldg_at<float>(xcb, (unsigned)tk.x * HC + c);
        f.uv.y = ldg_at<float>(xcb, (unsigned)tk.y * HC + c);
        return f;
    };
    // token of step `lstep` of the 16-step tile starting at t16 (staging of the projection rows)
    auto stage_tok = [&](int t16, int ln) -> int {
        const int t = max(t16 + (ln >> 2), 0);                     // (the tile before the chunk's first is asked for and never used)
        return idk[WHOLE ? t : min(t, L - 1)];
    };

    float accB[4][T8], accC[4][T8], accR[RMAX][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < T8; ++kk) { accB[i][kk] = 0.f; accC[i][kk] = 0.f; }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) { accR[r][0] = 0.f; accR[r][1] = 0.f; }

    int2 tk_cur = tile_tok(m_first, threadIdx.x);
    int2 tk_prev = tile_tok(max(m_first - 1, 0), threadIdx.x);
    int tk16 = stage_tok(tc0 + (m_first >> 1) * ST, threadIdx.x);
    Stream nx = fetch(m_first, 0, tk_cur, threadIdx.x);
    int m = m_first, j = 0;                                        // (tile, channel slot) of the iteration: counters, no division
#pragma unroll 1
    for (int it = 0; it < n_it; ++it) {
        // the lane index is made opaque once per iteration (see selscan_bwd_group_kernel: loop-invariant address variants are
        // otherwise hoisted and spilled)
        int ln = threadIdx.x;
        asm volatile("" : "+v"(ln));
        const int lane = ln, cl = ln >> 2, s = ln & 3;
        const int sub = m >> 1, odd = m & 1, ho = odd * T8;
        const int t16 = tc0 + sub * ST;                            // first step of the 16-step tile
        const bool last_slot = j == J - 1;
        const int mn = last_slot ? max(m - 1, 0) : m, jn = last_slot ? 0 : j + 1;   // the next iteration's (tile, slot)
        const int tm = t16 + ho;                                   // first step of this 8-step tile
        if (j == 0 && (odd || m == m_first)) {
            // first visit of the 16-step tile (tiles run in reverse): its 16 projection rows -> B / C as [n][t], rank rows [r][t]
            const int lstep = lane >> 2, lq = lane & 3;
            const unsigned ro = (unsigned)tk16 * XW;
            const float4 rb = ldg_at<float4>(xdb, ro + 4 + 4 * lq);
            const float4 rc = ldg_at<float4>(xdb, ro + 20 + 4 * lq);
            const float4 rr = ldg_at<float4>(xdb, ro);
            tk16 = stage_tok(t16 - ST, ln);                        // the next (earlier) 16-step tile's tokens, used 12 iterations on
            wave_lds_fence();
            {
                float *pb = sB + (4 * lq) * BP + lstep, *pc = sC + (4 * lq) * BP + lstep;
                pb[0] = rb.x; pb[BP] = rb.y; pb[2 * BP] = rb.z; pb[3 * BP] = rb.w;
                pc[0] = rc.x; pc[BP] = rc.y; pc[2 * BP] = rc.z; pc[3 * BP] = rc.w;
                if (lq == 0) { sR[lstep] = rr.x; sR[ST + lstep] = rr.y; sR[2 * ST + lstep] = rr.z; }
            }
            wave_lds_fence();
        }
        const bool inr2 = WHOLE || tm + 2 * s < L;                 // this lane's two steps lie inside the sequence
        Stream cur = nx;
        if (!WHOLE) cur.gy = make_float2(inr2 ? nx.gy.x : 0.f, inr2 ? nx.gy.y : 0.f);     // clamped rows beyond the sequence: dy = 0
        const unsigned c = cl + 16 * j;
        // this channel's small operands (L1 / L2 hits) are requested BEFORE the next iteration's streams: vmcnt retires in
        // order, so a wait for them must not sit behind the prefetch
        const float4 Av = ldg_at<float4>(Ak, c * NS + 4 * s);
        float wv[RK];
#pragma unroll
        for (int r = 0; r < RK; ++r) wv[r] = ldg_at<float>(Wk, c * RK + r);
        const float b_raw = ldg_at<float>(bk_, c), d_raw = ldg_at<float>(Dk, c);
        __builtin_amdgcn_sched_barrier(0);
        nx = fetch(mn, jn, last_slot ? tk_prev : tk_cur, ln);      // in flight until the next iteration's staging
        __builtin_amdgcn_sched_barrier(0);
        const float bias = dbias ? b_raw : 0.f;
        float Dd = Dv ? d_raw : 0.f;
        asm volatile("" : "+v"(Dd));                               // taken here (vmcnt leaves the prefetch alone), not lazily at its use
        float2 uf;                                                 // u of the two steps this lane finishes
        {
            // ---- activation and staging of this (tile, channel): lane (cl, s) activates steps 2s, 2s+1 ----
            wave_lds_fence();                                      // the previous channel's readers of the staging rows are done
            float2 raw = make_float2(0.f, 0.f);
#pragma unroll
            for (int r = 0; r < RK; ++r) {
                const float2 rv = *reinterpret_cast<const float2 *>(sR + r * ST + ho + 2 * s);
                raw.x += wv[r] * rv.x; raw.y += wv[r] * rv.y;
            }
            float2 da;
            da.x = act_delta(raw.x, bias, 1, inr2);
            da.y = act_delta(raw.y, bias, 1, inr2);
            *reinterpret_cast<float2 *>(sD + cl * SP8 + 2 * s) = da;
            *reinterpret_cast<float2 *>(sX + cl * SP8 + 2 * s) = make_float2(da.x * cur.uv.x, da.y * cur.uv.y);
            uf = cur.uv;
            *reinterpret_cast<float2 *>(sY + cl * SP8 + 2 * s) = cur.gy;
            wave_lds_fence();
        }

        const float Ar[4] = {Av.x, Av.y, Av.z, Av.w};
        const float hent[4] = {cur.hv.x, cur.hv.y, cur.hv.z, cur.hv.w};
        float dk[T8], xk[T8], yk8[T8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 a4 = *reinterpret_cast<const float4 *>(sD + cl * SP8 + 4 * q);
            const float4 x4 = *reinterpret_cast<const float4 *>(sX + cl * SP8 + 4 * q);
            const float4 y4 = *reinterpret_cast<const float4 *>(sY + cl * SP8 + 4 * q);
            dk[4 * q] = a4.x; dk[4 * q + 1] = a4.y; dk[4 * q + 2] = a4.z; dk[4 * q + 3] = a4.w;
            xk[4 * q] = x4.x; xk[4 * q + 1] = x4.y; xk[4 * q + 2] = x4.z; xk[4 * q + 3] = x4.w;
            yk8[4 * q] = y4.x; yk8[4 * q + 1] = y4.y; yk8[4 * q + 2] = y4.z; yk8[4 * q + 3] = y4.w;
        }
        const float4 q4 = sQ[j * 64 + lane], a4c = sA[j * 64 + lane];
        float qc[4] = {q4.x, q4.y, q4.z, q4.w}, dAc[4] = {a4c.x, a4c.y, a4c.z, a4c.w};
        float sT[T8], sG[T8];
#pragma unroll
        for (int kk = 0; kk < T8; ++kk) { sT[kk] = 0.f; sG[kk] = 0.f; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *sBn = sB + (4 * s + i) * BP + ho, *sCn = sC + (4 * s + i) * BP + ho;
            const float4 b0 = *reinterpret_cast<const float4 *>(sBn), b1 = *reinterpret_cast<const float4 *>(sBn + 4);
            const float bk[T8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            const float A2 = Ar[i] * LOG2E;
            float ak[T8], hp[T8 + 1];
            float hh = hent[i];
#pragma unroll
            for (int kk = 0; kk < T8; ++kk) {
                hp[kk] = hh;
                ak[kk] = fast_exp2(dk[kk] * A2);
                hh = ak[kk] * hh + xk[kk] * bk[kk];
            }
            hp[T8] = hh;
            const float4 c0 = *reinterpret_cast<const float4 *>(sCn), c1 = *reinterpret_cast<const float4 *>(sCn + 4);
            const float ck[T8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            float qq = qc[i], dAi = dAc[i];
#pragma unroll
            for (int kk = T8 - 1; kk >= 0; --kk) {
                const float gh = qq + yk8[kk] * ck[kk];                     // dL/dh_k
                qq = ak[kk] * gh;                                           // carried to step k-1
                const float t1 = qq * hp[kk];                               // dL/da_k * a_k
                sT[kk] += t1 * Ar[i];
                sG[kk] += gh * bk[kk];
                dAi += t1 * dk[kk];
                accB[i][kk] += gh * xk[kk];                                 // this channel's dB[k][n] term, summed in the lane
                accC[i][kk] += yk8[kk] * hp[kk + 1];                        // dC[k][n] term
            }
            qc[i] = qq;
            dAc[i] = dAi;
        }
        __builtin_amdgcn_sched_barrier(0);
        sQ[j * 64 + lane] = make_float4(qc[0], qc[1], qc[2], qc[3]);
        sA[j * 64 + lane] = make_float4(dAc[0], dAc[1], dAc[2], dAc[3]);
        // ---- finish d(delta') and du: quad reduce-scatter, lane s keeps steps 2s, 2s+1 ----
        {
            float Ts[2], Gs[2];
            {
                const bool hi = s & 2, lo = s & 1;
                float kt[4], kg[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    kt[x] = (hi ? sT[x + 4] : sT[x]) + dpp_quad_xor2(hi ? sT[x] : sT[x + 4]);
                    kg[x] = (hi ? sG[x + 4] : sG[x]) + dpp_quad_xor2(hi ? sG[x] : sG[x + 4]);
                }
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    Ts[y] = (lo ? kt[y + 2] : kt[y]) + dpp_quad_xor1(lo ? kt[y] : kt[y + 2]);
                    Gs[y] = (lo ? kg[y + 2] : kg[y]) + dpp_quad_xor1(lo ? kg[y] : kg[y + 2]);
                }
            }
            const float2 d2 = *reinterpret_cast<const float2 *>(sD + cl * SP8 + 2 * s);
            const float2 g2 = *reinterpret_cast<const float2 *>(sY + cl * SP8 + 2 * s);
            const float dl2[2] = {d2.x, d2.y}, uu2[2] = {uf.x, uf.y}, gg2[2] = {g2.x, g2.y};
            float odd2[2], odu2[2];
            float ev[2 + RK] = {0.f, 0.f, 0.f, 0.f, 0.f};                  // this lane's part of dD, d(bias), dW[0..2]
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float sp = 1.f - __expf(-dl2[e]);                     // d softplus(x)/dx = sigmoid(x) = 1 - exp(-softplus(x))
                odd2[e] = inr2 ? (Ts[e] + uu2[e] * Gs[e]) * sp : 0.f;       // (sT + u * sG) * softplus'
                odu2[e] = dl2[e] * Gs[e] + Dd * gg2[e];                     // delta' * sG + D * dy
                ev[1] += odd2[e];
                ev[0] += gg2[e] * uu2[e];
            }
            if (WHOLE || inr2) {
                stg_at<float>(dukb, (unsigned)tk_cur.x * DIMT + c, odu2[0]);
                stg_at<float>(dukb, (unsigned)tk_cur.y * DIMT + c, odu2[1]);
            }
#pragma unroll
            for (int r = 0; r < RK; ++r) {
                const float2 rv = *reinterpret_cast<const float2 *>(sR + r * ST + ho + 2 * s);
                ev[2 + r] = odd2[0] * rv.x + odd2[1] * rv.y;                // dWdt[d][r] += sum_t d(raw delta) dtr[r][t]
                accR[r][0] += odd2[0] * wv[r];                              // d(dtr)[r][t] += d(raw delta) Wdt[d][r]
                accR[r][1] += odd2[1] * wv[r];
            }
#pragma unroll
            for (int x = 0; x < 2 + RK; ++x) ev[x] = quad_sum(ev[x]);
            if (s == 0) {
                float2 *e = sE + (j * 16 + cl) * 3;
                float2 e0 = e[0], e1 = e[1], e2 = e[2];
                e0.x += ev[0]; e0.y += ev[1]; e1.x += ev[2]; e1.y += ev[3]; e2.x += ev[4];
                e[0] = e0; e[1] = e1; e[2] = e2;
            }
        }

        if (last_slot) {
            // ---- sums over the 16 channel lanes (lane bits 2..5), once per tile for all 6 slots (see selscan_bwd_group_kernel) ----
            const int p = lane >> 5, q = (lane >> 4) & 1, r4 = cl & 3;
            const int e0 = tm + 4 * (r4 & 1);
            const int4 t4 = *reinterpret_cast<const int4 *>(idk + ((WHOLE || e0 < L) ? e0 : 0));      // tokens of the 4 steps this lane stores
            float v[32];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < T8; ++kk) {
                    float a = accB[i][kk], cc = accC[i][kk];
                    swap32(a, cc);
                    v[i * 8 + kk] = a + cc;
                    accB[i][kk] = 0.f; accC[i][kk] = 0.f;
                }
#pragma unroll
            for (int x = 0; x < 16; ++x) { swap16(v[x], v[x + 16]); v[x] += v[x + 16]; }
#pragma unroll
            for (int x = 0; x < 16; ++x) { v[x] += row_ror4(v[x]); v[x] += row_ror8(v[x]); }
            float4 o;
            o.x = r4 == 0 ? v[0] : (r4 == 1 ? v[4] : (r4 == 2 ? v[8] : v[12]));
            o.y = r4 == 0 ? v[1] : (r4 == 1 ? v[5] : (r4 == 2 ? v[9] : v[13]));
            o.z = r4 == 0 ? v[2] : (r4 == 1 ? v[6] : (r4 == 2 ? v[10] : v[14]));
            o.w = r4 == 0 ? v[3] : (r4 == 1 ? v[7] : (r4 == 2 ? v[11] : v[15]));
            // lane (p, q, r4, s): state n = 4s + 2q + (r4 >> 1) of dB (p = 0) / dC (p = 1), steps e0 .. e0+3: per store instruction
            // the wave writes the 32 B | C gradient columns (128 contiguous bytes) of two token rows
            const unsigned col = 4 + 16 * p + 4 * s + 2 * q + (r4 >> 1);
            if (WHOLE || e0 < L) {
                stg_at<float>(dxb, (unsigned)t4.x * XW + col, o.x);
                stg_at<float>(dxb, (unsigned)t4.y * XW + col, o.y);
                stg_at<float>(dxb, (unsigned)t4.z * XW + col, o.z);
                stg_at<float>(dxb, (unsigned)t4.w * XW + col, o.w);
            }
            {
                // d(dtr)[r][steps 2s, 2s+1]: 8 values (r, e) -> reduce-scatter over lane bits 5, 4, all-reduce over bits 3, 2
                float w[8];
#pragma unroll
                for (int r = 0; r < RMAX; ++r) { w[2 * r] = accR[r][0]; w[2 * r + 1] = accR[r][1]; accR[r][0] = 0.f; accR[r][1] = 0.f; }
#pragma unroll
                for (int x = 0; x < 4; ++x) { swap32(w[x], w[x + 4]); w[x] += w[x + 4]; }
#pragma unroll
                for (int x = 0; x < 2; ++x) { swap16(w[x], w[x + 2]); w[x] += w[x + 2]; }
#pragma unroll
                for (int x = 0; x < 2; ++x) { w[x] += row_ror4(w[x]); w[x] += row_ror8(w[x]); }
                const unsigned r = 2 * p + q;                      // the rank row this lane ends up with; r = 3: the pad column (sums of zeros)
                if ((cl & 3) == 0 && (WHOLE || inr2)) {
                    stg_at<float>(dxb, (unsigned)tk_cur.x * XW + r, w[0]);
                    stg_at<float>(dxb, (unsigned)tk_cur.y * XW + r, w[1]);
                }
            }
            tk_cur = tk_prev;
            tk_prev = tile_tok(max(mn - 1, 0), ln);                // the tile after the next, used 5 iterations on
        }
        m = mn; j = jn;
    }

    // per-chunk partial sums of dA (16 per channel), dD, ddelta_bias, dWdt -> selscan_reduce_partials
    __syncthreads();
    {
        const int lane = threadIdx.x, cl = lane >> 2, s = lane & 3;
        for (int jj = 0; jj < J; ++jj) {
            float *prow = part + (crow * DIMT + k * HC + cl + 16 * jj) * PP;
            *reinterpret_cast<float4 *>(prow + 4 * s) = sA[jj * 64 + lane];
            if (s == 0) {
                const float2 *e = sE + (jj * 16 + cl) * 3;
                *reinterpret_cast<float4 *>(prow + NS) = make_float4(e[0].x, e[0].y, e[1].x, e[1].y);
                *reinterpret_cast<float4 *>(prow + NS + 4) = make_float4(e[2].x, 0.f, 0.f, 0.f);
            }
        }
    }
}

int tok_geom(TokGeom &gm, ScanGeom &sg, int batch, int L)
{
    // L % 4: token pairs / quads of a tile are inside the sequence or outside it as a whole; 32-bit row offsets
    if (batch <= 0 || batch > 65535 || L <= 0 || (L & 3) || (size_t)L * DIMT >= (1ull << 32)) return MLAGG_E_UNSUPPORTED;
    gm.batch = batch; gm.L = L; gm.nchunks = (L + TC - 1) / TC;
    return make_geom(sg, batch, DIMT, L, NS, KD, SCAN_CB);
}

}  // namespace

extern "C" int mlagg_msmm_scan_supported(int d_inner, int d_state, int dt_rank, int directions, int L)
{
    return d_inner == HC && d_state == NS && dt_rank == RK && directions == KD && L > 0 && (L & 3) == 0 &&
           (size_t)L * DIMT < (1ull << 32);
}

extern "C" size_t mlagg_msmm_scan_state_floats(int batch, int L)
{
    const size_t nchunks = (L + TC - 1) / TC;
    return (size_t)batch * nchunks * DIMT * (NS + 1 + NT8 * NS);          // chunk entry states, chunk delta sums, 8-step tile entry states
}

extern "C" size_t mlagg_msmm_scan_fwd_workspace_floats(int batch, int L) { return (size_t)batch * L * DIMT; }

extern "C" size_t mlagg_msmm_scan_bwd_workspace_floats(int batch, int L)
{
    const size_t nchunks = (L + TC - 1) / TC;
    return (size_t)batch * nchunks * DIMT * (NS + PP) + (size_t)batch * L * DIMT;
}

extern "C" int mlagg_msmm_scan_fwd(const float *xc, const float *xdbl, const int *idx, const float *Wdt, const float *A, const float *D,
                                   const float *delta_bias, float *y, float *state, float *workspace, int batch, int L, void *stream)
{
    if (!xc || !xdbl || !idx || !Wdt || !A || !y || !state || !workspace) return MLAGG_E_NULLPTR;
    TokGeom gm;
    ScanGeom sg;
    if (int rc = tok_geom(gm, sg, batch, L)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *cstate = state;
    float *cdsum = state + (size_t)batch * gm.nchunks * DIMT * NS;
    float *csub = cdsum + (size_t)batch * gm.nchunks * DIMT;
    float *yk = workspace;
    const dim3 grid(gm.nchunks, KD * NBLK, batch), block(128);
    { MLAGG_TIMED(K_TOK_FWD_LOCAL, st);
      hipLaunchKernelGGL(tok_fwd_kernel<false>, grid, block, 0, st, xc, xdbl, idx, Wdt, A, D, delta_bias, yk, cstate, cdsum, csub, gm); }
    { MLAGG_TIMED(K_SELSCAN_PREFIX, st);
      hipLaunchKernelGGL(selscan_chunk_prefix, dim3((DIMT * NS + 255) / 256, batch), dim3(256), 0, st, A, cstate, cdsum, sg, 0); }
    { MLAGG_TIMED(K_TOK_FWD_FINAL, st);
      hipLaunchKernelGGL(tok_fwd_kernel<true>, grid, block, 0, st, xc, xdbl, idx, Wdt, A, D, delta_bias, yk, cstate, cdsum, csub, gm); }
    if (int rc = (int)hipGetLastError()) return rc;
    return mlagg_block_sum(yk, y, (long)batch * L, KD, HC, stream);          // the four-way sum of reference M:534
}

extern "C" int mlagg_msmm_scan_bwd(const float *xc, const float *xdbl, const int *idx, const float *Wdt, const float *A, const float *D,
                                   const float *delta_bias, const float *dy, const float *state, float *dxc, float *dxdbl, float *dWdt,
                                   float *dA, float *dD, float *ddelta_bias, float *workspace, int batch, int L, void *stream)
{
    if (!xc || !xdbl || !idx || !Wdt || !A || !dy || !state || !dxc || !dxdbl || !dWdt || !dA || !workspace) return MLAGG_E_NULLPTR;
    TokGeom gm;
    ScanGeom sg;
    if (int rc = tok_geom(gm, sg, batch, L)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *cstate = state;
    const float *cdsum = state + (size_t)batch * gm.nchunks * DIMT * NS;
    const float *csub = cdsum + (size_t)batch * gm.nchunks * DIMT;
    float *cq = workspace;
    float *part = cq + (size_t)batch * gm.nchunks * DIMT * NS;
    float *duk = part + (size_t)batch * gm.nchunks * DIMT * PP;
    { MLAGG_TIMED(K_TOK_BWD_LOCAL, st);
      hipLaunchKernelGGL(tok_bwd_local_kernel, dim3(gm.nchunks, KD * NBLK, batch), dim3(128), 0, st, xdbl, idx, Wdt, A, delta_bias, dy, cq, gm); }
    { MLAGG_TIMED(K_SELSCAN_PREFIX, st);
      hipLaunchKernelGGL(selscan_chunk_prefix, dim3((DIMT * NS + 255) / 256, batch), dim3(256), 0, st, A, cq, cdsum, sg, 1); }
    { MLAGG_TIMED(K_TOK_BWD_GROUP, st);
      const dim3 gridg(gm.nchunks, KD, batch);
      if (L % TC == 0)
          hipLaunchKernelGGL(tok_bwd_group_kernel<true>, gridg, dim3(64), 0, st, xc, xdbl, idx, Wdt, A, D, delta_bias, dy, cstate, csub, cq,
                             duk, dxdbl, part, gm);
      else
          hipLaunchKernelGGL(tok_bwd_group_kernel<false>, gridg, dim3(64), 0, st, xc, xdbl, idx, Wdt, A, D, delta_bias, dy, cstate, csub, cq,
                             duk, dxdbl, part, gm); }
    { MLAGG_TIMED(K_SELSCAN_REDUCE, st);
      hipLaunchKernelGGL(selscan_reduce_partials, dim3((DIMT * PP + 63) / 64), dim3(1024), 0, st, part, dA, dD, ddelta_bias, dWdt, RK, sg); }
    if (int rc = (int)hipGetLastError()) return rc;
    return mlagg_block_sum(duk, dxc, (long)batch * L, KD, HC, stream);        // every token collects its four directions
}
