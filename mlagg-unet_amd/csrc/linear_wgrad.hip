// K5w -- weight / bias gradient of the token-major Linear layers:  dW[o][i] = sum_m dy[m][o] x[m][i],
// db[o] = sum_m dy[m][o], for M = batch * tokens in the 10^4..10^5 range and O, I in {48 .. 384}.
//
// These are the tall-skinny "reduce over tokens" GEMMs behind every nn.Linear of MLLABlock / Mlp /
// AggregatedAttention / SS2D_skip (reference nnUNetTrainer_MLAgg_2D_dt_MS.py:687-690, 887-907,
// MambaSkip.py:518,538,572-575).  The library GEMM picks 32x32 macro-tiles with a serial K loop for
// them (267 us for 96x96 at M = 163840 in the round-1 profile); the shape wants split-K over tokens.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32): the contraction index k is the TOKEN.  For one
// token pair, lane l supplies A[o = l & 31][k = l >> 5] = dy[m0 + (l >> 5)][o0 + (l & 31)] and
// B[k][i = l & 31] = x[m0 + (l >> 5)][i0 + (l & 31)]: both are 128-byte runs of one row, read straight
// from HBM into one VGPR each -- no LDS, no transpose.  A wave owns a slab of tokens and up to 3x3
// output tiles (144 accumulator registers), streams the slab once, and writes its partial block;
// a second kernel sums the partial blocks over slabs (16 row-groups, one float atomic each).
//
// Roofline: HBM-bound, algorithmic bytes 4 * M * (O + I) per launch (dy and x read once).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mlagg_hip.h"
#include "prof.h"
#include "bf16x3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TMAX = 3;            // output tiles per wave along O and along I
constexpr int RGROUPS = 16;        // row-groups of the partial reduction

struct WGeom {
    int M, O, I, dy_stride, x_stride;
    int slab, nslabs, ogroups, igroups;
};

__device__ __forceinline__ float ld_or_zero(const float *__restrict__ p, bool ok) { return ok ? *p : 0.f; }

template <int TO, int TI>
__global__ void __launch_bounds__(64)
linear_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ part, WGeom g)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    const int slab = blockIdx.x, og = blockIdx.y, ig = blockIdx.z;
    const int o0 = og * TMAX * 32, i0 = ig * TMAX * 32;
    f32x16 acc[TO][TI];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[TO];
    bool oko[TO], oki[TI];
#pragma unroll
    for (int a = 0; a < TO; ++a) { bsum[a] = 0.f; oko[a] = o0 + 32 * a + col < g.O; }
#pragma unroll
    for (int b = 0; b < TI; ++b) oki[b] = i0 + 32 * b + col < g.I;

    const int m_begin = slab * g.slab, m_end = min(m_begin + g.slab, g.M);
    const float *dyp = dy + (size_t)(m_begin + kh) * g.dy_stride + o0 + col;
    const float *xp = x + (size_t)(m_begin + kh) * g.x_stride + i0 + col;
    constexpr int UNR = 4;
    int m = m_begin;
    // software pipeline: the operands of the NEXT 4 token pairs are in flight while the 36 MFMAs of the
    // current 4 pairs issue (a wave alone on its SIMD would otherwise wait out every HBM round trip).
    // The pipelined loads are UNCONDITIONAL: columns beyond O / I read a clamped (valid) column and feed accumulator rows /
    // columns that are never stored.  With `ok ? *p : 0` every load sat in its own exec-masked block, the compiler could not
    // count them and put `s_waitcnt vmcnt(0)` in front of each consume -- which waits for the NEXT step's loads too.
    const float *dyb = dy + (size_t)m_begin * g.dy_stride;             // uniform row base; lane part as a 32-bit byte offset
    const float *xb = x + (size_t)m_begin * g.x_stride;
    unsigned offa[TO], offb[TI];
#pragma unroll
    for (int a = 0; a < TO; ++a) offa[a] = 4u * (unsigned)(kh * g.dy_stride + min(o0 + 32 * a + col, g.O - 1));
#pragma unroll
    for (int b = 0; b < TI; ++b) offb[b] = 4u * (unsigned)(kh * g.x_stride + min(i0 + 32 * b + col, g.I - 1));
    float av[2][UNR][TO], bv[2][UNR][TI];
    auto fetch = [&](float (&A)[UNR][TO], float (&Bv)[UNR][TI]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const char *da = reinterpret_cast<const char *>(dyb + (size_t)(2 * u) * g.dy_stride);
            const char *db = reinterpret_cast<const char *>(xb + (size_t)(2 * u) * g.x_stride);
#pragma unroll
            for (int a = 0; a < TO; ++a) A[u][a] = *reinterpret_cast<const float *>(da + (size_t)offa[a]);
#pragma unroll
            for (int b = 0; b < TI; ++b) Bv[u][b] = *reinterpret_cast<const float *>(db + (size_t)offb[b]);
        }
        dyb += (size_t)(2 * UNR) * g.dy_stride;
        xb += (size_t)(2 * UNR) * g.x_stride;
        dyp += (size_t)(2 * UNR) * g.dy_stride;
        xp += (size_t)(2 * UNR) * g.x_stride;
    };
    auto consume = [&](const float (&A)[UNR][TO], const float (&Bv)[UNR][TI]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int a = 0; a < TO; ++a) {
                bsum[a] += A[u][a];
#pragma unroll
                for (int b = 0; b < TI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u][a], Bv[u][b], acc[a][b], 0, 0, 0);
            }
        }
    };
    const int nfull = (m_end - m_begin) / (2 * UNR);       // iterations with all 8 rows in range
    if (nfull > 0) {
        fetch(av[0], bv[0]);
        int it = 0;
        for (; it + 2 < nfull; it += 2) {
            fetch(av[1], bv[1]);
            consume(av[0], bv[0]);
            fetch(av[0], bv[0]);
            consume(av[1], bv[1]);
        }
        if (it + 1 < nfull) {
            fetch(av[1], bv[1]);
            consume(av[0], bv[0]);
            consume(av[1], bv[1]);
        } else {
            consume(av[0], bv[0]);
        }
        m += nfull * 2 * UNR;
    }
    for (; m < m_end; m += 2) {                         // tail pairs (second row may be out of range)
        const bool rowok = m + kh < m_end;
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            const float av = ld_or_zero(dyp + 32 * a, oko[a] && rowok);
            bsum[a] += av;
#pragma unroll
            for (int b = 0; b < TI; ++b) {
                const float bv = ld_or_zero(xp + 32 * b, oki[b] && rowok);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
            }
        }
        dyp += (size_t)2 * g.dy_stride;
        xp += (size_t)2 * g.x_stride;
    }
    // partial block of this slab: part[slab][O*I + O]; D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*kh
    float *prow = part + (size_t)slab * ((size_t)g.O * g.I + g.O);
#pragma unroll
    for (int a = 0; a < TO; ++a) {
#pragma unroll
        for (int b = 0; b < TI; ++b) {
            if (!oki[b]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (o < g.O) prow[(size_t)o * g.I + i0 + 32 * b + col] = acc[a][b][r];
            }
        }
        if (ig == 0) {
            const float s = bsum[a] + __shfl_down(bsum[a], 32, 64);
            if (kh == 0 && oko[a]) prow[(size_t)g.O * g.I + o0 + 32 * a + col] = s;
        }
    }
}

// The same product on v_mfma_f32_32x32x16_bf16 with every fp32 operand as three bf16 pieces (bf16x3.h): as accurate as the fp32
// instruction, 2.7x less matrix-pipe time.  The token is still the contraction: a lane's operand is now EIGHT consecutive tokens of
// its column -- values of loads it issues itself, no LDS and no transpose here either.  Column mapping: tile j of a T-tile group
// takes the columns base + T * lane + j, so ONE T-float load per token row feeds all T tiles (T = 3: global_load_dwordx3, the
// half-wave reads the row's 384 bytes in one instruction) and one T-float store per accumulator row writes them back contiguously.
// 16 loads per 16-token block instead of 48: two blocks in flight stay under the 63-entry vmcnt counter (with scalar loads the
// compiler had to wait for the older block before it could issue the rest of the next).  Per block and 3 x 3 tiles: 16 loads, 264
// VALU instructions of splitting (filler under the matrix pipe: one wave per SIMD), 54 MFMAs of 8 passes instead of 72 of 16.
template <int T>
struct __attribute__((packed, aligned(4))) FVec {
    float v[T];
};

template <int TO, int TI>
__global__ void __launch_bounds__(64)
linear_wgrad_x3_kernel(const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ part, WGeom g)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    const int slab = blockIdx.x, og = blockIdx.y, ig = blockIdx.z;
    const int o0 = og * TMAX * 32, i0 = ig * TMAX * 32;
    f32x16 acc[TO][TI];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[TO];
#pragma unroll
    for (int a = 0; a < TO; ++a) bsum[a] = 0.f;

    const int m_begin = slab * g.slab, m_end = min(m_begin + g.slab, g.M);
    // the lane's first column, clamped so that its T-float load stays inside the row (columns past O / I feed accumulator rows /
    // columns that are never stored); uniform row base + per-lane 32-bit byte offset
    const int oc = min(o0 + TO * col, g.O - TO), ic = min(i0 + TI * col, g.I - TI);
    const float *dyb = dy + (size_t)m_begin * g.dy_stride;
    const float *xb = x + (size_t)m_begin * g.x_stride;
    const unsigned offa = 4u * (unsigned)(8 * kh * g.dy_stride + oc), offb = 4u * (unsigned)(8 * kh * g.x_stride + ic);
    // Three operand buffers: the loads of blocks j + 1 and j + 2 are in flight while block j is consumed -- with one wave per SIMD
    // nothing else covers the HBM round trip (longer than one block's ~1900 cycles of matrix work).  Every fetch is unconditional
    // (block index clamped to the last full block: at most two redundant fetches per slab).
    FVec<TO> av[3][8];
    FVec<TI> bv[3][8];
    const int nfull = (m_end - m_begin) / 16;              // blocks with all 16 tokens in range
    auto fetch = [&](FVec<TO> (&A)[8], FVec<TI> (&Bv)[8], int blk) {
        const size_t row = (size_t)16 * (size_t)min(blk, nfull - 1);
        const char *da = reinterpret_cast<const char *>(dyb + row * g.dy_stride);
        const char *db = reinterpret_cast<const char *>(xb + row * g.x_stride);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            A[t] = *reinterpret_cast<const FVec<TO> *>(da + (size_t)t * 4u * g.dy_stride + (size_t)offa);
            Bv[t] = *reinterpret_cast<const FVec<TI> *>(db + (size_t)t * 4u * g.x_stride + (size_t)offb);
        }
    };
    auto consume = [&](const FVec<TO> (&A)[8], const FVec<TI> (&Bv)[8]) {
        uint4 bq[TI][3];
#pragma unroll
        for (int b = 0; b < TI; ++b) {
            bf16x3::split3(Bv[0].v[b], Bv[1].v[b], bq[b][0].x, bq[b][1].x, bq[b][2].x);
            bf16x3::split3(Bv[2].v[b], Bv[3].v[b], bq[b][0].y, bq[b][1].y, bq[b][2].y);
            bf16x3::split3(Bv[4].v[b], Bv[5].v[b], bq[b][0].z, bq[b][1].z, bq[b][2].z);
            bf16x3::split3(Bv[6].v[b], Bv[7].v[b], bq[b][0].w, bq[b][1].w, bq[b][2].w);
        }
        uint4 aq[TO][3];
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            bf16x3::split3(A[0].v[a], A[1].v[a], aq[a][0].x, aq[a][1].x, aq[a][2].x);
            bf16x3::split3(A[2].v[a], A[3].v[a], aq[a][0].y, aq[a][1].y, aq[a][2].y);
            bf16x3::split3(A[4].v[a], A[5].v[a], aq[a][0].z, aq[a][1].z, aq[a][2].z);
            bf16x3::split3(A[6].v[a], A[7].v[a], aq[a][0].w, aq[a][1].w, aq[a][2].w);
            bsum[a] += ((A[0].v[a] + A[1].v[a]) + (A[2].v[a] + A[3].v[a])) + ((A[4].v[a] + A[5].v[a]) + (A[6].v[a] + A[7].v[a]));
        }
        bf16x3::mfma_tiles<TO, TI>(aq, bq, acc);
    };
    if (nfull > 0) {
        fetch(av[0], bv[0], 0);
        fetch(av[1], bv[1], 1);
        int it = 0;
        for (; it + 3 <= nfull; it += 3) {                 // buffers 0, 1 hold blocks it, it + 1
            fetch(av[2], bv[2], it + 2);
            consume(av[0], bv[0]);
            fetch(av[0], bv[0], it + 3);
            consume(av[1], bv[1]);
            fetch(av[1], bv[1], it + 4);
            consume(av[2], bv[2]);
        }
        if (it < nfull) consume(av[0], bv[0]);
        if (it + 1 < nfull) consume(av[1], bv[1]);
    }
    if (m_begin + nfull * 16 < m_end) {                    // last, partial block: clamped rows, values beyond the slab dropped
        const int mb = m_begin + nfull * 16;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int m = mb + 8 * kh + t;
            const bool ok = m < m_end;
            const size_t mr = (size_t)min(m, g.M - 1);
            const FVec<TO> va = *reinterpret_cast<const FVec<TO> *>(dy + mr * g.dy_stride + oc);
            const FVec<TI> vb = *reinterpret_cast<const FVec<TI> *>(x + mr * g.x_stride + ic);
#pragma unroll
            for (int a = 0; a < TO; ++a) av[0][t].v[a] = ok ? va.v[a] : 0.f;
#pragma unroll
            for (int b = 0; b < TI; ++b) bv[0][t].v[b] = ok ? vb.v[b] : 0.f;
        }
        consume(av[0], bv[0]);
    }
    // partial block of this slab: part[slab][O*I + O].  D layout of tile (a, b): column = lane & 31, row R = (r&3) + 8*(r>>2) +
    // 4*kh, i.e. dW[base_o(R) + a][base_i(col) + b] with base(.) the (clamped) first column of that lane: the TI tiles of a row
    // leave as one store.  A clamped lane repeats columns of its left neighbours below its natural base: those are skipped.
    float *prow = part + (size_t)slab * ((size_t)g.O * g.I + g.O);
    const int ib = i0 + TI * col;
#pragma unroll
    for (int a = 0; a < TO; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nat = o0 + TO * ((r & 3) + 8 * (r >> 2) + 4 * kh);
            const int o = min(nat, g.O - TO) + a;
            if (o < nat) continue;
            float *dst = prow + (size_t)o * g.I + ic;
            if (ic == ib) {
                FVec<TI> v;
#pragma unroll
                for (int b = 0; b < TI; ++b) v.v[b] = acc[a][b][r];
                *reinterpret_cast<FVec<TI> *>(dst) = v;
            } else {
#pragma unroll
                for (int b = 0; b < TI; ++b)
                    if (ic + b >= ib) dst[b] = acc[a][b][r];
            }
        }
        if (ig == 0) {
            const float sacc = bsum[a] + __shfl_down(bsum[a], 32, 64);
            if (kh == 0 && oc + a >= o0 + TO * col) prow[(size_t)g.O * g.I + oc + a] = sacc;
        }
    }
}

// dW[O*I] (+ db[O]) = sum over slabs of part rows.  A workgroup owns 64 consecutive columns (256-byte row segments) and
// splits the slabs over RGROUPS row-groups, finished through LDS: every output is written once -- no pre-zeroing of dW / db
// (the first version added 16 partial sums per output with float atomics behind a hipMemsetAsync: 96 fills per step).
__global__ void __launch_bounds__(64 * RGROUPS)
linear_wgrad_reduce_kernel(const float *__restrict__ part, int nslabs, int OI, int O, float *__restrict__ dW,
                           float *__restrict__ db)
{
    __shared__ float red[RGROUPS][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = OI + O;
    const int idx = blockIdx.x * 64 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < n) {
        int r = rg;
        for (; r + 3 * RGROUPS < nslabs; r += 4 * RGROUPS) {
            s0 += part[(size_t)r * n + idx];
            s1 += part[(size_t)(r + RGROUPS) * n + idx];
            s2 += part[(size_t)(r + 2 * RGROUPS) * n + idx];
            s3 += part[(size_t)(r + 3 * RGROUPS) * n + idx];
        }
        for (; r < nslabs; r += RGROUPS) s0 += part[(size_t)r * n + idx];
    }
    red[rg][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && idx < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RGROUPS; ++k) s += red[k][cx];
        if (idx < OI) dW[idx] = s;
        else if (db) db[idx - OI] = s;
    }
}

int make_geom(WGeom &g, int M, int O, int I, int dys, int xs)
{
    if (M <= 0 || O <= 0 || I <= 0 || dys < O || xs < I) return MLAGG_E_UNSUPPORTED;
    g.M = M; g.O = O; g.I = I; g.dy_stride = dys; g.x_stride = xs;
    g.ogroups = (O + TMAX * 32 - 1) / (TMAX * 32);
    g.igroups = (I + TMAX * 32 - 1) / (TMAX * 32);
    // slabs of an even number of tokens, at least 64
    // token slabs: ONE wave per SIMD (1024).  Round 1 ran ~2 per SIMD to hide the operand latency; with the pipelined loads
    // really in flight (see the kernel) one wave does, and half as many partial blocks are written and reduced: 1207 -> 1098 us
    // over the 15 shapes of tools/bench_linear.py (512: 1603, 1536: 1345 -- uneven -- 4096: 1522).  MLAGG_K5W_WAVES overrides.
    static const int target = [] { const char *e = getenv("MLAGG_K5W_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    int slab = (int)(((long long)M * g.ogroups * g.igroups + target - 1) / target);
    slab = ((slab + 15) / 16) * 16;       // whole 16-token blocks of the 16-deep instruction (and 8 pairs of the fp32 one)
    if (slab < 64) slab = 64;
    g.slab = slab;
    g.nslabs = (M + slab - 1) / slab;
    if (g.ogroups > 65535 || g.igroups > 65535) return MLAGG_E_UNSUPPORTED;
    return 0;
}

template <int TO, int TI>
void launch(const float *dy, const float *x, float *part, const WGeom &g, hipStream_t st, bool x3)
{
    if (x3)
        hipLaunchKernelGGL((linear_wgrad_x3_kernel<TO, TI>), dim3(g.nslabs, g.ogroups, g.igroups), dim3(64), 0, st, dy, x, part, g);
    else
        hipLaunchKernelGGL((linear_wgrad_kernel<TO, TI>), dim3(g.nslabs, g.ogroups, g.igroups), dim3(64), 0, st, dy, x, part, g);
}

}  // namespace

extern "C" size_t mlagg_linear_wgrad_workspace_floats(int M, int O, int I)
{
    WGeom g;
    if (make_geom(g, M, O, I, O, I)) return 0;
    return (size_t)g.nslabs * ((size_t)O * I + O);
}

namespace {
int wgrad(const float *dy, int dy_stride, const float *x, int x_stride, float *dW, float *db, float *workspace, int M, int O, int I,
          bool x3, void *stream)
{
    if (!dy || !x || !dW || !workspace) return MLAGG_E_NULLPTR;
    WGeom g;
    if (int rc = make_geom(g, M, O, I, dy_stride, x_stride)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // tiles per wave: full 3x3 groups when a dimension exceeds 96, else exactly what the dimension needs
    const int to = g.ogroups > 1 ? TMAX : (O + 31) / 32;
    const int ti = g.igroups > 1 ? TMAX : (I + 31) / 32;
    {
        MLAGG_TIMED(K_LINEAR_WGRAD, st);
        switch (to * 4 + ti) {
        case 1 * 4 + 1: launch<1, 1>(dy, x, workspace, g, st, x3); break;
        case 1 * 4 + 2: launch<1, 2>(dy, x, workspace, g, st, x3); break;
        case 1 * 4 + 3: launch<1, 3>(dy, x, workspace, g, st, x3); break;
        case 2 * 4 + 1: launch<2, 1>(dy, x, workspace, g, st, x3); break;
        case 2 * 4 + 2: launch<2, 2>(dy, x, workspace, g, st, x3); break;
        case 2 * 4 + 3: launch<2, 3>(dy, x, workspace, g, st, x3); break;
        case 3 * 4 + 1: launch<3, 1>(dy, x, workspace, g, st, x3); break;
        case 3 * 4 + 2: launch<3, 2>(dy, x, workspace, g, st, x3); break;
        default: launch<3, 3>(dy, x, workspace, g, st, x3); break;
        }
    }
    const int n = O * I + O;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(64 * RGROUPS), 0, st, workspace,
                       g.nslabs, O * I, O, dW, db);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mlagg_linear_wgrad(const float *dy, int dy_stride, const float *x, int x_stride, float *dW, float *db,
                                  float *workspace, int M, int O, int I, void *stream)
{
    return wgrad(dy, dy_stride, x, x_stride, dW, db, workspace, M, O, I, false, stream);
}

extern "C" int mlagg_linear_wgrad_x3(const float *dy, int dy_stride, const float *x, int x_stride, float *dW, float *db,
                                     float *workspace, int M, int O, int I, void *stream)
{
    return wgrad(dy, dy_stride, x, x_stride, dW, db, workspace, M, O, I, true, stream);
}
