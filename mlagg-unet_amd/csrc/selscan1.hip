// K1s -- selective scan with ONE state per channel (d_state = 1) on token-major volumes, scan orders applied in-kernel.
//
// Replaces, for the 3-D network of reference variants/mamba/UMambaEnc_SS3D.py (SS3D built with d_state = 1, expand = 2 at
// :640-655), the whole of `SS3D.forward_corev0` (:244-296) behind x_proj -- the 14 stack / permute / flip / cat copies that build
// the 12 scan sequences, the dt einsum, `selective_scan_fn` and the 11 inverse permutations -- and its autograd:
//     delta_l = softplus(Wdt[d] . dtr[:, l] + bias[d]),  h_l = exp(delta_l A[d]) h_{l-1} + delta_l B_l u_l,  y_l = C_l h_l + D[d] u_l
// u is read straight from the token-major convolution output through the int32 permutation table of the direction
// (tok[b][idx[k][l]][c]: a 256-byte row per step and 64 channels, coalesced whatever the order) and y_k is written at the natural
// position of its token, so the (B, K*C, L) scan-order copies of u and y (15 GB each at 2 x 96x160x160 tokens) never exist.
//
// Decomposition (wave64, one wave per workgroup): lane = channel, time is serial inside a chunk of CH steps, chunks run in
// parallel and are stitched by a prefix over the per-chunk affine maps (h -> exp(A sum(delta)) h + s): three passes forward
// (chunk sums, prefix, final) and three backward (reverse chunk sums, prefix, main).  With one state per channel a step is
// ~15 VALU instructions for 12 bytes of HBM traffic: the kernels are bandwidth-bound, unlike the 16-state K1.
// The per-step parameters (token index, B_l, C_l, dtr[:, l]: the same for every channel) are staged by the wave's lanes into a
// wave-private LDS tile and read back as broadcasts.  Backward: per-step sums over channels (dB_l, dC_l, d(dtr)[:, l]) go
// through an LDS transpose (lane = channel writes, lane = (quantity, step) sums a 64-float row); sums over time (dA, dD, dbias,
// dWdt) stay in registers and leave as per-chunk partial rows reduced in a fixed order.  No atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "internal.h"
#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int TS = 64;                 // steps per parameter tile in the forward / local kernels
constexpr int TB = 16;                 // steps per backward tile = interval of the saved states
constexpr int UN = 16;                 // row loads in flight per wave in the forward kernels
constexpr int RP = 68;                 // pitch of a row of the reduction tile (floats): conflict-free 16-byte reads

struct S1 {
    int B, L, C, K, R, CH, nchunk, nblk;
    long tok_stride, dout_stride;
};

__host__ __device__ inline int param_pitch(int R) { return (3 + R + 3) & ~3; }       // [idx, B, C, dt_0..dt_{R-1}], 16-byte rows

inline int chunk_len(int B, int K, int L)
{
    int ch = 2048;
    while (ch > 64 && (long)B * K * ((L + ch - 1) / ch) < 2048) ch >>= 1;
    return ch;
}

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

// softplus and sigmoid of the raw delta from one exponential (csrc/selscan.hip softplus_f: series below e = 0.01)
__device__ __forceinline__ float softplus1(float x, float &e)
{
    e = __expf(-fabsf(x));
    const float small = e * (1.f - e * (0.5f - e * (1.f / 3.f)));
    const float big = __builtin_amdgcn_logf(1.f + e) * 0.6931471805599453f;
    return fmaxf(x, 0.f) + (e < 0.01f ? small : big);
}

// lanes 0..n-1 fetch the parameters of steps l0..l0+n-1 of (b, k) and park them in the wave's LDS tile (step-major rows)
template <int R>
__device__ __forceinline__ void stage_params(float *sP, const S1 &g, const int *__restrict__ idx, const float *__restrict__ dtr,
                                             const float *__restrict__ Bs, const float *__restrict__ Cs, int b, int k, int l0, int n)
{
    constexpr int PQ = (3 + R + 3) & ~3;
    const int lane = threadIdx.x;
    if (lane < n) {
        const int l = min(l0 + lane, g.L - 1);
        const size_t bk = (size_t)b * g.K + k;
        float *row = sP + lane * PQ;
        row[0] = __int_as_float(idx[(size_t)k * g.L + l]);
        row[1] = Bs[bk * g.L + l];
        row[2] = Cs[bk * g.L + l];
#pragma unroll
        for (int r = 0; r < R; ++r) row[3 + r] = dtr[(bk * R + r) * g.L + l];
    }
}

template <int R>
struct StepParams {
    int token;
    float Bt, Ct, dt[R];
};

template <int R>
__device__ __forceinline__ StepParams<R> read_params(const float *sP, int t)
{
    constexpr int PQ = (3 + R + 3) & ~3;
    StepParams<R> p;
    const float4 *row = reinterpret_cast<const float4 *>(sP + t * PQ);
    float v[PQ];
#pragma unroll
    for (int i = 0; i < PQ / 4; ++i) {
        const float4 q = row[i];
        v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
    p.token = __builtin_amdgcn_readfirstlane(__float_as_int(v[0]));
    p.Bt = v[1];
    p.Ct = v[2];
#pragma unroll
    for (int r = 0; r < R; ++r) p.dt[r] = v[3 + r];
    return p;
}

// ------------------------------------------------------------------------------------------------------------------
// forward: FINAL = false: chunk summaries (end state from a zero start, sum of delta); FINAL = true: the real sweep from the
// prefixed chunk-entry state, writes y_k at the token's natural position and the state entering every TB-step tile
// ------------------------------------------------------------------------------------------------------------------
template <int R, bool FINAL>
__global__ void __launch_bounds__(64)
sel1_fwd_kernel(const float *__restrict__ tok, const int *__restrict__ idx, const float *__restrict__ dtr, const float *__restrict__ Bs,
                const float *__restrict__ Cs, const float *__restrict__ Wdt, const float *__restrict__ A, const float *__restrict__ D,
                const float *__restrict__ bias, float *__restrict__ yk, float *__restrict__ Hin, float *__restrict__ Dsum,
                float *__restrict__ Hs, S1 g)
{
    constexpr int PQ = (3 + R + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float sP[TS * PQ];
    const int chunk = blockIdx.x, k = blockIdx.y / g.nblk, cb = blockIdx.y - k * g.nblk, b = blockIdx.z;
    const int lane = threadIdx.x, c = cb * 64 + lane, kc = k * g.C + c;
    const float Ac = A[kc], Dc = D[kc], bc = bias[kc];
    float W[R];
#pragma unroll
    for (int r = 0; r < R; ++r) W[r] = Wdt[(size_t)kc * R + r];
    const size_t sidx = (((size_t)b * g.K + k) * g.nchunk + chunk) * g.C + c;
    float h = FINAL ? Hin[sidx] : 0.f, sumd = 0.f;
    const float *tb = tok + (size_t)b * g.L * g.tok_stride + c;
    const size_t yrow = (size_t)g.K * g.C;
    float *yb = yk + (size_t)k * g.C + c;
    const int lbeg = chunk * g.CH, lend = min(g.L, lbeg + g.CH);
    for (int l0 = lbeg; l0 < lend; l0 += TS) {
        const int n = min(TS, lend - l0);
        wave_fence();
        stage_params<R>(sP, g, idx, dtr, Bs, Cs, b, k, l0, n);
        wave_fence();
        if (n == TS) {
#pragma unroll 1
            for (int t0 = 0; t0 < TS; t0 += UN) {
                StepParams<R> p[UN];
                float u[UN];
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    p[j] = read_params<R>(sP, t0 + j);
                    u[j] = tb[(size_t)p[j].token * g.tok_stride];
                }
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    if (FINAL && ((t0 + j) % TB) == 0)
                        Hs[(((size_t)b * g.K + k) * ((g.L + TB - 1) / TB) + (l0 + t0 + j) / TB) * g.C + c] = h;
                    float raw = bc, e;
#pragma unroll
                    for (int r = 0; r < R; ++r) raw = fmaf(W[r], p[j].dt[r], raw);
                    const float dl = softplus1(raw, e);
                    h = fmaf(__expf(dl * Ac), h, dl * p[j].Bt * u[j]);
                    if (FINAL) yb[((size_t)b * g.L + p[j].token) * yrow] = fmaf(p[j].Ct, h, Dc * u[j]);
                    else sumd += dl;
                }
            }
        } else {
            for (int t = 0; t < n; ++t) {            // ragged tail of the sequence: one step at a time
                const StepParams<R> p = read_params<R>(sP, t);
                const float u = tb[(size_t)p.token * g.tok_stride];
                if (FINAL && (t % TB) == 0) Hs[(((size_t)b * g.K + k) * ((g.L + TB - 1) / TB) + (l0 + t) / TB) * g.C + c] = h;
                float raw = bc, e;
#pragma unroll
                for (int r = 0; r < R; ++r) raw = fmaf(W[r], p.dt[r], raw);
                const float dl = softplus1(raw, e);
                h = fmaf(__expf(dl * Ac), h, dl * p.Bt * u);
                if (FINAL) yb[((size_t)b * g.L + p.token) * yrow] = fmaf(p.Ct, h, Dc * u);
                else sumd += dl;
            }
        }
    }
    if (!FINAL) { Hin[sidx] = h; Dsum[sidx] = sumd; }
}

// prefix over the chunks of one (b, k, channel): carry' = exp(A sum(delta)) carry + s.  In place: S[chunk] becomes the value
// ENTERING the chunk (forward: from the left; REVERSE: from the right).
template <bool REVERSE>
__global__ void __launch_bounds__(64)
sel1_prefix_kernel(float *__restrict__ S, const float *__restrict__ Dsum, const float *__restrict__ A, S1 g)
{
    const int k = blockIdx.x / g.nblk, cb = blockIdx.x - k * g.nblk, b = blockIdx.y;
    const int c = cb * 64 + threadIdx.x;
    const float Ac = A[k * g.C + c];
    const size_t base = ((size_t)b * g.K + k) * g.nchunk * g.C + c;
    float carry = 0.f;
    constexpr int PB = 16;
    for (int i0 = 0; i0 < g.nchunk; i0 += PB) {
        float s[PB], d[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int i = min(i0 + j, g.nchunk - 1), ch = REVERSE ? g.nchunk - 1 - i : i;
            s[j] = S[base + (size_t)ch * g.C];
            d[j] = Dsum[base + (size_t)ch * g.C];
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            if (i0 + j < g.nchunk) {
                const int ch = REVERSE ? g.nchunk - 1 - (i0 + j) : i0 + j;
                S[base + (size_t)ch * g.C] = carry;
                carry = fmaf(__expf(Ac * d[j]), carry, s[j]);
            }
        }
    }
}

// backward, pass 1: q entering the chunk's FIRST step from the right end of the chunk with a zero carry:
// gh_l = dy_l C_l + q_{l+1}, q_l = a_l gh_l, swept from the chunk's last step down
template <int R>
__global__ void __launch_bounds__(64)
sel1_bwd_local_kernel(const float *__restrict__ dout, const int *__restrict__ idx, const float *__restrict__ dtr,
                      const float *__restrict__ Bs, const float *__restrict__ Cs, const float *__restrict__ Wdt,
                      const float *__restrict__ A, const float *__restrict__ bias, float *__restrict__ Q, S1 g)
{
    constexpr int PQ = (3 + R + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float sP[TS * PQ];
    const int chunk = blockIdx.x, k = blockIdx.y / g.nblk, cb = blockIdx.y - k * g.nblk, b = blockIdx.z;
    const int lane = threadIdx.x, c = cb * 64 + lane, kc = k * g.C + c;
    const float Ac = A[kc], bc = bias[kc];
    float W[R];
#pragma unroll
    for (int r = 0; r < R; ++r) W[r] = Wdt[(size_t)kc * R + r];
    const float *db = dout + (size_t)b * g.L * g.dout_stride + c;
    const int lbeg = chunk * g.CH, lend = min(g.L, lbeg + g.CH);
    const int ntile = (lend - lbeg + TS - 1) / TS;
    float q = 0.f;
    for (int ti = ntile - 1; ti >= 0; --ti) {
        const int l0 = lbeg + ti * TS, n = min(TS, lend - l0);
        wave_fence();
        stage_params<R>(sP, g, idx, dtr, Bs, Cs, b, k, l0, n);
        wave_fence();
        if (n == TS) {
#pragma unroll 1
            for (int t0 = TS - UN; t0 >= 0; t0 -= UN) {
                StepParams<R> p[UN];
                float dy[UN];
#pragma unroll
                for (int j = UN - 1; j >= 0; --j) {
                    p[j] = read_params<R>(sP, t0 + j);
                    dy[j] = db[(size_t)p[j].token * g.dout_stride];
                }
#pragma unroll
                for (int j = UN - 1; j >= 0; --j) {
                    float raw = bc, e;
#pragma unroll
                    for (int r = 0; r < R; ++r) raw = fmaf(W[r], p[j].dt[r], raw);
                    const float dl = softplus1(raw, e);
                    q = __expf(dl * Ac) * fmaf(dy[j], p[j].Ct, q);
                }
            }
        } else {
            for (int t = n - 1; t >= 0; --t) {
                const StepParams<R> p = read_params<R>(sP, t);
                const float dy = db[(size_t)p.token * g.dout_stride];
                float raw = bc, e;
#pragma unroll
                for (int r = 0; r < R; ++r) raw = fmaf(W[r], p.dt[r], raw);
                const float dl = softplus1(raw, e);
                q = __expf(dl * Ac) * fmaf(dy, p.Ct, q);
            }
        }
    }
    Q[(((size_t)b * g.K + k) * g.nchunk + chunk) * g.C + c] = q;
}

// backward, pass 3: per TB-step tile in reverse: re-run the tile forward from its saved entry state, sweep it backwards, write
// du_k at the token's natural position, reduce the per-step channel sums through LDS, keep the per-channel time sums.
//   outputs per step (scan order): dB (B, K, L), dC (B, K, L), ddtr (B, K, R, L) -- or, with more than one channel block, their
//   per-block partials [B][K][nblk][2 + R][L] in `stepws` (summed by sel1_step_reduce_kernel)
//   part: per-chunk partial rows [B][nchunk][K][C][3 + R] = dA, dD, dbias, dWdt[0..R)
template <int R>
__global__ void __launch_bounds__(64)
sel1_bwd_kernel(const float *__restrict__ tok, const float *__restrict__ dout, const int *__restrict__ idx, const float *__restrict__ dtr,
                const float *__restrict__ Bs, const float *__restrict__ Cs, const float *__restrict__ Wdt, const float *__restrict__ A,
                const float *__restrict__ D, const float *__restrict__ bias, const float *__restrict__ Hs, const float *__restrict__ Qin,
                float *__restrict__ duk, float *__restrict__ dBs, float *__restrict__ dCs, float *__restrict__ ddtr,
                float *__restrict__ stepws, float *__restrict__ part, float *__restrict__ dump, S1 g)
{
    constexpr int PQ = (3 + R + 3) & ~3;
    constexpr int NQ = 2 + R;                          // per-step quantities: dB, dC, ddt_r
    __shared__ __attribute__((aligned(16))) float sP[TB * PQ];
    __shared__ __attribute__((aligned(16))) float sR[64 * RP];
    const int chunk = blockIdx.x, k = blockIdx.y / g.nblk, cb = blockIdx.y - k * g.nblk, b = blockIdx.z;
    const int lane = threadIdx.x, c = cb * 64 + lane, kc = k * g.C + c;
    const float Ac = A[kc], Dc = D[kc], bc = bias[kc];
    float W[R], dW[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { W[r] = Wdt[(size_t)kc * R + r]; dW[r] = 0.f; }
    float dA = 0.f, dD = 0.f, dbias = 0.f;
    const size_t bk = (size_t)b * g.K + k;
    float q = Qin[(bk * g.nchunk + chunk) * g.C + c];
    const float *tb = tok + (size_t)b * g.L * g.tok_stride + c;
    const float *db = dout + (size_t)b * g.L * g.dout_stride + c;
    const size_t yrow = (size_t)g.K * g.C;
    float *dub = duk + (size_t)k * g.C + c;
    const int ntile_all = (g.L + TB - 1) / TB;
    const int lbeg = chunk * g.CH, lend = min(g.L, lbeg + g.CH);
    const int ntile = (lend - lbeg + TB - 1) / TB;
    // where lane o = (quantity o / TB, step o % TB) of a reduction round stores its sum
    const int oq = lane / TB, ot = lane % TB;
    for (int ti = ntile - 1; ti >= 0; --ti) {
        const int l0 = lbeg + ti * TB, n = min(TB, lend - l0);
        wave_fence();
        stage_params<R>(sP, g, idx, dtr, Bs, Cs, b, k, l0, n);
        wave_fence();
        // the parameters are re-read from LDS in every phase (broadcast reads are cheap; TB x (3 + R) live registers are not)
        float u[TB], dy[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const StepParams<R> p = read_params<R>(sP, min(j, n - 1));
            u[j] = tb[(size_t)p.token * g.tok_stride];
            dy[j] = db[(size_t)p.token * g.dout_stride];
        }
        float h = Hs[(bk * ntile_all + l0 / TB) * g.C + c];
        float dl[TB], hp[TB], av[TB], sg[TB], bu[TB];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const StepParams<R> p = read_params<R>(sP, min(j, n - 1));
            float raw = bc, e;
#pragma unroll
            for (int r = 0; r < R; ++r) raw = fmaf(W[r], p.dt[r], raw);
            const float d0 = softplus1(raw, e);
            const float inv = __builtin_amdgcn_rcpf(1.f + e);
            const bool ok = j < n;
            dl[j] = ok ? d0 : 0.f;                      // a step past the end of the sequence is the identity map
            dy[j] = ok ? dy[j] : 0.f;
            sg[j] = raw >= 0.f ? inv : e * inv;          // sigmoid(raw) = d softplus / d raw
            hp[j] = h;
            av[j] = __expf(dl[j] * Ac);
            bu[j] = p.Bt * u[j];
            h = fmaf(av[j], h, dl[j] * bu[j]);
        }
        float pB[TB], pC[TB], pd[TB];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = TB - 1; j >= 0; --j) {
            const StepParams<R> p = read_params<R>(sP, min(j, n - 1));
            const float hj = fmaf(av[j], hp[j], dl[j] * bu[j]);
            const float gh = fmaf(dy[j], p.Ct, q);
            pC[j] = dy[j] * hj;
            pB[j] = gh * dl[j] * u[j];
            const float du = fmaf(dy[j], Dc, gh * dl[j] * p.Bt);
            const float t1 = gh * hp[j] * av[j];
            const float ddl = fmaf(gh, bu[j], t1 * Ac);
            dA = fmaf(t1, dl[j], dA);
            dD = fmaf(dy[j], u[j], dD);
            const float draw = (j < n) ? ddl * sg[j] : 0.f;
            pd[j] = draw;
            dbias += draw;
#pragma unroll
            for (int r = 0; r < R; ++r) dW[r] = fmaf(draw, p.dt[r], dW[r]);
            q = av[j] * gh;
            float *dst = (j < n) ? dub + ((size_t)b * g.L + p.token) * yrow : dump + lane;
            *dst = du;
        }
        // channel sums per step: rounds of 64 / TB = 4 quantities through the LDS tile
        constexpr int QPR = 64 / TB;
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += QPR) {
            wave_fence();
#pragma unroll
            for (int qq = 0; qq < QPR; ++qq) {
                const int qi = q0 + qq;
                if (qi < NQ) {
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        float v;
                        if (qi == 0) v = pB[j];
                        else if (qi == 1) v = pC[j];
                        else v = pd[j] * W[(qi - 2) < R ? (qi - 2) : 0];
                        sR[(qq * TB + j) * RP + lane] = v;
                    }
                }
            }
            wave_fence();
            const float4 *row = reinterpret_cast<const float4 *>(sR + lane * RP);
            float4 s4 = row[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                const float4 v = row[i];
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
            const float s = (s4.x + s4.y) + (s4.z + s4.w);
            const int qi = q0 + oq;
            float *dst = dump + 64 + lane;
            if (qi < NQ && ot < n) {
                if (g.nblk > 1) dst = stepws + (((bk * g.nblk + cb) * NQ + qi) * (size_t)g.L) + l0 + ot;
                else if (qi == 0) dst = dBs + bk * g.L + l0 + ot;
                else if (qi == 1) dst = dCs + bk * g.L + l0 + ot;
                else dst = ddtr + (bk * R + (qi - 2)) * g.L + l0 + ot;
            }
            *dst = s;
        }
    }
    float *pr = part + ((((size_t)b * g.nchunk + chunk) * g.K + k) * g.C + c) * (3 + R);
    pr[0] = dA; pr[1] = dD; pr[2] = dbias;
#pragma unroll
    for (int r = 0; r < R; ++r) pr[3 + r] = dW[r];
}

// sums the per-channel-block partials of the per-step outputs: stepws [B*K][nblk][NQ][L] -> dB, dC (B*K, L), ddtr (B*K, R, L)
__global__ void __launch_bounds__(256)
sel1_step_reduce_kernel(const float *__restrict__ stepws, float *__restrict__ dBs, float *__restrict__ dCs, float *__restrict__ ddtr,
                        int nblk, int R, int L)
{
    const int l = blockIdx.x * 256 + threadIdx.x, qi = blockIdx.y, NQ = 2 + R;
    const size_t bk = blockIdx.z;
    if (l >= L) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += stepws[((bk * nblk + i) * NQ + qi) * (size_t)L + l];
    if (qi == 0) dBs[bk * L + l] = s;
    else if (qi == 1) dCs[bk * L + l] = s;
    else ddtr[(bk * R + (qi - 2)) * L + l] = s;
}

int check(const S1 &g)
{
    if (g.B <= 0 || g.L <= 0 || g.C <= 0 || g.K <= 0 || g.B > 65535) return MLAGG_E_UNSUPPORTED;
    if (g.C % 64 || (long)g.K * g.nblk > 65535 || g.tok_stride < g.C) return MLAGG_E_UNSUPPORTED;
    return 0;
}

S1 geom(int B, int L, int C, int K, int R, long tok_stride, long dout_stride)
{
    S1 g;
    g.B = B; g.L = L; g.C = C; g.K = K; g.R = R;
    g.CH = chunk_len(B, K, L);
    g.nchunk = (L + g.CH - 1) / g.CH;
    g.nblk = (C + 63) / 64;
    g.tok_stride = tok_stride; g.dout_stride = dout_stride;
    return g;
}

#define SEL1_DISPATCH_R(R, MACRO)                     \
    switch (R) {                                      \
    case 1: MACRO(1); break;                          \
    case 2: MACRO(2); break;                          \
    case 3: MACRO(3); break;                          \
    case 4: MACRO(4); break;                          \
    case 8: MACRO(8); break;                          \
    case 16: MACRO(16); break;                        \
    case 20: MACRO(20); break;                        \
    default: return MLAGG_E_UNSUPPORTED;              \
    }

}  // namespace

extern "C" int mlagg_selscan1_chunk(int B, int L, int K) { return chunk_len(B, K, L); }

extern "C" size_t mlagg_selscan1_state_floats(int B, int L, int C, int K)
{
    if (B <= 0 || L <= 0 || C <= 0 || K <= 0) return 0;
    const int ch = chunk_len(B, K, L);
    const size_t nchunk = (L + ch - 1) / ch, ntile = (L + TB - 1) / TB;
    return (size_t)B * K * C * (2 * nchunk + ntile) + 128;
}

extern "C" int mlagg_selscan1_fwd(const float *tok, long tok_stride, const int *idx, const float *dtr, const float *Bs,
                                  const float *Cs, const float *Wdt, int R, const float *A, const float *D, const float *bias,
                                  float *yk, float *state, int B, int L, int C, int K, void *stream)
{
    if (!tok || !idx || !dtr || !Bs || !Cs || !Wdt || !A || !D || !bias || !yk || !state) return MLAGG_E_NULLPTR;
    const S1 g = geom(B, L, C, K, R, tok_stride, 0);
    if (int rc = check(g)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t per = (size_t)B * K * g.nchunk * C;
    float *Hin = state, *Dsum = state + per, *Hs = state + 2 * per;
    const dim3 grid(g.nchunk, K * g.nblk, B), block(64);
#define SEL1_FWD(RR)                                                                                                             \
    {                                                                                                                            \
        {                                                                                                                        \
            mlagg_prof::Scope sc_(RR == 2 ? mlagg_prof::K_SEL1_FWD_LOCAL_R2 : mlagg_prof::K_SEL1_FWD_LOCAL, st);                                                                                \
            hipLaunchKernelGGL((sel1_fwd_kernel<RR, false>), grid, block, 0, st, tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, yk, Hin, \
                               Dsum, Hs, g);                                                                                     \
        }                                                                                                                        \
        {                                                                                                                        \
            MLAGG_TIMED(K_SEL1_PREFIX, st);                                                                                   \
            hipLaunchKernelGGL(sel1_prefix_kernel<false>, dim3(K * g.nblk, B), block, 0, st, Hin, Dsum, A, g);                   \
        }                                                                                                                        \
        {                                                                                                                        \
            mlagg_prof::Scope sc_(RR == 2 ? mlagg_prof::K_SEL1_FWD_FINAL_R2 : mlagg_prof::K_SEL1_FWD_FINAL, st);                                                                                \
            hipLaunchKernelGGL((sel1_fwd_kernel<RR, true>), grid, block, 0, st, tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, yk, Hin,  \
                               Dsum, Hs, g);                                                                                     \
        }                                                                                                                        \
    }
    SEL1_DISPATCH_R(R, SEL1_FWD)
#undef SEL1_FWD
    return (int)hipGetLastError();
}

extern "C" size_t mlagg_selscan1_bwd_workspace_floats(int B, int L, int C, int K, int R)
{
    if (B <= 0 || L <= 0 || C <= 0 || K <= 0 || R <= 0) return 0;
    const int ch = chunk_len(B, K, L);
    const size_t nchunk = (L + ch - 1) / ch, nblk = (C + 63) / 64;
    return (size_t)B * K * nchunk * C * (1 + 3 + R) + (nblk > 1 ? (size_t)B * K * nblk * (2 + R) * L : 0) + 256;
}

extern "C" int mlagg_selscan1_bwd(const float *tok, long tok_stride, const int *idx, const float *dtr, const float *Bs,
                                  const float *Cs, const float *Wdt, int R, const float *A, const float *D, const float *bias,
                                  const float *dout, long dout_stride, const float *state, float *duk, float *ddtr, float *dBs,
                                  float *dCs, float *dparams, float *workspace, int B, int L, int C, int K, void *stream)
{
    if (!tok || !idx || !dtr || !Bs || !Cs || !Wdt || !A || !D || !bias || !dout || !state || !duk || !ddtr || !dBs || !dCs ||
        !dparams || !workspace)
        return MLAGG_E_NULLPTR;
    const S1 g = geom(B, L, C, K, R, tok_stride, dout_stride);
    if (int rc = check(g)) return rc;
    if (dout_stride < C) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t per = (size_t)B * K * g.nchunk * C;
    const float *Dsum = state + per, *Hs = state + 2 * per;
    float *Q = workspace, *part = workspace + per, *dump = part + per * (3 + R), *stepws = dump + 256;
    const dim3 grid(g.nchunk, K * g.nblk, B), block(64);
#define SEL1_BWD(RR)                                                                                                              \
    {                                                                                                                             \
        {                                                                                                                         \
            mlagg_prof::Scope sc_(RR == 2 ? mlagg_prof::K_SEL1_BWD_LOCAL_R2 : mlagg_prof::K_SEL1_BWD_LOCAL, st);                                                                                 \
            hipLaunchKernelGGL(sel1_bwd_local_kernel<RR>, grid, block, 0, st, dout, idx, dtr, Bs, Cs, Wdt, A, bias, Q, g);        \
        }                                                                                                                         \
        {                                                                                                                         \
            MLAGG_TIMED(K_SEL1_PREFIX, st);                                                                                    \
            hipLaunchKernelGGL(sel1_prefix_kernel<true>, dim3(K * g.nblk, B), block, 0, st, Q, Dsum, A, g);                       \
        }                                                                                                                         \
        {                                                                                                                         \
            mlagg_prof::Scope sc_(RR == 2 ? mlagg_prof::K_SEL1_BWD_R2 : mlagg_prof::K_SEL1_BWD, st);                                                                                       \
            hipLaunchKernelGGL(sel1_bwd_kernel<RR>, grid, block, 0, st, tok, dout, idx, dtr, Bs, Cs, Wdt, A, D, bias, Hs, Q, duk,  \
                               dBs, dCs, ddtr, stepws, part, dump, g);                                                            \
        }                                                                                                                         \
    }
    SEL1_DISPATCH_R(R, SEL1_BWD)
#undef SEL1_BWD
    MLAGG_TIMED(K_SEL1_REDUCE, st);
    if (g.nblk > 1)
        hipLaunchKernelGGL(sel1_step_reduce_kernel, dim3((L + 255) / 256, 2 + R, B * K), dim3(256), 0, st, stepws, dBs, dCs, ddtr,
                           g.nblk, R, L);
    const int cols = K * C * (3 + R);
    hipLaunchKernelGGL(mlagg_internal::column_sum_kernel<false>, dim3((cols + 63) / 64), dim3(1024), 0, st, part, B * g.nchunk, cols,
                       cols, dparams);
    return (int)hipGetLastError();
}
