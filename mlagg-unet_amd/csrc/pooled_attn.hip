// K4 -- fused pooled (global-branch) differential attention + RMSNorm for gfx950.
//
// Replaces the `local=False` branch of AggregatedAttention.forward after the pooled keys/values
// exist (reference nnUNetTrainer_MLAgg_2D_dt_MS.py:733-760: four flash_attn_func launches that
// each recompute q k^T, or T:762-777: an eager path that materialises the (B, 2nh, N, P) maps).
// One launch computes both softmax maps of a head pair, their lambda-weighted difference, A V, the
// RMSNorm and the 0.2 gain; K and V of the (batch, head) stay resident in LDS for all tokens of the
// workgroup (P <= 320 keys x 96 floats <= 120 KiB of the 160 KiB LDS).
//
// Forward / backward-1: a DPP lane pair owns (token, head); lane r holds map r's 24 q channels and
// output channels [24r, 24r+24).  Keys stream from LDS as broadcasts (every lane of a wave reads one
// of two addresses), so the loop is pure FMA + 2 v_exp per key.  Two passes over the keys (online
// max/sum, then weights) keep 24 accumulators per lane instead of 2 x 48.
// Backward-2 flips the ownership: a lane owns a KEY (k, v, dk, dv rows in VGPRs) and streams token
// tiles from LDS, so dK/dV need no cross-lane reduction; workgroups cover token chunks and finish
// with one float atomic per element (P x 96 per chunk).
//
// fp32 VALU throughout: at head_dim 24 and P <= 320 the op is HBM/LDS-latency bound in fp32 and the
// f32 MFMA rate equals the VALU rate on gfx950 (MI355X_MICROARCH: 1/16 of bf16), so MFMA buys nothing
// here until the bf16 variant (config 5) lands.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"
#include "internal.h"

namespace {

constexpr int HD = 24;        // head_dim
constexpr int HD2 = 48;
constexpr int TOK_PER_BLOCK = 128;        // forward / backward-1: 256 threads = 128 lane pairs
constexpr int WS_PER_UNIT = 52;           // d(o)[48], D1, D2, pad
constexpr float RMS_EPS = 1e-5f;
constexpr float OUT_GAIN = 0.2f;

struct Geom {
    int batch, N, P, nh, d;
    int q_stride, kp_stride, vp_stride, out_stride;
    float scale;
};

__device__ __forceinline__ float dpp_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

template <int NF>
__device__ __forceinline__ void loadv(const float *__restrict__ p, float (&x)[NF])
{
#pragma unroll
    for (int i = 0; i < NF / 4; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(p + 4 * i);
        x[4 * i] = a.x; x[4 * i + 1] = a.y; x[4 * i + 2] = a.z; x[4 * i + 3] = a.w;
    }
}

template <int NF>
__device__ __forceinline__ void storev(float *__restrict__ p, const float (&x)[NF])
{
#pragma unroll
    for (int i = 0; i < NF / 4; ++i)
        *reinterpret_cast<float4 *>(p + 4 * i) = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
}

template <int NF>
__device__ __forceinline__ float dotv(const float (&a)[NF], const float *__restrict__ lds)
{
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NF / 4; ++i) {
        const float4 b = *reinterpret_cast<const float4 *>(lds + 4 * i);
        acc += a[4 * i] * b.x + a[4 * i + 1] * b.y + a[4 * i + 2] * b.z + a[4 * i + 3] * b.w;
    }
    return acc;
}

template <int NF>
__device__ __forceinline__ void axpyv(float (&acc)[NF], float w, const float *__restrict__ lds)
{
#pragma unroll
    for (int i = 0; i < NF / 4; ++i) {
        const float4 b = *reinterpret_cast<const float4 *>(lds + 4 * i);
        acc[4 * i] += w * b.x; acc[4 * i + 1] += w * b.y; acc[4 * i + 2] += w * b.z; acc[4 * i + 3] += w * b.w;
    }
}

// stage K and V rows of (batch b, head h) into LDS: sK[p][48], sV[p][48]
__device__ __forceinline__ void stage_kv(const Geom &g, const float *__restrict__ kp, const float *__restrict__ vp,
                                         int b, int h, float *sK, float *sV)
{
    const int n4 = g.P * (HD2 / 4);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        const int p = i / (HD2 / 4), c4 = i - p * (HD2 / 4);
        const float4 kk = *reinterpret_cast<const float4 *>(kp + ((size_t)b * g.P + p) * g.kp_stride + h * HD2 + 4 * c4);
        const float4 vv = *reinterpret_cast<const float4 *>(vp + ((size_t)b * g.P + p) * g.vp_stride + h * HD2 + 4 * c4);
        *reinterpret_cast<float4 *>(sK + p * HD2 + 4 * c4) = kk;
        *reinterpret_cast<float4 *>(sV + p * HD2 + 4 * c4) = vv;
    }
}

__global__ void __launch_bounds__(256)
pooled_attn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ kp, const float *__restrict__ vp,
                       const float *__restrict__ lamp, const float *__restrict__ subln_w, float *__restrict__ out,
                       float *__restrict__ lse, float *__restrict__ o_pre, Geom g)
{
    extern __shared__ float4 smem4[];
    float *sK = reinterpret_cast<float *>(smem4);
    float *sV = sK + g.P * HD2;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_kv(g, kp, vp, b, h, sK, sV);
    __syncthreads();
    const int t = blockIdx.x * TOK_PER_BLOCK + (threadIdx.x >> 1);
    const int r = threadIdx.x & 1;
    if (t >= g.N) return;                       // lane pairs leave together
    const float lam = lamp[0];
    const size_t tok = (size_t)b * g.N + t;
    float qv[HD];
    loadv<HD>(q + tok * g.q_stride + h * HD2 + HD * r, qv);
#pragma unroll
    for (int e = 0; e < HD; ++e) qv[e] *= g.scale;
    // pass 1: online max / sum of this lane's map
    float m = -3.0e38f, z = 0.f;
    for (int p = 0; p < g.P; ++p) {
        const float l = dotv<HD>(qv, sK + p * HD2 + HD * r);
        const float mn = fmaxf(m, l);
        z = z * __expf(m - mn) + __expf(l - mn);
        m = mn;
    }
    const float lse_r = m + __logf(z);
    // pass 2: weights and A V for output channels [24r, 24r + 24)
    float o[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) o[e] = 0.f;
    for (int p = 0; p < g.P; ++p) {
        const float s = __expf(dotv<HD>(qv, sK + p * HD2 + HD * r) - lse_r);
        const float so = dpp_xor1(s);
        const float w = r == 0 ? s - lam * so : so - lam * s;
        axpyv<HD>(o, w, sV + p * HD2 + HD * r);
    }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < HD; ++e) ss += o[e] * o[e];
    ss += dpp_xor1(ss);
    const float rstd = rsqrtf(ss * (1.f / HD2) + RMS_EPS);
    if (o_pre) storev<HD>(o_pre + tok * g.d + h * HD2 + HD * r, o);
    if (lse) lse[(tok * g.nh + h) * 2 + r] = lse_r;
    float res[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) res[e] = OUT_GAIN * subln_w[HD * r + e] * o[e] * rstd;
    storev<HD>(out + tok * g.out_stride + h * HD2 + HD * r, res);
}

// backward-1: per (token, head): d(o) through the RMSNorm, the softmax-backward row sums D1 / D2, dq,
// and the parameter gradients d(lambda), d(subln weight).
__global__ void __launch_bounds__(256)
pooled_attn_bwd1_kernel(const float *__restrict__ q, const float *__restrict__ kp, const float *__restrict__ vp,
                        const float *__restrict__ lamp, const float *__restrict__ subln_w,
                        const float *__restrict__ dout, int dout_stride, const float *__restrict__ lse,
                        const float *__restrict__ o_pre, float *__restrict__ dq, int dq_stride,
                        float *__restrict__ ws, float *__restrict__ pgrad, Geom g)
{
    extern __shared__ float4 smem4[];
    float *sK = reinterpret_cast<float *>(smem4);
    float *sV = sK + g.P * HD2;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_kv(g, kp, vp, b, h, sK, sV);
    __syncthreads();
    const int tl = threadIdx.x >> 1;
    const int t = blockIdx.x * TOK_PER_BLOCK + tl;
    const int r = threadIdx.x & 1;
    const bool act = t < g.N;
    const float lam = lamp[0];
    float dwp[HD];                 // d(subln_w) partials of this lane's 24 channels
#pragma unroll
    for (int e = 0; e < HD; ++e) dwp[e] = 0.f;
    float dlam_p = 0.f;
    if (act) {
        const size_t tok = (size_t)b * g.N + t;
        float qv[HD], dO[HD];
        loadv<HD>(q + tok * g.q_stride + h * HD2 + HD * r, qv);
#pragma unroll
        for (int e = 0; e < HD; ++e) qv[e] *= g.scale;
        {
            float ov[HD], dy[HD];
            loadv<HD>(o_pre + tok * g.d + h * HD2 + HD * r, ov);
            loadv<HD>(dout + tok * dout_stride + h * HD2 + HD * r, dy);
            float ss = 0.f, dot = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                ss += ov[e] * ov[e];
                dot += subln_w[HD * r + e] * dy[e] * ov[e];
            }
            ss += dpp_xor1(ss);
            dot += dpp_xor1(dot);
            const float rstd = rsqrtf(ss * (1.f / HD2) + RMS_EPS);
            dot *= (1.f / HD2) * rstd * rstd;
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                dwp[e] = OUT_GAIN * dy[e] * ov[e] * rstd;
                dO[e] = OUT_GAIN * rstd * (subln_w[HD * r + e] * dy[e] - ov[e] * dot);
            }
        }
        const float lse_r = lse[(tok * g.nh + h) * 2 + r];
        // One sweep over the keys:  s = softmax weight, dw_p = d(o) . v_p, ds_1 = dw, ds_2 = -lam dw,
        //   D_r = sum_p ds s,   U = sum_p (s ds) k_p,   V = sum_p s k_p,   and then  dq_r = scale (U - D_r V)
        // (the two-loop form re-evaluated s and dw for every key: 5 length-24 dot products / axpys and 2 exps per key
        // instead of 4 and 1, and read K and V from LDS twice).
        float Dr = 0.f;
        float U[HD], Vv[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) { U[e] = 0.f; Vv[e] = 0.f; }
        for (int p = 0; p < g.P; ++p) {
            const float *kp = sK + p * HD2 + HD * r;
            const float s = __expf(dotv<HD>(qv, kp) - lse_r);
            float dw = dotv<HD>(dO, sV + p * HD2 + HD * r);
            dw += dpp_xor1(dw);
            const float ds = r == 0 ? dw : -lam * dw;
            const float sds = s * ds;
            Dr += sds;
            if (r == 1) dlam_p -= dw * s;
            axpyv<HD>(U, sds, kp);
            axpyv<HD>(Vv, s, kp);
        }
        float dqv[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) dqv[e] = g.scale * (U[e] - Dr * Vv[e]);
        storev<HD>(dq + tok * dq_stride + h * HD2 + HD * r, dqv);
        float *wrow = ws + (tok * g.nh + h) * WS_PER_UNIT;
        storev<HD>(wrow + HD * r, dO);
        wrow[HD2 + r] = Dr;
    }
    // block reduction of d(subln_w) (48 columns) and d(lambda) (column 48) through LDS
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem4);          // [TOK_PER_BLOCK][49], K/V no longer needed
#pragma unroll
    for (int e = 0; e < HD; ++e) red[tl * 49 + HD * r + e] = dwp[e];
    if (r == 1) red[tl * 49 + 48] = dlam_p;
    __syncthreads();
    if (threadIdx.x < 49) {
        float s = 0.f;
        for (int i = 0; i < TOK_PER_BLOCK; ++i) s += red[i * 49 + threadIdx.x];
        // one partial row [d(subln_w) (48) | d(lambda)] per workgroup; the launcher's column sum adds them in a fixed order
        pgrad[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 49 + threadIdx.x] = s;
    }
}

// backward-2: lane = key.  A workgroup of 4 waves covers TPB tokens of one (batch, head, 64-key block);
// each wave streams its share of the tokens through a private LDS tile while dK / dV rows accumulate in
// VGPRs, the 4 waves are summed through LDS and the workgroup writes ONE partial block -- no atomics
// (256 adders per address made the atomic form contention-bound: 0.5 ms per launch in the first profile).
// pooled_attn_bwd2_reduce_kernel sums the partial blocks.
constexpr int B2_TILE = 32;                      // tokens per LDS tile of one wave
constexpr int B2_TILE_FLOATS = B2_TILE * (HD2 + HD2 + 4);
constexpr int B2_TPB = 128;                      // tokens per workgroup (512 left 1 workgroup per CU: every stage took ~85 us)
constexpr int B2_RED_PITCH = 65;                 // [wave][channel][key] reduction image, conflict-free both ways

__global__ void __launch_bounds__(256)
pooled_attn_bwd2_kernel(const float *__restrict__ q, const float *__restrict__ kp, const float *__restrict__ vp,
                        const float *__restrict__ lamp, const float *__restrict__ lse, const float *__restrict__ ws,
                        float *__restrict__ part, Geom g)
{
    // A DPP lane pair owns a key: lane r holds map r's 24 k channels, half of v, and the matching halves of
    // dK / dV (96 persistent VGPRs instead of 192 -> 4 waves per SIMD instead of AGPR spilling at 1).
    // A wave covers 32 keys; waves 0,1 take keys 0..63 for the first half of the workgroup's tokens,
    // waves 2,3 the same keys for the second half.
    __shared__ float4 lds4[2 * B2_TILE_FLOATS / 4 + 2 * HD2 * B2_RED_PITCH / 4 + 4];
    float *lds = reinterpret_cast<float *>(lds4);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = wave >> 1;                         // token half of the workgroup
    const int r = lane & 1;
    float *sQ = lds + half * B2_TILE_FLOATS;            // [tile][48] scaled q   (one tile per token half)
    float *sO = sQ + B2_TILE * HD2;                     // [tile][48] d(o)
    float *sS = sO + B2_TILE * HD2;                     // [tile][4]  lse1, lse2, D1, D2
    const int h = blockIdx.y, b = blockIdx.z;
    const int pblocks = (g.P + 63) / 64;
    const int pb = blockIdx.x % pblocks, tb = blockIdx.x / pblocks;
    const int p = pb * 64 + (wave & 1) * 32 + (lane >> 1);
    const bool act = p < g.P;
    const float lam = lamp[0];
    float kr[HD], vh[HD], dk[HD], dv[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) { kr[e] = 0.f; vh[e] = 0.f; dk[e] = 0.f; dv[e] = 0.f; }
    if (act) {
        loadv<HD>(kp + ((size_t)b * g.P + p) * g.kp_stride + h * HD2 + HD * r, kr);
        loadv<HD>(vp + ((size_t)b * g.P + p) * g.vp_stride + h * HD2 + HD * r, vh);
    }
    const int t_begin = tb * B2_TPB + half * (B2_TPB / 2);
    const int t_end = min(t_begin + B2_TPB / 2, g.N);
    const int ltid = threadIdx.x & 127;                 // the two waves of a token half stage its tile together
    for (int t0 = t_begin; t0 < t_begin + B2_TPB / 2; t0 += B2_TILE) {     // uniform trip count: barriers inside
        const int nt = max(0, min(B2_TILE, t_end - t0));
        __syncthreads();
        for (int i = ltid; i < nt * (HD2 / 4); i += 128) {
            const int tt = i / (HD2 / 4), c4 = i - tt * (HD2 / 4);
            const size_t tok = (size_t)b * g.N + t0 + tt;
            float4 a = *reinterpret_cast<const float4 *>(q + tok * g.q_stride + h * HD2 + 4 * c4);
            a.x *= g.scale; a.y *= g.scale; a.z *= g.scale; a.w *= g.scale;
            *reinterpret_cast<float4 *>(sQ + tt * HD2 + 4 * c4) = a;
            *reinterpret_cast<float4 *>(sO + tt * HD2 + 4 * c4) =
                *reinterpret_cast<const float4 *>(ws + (tok * g.nh + h) * WS_PER_UNIT + 4 * c4);
        }
        if (ltid < nt) {
            const size_t tok = (size_t)b * g.N + t0 + ltid;
            const float *wrow = ws + (tok * g.nh + h) * WS_PER_UNIT;
            *reinterpret_cast<float4 *>(sS + 4 * ltid) =
                make_float4(lse[(tok * g.nh + h) * 2], lse[(tok * g.nh + h) * 2 + 1], wrow[HD2], wrow[HD2 + 1]);
        }
        __syncthreads();
        for (int tt = 0; tt < nt; ++tt) {
            const float4 st = *reinterpret_cast<const float4 *>(sS + 4 * tt);
            const float *qt = sQ + tt * HD2 + HD * r, *ot = sO + tt * HD2 + HD * r;
            float la[4] = {0.f, 0.f, 0.f, 0.f}, da[4] = {0.f, 0.f, 0.f, 0.f};
            float qv[HD], ov[HD];
#pragma unroll
            for (int e = 0; e < HD; e += 4) {
                const float4 q4 = *reinterpret_cast<const float4 *>(qt + e);
                const float4 o4 = *reinterpret_cast<const float4 *>(ot + e);
                qv[e] = q4.x; qv[e + 1] = q4.y; qv[e + 2] = q4.z; qv[e + 3] = q4.w;
                ov[e] = o4.x; ov[e + 1] = o4.y; ov[e + 2] = o4.z; ov[e + 3] = o4.w;
            }
#pragma unroll
            for (int e = 0; e < HD; e += 4)
#pragma unroll
                for (int j = 0; j < 4; ++j) { la[j] += qv[e + j] * kr[e + j]; da[j] += ov[e + j] * vh[e + j]; }
            const float l = (la[0] + la[1]) + (la[2] + la[3]);          // this lane's map logit
            float dw = (da[0] + da[1]) + (da[2] + da[3]);
            dw += dpp_xor1(dw);                                          // d(o) . v over all 48 channels
            const float s = __expf(l - (r ? st.y : st.x));
            const float so = dpp_xor1(s);
            const float w = r ? so - lam * s : s - lam * so;             // s1 - lam s2 on both lanes
            const float dl = r ? s * (-lam * dw - st.w) : s * (dw - st.z);   // dL/d(logit of this lane's map)
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                dk[e] += dl * qv[e];        // q tile is pre-scaled: d(logit)/dk = scaled q row
                dv[e] += w * ov[e];
            }
        }
    }
    // sum the two token halves through LDS, dK then dV; image [half][channel 0..47][key 0..63], pitch 65
    float *red = lds + 2 * B2_TILE_FLOATS;
    float *pbase = part + ((((size_t)b * g.nh + h) * gridDim.x + blockIdx.x) * 2) * (64 * HD2);
    const int kl = (wave & 1) * 32 + (lane >> 1);       // key inside the 64-key block
#pragma unroll 1
    for (int kind = 0; kind < 2; ++kind) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < HD; ++e) red[((half * HD2) + HD * r + e) * B2_RED_PITCH + kl] = kind ? dv[e] : dk[e];
        __syncthreads();
        for (int o = threadIdx.x; o < 64 * HD2; o += 256) {
            const int pp = o / HD2, e = o - pp * HD2;
            pbase[(size_t)kind * (64 * HD2) + o] = red[e * B2_RED_PITCH + pp] + red[(HD2 + e) * B2_RED_PITCH + pp];
        }
    }
}

// part[(b, h)][block][kind][64 keys][48] -> dkp / dvp (batch, P, d); one thread per output element
__global__ void pooled_attn_bwd2_reduce_kernel(const float *__restrict__ part, int nblocks, int pblocks,
                                               float *__restrict__ dkp, int dkp_stride, float *__restrict__ dvp,
                                               int dvp_stride, Geom g)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over P * 48
    const int h = blockIdx.y, b = blockIdx.z;
    if (idx >= g.P * HD2) return;
    const int p = idx / HD2, e = idx - p * HD2;
    const int pb = p >> 6, pl = p & 63;
    const float *base = part + (((size_t)b * g.nh + h) * nblocks) * 2 * (64 * HD2);
    float sk = 0.f, sv = 0.f;
    for (int blk = pb; blk < nblocks; blk += pblocks) {
        const float *pr = base + (size_t)blk * 2 * (64 * HD2) + pl * HD2 + e;
        sk += pr[0];
        sv += pr[64 * HD2];
    }
    dkp[((size_t)b * g.P + p) * dkp_stride + h * HD2 + e] = sk;
    dvp[((size_t)b * g.P + p) * dvp_stride + h * HD2 + e] = sv;
}

int make_geom(Geom &g, int batch, int N, int P, int nh, int qs, int kps, int vps, int outs, float scale)
{
    if (batch <= 0 || N <= 0 || P <= 0 || nh <= 0 || batch > 65535 || nh > 65535) return MLAGG_E_UNSUPPORTED;
    g.batch = batch; g.N = N; g.P = P; g.nh = nh; g.d = nh * HD2;
    g.q_stride = qs; g.kp_stride = kps; g.vp_stride = vps; g.out_stride = outs; g.scale = scale;
    if (qs < g.d || kps < g.d || vps < g.d || outs < g.d || ((qs | kps | vps | outs) & 3)) return MLAGG_E_UNSUPPORTED;
    if (scale == 0.f) return MLAGG_E_UNSUPPORTED;
    return 0;
}

size_t kv_lds_bytes(const Geom &g, bool with_reduce)
{
    size_t fl = (size_t)2 * g.P * HD2;
    if (with_reduce && fl < (size_t)TOK_PER_BLOCK * 49) fl = (size_t)TOK_PER_BLOCK * 49;
    return fl * sizeof(float);
}

template <typename K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes > 160 * 1024) return MLAGG_E_UNSUPPORTED;      // P > ~420 keys: not on the MLAgg path
    if (bytes > 48 * 1024)
        return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return 0;
}

}  // namespace

extern "C" int mlagg_pooled_attn_fwd(const float *q, int q_stride, const float *kp, int kp_stride, const float *vp,
                                     int vp_stride, const float *lam, const float *subln_w, float *out,
                                     int out_stride, float *lse, float *o_pre, int batch, int N, int P, int nh,
                                     float scale, void *stream)
{
    if (!q || !kp || !vp || !lam || !subln_w || !out) return MLAGG_E_NULLPTR;
    Geom g;
    if (int rc = make_geom(g, batch, N, P, nh, q_stride, kp_stride, vp_stride, out_stride, scale)) return rc;
    const size_t lds = kv_lds_bytes(g, false);
    if (int rc = allow_lds(pooled_attn_fwd_kernel, lds)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    { MLAGG_TIMED(K_POOLED_FWD, st); hipLaunchKernelGGL(pooled_attn_fwd_kernel, dim3((N + TOK_PER_BLOCK - 1) / TOK_PER_BLOCK, nh, batch), dim3(256),
                       lds, st, q, kp, vp, lam, subln_w, out, lse, o_pre, g); }
    return (int)hipGetLastError();
}

extern "C" size_t mlagg_pooled_attn_bwd_workspace_floats(int batch, int N, int P, int nh)
{
    const size_t nblocks = (size_t)((P + 63) / 64) * ((N + B2_TPB - 1) / B2_TPB);
    return (size_t)batch * N * nh * WS_PER_UNIT + (size_t)batch * nh * nblocks * 2 * (64 * HD2) +
           (size_t)batch * nh * ((N + TOK_PER_BLOCK - 1) / TOK_PER_BLOCK) * 49;
}

extern "C" int mlagg_pooled_attn_bwd(const float *q, int q_stride, const float *kp, int kp_stride, const float *vp,
                                     int vp_stride, const float *lam, const float *subln_w, const float *dout,
                                     int dout_stride, const float *lse, const float *o_pre, float *dq,
                                     int dq_stride, float *dkp, int dkp_stride, float *dvp, int dvp_stride,
                                     float *dlam, float *dsubln_w, float *workspace, int batch, int N, int P,
                                     int nh, float scale, void *stream)
{
    if (!q || !kp || !vp || !lam || !subln_w || !dout || !lse || !o_pre || !dq || !dkp || !dvp || !dlam ||
        !dsubln_w || !workspace)
        return MLAGG_E_NULLPTR;
    Geom g;
    if (int rc = make_geom(g, batch, N, P, nh, q_stride, kp_stride, vp_stride, dout_stride, scale)) return rc;
    if (dq_stride < g.d || dkp_stride < g.d || dvp_stride < g.d || ((dq_stride | dkp_stride | dvp_stride) & 3))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = kv_lds_bytes(g, true);
    if (int rc = allow_lds(pooled_attn_bwd1_kernel, lds)) return rc;
    const int pblocks = (P + 63) / 64;
    const int tblocks = (N + B2_TPB - 1) / B2_TPB;
    float *part = workspace + (size_t)batch * N * nh * WS_PER_UNIT;
    float *pgrad = part + (size_t)batch * nh * pblocks * tblocks * 2 * (64 * HD2);
    const int nb1 = (N + TOK_PER_BLOCK - 1) / TOK_PER_BLOCK;
    { MLAGG_TIMED(K_POOLED_BWD1, st); hipLaunchKernelGGL(pooled_attn_bwd1_kernel, dim3(nb1, nh, batch), dim3(256),
                       lds, st, q, kp, vp, lam, subln_w, dout, dout_stride, lse, o_pre, dq, dq_stride, workspace, pgrad, g);
      hipLaunchKernelGGL(mlagg_internal::column_sum_split_kernel<0>, dim3(1), dim3(1024), 0, st, pgrad, nb1 * nh * batch, 49, 49, 48,
                         dsubln_w, dlam); }
    {
        MLAGG_TIMED(K_POOLED_BWD2, st);
        hipLaunchKernelGGL(pooled_attn_bwd2_kernel, dim3(pblocks * tblocks, nh, batch), dim3(256), 0, st, q, kp, vp,
                           lam, lse, workspace, part, g);
        hipLaunchKernelGGL(pooled_attn_bwd2_reduce_kernel, dim3((P * HD2 + 255) / 256, nh, batch), dim3(256), 0, st,
                           part, pblocks * tblocks, pblocks, dkp, dkp_stride, dvp, dvp_stride, g);
    }
    return (int)hipGetLastError();
}
