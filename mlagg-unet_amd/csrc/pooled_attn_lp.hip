// K4lp -- pooled (global-branch) differential attention on the 16-bit matrix cores: the form of K4 that runs under the reference's
// autocast step (nnUNetTrainer.py:848; BASELINE configs[2] bf16, configs[4] fp16 + the shipped flash path).
//
// Replaces the four `flash_attn_func(q_j, k_j, v_i)` launches per pooled branch of AggregatedAttention.forward
// (nnUNetTrainer_MLAgg_2D_dt_MS.py:733-751: fp16 / bf16 tensors, head_dim 24, 16..320 pooled keys), the lambda-weighted
// subtraction (:753-756), the RMSNorm (:757-759) and the 0.2 gain (:760) -- and their backward -- by three MFMA kernels around
// K / V tiles resident in LDS.  Operands (q * scale, k, v, softmax weights, d(o)) are rounded to fp16 / bf16 on their way into
// LDS / registers -- what flash-attn's 16-bit tensors hold -- products run on v_mfma_f32_32x32x16_{f16,bf16} with fp32 sums,
// softmax, RMSNorm and every tensor in HBM stay fp32.  The fp32 K4 (csrc/pooled_attn.hip, VALU) remains the fp32-mode kernel.
//
// Score layout.  Forward and the query-gradient kernel compute the TRANSPOSED tile S^T = K (Q scale)^T: the accumulator of lane
// (token column j, half h) then holds 16 keys of ONE token, so the softmax row statistics are in-lane sums and the weights can be
// fed straight back as the B operand of the next product (O^T = V^T W^T, dq^T = K^T dS^T) -- the 8 accumulator registers of an
// instruction are 8 keys in a fixed permuted order, and the LDS images of V^T / K^T are stored in that order at staging time.
// The key-gradient kernel computes S = Q K^T (lane = key column): there the token is the contraction index of
// dV^T = d(o)^T W and dK^T = Q^T dS, and the per-token constants (log-sum-exp, D) are read per accumulator row.
// The softmax-backward row sums need no pass over the keys: with o1 = P1 V and o2 = P2 V saved by the forward,
// D1 = sum_k P1 dP1 = d(o) . o1, D2 = sum_k P2 dP2 = -lambda d(o) . o2 and d(lambda) = -sum_t d(o) . o2 are per-token dot products
// formed in the RMSNorm-backward prologue.
//
// Roofline: matrix-core / exp bound (6 * 48 * P flop and 4 P exponentials per token and head forward), not HBM.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "internal.h"
#include "mlagg_hip.h"
#include "prof.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int HD = 24, HD2 = 48;
constexpr int KP = 40;                  // u16 per row of a [row][32 k] operand image: 80 bytes, conflict-free 16-byte reads
constexpr int VP = 56;                  // u16 per row of a [row][48 k] operand image: 112 bytes
constexpr int WS_PER_UNIT = 52;         // d(o)[48], D1, D2, pad  (per token and head)
constexpr int TOK_WG = 256;             // tokens per workgroup of the token-parallel kernels: 8 waves x 32
constexpr float RMS_EPS = 1e-5f, OUT_GAIN = 0.2f, NEG = -3.0e38f;

struct LGeom {
    int batch, N, P, nh, d, ntile;
    int q_stride, kp_stride, vp_stride, out_stride;
    float scale;
};

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
__device__ __forceinline__ u16 cvt1(float a) { return (u16)(pack2<BF16>(a, 0.f) & 0xffff); }

template <bool BF16>
__device__ __forceinline__ uint4 pack8(const float *v)
{
    return make_uint4(pack2<BF16>(v[0], v[1]), pack2<BF16>(v[2], v[3]), pack2<BF16>(v[4], v[5]), pack2<BF16>(v[6], v[7]));
}

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(const uint4 &a, const uint4 &b, f32x16 c)
{
    if (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a), *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row index of accumulator register v in lane half kh (D layout of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int v, int kh) { return (v & 3) + 8 * (v >> 2) + 4 * kh; }

// position of contraction element `within` (0..31) of a 32-wide block in the PERMUTED operand images: instruction q (0, 1), lane
// half h, element e (0..7) -- the order in which the accumulator registers 8q..8q+7 of half h hold their rows
__device__ __forceinline__ void perm_pos(int within, int &q, int &h, int &e)
{
    const int m = within >> 3, n = within & 3;
    h = (within >> 2) & 1;
    q = m >> 1;
    e = ((m & 1) << 2) | n;
}

// ---- LDS images of K and V of one (batch, head) ------------------------------------------------------------------------------
// sKrow [2 maps][Ppad][KP]   : row = key, k = dim (24, zero-padded to 32)        A operand of S^T = K Q^T
// sVrow [Ppad][VP]           : row = key, k = value channel (48)                 A operand of dW^T = V d(o)^T
// sKt   [2][ntile][2][2][32][8]: row = dim, contraction = key (permuted)         A operand of dq^T = K^T dS^T
// sVt   [2 mt][ntile][2][2][32][8]: row = value channel (48 in two 32-row blocks), contraction = key (permuted)   A of O^T = V^T W^T
template <bool BF16, bool ROWS, bool VROWS, bool KT, bool VT>
__device__ __forceinline__ void stage_kv(const LGeom &g, const float *__restrict__ kp, const float *__restrict__ vp, int b, int h,
                                         u16 *sKrow, u16 *sVrow, u16 *sKt, u16 *sVt)
{
    const int Ppad = g.ntile * 32;
    for (int i = threadIdx.x; i < Ppad * 16; i += blockDim.x) {           // 16 groups of 4 channels per key: 8 of K (2 x 32), 12 of V
        const int p = i >> 4, c4 = i & 15;
        const bool ok = p < g.P;
        const int tile = p >> 5;
        int q, hh, e;
        perm_pos(p & 31, q, hh, e);
        if (c4 < 16) {                                                    // K: map r = c4 / 8, dims 4 (c4 % 8) .. + 4 (>= 24: zero)
            const int r = c4 >> 3, d0 = 4 * (c4 & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && d0 < HD) v = *reinterpret_cast<const float4 *>(kp + ((size_t)b * g.P + p) * g.kp_stride + h * HD2 + HD * r + d0);
            if (ROWS)
                *reinterpret_cast<uint2 *>(sKrow + ((size_t)r * Ppad + p) * KP + d0) = make_uint2(pack2<BF16>(v.x, v.y), pack2<BF16>(v.z, v.w));
            if (KT) {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sKt[(((((size_t)r * g.ntile + tile) * 2 + q) * 2 + hh) * 32 + d0 + j) * 8 + e] = cvt1<BF16>(vv[j]);
            }
        }
    }
    for (int i = threadIdx.x; i < Ppad * 16; i += blockDim.x) {           // V: channels 4 c4 .. + 4 (>= 48: zero)
        const int p = i >> 4, c4 = i & 15;
        const bool ok = p < g.P;
        const int tile = p >> 5, d0 = 4 * c4;
        int q, hh, e;
        perm_pos(p & 31, q, hh, e);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && d0 < HD2) v = *reinterpret_cast<const float4 *>(vp + ((size_t)b * g.P + p) * g.vp_stride + h * HD2 + d0);
        if (VROWS && d0 < HD2)
            *reinterpret_cast<uint2 *>(sVrow + (size_t)p * VP + d0) = make_uint2(pack2<BF16>(v.x, v.y), pack2<BF16>(v.z, v.w));
        if (VT) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int vd = d0 + j, mt = vd >> 5;
                sVt[(((((size_t)mt * g.ntile + tile) * 2 + q) * 2 + hh) * 32 + (vd & 31)) * 8 + e] = cvt1<BF16>(vv[j]);
            }
        }
    }
}

// the B operand Q^T of one token: lane (token, kh) holds dims 16 p + 8 kh .. + 8 of map r, scaled, rounded; dims >= 24 are zero
template <bool BF16>
__device__ __forceinline__ void load_q_operand(const float *__restrict__ qrow, float scale, int kh, uint4 (&qB)[2][2])
{
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int d0 = 16 * p + 8 * kh;
            float v[8];
            if (d0 < HD) {
                const float4 a = *reinterpret_cast<const float4 *>(qrow + HD * r + d0), c = *reinterpret_cast<const float4 *>(qrow + HD * r + d0 + 4);
                v[0] = a.x * scale; v[1] = a.y * scale; v[2] = a.z * scale; v[3] = a.w * scale;
                v[4] = c.x * scale; v[5] = c.y * scale; v[6] = c.z * scale; v[7] = c.w * scale;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            qB[r][p] = pack8<BF16>(v);
        }
}

// S^T tile of map r: rows = keys tile * 32 .., column = the lane's token; keys >= P masked to -inf
template <bool BF16>
__device__ __forceinline__ f32x16 score_t(const u16 *sKrow, const LGeom &g, int r, int tile, int col, int kh, const uint4 (&qB)[2][2])
{
    const u16 *row = sKrow + ((size_t)r * g.ntile * 32 + tile * 32 + col) * KP + 8 * kh;
    f32x16 s = mfma16<BF16>(*reinterpret_cast<const uint4 *>(row), qB[r][0], zero16());
    s = mfma16<BF16>(*reinterpret_cast<const uint4 *>(row + 16), qB[r][1], s);
    if (tile * 32 + 32 > g.P) {
#pragma unroll
        for (int v = 0; v < 16; ++v)
            if (tile * 32 + acc_row(v, kh) >= g.P) s[v] = NEG;
    }
    return s;
}

// ------------------------------------------------------------------------------------------------------------------------------
// forward: out = 0.2 w RMSNorm(o1 - lambda o2), o_r = softmax(q_r k_r^T scale) v; saves lse (B, N, nh, 2), o1, o2 (B, N, d)
// ------------------------------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ void __launch_bounds__(512)
pooled_lp_fwd_kernel(const float *__restrict__ q, const float *__restrict__ kp, const float *__restrict__ vp,
                     const float *__restrict__ lamp, const float *__restrict__ subln_w, float *__restrict__ out,
                     float *__restrict__ lse, float *__restrict__ o1, float *__restrict__ o2, LGeom g)
{
    extern __shared__ uint4 smem[];
    u16 *sKrow = reinterpret_cast<u16 *>(smem);
    u16 *sVt = sKrow + (size_t)2 * g.ntile * 32 * KP;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_kv<BF16, true, false, false, true>(g, kp, vp, b, h, sKrow, nullptr, nullptr, sVt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, kh = lane >> 5;
    const int t = blockIdx.x * TOK_WG + wave * 32 + col;
    if (blockIdx.x * TOK_WG + wave * 32 >= g.N) return;                   // whole waves leave together
    const bool valid = t < g.N;
    const size_t tok = (size_t)b * g.N + min(t, g.N - 1);
    uint4 qB[2][2];
    load_q_operand<BF16>(q + tok * g.q_stride + h * HD2, g.scale, kh, qB);
    // pass 1: log-sum-exp of both maps
    float lser[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float m = NEG, l = 0.f;
        for (int tile = 0; tile < g.ntile; ++tile) {
            const f32x16 s = score_t<BF16>(sKrow, g, r, tile, col, kh, qB);
            float mx = s[0];
#pragma unroll
            for (int v = 1; v < 16; ++v) mx = fmaxf(mx, s[v]);
            const float mn = fmaxf(m, mx);
            float a = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) a += __expf(s[v] - mn);
            l = l * __expf(m - mn) + a;
            m = mn;
        }
        const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, mo);
        lser[r] = M + __logf(l * __expf(m - M) + lo * __expf(mo - M));
    }
    // pass 2: o_r^T = V^T softmax^T
    f32x16 oacc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) { oacc[r][0] = zero16(); oacc[r][1] = zero16(); }
    for (int tile = 0; tile < g.ntile; ++tile) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const f32x16 s = score_t<BF16>(sKrow, g, r, tile, col, kh, qB);
            float w[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) w[v] = __expf(s[v] - lser[r]);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const uint4 wB = pack8<BF16>(w + 8 * qq);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(sVt + (((((size_t)mt * g.ntile + tile) * 2 + qq) * 2 + kh) * 32 + col) * 8);
                    oacc[r][mt] = mfma16<BF16>(a, wB, oacc[r][mt]);
                }
            }
        }
    }
    // epilogue: rows of oacc[r][mt] are value channels 32 mt + acc_row(v, kh) (mt = 1: rows < 16 only), column = the token
    const float lam = lamp[0];
    float ss = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int v = 0; v < (mt ? 8 : 16); ++v) {
            const float o = oacc[0][mt][v] - lam * oacc[1][mt][v];
            ss += o * o;
        }
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = rsqrtf(ss * (1.f / HD2) + RMS_EPS);
    if (valid) {
        const size_t ob = tok * g.out_stride + h * HD2, sb = tok * g.d + h * HD2;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < (mt ? 2 : 4); ++g4) {
                const int vd = 32 * mt + 8 * g4 + 4 * kh;
                float4 a, c, y;
                a.x = oacc[0][mt][4 * g4]; a.y = oacc[0][mt][4 * g4 + 1]; a.z = oacc[0][mt][4 * g4 + 2]; a.w = oacc[0][mt][4 * g4 + 3];
                c.x = oacc[1][mt][4 * g4]; c.y = oacc[1][mt][4 * g4 + 1]; c.z = oacc[1][mt][4 * g4 + 2]; c.w = oacc[1][mt][4 * g4 + 3];
                const float4 wv = *reinterpret_cast<const float4 *>(subln_w + vd);
                y.x = OUT_GAIN * wv.x * (a.x - lam * c.x) * rstd; y.y = OUT_GAIN * wv.y * (a.y - lam * c.y) * rstd;
                y.z = OUT_GAIN * wv.z * (a.z - lam * c.z) * rstd; y.w = OUT_GAIN * wv.w * (a.w - lam * c.w) * rstd;
                *reinterpret_cast<float4 *>(out + ob + vd) = y;
                if (o1) { *reinterpret_cast<float4 *>(o1 + sb + vd) = a; *reinterpret_cast<float4 *>(o2 + sb + vd) = c; }
            }
        if (lse && kh == 0) { lse[(tok * g.nh + h) * 2] = lser[0]; lse[(tok * g.nh + h) * 2 + 1] = lser[1]; }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward prologue (per token and head, fp32): d(o) through the 0.2 gain and the RMSNorm, D1, D2, and per-workgroup partial rows
// [d(subln_w) (48) | d(lambda)] (summed in a fixed order by the launcher)
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pooled_lp_prep_kernel(const float *__restrict__ dout, int dout_stride, const float *__restrict__ o1, const float *__restrict__ o2,
                      const float *__restrict__ lamp, const float *__restrict__ subln_w, float *__restrict__ ws,
                      float *__restrict__ pgrad, LGeom g)
{
    __shared__ float red[4][49];
    const long unit = (long)blockIdx.x * 256 + threadIdx.x, nunit = (long)g.batch * g.N * g.nh;
    const float lam = lamp[0];
    float dw[HD2], dl = 0.f;
#pragma unroll
    for (int e = 0; e < HD2; ++e) dw[e] = 0.f;
    if (unit < nunit) {
        const long tok = unit / g.nh;
        const int h = (int)(unit - tok * g.nh);
        const float *gp = dout + tok * dout_stride + h * HD2, *p1 = o1 + tok * g.d + h * HD2, *p2 = o2 + tok * g.d + h * HD2;
        float o[HD2], a2[HD2], gy[HD2];
        float ss = 0.f, sgo = 0.f;
#pragma unroll
        for (int i = 0; i < HD2 / 4; ++i) {
            const float4 a = *reinterpret_cast<const float4 *>(p1 + 4 * i), c = *reinterpret_cast<const float4 *>(p2 + 4 * i);
            const float4 gq = *reinterpret_cast<const float4 *>(gp + 4 * i), wv = *reinterpret_cast<const float4 *>(subln_w + 4 * i);
            const float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * i + j;
                o[e] = av[j] - lam * cv[j];
                a2[e] = cv[j];
                gy[e] = gv[j];                               // raw upstream gradient; c_e = 0.2 w_e applied below
                ss += o[e] * o[e];
                sgo += OUT_GAIN * ww[j] * gv[j] * o[e];
            }
        }
        const float var = ss * (1.f / HD2) + RMS_EPS, rstd = rsqrtf(var);
        const float k = sgo * rstd * rstd * (1.f / HD2);
        float D1 = 0.f, D2 = 0.f;
        float *wrow = ws + unit * WS_PER_UNIT;
#pragma unroll
        for (int i = 0; i < HD2 / 4; ++i) {
            const float4 a = *reinterpret_cast<const float4 *>(p1 + 4 * i), wv = *reinterpret_cast<const float4 *>(subln_w + 4 * i);
            const float av[4] = {a.x, a.y, a.z, a.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
            float dO[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * i + j;
                dO[j] = rstd * (OUT_GAIN * ww[j] * gy[e] - o[e] * k);
                dw[e] = OUT_GAIN * gy[e] * o[e] * rstd;
                D1 += dO[j] * av[j];
                D2 += dO[j] * a2[e];
            }
            *reinterpret_cast<float4 *>(wrow + 4 * i) = make_float4(dO[0], dO[1], dO[2], dO[3]);
        }
        dl = -D2;                                             // d(lambda) = -d(o) . o2
        wrow[HD2] = D1;
        wrow[HD2 + 1] = -lam * D2;                            // D2 of the softmax backward: sum_k P2 dP2 with dP2 = -lambda dW
    }
    // workgroup partial of d(subln_w), d(lambda): wave butterflies, one LDS row per wave, fixed-order sum
#pragma unroll
    for (int e = 0; e < HD2; ++e) {
        float v = dw[e];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = v;
    }
    {
        float v = dl;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][HD2] = v;
    }
    __syncthreads();
    if (threadIdx.x < 49) pgrad[(size_t)blockIdx.x * 49 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward, query side (S^T layout): dq_r = scale * dS_r K_r,  dS_r = P_r (dP_r - D_r),  dP_1 = dW, dP_2 = -lambda dW, dW = d(o) V^T
// ------------------------------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ void __launch_bounds__(512)
pooled_lp_bwd_q_kernel(const float *__restrict__ q, const float *__restrict__ kp, const float *__restrict__ vp,
                       const float *__restrict__ lamp, const float *__restrict__ lse, const float *__restrict__ ws,
                       float *__restrict__ dq, int dq_stride, LGeom g)
{
    extern __shared__ uint4 smem[];
    u16 *sKrow = reinterpret_cast<u16 *>(smem);
    u16 *sVrow = sKrow + (size_t)2 * g.ntile * 32 * KP;
    u16 *sKt = sVrow + (size_t)g.ntile * 32 * VP;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_kv<BF16, true, true, true, false>(g, kp, vp, b, h, sKrow, sVrow, sKt, nullptr);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, kh = lane >> 5;
    const int t = blockIdx.x * TOK_WG + wave * 32 + col;
    if (blockIdx.x * TOK_WG + wave * 32 >= g.N) return;
    const bool valid = t < g.N;
    const size_t tok = (size_t)b * g.N + min(t, g.N - 1);
    const size_t unit = tok * g.nh + h;
    uint4 qB[2][2];
    load_q_operand<BF16>(q + tok * g.q_stride + h * HD2, g.scale, kh, qB);
    const float lam = lamp[0];
    const float lser[2] = {lse[unit * 2], lse[unit * 2 + 1]};
    const float *wrow = ws + unit * WS_PER_UNIT;
    const float Dr[2] = {wrow[HD2], wrow[HD2 + 1]};
    uint4 dOB[3];                                              // B operand d(o)^T: value channels 16 p + 8 kh .. + 8 of the lane's token
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float4 a = *reinterpret_cast<const float4 *>(wrow + 16 * p + 8 * kh), c = *reinterpret_cast<const float4 *>(wrow + 16 * p + 8 * kh + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        dOB[p] = pack8<BF16>(v);
    }
    f32x16 dqacc[2] = {zero16(), zero16()};
    for (int tile = 0; tile < g.ntile; ++tile) {
        // dW^T tile: rows = keys, k = value channel
        const u16 *vrow = sVrow + (size_t)(tile * 32 + col) * VP + 8 * kh;
        f32x16 dW = mfma16<BF16>(*reinterpret_cast<const uint4 *>(vrow), dOB[0], zero16());
        dW = mfma16<BF16>(*reinterpret_cast<const uint4 *>(vrow + 16), dOB[1], dW);
        dW = mfma16<BF16>(*reinterpret_cast<const uint4 *>(vrow + 32), dOB[2], dW);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const f32x16 s = score_t<BF16>(sKrow, g, r, tile, col, kh, qB);
            const float sg = r == 0 ? 1.f : -lam;
            float ds[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) ds[v] = __expf(s[v] - lser[r]) * (sg * dW[v] - Dr[r]);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const uint4 dB = pack8<BF16>(ds + 8 * qq);
                const uint4 a = *reinterpret_cast<const uint4 *>(sKt + (((((size_t)r * g.ntile + tile) * 2 + qq) * 2 + kh) * 32 + col) * 8);
                dqacc[r] = mfma16<BF16>(a, dB, dqacc[r]);
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g4 = 0; g4 < 3; ++g4) {                   // rows (dims) 8 g4 + 4 kh .. + 4 < 24
                const int dim = 8 * g4 + 4 * kh;
                *reinterpret_cast<float4 *>(dq + tok * dq_stride + h * HD2 + HD * r + dim) =
                    make_float4(g.scale * dqacc[r][4 * g4], g.scale * dqacc[r][4 * g4 + 1], g.scale * dqacc[r][4 * g4 + 2], g.scale * dqacc[r][4 * g4 + 3]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward, key side (S layout, lane = key): a wave owns one 32-key tile, the workgroup a chunk of tokens staged 32 at a time:
//   dV^T += d(o)^T W,  W = P1 - lambda P2;   dK_r^T += (Q_r scale)^T dS_r.   Partial blocks part[b][h][chunk][key][96] = dK (2 x 24) | dV (48)
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int KV_TOK = 32;
template <bool BF16>
__global__ void __launch_bounds__(640)
pooled_lp_bwd_kv_kernel(const float *__restrict__ q, const float *__restrict__ kp, const float *__restrict__ vp,
                        const float *__restrict__ lamp, const float *__restrict__ lse, const float *__restrict__ ws,
                        float *__restrict__ part, int chunk_tokens, int nchunk, LGeom g)
{
    __shared__ __attribute__((aligned(16))) u16 sQ[2][KV_TOK][KP];                // row = token, k = dim                (A of S)
    __shared__ __attribute__((aligned(16))) u16 sdO[KV_TOK][VP];                  // row = token, k = value channel      (A of dW)
    __shared__ __attribute__((aligned(16))) u16 sQt[2][2][2][32][8];              // row = dim, contraction = token      (A of dK^T)
    __shared__ __attribute__((aligned(16))) u16 sdOt[2][2][2][32][8];             // row = value channel, contr. = token (A of dV^T)
    __shared__ __attribute__((aligned(16))) float sC[4][KV_TOK];                  // lse1, lse2, D1, D2 per token
    const int h = blockIdx.y, b = blockIdx.z, chunk = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, kh = lane >> 5;
    const int tile = wave;                                                       // blockDim = 64 * ntile
    const int key = tile * 32 + col;
    const float lam = lamp[0];
    // B operands of this wave's keys: K_r^T (k = dim), V^T (k = value channel)
    uint4 kB[2][2], vB[3];
    {
        const bool ok = key < g.P;
        const float *krow = kp + ((size_t)b * g.P + min(key, g.P - 1)) * g.kp_stride + h * HD2;
        const float *vrow = vp + ((size_t)b * g.P + min(key, g.P - 1)) * g.vp_stride + h * HD2;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int d0 = 16 * p + 8 * kh;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (ok && d0 + j < HD) ? krow[HD * r + min(d0 + j, HD - 1)] : 0.f;
                kB[r][p] = pack8<BF16>(v);
            }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? vrow[16 * p + 8 * kh + j] : 0.f;
            vB[p] = pack8<BF16>(v);
        }
    }
    // dims 24..31 of the token-row q image are contraction positions that the staging below never writes: they must read as zero
    // (0 x NaN = NaN); the pad ROWS of the transposed images only feed accumulator rows that are never stored
    for (int i = threadIdx.x; i < 2 * KV_TOK * 8; i += blockDim.x) sQ[i / (KV_TOK * 8)][(i / 8) % KV_TOK][HD + (i & 7)] = 0;
    f32x16 dKacc[2] = {zero16(), zero16()}, dVacc[2] = {zero16(), zero16()};
    const int t_lo = chunk * chunk_tokens, t_hi = min(g.N, t_lo + chunk_tokens);
    for (int t0 = t_lo; t0 < t_hi; t0 += KV_TOK) {
        __syncthreads();                                                         // the previous tile's operand reads are done
        // stage 32 tokens: 2 x 24 q dims (scaled) + 48 d(o) channels + 4 constants each; tokens past the chunk are zero / lse = +inf
        for (int i = threadIdx.x; i < KV_TOK * 28; i += blockDim.x) {
            const int tt = i / 28, c = i - tt * 28, t = t0 + tt;
            const bool ok = t < t_hi;
            const size_t tok = (size_t)b * g.N + min(t, g.N - 1), unit = tok * g.nh + h;
            int qq, hh, e;
            perm_pos(tt, qq, hh, e);
            if (c < 12) {                                                        // q: map c / 6, dims 4 (c % 6) .. + 4
                const int r = c / 6, d0 = 4 * (c - 6 * r);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float4 *>(q + tok * g.q_stride + h * HD2 + HD * r + d0);
                const float vv[4] = {v.x * g.scale, v.y * g.scale, v.z * g.scale, v.w * g.scale};
                *reinterpret_cast<uint2 *>(&sQ[r][tt][d0]) = make_uint2(pack2<BF16>(vv[0], vv[1]), pack2<BF16>(vv[2], vv[3]));
#pragma unroll
                for (int j = 0; j < 4; ++j) sQt[r][qq][hh][d0 + j][e] = cvt1<BF16>(vv[j]);
            } else if (c < 24) {                                                 // d(o): channels 4 (c - 12) .. + 4
                const int d0 = 4 * (c - 12);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float4 *>(ws + unit * WS_PER_UNIT + d0);
                const float vv[4] = {v.x, v.y, v.z, v.w};
                *reinterpret_cast<uint2 *>(&sdO[tt][d0]) = make_uint2(pack2<BF16>(vv[0], vv[1]), pack2<BF16>(vv[2], vv[3]));
#pragma unroll
                for (int j = 0; j < 4; ++j) sdOt[(d0 + j) >> 5][qq][hh][(d0 + j) & 31][e] = cvt1<BF16>(vv[j]);
            } else if (c < 26) {                                                 // lse of map c - 24 (+inf: weight 0 for a padded token)
                sC[c - 24][tt] = ok ? lse[unit * 2 + (c - 24)] : 3.0e38f;
            } else {
                sC[c - 24][tt] = ok ? ws[unit * WS_PER_UNIT + HD2 + (c - 26)] : 0.f;
            }
        }
        __syncthreads();
        // S = Q K^T: rows = tokens, column = this lane's key
        f32x16 dW = mfma16<BF16>(*reinterpret_cast<const uint4 *>(&sdO[col][8 * kh]), vB[0], zero16());
        dW = mfma16<BF16>(*reinterpret_cast<const uint4 *>(&sdO[col][16 + 8 * kh]), vB[1], dW);
        dW = mfma16<BF16>(*reinterpret_cast<const uint4 *>(&sdO[col][32 + 8 * kh]), vB[2], dW);
        float w[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) w[v] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x16 s = mfma16<BF16>(*reinterpret_cast<const uint4 *>(&sQ[r][col][8 * kh]), kB[r][0], zero16());
            s = mfma16<BF16>(*reinterpret_cast<const uint4 *>(&sQ[r][col][16 + 8 * kh]), kB[r][1], s);
            const float sg = r == 0 ? 1.f : -lam;
            float ds[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = acc_row(v, kh);
                const float p = key < g.P ? __expf(s[v] - sC[r][row]) : 0.f;
                ds[v] = p * (sg * dW[v] - sC[2 + r][row]);
                w[v] += sg * p;
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const uint4 dB = pack8<BF16>(ds + 8 * qq);
                dKacc[r] = mfma16<BF16>(*reinterpret_cast<const uint4 *>(&sQt[r][qq][kh][col][0]), dB, dKacc[r]);
            }
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const uint4 wB = pack8<BF16>(w + 8 * qq);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                dVacc[mt] = mfma16<BF16>(*reinterpret_cast<const uint4 *>(&sdOt[mt][qq][kh][col][0]), wB, dVacc[mt]);
        }
    }
    if (key < g.P) {
        float *prow = part + ((((size_t)b * g.nh + h) * nchunk + chunk) * g.P + key) * 96;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g4 = 0; g4 < 3; ++g4)
                *reinterpret_cast<float4 *>(prow + HD * r + 8 * g4 + 4 * kh) =
                    make_float4(dKacc[r][4 * g4], dKacc[r][4 * g4 + 1], dKacc[r][4 * g4 + 2], dKacc[r][4 * g4 + 3]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < (mt ? 2 : 4); ++g4)
                *reinterpret_cast<float4 *>(prow + HD2 + 32 * mt + 8 * g4 + 4 * kh) =
                    make_float4(dVacc[mt][4 * g4], dVacc[mt][4 * g4 + 1], dVacc[mt][4 * g4 + 2], dVacc[mt][4 * g4 + 3]);
    }
}

// dkp / dvp (B, P, d) = sum over token chunks of the partial blocks, fixed order
__global__ void __launch_bounds__(256)
pooled_lp_kv_reduce_kernel(const float *__restrict__ part, int nchunk, float *__restrict__ dkp, int dkp_stride,
                           float *__restrict__ dvp, int dvp_stride, LGeom g)
{
    const int h = blockIdx.y, b = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.P * 96) return;
    const int key = i / 96, c = i - key * 96;
    const float *p = part + (((size_t)b * g.nh + h) * nchunk * g.P + key) * 96 + c;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < nchunk; k += 2) { s0 += p[(size_t)k * g.P * 96]; s1 += p[(size_t)(k + 1) * g.P * 96]; }
    if (k < nchunk) s0 += p[(size_t)k * g.P * 96];
    const float s = s0 + s1;
    if (c < HD2) dkp[((size_t)b * g.P + key) * dkp_stride + h * HD2 + c] = s;
    else dvp[((size_t)b * g.P + key) * dvp_stride + h * HD2 + (c - HD2)] = s;
}

int make_geom(LGeom &g, int batch, int N, int P, int nh, int qs, int kps, int vps, int outs, float scale)
{
    if (batch <= 0 || N <= 0 || P <= 0 || nh <= 0 || batch > 65535 || nh > 65535 || P > 320) return MLAGG_E_UNSUPPORTED;
    g.batch = batch; g.N = N; g.P = P; g.nh = nh; g.d = nh * HD2; g.ntile = (P + 31) / 32;
    g.q_stride = qs; g.kp_stride = kps; g.vp_stride = vps; g.out_stride = outs; g.scale = scale;
    if (qs < g.d || kps < g.d || vps < g.d || outs < g.d || ((qs | kps | vps | outs) & 3) || scale == 0.f) return MLAGG_E_UNSUPPORTED;
    return 0;
}

template <typename K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes > 160 * 1024) return MLAGG_E_UNSUPPORTED;
    if (bytes > 48 * 1024)
        return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return 0;
}

// tokens per workgroup of the key-side kernel: enough workgroups that batch * heads * chunks * key tiles (= waves) fill the chip
// four times over (with 49 or 64 keys a workgroup is only two waves: 0.73 ms per step at 224 x 224 with 256-token chunks)
inline int kv_chunk_tokens(int batch, int nh, int N, int P)
{
    const int ntile = (P + 31) / 32;
    int ch = 4096;
    while (ch > 64 && (long)batch * nh * ((N + ch - 1) / ch) * ntile < 4096) ch >>= 1;
    return ch;
}

}  // namespace

extern "C" int mlagg_pooled_attn_lp_fwd(const float *q, int q_stride, const float *kp, int kp_stride, const float *vp, int vp_stride,
                                        const float *lam, const float *subln_w, float *out, int out_stride, float *lse, float *o1,
                                        float *o2, int batch, int N, int P, int nh, float scale, int dtype, void *stream)
{
    if (!q || !kp || !vp || !lam || !subln_w || !out || ((o1 == nullptr) != (o2 == nullptr))) return MLAGG_E_NULLPTR;
    if (dtype != MLAGG_DTYPE_BF16 && dtype != MLAGG_DTYPE_F16) return MLAGG_E_UNSUPPORTED;
    LGeom g;
    if (int rc = make_geom(g, batch, N, P, nh, q_stride, kp_stride, vp_stride, out_stride, scale)) return rc;
    const size_t lds = (size_t)2 * g.ntile * 32 * KP * 2 + (size_t)2 * g.ntile * 2 * 2 * 32 * 8 * 2;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((N + TOK_WG - 1) / TOK_WG, nh, batch);
    MLAGG_TIMED(K_POOLED_FWD, st);
    if (dtype == MLAGG_DTYPE_BF16) {
        if (int rc = allow_lds(pooled_lp_fwd_kernel<true>, lds)) return rc;
        hipLaunchKernelGGL(pooled_lp_fwd_kernel<true>, grid, dim3(512), lds, st, q, kp, vp, lam, subln_w, out, lse, o1, o2, g);
    } else {
        if (int rc = allow_lds(pooled_lp_fwd_kernel<false>, lds)) return rc;
        hipLaunchKernelGGL(pooled_lp_fwd_kernel<false>, grid, dim3(512), lds, st, q, kp, vp, lam, subln_w, out, lse, o1, o2, g);
    }
    return (int)hipGetLastError();
}

extern "C" size_t mlagg_pooled_attn_lp_bwd_workspace_floats(int batch, int N, int P, int nh)
{
    if (batch <= 0 || N <= 0 || P <= 0 || nh <= 0) return 0;
    const size_t units = (size_t)batch * N * nh;
    const size_t nchunk = (N + kv_chunk_tokens(batch, nh, N, P) - 1) / kv_chunk_tokens(batch, nh, N, P);
    return units * WS_PER_UNIT + ((units + 255) / 256) * 49 + (size_t)batch * nh * nchunk * P * 96;
}

extern "C" int mlagg_pooled_attn_lp_bwd(const float *q, int q_stride, const float *kp, int kp_stride, const float *vp, int vp_stride,
                                        const float *lam, const float *subln_w, const float *dout, int dout_stride, const float *lse,
                                        const float *o1, const float *o2, float *dq, int dq_stride, float *dkp, int dkp_stride,
                                        float *dvp, int dvp_stride, float *dlam, float *dsubln_w, float *workspace, int batch, int N,
                                        int P, int nh, float scale, int dtype, void *stream)
{
    if (!q || !kp || !vp || !lam || !subln_w || !dout || !lse || !o1 || !o2 || !dq || !dkp || !dvp || !dlam || !dsubln_w || !workspace)
        return MLAGG_E_NULLPTR;
    if (dtype != MLAGG_DTYPE_BF16 && dtype != MLAGG_DTYPE_F16) return MLAGG_E_UNSUPPORTED;
    LGeom g;
    if (int rc = make_geom(g, batch, N, P, nh, q_stride, kp_stride, vp_stride, dout_stride, scale)) return rc;
    if (dq_stride < g.d || dkp_stride < g.d || dvp_stride < g.d || ((dq_stride | dkp_stride | dvp_stride) & 3)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t units = (size_t)batch * N * nh;
    const int nprep = (int)((units + 255) / 256);
    float *ws = workspace, *pgrad = ws + units * WS_PER_UNIT, *part = pgrad + (size_t)nprep * 49;
    const int ch = kv_chunk_tokens(batch, nh, N, P), nchunk = (N + ch - 1) / ch;
    {
        MLAGG_TIMED(K_POOLED_BWD1, st);
        hipLaunchKernelGGL(pooled_lp_prep_kernel, dim3(nprep), dim3(256), 0, st, dout, dout_stride, o1, o2, lam, subln_w, ws, pgrad, g);
        hipLaunchKernelGGL(mlagg_internal::column_sum_split_kernel<0>, dim3(1), dim3(1024), 0, st, pgrad, nprep, 49, 49, 48, dsubln_w, dlam);
        const size_t lds = (size_t)2 * g.ntile * 32 * KP * 2 + (size_t)g.ntile * 32 * VP * 2 + (size_t)2 * g.ntile * 2 * 2 * 32 * 8 * 2;
        const dim3 grid((N + TOK_WG - 1) / TOK_WG, nh, batch);
        if (dtype == MLAGG_DTYPE_BF16) {
            if (int rc = allow_lds(pooled_lp_bwd_q_kernel<true>, lds)) return rc;
            hipLaunchKernelGGL(pooled_lp_bwd_q_kernel<true>, grid, dim3(512), lds, st, q, kp, vp, lam, lse, ws, dq, dq_stride, g);
        } else {
            if (int rc = allow_lds(pooled_lp_bwd_q_kernel<false>, lds)) return rc;
            hipLaunchKernelGGL(pooled_lp_bwd_q_kernel<false>, grid, dim3(512), lds, st, q, kp, vp, lam, lse, ws, dq, dq_stride, g);
        }
    }
    {
        MLAGG_TIMED(K_POOLED_BWD2, st);
        const dim3 grid(nchunk, nh, batch), block(64 * g.ntile);
        if (dtype == MLAGG_DTYPE_BF16)
            hipLaunchKernelGGL(pooled_lp_bwd_kv_kernel<true>, grid, block, 0, st, q, kp, vp, lam, lse, ws, part, ch, nchunk, g);
        else
            hipLaunchKernelGGL(pooled_lp_bwd_kv_kernel<false>, grid, block, 0, st, q, kp, vp, lam, lse, ws, part, ch, nchunk, g);
        hipLaunchKernelGGL(pooled_lp_kv_reduce_kernel, dim3((P * 96 + 255) / 256, nh, batch), dim3(256), 0, st, part, nchunk, dkp,
                           dkp_stride, dvp, dvp_stride, g);
    }
    return (int)hipGetLastError();
}
