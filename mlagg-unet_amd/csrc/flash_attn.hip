// Boundary #3 -- `flash_attn.flash_attn_func(q, k, v, causal=False)` as the reference calls it for the pooled branch of
// AggregatedAttention (nnUNetTrainer_MLAgg_2D_dt_MS.py:173, 745-750: four calls per module, fp16 / bf16, head_dim 24,
// P = (H / sr)(W / sr) <= 320 pooled keys): out = softmax(q k^T * scale) v per (batch, head).
//
// This serves the SHIM (mlagg_unet_amd.shims.flash_attn_func: the reference's own model file running unmodified on
// MI355X); the product network does not come through here -- its pooled branch is the single fused K4 launch.
//
// Shape of the problem: short key sets, tiny head_dim, many query tokens (N up to 81920 per image): a stream over q /
// out rows with K and V of one (batch, head) resident in LDS (P x 24 x 2 floats <= 60 KiB).  16-bit tensors in HBM,
// fp32 arithmetic:
//   forward     lane = (token, head): two sweeps over the keys in LDS (max / sum, then weights), q and the 24
//               accumulators in VGPRs, LDS reads are wave-wide broadcasts;
//   backward-1  lane = (token, head): D = <dout, out>, p_j and d(s_j) recomputed per key, dq accumulated in VGPRs;
//   backward-2  lane = key: k, v, dk, dv rows in VGPRs, token tiles (q, dout, lse, D) streamed through LDS, so dk / dv
//               need no cross-lane sums; token chunks accumulate into an fp32 workspace with float atomics (P x 48
//               per chunk).
// HBM-bound by 2 * 2 * e bytes per (token, head) forward.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int E = 24;                  // head_dim of every MLAgg-UNet stage (reference T:634)
constexpr int TOK = 256;               // tokens per workgroup (forward, backward-1)
constexpr int CH2 = 512;               // tokens per workgroup chunk (backward-2)
constexpr int TT = 32;                 // tokens per LDS tile (backward-2)
constexpr int TW = 2 * E + 4;          // floats per staged token: q[24], dout[24], lse, D, 2 of padding: rows stay 16-byte aligned for the float4 reads

struct FGeom {
    int B, N, P, nh;
    float scale;
};

template <bool BF16>
__device__ __forceinline__ float ld16(const unsigned short *p)
{
    if (BF16) return __uint_as_float((unsigned)(*p) << 16);
    return __half2float(*reinterpret_cast<const __half *>(p));
}
template <bool BF16>
__device__ __forceinline__ unsigned short st16(float v)
{
    if (BF16) {
        const __hip_bfloat16 b = __float2bfloat16(v);
        return *reinterpret_cast<const unsigned short *>(&b);
    }
    const __half h = __float2half_rn(v);
    return *reinterpret_cast<const unsigned short *>(&h);
}
// 24 consecutive 16-bit values (48 bytes, 16-byte aligned rows) -> fp32
template <bool BF16>
__device__ __forceinline__ void load_row(const unsigned short *__restrict__ p, float (&x)[E])
{
#pragma unroll
    for (int i = 0; i < E / 8; ++i) {
        const uint4 w = *reinterpret_cast<const uint4 *>(p + 8 * i);
        const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned short lo = (unsigned short)(u[j] & 0xffff), hi = (unsigned short)(u[j] >> 16);
            x[8 * i + 2 * j] = ld16<BF16>(&lo);
            x[8 * i + 2 * j + 1] = ld16<BF16>(&hi);
        }
    }
}
template <bool BF16>
__device__ __forceinline__ void store_row(unsigned short *__restrict__ p, const float (&x)[E])
{
#pragma unroll
    for (int i = 0; i < E / 8; ++i) {
        unsigned u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            u[j] = (unsigned)st16<BF16>(x[8 * i + 2 * j]) | ((unsigned)st16<BF16>(x[8 * i + 2 * j + 1]) << 16);
        *reinterpret_cast<uint4 *>(p + 8 * i) = make_uint4(u[0], u[1], u[2], u[3]);
    }
}

__device__ __forceinline__ float dot_lds(const float (&a)[E], const float *__restrict__ lds)
{
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < E / 4; ++i) {
        const float4 b = *reinterpret_cast<const float4 *>(lds + 4 * i);
        acc += a[4 * i] * b.x + a[4 * i + 1] * b.y + a[4 * i + 2] * b.z + a[4 * i + 3] * b.w;
    }
    return acc;
}
__device__ __forceinline__ void axpy_lds(float (&acc)[E], float w, const float *__restrict__ lds)
{
#pragma unroll
    for (int i = 0; i < E / 4; ++i) {
        const float4 b = *reinterpret_cast<const float4 *>(lds + 4 * i);
        acc[4 * i] += w * b.x; acc[4 * i + 1] += w * b.y; acc[4 * i + 2] += w * b.z; acc[4 * i + 3] += w * b.w;
    }
}

// K and V rows of (batch b, head h): (B, P, nh, E) 16-bit -> sK[p][E], sV[p][E] fp32
template <bool BF16>
__device__ __forceinline__ void stage_kv(const FGeom &g, const unsigned short *__restrict__ k, const unsigned short *__restrict__ v,
                                         int b, int h, float *sK, float *sV)
{
    for (int i = threadIdx.x; i < g.P * (E / 8); i += blockDim.x) {
        const int p = i / (E / 8), c = i - p * (E / 8);
        const size_t off = (((size_t)b * g.P + p) * g.nh + h) * E + 8 * c;
        const uint4 kw = *reinterpret_cast<const uint4 *>(k + off), vw = *reinterpret_cast<const uint4 *>(v + off);
        const unsigned ku[4] = {kw.x, kw.y, kw.z, kw.w}, vu[4] = {vw.x, vw.y, vw.z, vw.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned short k0 = (unsigned short)(ku[j] & 0xffff), k1 = (unsigned short)(ku[j] >> 16);
            const unsigned short v0 = (unsigned short)(vu[j] & 0xffff), v1 = (unsigned short)(vu[j] >> 16);
            sK[p * E + 8 * c + 2 * j] = ld16<BF16>(&k0); sK[p * E + 8 * c + 2 * j + 1] = ld16<BF16>(&k1);
            sV[p * E + 8 * c + 2 * j] = ld16<BF16>(&v0); sV[p * E + 8 * c + 2 * j + 1] = ld16<BF16>(&v1);
        }
    }
}

template <bool BF16>
__global__ void __launch_bounds__(TOK)
flash_fwd_kernel(const unsigned short *__restrict__ q, const unsigned short *__restrict__ k, const unsigned short *__restrict__ v,
                 unsigned short *__restrict__ out, float *__restrict__ lse, FGeom g)
{
    extern __shared__ float4 smem4[];
    float *sK = reinterpret_cast<float *>(smem4), *sV = sK + g.P * E;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_kv<BF16>(g, k, v, b, h, sK, sV);
    __syncthreads();
    const int t = blockIdx.x * TOK + threadIdx.x;
    if (t >= g.N) return;
    const size_t row = (((size_t)b * g.N + t) * g.nh + h) * E;
    float qv[E];
    load_row<BF16>(q + row, qv);
#pragma unroll
    for (int e = 0; e < E; ++e) qv[e] *= g.scale;
    float m = -3.0e38f, z = 0.f;
    for (int p = 0; p < g.P; ++p) {
        const float l = dot_lds(qv, sK + p * E);
        const float mn = fmaxf(m, l);
        z = z * __expf(m - mn) + __expf(l - mn);
        m = mn;
    }
    const float L = m + __logf(z);
    float o[E];
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = 0.f;
    for (int p = 0; p < g.P; ++p) axpy_lds(o, __expf(dot_lds(qv, sK + p * E) - L), sV + p * E);
    store_row<BF16>(out + row, o);
    if (lse) lse[((size_t)b * g.nh + h) * g.N + t] = L;
}

// backward-1: dq, and D = <dout, out> for backward-2
template <bool BF16>
__global__ void __launch_bounds__(TOK)
flash_bwd_q_kernel(const unsigned short *__restrict__ q, const unsigned short *__restrict__ k, const unsigned short *__restrict__ v,
                   const unsigned short *__restrict__ out, const unsigned short *__restrict__ dout, const float *__restrict__ lse,
                   unsigned short *__restrict__ dq, float *__restrict__ Dws, FGeom g)
{
    extern __shared__ float4 smem4[];
    float *sK = reinterpret_cast<float *>(smem4), *sV = sK + g.P * E;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_kv<BF16>(g, k, v, b, h, sK, sV);
    __syncthreads();
    const int t = blockIdx.x * TOK + threadIdx.x;
    if (t >= g.N) return;
    const size_t row = (((size_t)b * g.N + t) * g.nh + h) * E;
    float qv[E], gv[E], ov[E];
    load_row<BF16>(q + row, qv);
    load_row<BF16>(dout + row, gv);
    load_row<BF16>(out + row, ov);
    float D = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { D += gv[e] * ov[e]; qv[e] *= g.scale; }
    const float L = lse[((size_t)b * g.nh + h) * g.N + t];
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    for (int p = 0; p < g.P; ++p) {
        const float pr = __expf(dot_lds(qv, sK + p * E) - L);
        const float ds = pr * (dot_lds(gv, sV + p * E) - D);
        axpy_lds(acc, ds, sK + p * E);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] *= g.scale;
    store_row<BF16>(dq + row, acc);
    Dws[((size_t)b * g.nh + h) * g.N + t] = D;
}

// backward-2: lane = key.  dkv: fp32 (B, P, nh, 2, E) accumulated over token chunks with float atomics.
template <bool BF16>
__global__ void __launch_bounds__(512)
flash_bwd_kv_kernel(const unsigned short *__restrict__ q, const unsigned short *__restrict__ k, const unsigned short *__restrict__ v,
                    const unsigned short *__restrict__ dout, const float *__restrict__ lse, const float *__restrict__ Dws,
                    float *__restrict__ dkv, FGeom g)
{
    __shared__ __attribute__((aligned(16))) float sT[TT * TW];
    const int h = blockIdx.y, b = blockIdx.z;
    const int p = threadIdx.x;
    const bool act = p < g.P;
    float kv[E], vv[E], dk[E], dv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { kv[e] = vv[e] = dk[e] = dv[e] = 0.f; }
    if (act) {
        const size_t off = (((size_t)b * g.P + p) * g.nh + h) * E;
        load_row<BF16>(k + off, kv);
        load_row<BF16>(v + off, vv);
    }
    const int t0 = blockIdx.x * CH2, t1 = min(g.N, t0 + CH2);
    for (int tb = t0; tb < t1; tb += TT) {
        __syncthreads();
        // stage TT tokens: 6 uint4 (q: 3, dout: 3) + 2 scalars per token
        for (int i = threadIdx.x; i < TT * 8; i += blockDim.x) {
            const int tt = i >> 3, c = i & 7, t = tb + tt;
            float *dst = sT + tt * TW;
            if (c < 6) {
                float x8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (t < t1) {
                    const unsigned short *src = (c < 3 ? q : dout) + (((size_t)b * g.N + t) * g.nh + h) * E + 8 * (c % 3);
                    const uint4 w = *reinterpret_cast<const uint4 *>(src);
                    const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned short lo = (unsigned short)(u[j] & 0xffff), hi = (unsigned short)(u[j] >> 16);
                        x8[2 * j] = ld16<BF16>(&lo); x8[2 * j + 1] = ld16<BF16>(&hi);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[(c < 3 ? 0 : E) + 8 * (c % 3) + j] = x8[j];
            } else if (c == 6) {
                dst[2 * E] = t < t1 ? lse[((size_t)b * g.nh + h) * g.N + t] : 3.0e38f;       // exp(s - inf) = 0: padded token
            } else {
                dst[2 * E + 1] = t < t1 ? Dws[((size_t)b * g.nh + h) * g.N + t] : 0.f;
            }
        }
        __syncthreads();
        if (act) {
            for (int tt = 0; tt < TT; ++tt) {
                const float *row = sT + tt * TW;
                const float pr = __expf(dot_lds(kv, row) * g.scale - row[2 * E]);
                const float ds = pr * (dot_lds(vv, row + E) - row[2 * E + 1]) * g.scale;
                axpy_lds(dv, pr, row + E);
                axpy_lds(dk, ds, row);
            }
        }
    }
    if (act) {
        float *dst = dkv + (((size_t)b * g.P + p) * g.nh + h) * 2 * E;
#pragma unroll
        for (int e = 0; e < E; ++e) { atomicAdd(dst + e, dk[e]); atomicAdd(dst + E + e, dv[e]); }
    }
}

template <typename K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return 0;
}

int check(const FGeom &g, int head_dim, int dtype)
{
    if (head_dim != E || g.B <= 0 || g.N <= 0 || g.P <= 0 || g.nh <= 0) return MLAGG_E_UNSUPPORTED;
    if (dtype != MLAGG_DTYPE_BF16 && dtype != MLAGG_DTYPE_F16) return MLAGG_E_UNSUPPORTED;
    if (g.P > 512 || g.nh > 65535 || g.B > 65535) return MLAGG_E_UNSUPPORTED;     // 512 keys: 96 KiB of LDS, one lane per key
    return 0;
}

}  // namespace

extern "C" size_t mlagg_flash_attn_bwd_workspace_floats(int B, int N, int P, int nh, int head_dim)
{
    return (size_t)B * nh * N + (size_t)B * P * nh * 2 * head_dim;        // D, then the fp32 dk | dv accumulators
}

extern "C" int mlagg_flash_attn_fwd(const void *q, const void *k, const void *v, void *out, float *lse, int B, int N, int P,
                                    int nh, int head_dim, float softmax_scale, int dtype, void *stream)
{
    if (!q || !k || !v || !out) return MLAGG_E_NULLPTR;
    FGeom g{B, N, P, nh, softmax_scale};
    if (int rc = check(g, head_dim, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((N + TOK - 1) / TOK, nh, B), block(TOK);
    const size_t lds = (size_t)2 * P * E * sizeof(float);
    auto Q = static_cast<const unsigned short *>(q), K = static_cast<const unsigned short *>(k), V = static_cast<const unsigned short *>(v);
    auto O = static_cast<unsigned short *>(out);
    if (int rc = dtype == MLAGG_DTYPE_BF16 ? allow_lds(flash_fwd_kernel<true>, lds) : allow_lds(flash_fwd_kernel<false>, lds)) return rc;
    MLAGG_TIMED(K_FLASH_FWD, st);
    if (dtype == MLAGG_DTYPE_BF16) hipLaunchKernelGGL(flash_fwd_kernel<true>, grid, block, lds, st, Q, K, V, O, lse, g);
    else hipLaunchKernelGGL(flash_fwd_kernel<false>, grid, block, lds, st, Q, K, V, O, lse, g);
    return (int)hipGetLastError();
}

// dq (B, N, nh, E) in `dtype`; dkv_ws: the workspace; its tail holds fp32 (B, P, nh, 2, E): [.., 0, :] = dk, [.., 1, :] = dv
extern "C" int mlagg_flash_attn_bwd(const void *q, const void *k, const void *v, const void *out, const void *dout,
                                    const float *lse, void *dq, float *workspace, int B, int N, int P, int nh, int head_dim,
                                    float softmax_scale, int dtype, void *stream)
{
    if (!q || !k || !v || !out || !dout || !lse || !dq || !workspace) return MLAGG_E_NULLPTR;
    FGeom g{B, N, P, nh, softmax_scale};
    if (int rc = check(g, head_dim, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *Dws = workspace, *dkv = workspace + (size_t)B * nh * N;
    (void)hipMemsetAsync(dkv, 0, (size_t)B * P * nh * 2 * E * sizeof(float), st);
    auto Q = static_cast<const unsigned short *>(q), K = static_cast<const unsigned short *>(k), V = static_cast<const unsigned short *>(v);
    auto O = static_cast<const unsigned short *>(out), G = static_cast<const unsigned short *>(dout);
    auto DQ = static_cast<unsigned short *>(dq);
    const size_t lds = (size_t)2 * P * E * sizeof(float);
    const dim3 grid1((N + TOK - 1) / TOK, nh, B), grid2((N + CH2 - 1) / CH2, nh, B);
    const int threads2 = ((P + 63) / 64) * 64;
    if (int rc = dtype == MLAGG_DTYPE_BF16 ? allow_lds(flash_bwd_q_kernel<true>, lds) : allow_lds(flash_bwd_q_kernel<false>, lds)) return rc;
    MLAGG_TIMED(K_FLASH_BWD, st);
    if (dtype == MLAGG_DTYPE_BF16) {
        hipLaunchKernelGGL(flash_bwd_q_kernel<true>, grid1, dim3(TOK), lds, st, Q, K, V, O, G, lse, DQ, Dws, g);
        hipLaunchKernelGGL(flash_bwd_kv_kernel<true>, grid2, dim3(threads2), 0, st, Q, K, V, G, lse, Dws, dkv, g);
    } else {
        hipLaunchKernelGGL(flash_bwd_q_kernel<false>, grid1, dim3(TOK), lds, st, Q, K, V, O, G, lse, DQ, Dws, g);
        hipLaunchKernelGGL(flash_bwd_kv_kernel<false>, grid2, dim3(threads2), 0, st, Q, K, V, G, lse, Dws, dkv, g);
    }
    return (int)hipGetLastError();
}
