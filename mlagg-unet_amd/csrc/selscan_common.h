// Shared pieces of the selective-scan kernels (csrc/selscan.hip: the (B, D, L) boundary form; csrc/selscan_tok.hip: the token-major
// MSMM form): constants, wave64 helpers, the chunk prefix and the per-chunk partial reduction.  Included by both translation units;
// everything lives in an anonymous namespace.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int NS = 16;    // d_state (fixed: reference uses 16, MambaSkip.py:271)
constexpr int ST = 16;    // steps per LDS sub-tile
constexpr int TC = 64;    // steps per chunk
constexpr int NSUB = TC / ST;
constexpr int NT8 = TC / 8 - 1;   // saved entry states per chunk besides the chunk's own: one per 8-step tile (round 2; was per 16)
constexpr int UP = ST + 4;   // row pitch of the u / delta / dy tiles (floats): conflict-free b128 reads
constexpr int BP = 20;       // pitch of the [t][n] B / C tiles
constexpr int PP = 24;       // pitch of a per-chunk partial row: dA[16], dD, ddelta_bias, dWdt[4], pad
constexpr int RMAX = 4;      // largest supported rank of the low-rank delta projection
constexpr float LOG2E = 1.4426950408889634f;
// Channels per workgroup.  A group has 96 channels, but 32-channel workgroups (2 waves) measured fastest for
// every pass: forward 0.49 -> 0.34 ms and 0.62 -> 0.43 ms, backward main 3.6 -> 2.55 ms at config 2.  Small
// workgroups keep all four SIMDs of a CU evenly loaded (6-wave workgroups land 2/2/1/1) and give the
// scheduler independent workgroups to overlap one's barrier/LDS phases with another's arithmetic; the B/C
// tile is then staged by 3 workgroups (L2 hits) and backward sums dB/dC with 3-way float atomics.
constexpr int SCAN_CB = 32;
constexpr int BWD_CB = SCAN_CB;

struct ScanGeom {
    int batch, dim, L, G, Hc, CB, nblk, nchunks;
};

// softplus(x) = max(x, 0) + log1p(exp(-|x|)).  e = exp(-|x|) is in (0, 1]; for small e the series
// e - e^2/2 + e^3/3 (truncation < e^4/4 <= 2.5e-9 at e = 0.01) avoids the cancellation of log(1 + e),
// elsewhere v_log_f32 on 1 + e is accurate to ~1 ulp of a value in [0.01, 0.69].  ~10 VALU ops instead
// of the ~100 of libm's expf + log1pf, which were 45 % of the forward kernels' instructions (round-1 PMC).
__device__ __forceinline__ float softplus_f(float x)
{
    const float e = __expf(-fabsf(x));
    const float small = e * (1.f - e * (0.5f - e * (1.f / 3.f)));
    const float big = __builtin_amdgcn_logf(1.f + e) * 0.6931471805599453f;   // bare v_log_f32 (log2): 1 + e >= 1, no denormal path needed
    return fmaxf(x, 0.f) + (e < 0.01f ? small : big);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// Keep a prefetched value where it is: a fake read-modify of the register.  Without it the compiler is free to SINK a load
// of a `const __restrict__` row down to its use (it did, past barriers), which turns "all loads of the chunk up front" back
// into one exposed round trip per sub-tile.
__device__ __forceinline__ void pin4(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// Workgroup barrier for kernels whose waves talk through LDS only: __syncthreads() also releases GLOBAL memory, i.e. waits
// (vmcnt(0)) for every store of the sub-tile before the next one may start.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float dpp_quad_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += dpp_quad_xor1(v);
    v += dpp_quad_xor2(v);
    return v;
}

// pass 2: exclusive prefix over chunks of h -> exp(A * dsum_c) h + s_c.  In place: s_c becomes the
// state ENTERING chunk c.  reverse = true runs from the last chunk (backward's q carries).
__global__ void selscan_chunk_prefix(const float *__restrict__ A, float *__restrict__ cstate,
                                     const float *__restrict__ cdsum, ScanGeom gm, int reverse)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int d = idx >> 4, n = idx & 15, b = blockIdx.y;
    if (d >= gm.dim) return;
    const float A2 = A[d * NS + n] * LOG2E;
    float H = 0.f;
    const size_t base = (size_t)b * gm.nchunks * gm.dim + d;
    // groups of 8 chunks: all 16 loads of a group are issued before its 8 dependent updates (the loop is a latency
    // chain of nchunks steps on ~1 wave per SIMD; with the in-place store between them the loads did not overlap)
    constexpr int GR = 8;
    for (int i0 = 0; i0 < gm.nchunks; i0 += GR) {
        float S[GR], P[GR];
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            const int i = i0 + j;
            if (i < gm.nchunks) {
                const size_t row = base + (size_t)(reverse ? gm.nchunks - 1 - i : i) * gm.dim;
                S[j] = cstate[row * NS + n];
                P[j] = cdsum[row];
            } else {
                S[j] = 0.f; P[j] = 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            const int i = i0 + j;
            if (i < gm.nchunks) {
                const size_t row = base + (size_t)(reverse ? gm.nchunks - 1 - i : i) * gm.dim;
                cstate[row * NS + n] = H;
                H = fast_exp2(A2 * P[j]) * H + S[j];
            }
        }
    }
}

__device__ __forceinline__ void swap32(float &a, float &b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &a, float &b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float row_ror4(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_ror8(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}

constexpr int T8 = 8;            // steps per tile of the group kernel
constexpr int SP8 = 8;           // pitch of the 8-step dy rows (b128 reads of one 16-lane group still hit 16 distinct banks)
constexpr int JMAX = 6;          // channel slots per lane: groups of up to 96 channels
__device__ __forceinline__ float act_delta(float raw, float bias, int softplus, bool inr)
{
    float x = raw + bias;
    if (softplus) x = softplus_f(x);
    return inr ? x : 0.f;
}

// Loads / stores of the group kernel.  VEC (L % 4 == 0: every MLAgg-UNet shape): branch-free -- the address is clamped
// and the value selected, so that no conditional block (and no s_waitcnt at its end) sits between a prefetch and its use.
// uniform base + a 32-bit BYTE offset per lane: the form the `global_load ... v_off, s[base:base+1]` encoding takes (an
// element offset scaled in 64 bits costs a v_mov + v_lshl_add_u64 per access)
template <typename T>
__device__ __forceinline__ T ldg_at(const float *__restrict__ base, unsigned elem)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(elem * 4u));
}
template <typename T>
__device__ __forceinline__ void stg_at(float *__restrict__ base, unsigned elem, const T &v)
{
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (size_t)(elem * 4u)) = v;
}
// One wave per workgroup: LDS instructions of a wave execute in order, so all the kernel needs between a producer and a
// consumer of an LDS row is that the COMPILER keeps the order.  (__syncthreads() would also drain vmcnt and with it the
// prefetched streams.)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }
// component-wise: a select between float4 AGGREGATES is lowered through a scratch (stack) slot
__device__ __forceinline__ float4 keep4(bool c, const float4 &v)
{
    return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f);
}

// part[b][chunk][d][PP] -> dA[d][16], dD[d], ddbias[d], dWdt[d][R]: column sums of the (batch * nchunks) x (dim * PP)
// matrix.  A workgroup owns 64 consecutive columns (256-byte row segments, coalesced) and splits the rows over 16
// row-groups (the first version gave one workgroup per channel 88-byte segments at a 37 KB stride: 0.14 ms).
__global__ void __launch_bounds__(1024)
selscan_reduce_partials(const float *__restrict__ part, float *__restrict__ dA, float *__restrict__ dD,
                        float *__restrict__ ddbias, float *__restrict__ dWdt, int R, ScanGeom gm)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx, cols = gm.dim * PP;
    const int rows = gm.batch * gm.nchunks;
    float s0 = 0.f, s1 = 0.f;
    if (col < cols) {
        int r = rg;
        for (; r + 16 < rows; r += 32) {
            s0 += part[(size_t)r * cols + col];
            s1 += part[(size_t)(r + 16) * cols + col];
        }
        if (r < rows) s0 += part[(size_t)r * cols + col];
    }
    red[rg][cx] = s0 + s1;
    __syncthreads();
    if (rg == 0 && col < cols) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cx];
        const int d = col / PP, c = col - d * PP;
        if (c < NS) dA[d * NS + c] = s;
        else if (c == NS) { if (dD) dD[d] = s; }
        else if (c == NS + 1) { if (ddbias) ddbias[d] = s; }
        else if (dWdt && c - NS - 2 < R) dWdt[(size_t)d * R + c - NS - 2] = s;
    }
}

int make_geom(ScanGeom &gm, int batch, int dim, int L, int N, int G, int max_cb = 128)
{
    if (N != NS || batch <= 0 || dim <= 0 || L <= 0 || G <= 0 || dim % G != 0) return MLAGG_E_UNSUPPORTED;
    if (batch > 65535) return MLAGG_E_UNSUPPORTED;
    gm.batch = batch; gm.dim = dim; gm.L = L; gm.G = G; gm.Hc = dim / G;
    gm.nblk = (gm.Hc + max_cb - 1) / max_cb;
    gm.CB = (gm.Hc + gm.nblk - 1) / gm.nblk;
    gm.nchunks = (L + TC - 1) / TC;
    if ((size_t)G * gm.nblk > 65535) return MLAGG_E_UNSUPPORTED;
    return 0;
}

}  // namespace
