// K1 -- selective scan for gfx950 (MI355X), forward and backward.
//
// What it replaces: mamba-ssm's selective_scan_cuda.{fwd,bwd} behind `selective_scan_fn` as called
// by SS2D_skip.forward_corev0 (reference MambaSkip.py:445-451).  Not a port of that CUDA kernel:
// the decomposition below is built around wave64, DPP and LDS on CDNA4.
//
// Decomposition (chunk-parallel, three launches per direction):
//   * time is cut into chunks of TC = 64 steps; a workgroup owns (batch b, group g, chunk) and ALL
//     H = dim/G channels of the group, so the B/C tile of the chunk is staged in LDS once and the
//     backward's dB/dC sums over channels never leave the workgroup (no global atomics);
//   * a lane owns (channel c, state quad s): 4 of the 16 states of one channel, carried in VGPRs and
//     advanced serially in time -- 5 VALU ops per (step, state), no cross-lane scan in the hot loop;
//     the 4 lanes of a channel are a DPP quad, so y / d(delta) / du are quad reductions;
//   * pass 1 runs every chunk from a zero state and emits (sum of delta, end state); pass 2 is the
//     parallel-prefix over chunks of the affine maps h -> exp(A * sum_delta) h + s (one lane per
//     (b, d, n), 340 dependent FMAs at 256x256); pass 3 re-runs each chunk from its true entry
//     state and writes y.  The entry states are what backward needs, so they double as the
//     saved-for-backward tensor (`chunk_state`).
//   * HBM access: u/delta rows are read as 64-byte segments (4 lanes x float4) per channel and
//     sub-tile, B/C as 64-byte segments per state row, transposed to [t][n] through LDS so a lane
//     fetches its 4 states with one ds_read_b128 broadcast across the 16 channels of the wave.
//
// Roofline: HBM-bound by design target.  Algorithmic bytes per (b, l): 4*(3*dim + 2*G*N) forward,
// 4*(5*dim + 4*G*N) backward.  This 3-pass form reads u/delta twice.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int NS = 16;    // d_state (fixed: reference uses 16, MambaSkip.py:271)
constexpr int ST = 16;    // steps per LDS sub-tile
constexpr int TC = 64;    // steps per chunk
constexpr int NSUB = TC / ST;
constexpr int UP = ST + 4;   // row pitch of the u / delta / dy tiles (floats): conflict-free b128 reads
constexpr int BP = 20;       // pitch of the [t][n] B / C tiles
constexpr int PP = 20;       // pitch of a per-chunk partial row: dA[16], dD, ddelta_bias, pad
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
    const float big = __logf(1.f + e);
    return fmaxf(x, 0.f) + (e < 0.01f ? small : big);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

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

__device__ __forceinline__ float4 load4(const float *__restrict__ row, int t, int L, bool vec)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
        if (t < L) v = *reinterpret_cast<const float4 *>(row + t);
    } else {
        if (t < L) v.x = row[t];
        if (t + 1 < L) v.y = row[t + 1];
        if (t + 2 < L) v.z = row[t + 2];
        if (t + 3 < L) v.w = row[t + 3];
    }
    return v;
}

__device__ __forceinline__ void store4(float *__restrict__ row, int t, int L, bool vec, float4 v)
{
    if (vec) {
        if (t < L) *reinterpret_cast<float4 *>(row + t) = v;
    } else {
        if (t < L) row[t] = v.x;
        if (t + 1 < L) row[t + 1] = v.y;
        if (t + 2 < L) row[t + 2] = v.z;
        if (t + 3 < L) row[t + 3] = v.w;
    }
}

__device__ __forceinline__ float f4get(const float4 &v, int j)
{
    return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

// delta' = softplus(delta + bias) for in-range steps, 0 beyond L (a = 1, b = 0: identity step)
__device__ __forceinline__ float4 activate_delta(float4 raw, float bias, int softplus, int t, int L)
{
    float r[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = r[j] + bias;
        if (softplus) x = softplus_f(x);
        r[j] = (t + j < L) ? x : 0.f;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

__device__ __forceinline__ void rot4(float (&x)[4])
{
    const float t = x[0];
    x[0] = x[1]; x[1] = x[2]; x[2] = x[3]; x[3] = t;
}

struct LaneId {
    int cl, s, b, g, d, chunk;
    bool act;
};

__device__ __forceinline__ LaneId lane_id(const ScanGeom &gm)
{
    LaneId id;
    id.cl = threadIdx.x >> 2;
    id.s = threadIdx.x & 3;
    id.chunk = blockIdx.x;
    id.g = blockIdx.y / gm.nblk;
    const int blk = blockIdx.y - id.g * gm.nblk;
    id.b = blockIdx.z;
    const int cg = blk * gm.CB + id.cl;
    id.act = id.cl < gm.CB && cg < gm.Hc;
    id.d = id.g * gm.Hc + (id.act ? cg : 0);
    return id;
}

// ------------------------------------------------------------------------------------------
// forward pass 1 (FINAL = false): chunk from zero state -> (end state, sum delta')
// forward pass 3 (FINAL = true) : chunk from its entry state -> y
// ------------------------------------------------------------------------------------------
template <bool FINAL>
__global__ void selscan_fwd_kernel(const float *__restrict__ u, const float *__restrict__ delta,
                                   const float *__restrict__ A, const float *__restrict__ Bm,
                                   const float *__restrict__ Cm, const float *__restrict__ Dv,
                                   const float *__restrict__ dbias, float *__restrict__ out,
                                   float *__restrict__ cstate, float *__restrict__ cdsum, ScanGeom gm,
                                   int softplus)
{
    extern __shared__ float4 smem4[];
    float *su = reinterpret_cast<float *>(smem4);
    float *sd = su + gm.CB * UP;
    float *sB = sd + gm.CB * UP;
    float *sC = sB + ST * BP;

    const LaneId id = lane_id(gm);
    const int tid = threadIdx.x, L = gm.L;
    const bool vec = (L & 3) == 0;
    const float *urow = u + ((size_t)id.b * gm.dim + id.d) * L;
    const float *drow = delta + ((size_t)id.b * gm.dim + id.d) * L;
    const float *bcrow = nullptr;
    if (tid < 64)
        bcrow = Bm + (((size_t)id.b * gm.G + id.g) * NS + (tid >> 2)) * L;
    else if (FINAL && tid < 128)
        bcrow = Cm + (((size_t)id.b * gm.G + id.g) * NS + ((tid - 64) >> 2)) * L;
    const float bias = (dbias && id.act) ? dbias[id.d] : 0.f;
    const float Dd = (FINAL && Dv && id.act) ? Dv[id.d] : 0.f;
    const size_t srow = ((size_t)id.b * gm.nchunks + id.chunk) * gm.dim + id.d;

    float A2[4], h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A2[i] = id.act ? A[id.d * NS + 4 * id.s + i] * LOG2E : 0.f;
        h[i] = 0.f;
    }
    if (FINAL && id.act) {
        const float4 h0 = *reinterpret_cast<const float4 *>(cstate + srow * NS + 4 * id.s);
        h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w;
    }
    float dsum = 0.f;
    const int tc0 = id.chunk * TC;

    // Every load of the chunk is issued up front (4 sub-tiles x {u, delta, B|C} = 12 float4 per lane in
    // flight): the kernel is latency-bound on 64-byte row segments, so memory-level parallelism, not
    // LDS capacity, is what the chunk needs; LDS still only ever holds one 16-step sub-tile.
    float4 pu[NSUB], pd[NSUB], pbc[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t0 = tc0 + sub * ST;
        pu[sub] = pd[sub] = pbc[sub] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id.act) { pu[sub] = load4(urow, t0 + 4 * id.s, L, vec); pd[sub] = load4(drow, t0 + 4 * id.s, L, vec); }
        if (bcrow) pbc[sub] = load4(bcrow, t0 + 4 * (tid & 3), L, vec);
    }

#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t0 = tc0 + sub * ST;
        __syncthreads();
        if (id.act) {
            *reinterpret_cast<float4 *>(su + id.cl * UP + 4 * id.s) = pu[sub];
            *reinterpret_cast<float4 *>(sd + id.cl * UP + 4 * id.s) =
                activate_delta(pd[sub], bias, softplus, t0 + 4 * id.s, L);
        }
        if (bcrow) {
            float *dst = (tid < 64 ? sB : sC) + (4 * (tid & 3)) * BP + ((tid & 63) >> 2);
            dst[0] = pbc[sub].x; dst[BP] = pbc[sub].y; dst[2 * BP] = pbc[sub].z; dst[3 * BP] = pbc[sub].w;
        }
        __syncthreads();
        if (id.act) {
            float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 dv = *reinterpret_cast<const float4 *>(sd + id.cl * UP + 4 * q);
                const float4 uv = *reinterpret_cast<const float4 *>(su + id.cl * UP + 4 * q);
                float yq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = 4 * q + j;
                    const float dl = f4get(dv, j), uu = f4get(uv, j);
                    const float dlu = dl * uu;
                    const float4 Bq = *reinterpret_cast<const float4 *>(sB + t * BP + 4 * id.s);
                    h[0] = fast_exp2(dl * A2[0]) * h[0] + dlu * Bq.x;
                    h[1] = fast_exp2(dl * A2[1]) * h[1] + dlu * Bq.y;
                    h[2] = fast_exp2(dl * A2[2]) * h[2] + dlu * Bq.z;
                    h[3] = fast_exp2(dl * A2[3]) * h[3] + dlu * Bq.w;
                    if (FINAL) {
                        const float4 Cq = *reinterpret_cast<const float4 *>(sC + t * BP + 4 * id.s);
                        const float yp = Cq.x * h[0] + Cq.y * h[1] + Cq.z * h[2] + Cq.w * h[3];
                        yq[j] = quad_sum(yp) + Dd * uu;
                    } else {
                        dsum += dl;
                    }
                }
                if (FINAL && id.s == q) yv = make_float4(yq[0], yq[1], yq[2], yq[3]);
            }
            if (FINAL) store4(out + ((size_t)id.b * gm.dim + id.d) * L, t0 + 4 * id.s, L, vec, yv);
        }
    }
    if (!FINAL && id.act) {
        *reinterpret_cast<float4 *>(cstate + srow * NS + 4 * id.s) = make_float4(h[0], h[1], h[2], h[3]);
        if (id.s == 0) cdsum[srow] = dsum;
    }
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
#pragma unroll 4
    for (int i = 0; i < gm.nchunks; ++i) {
        const int c = reverse ? gm.nchunks - 1 - i : i;
        const size_t row = base + (size_t)c * gm.dim;
        const float S = cstate[row * NS + n];
        const float P = fast_exp2(A2 * cdsum[row]);
        cstate[row * NS + n] = H;
        H = P * H + S;
    }
}

// ------------------------------------------------------------------------------------------
// backward pass 1: reverse-local summaries.  q_l = a_l (q_{l+1} + dy_l C_l) run from q = 0 at the
// chunk end; the value at the chunk start is the affine offset of the chunk (slope is the same
// exp(A * dsum_c) as forward).
// ------------------------------------------------------------------------------------------
__global__ void selscan_bwd_local_kernel(const float *__restrict__ delta, const float *__restrict__ A,
                                         const float *__restrict__ Cm, const float *__restrict__ dbias,
                                         const float *__restrict__ dout, float *__restrict__ cq, ScanGeom gm,
                                         int softplus)
{
    extern __shared__ float4 smem4[];
    float *sg = reinterpret_cast<float *>(smem4);
    float *sd = sg + gm.CB * UP;
    float *sC = sd + gm.CB * UP;

    const LaneId id = lane_id(gm);
    const int tid = threadIdx.x, L = gm.L;
    const bool vec = (L & 3) == 0;
    const float *grow = dout + ((size_t)id.b * gm.dim + id.d) * L;
    const float *drow = delta + ((size_t)id.b * gm.dim + id.d) * L;
    const float *crow = tid < 64 ? Cm + (((size_t)id.b * gm.G + id.g) * NS + (tid >> 2)) * L : nullptr;
    const float bias = (dbias && id.act) ? dbias[id.d] : 0.f;

    float A2[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A2[i] = id.act ? A[id.d * NS + 4 * id.s + i] * LOG2E : 0.f;
        q[i] = 0.f;
    }
    const int tc0 = id.chunk * TC;
    float4 pg[NSUB], pd[NSUB], pc[NSUB];          // all loads of the chunk in flight at once (see forward)
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t0 = tc0 + sub * ST;
        pg[sub] = pd[sub] = pc[sub] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id.act) { pg[sub] = load4(grow, t0 + 4 * id.s, L, vec); pd[sub] = load4(drow, t0 + 4 * id.s, L, vec); }
        if (crow) pc[sub] = load4(crow, t0 + 4 * (tid & 3), L, vec);
    }
#pragma unroll
    for (int sub = NSUB - 1; sub >= 0; --sub) {
        const int t0 = tc0 + sub * ST;
        __syncthreads();
        if (id.act) {
            *reinterpret_cast<float4 *>(sg + id.cl * UP + 4 * id.s) = pg[sub];
            *reinterpret_cast<float4 *>(sd + id.cl * UP + 4 * id.s) =
                activate_delta(pd[sub], bias, softplus, t0 + 4 * id.s, L);
        }
        if (crow) {
            float *dst = sC + (4 * (tid & 3)) * BP + (tid >> 2);
            dst[0] = pc[sub].x; dst[BP] = pc[sub].y; dst[2 * BP] = pc[sub].z; dst[3 * BP] = pc[sub].w;
        }
        __syncthreads();
        if (id.act) {
#pragma unroll
            for (int qq = 3; qq >= 0; --qq) {
                const float4 dv = *reinterpret_cast<const float4 *>(sd + id.cl * UP + 4 * qq);
                const float4 gv = *reinterpret_cast<const float4 *>(sg + id.cl * UP + 4 * qq);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int t = 4 * qq + j;
                    const float dl = f4get(dv, j), gy = f4get(gv, j);
                    const float4 Cq = *reinterpret_cast<const float4 *>(sC + t * BP + 4 * id.s);
                    q[0] = fast_exp2(dl * A2[0]) * (q[0] + gy * Cq.x);
                    q[1] = fast_exp2(dl * A2[1]) * (q[1] + gy * Cq.y);
                    q[2] = fast_exp2(dl * A2[2]) * (q[2] + gy * Cq.z);
                    q[3] = fast_exp2(dl * A2[3]) * (q[3] + gy * Cq.w);
                }
            }
        }
    }
    if (id.act) {
        const size_t srow = ((size_t)id.b * gm.nchunks + id.chunk) * gm.dim + id.d;
        *reinterpret_cast<float4 *>(cq + srow * NS + 4 * id.s) = make_float4(q[0], q[1], q[2], q[3]);
    }
}

// Sum over the 16 channel-lanes of a wave (lane = 4 * channel + quad) of NV values per lane,
// reduce-scatter style: after the call, for v in [0, NV/16) ... the caller reads the result of value
// index (k) from the lane whose channel bits select it.  Generic butterfly with ds_bpermute
// (LDS crossbar pipe, not LDS memory), so the VALU only pays the select + add.
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

// Sum 32 per-lane values over the 16 channels of a wave (lane = 4 * channel + state quad, so the channel is
// lane bits 2..5).  Lane bits 5 and 4: reduce-scatter with v_permlane32_swap / v_permlane16_swap (the swap
// hands each half exactly the operand it needs: no selects, no LDS crossbar); lane bits 3 and 2: the four
// lanes {i, i+4, i+8, i+12} of a DPP row are one orbit of row_ror:4, so two fused DPP adds all-reduce them.
// Result: v[0..7] hold the channel sums of original values  j + 8 * bit4 + 16 * bit5,  identical in the
// four lanes of an orbit.
__device__ __forceinline__ void channel_reduce32(float (&v)[32])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) { swap32(v[i], v[i + 16]); v[i] += v[i + 16]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { swap16(v[i], v[i + 8]); v[i] += v[i + 8]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] += row_ror4(v[i]); v[i] += row_ror8(v[i]); }
}

// ------------------------------------------------------------------------------------------
// backward pass 3: per chunk -- forward sweep to recover the states entering sub-tiles 1..3
// (register checkpoints), then sub-tiles in reverse: per state, re-run 16 steps forward keeping
// h_{k-1} and a_k in registers, run the reverse recurrence and form every gradient.
// dB/dC: butterfly over the wave's 16 channels, then ds_add_f32 into the workgroup's [t][n]
// accumulators, stored once per sub-tile (plain stores when one workgroup covers the group).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128, 3)
selscan_bwd_kernel(const float *__restrict__ u, const float *__restrict__ delta, const float *__restrict__ A,
                   const float *__restrict__ Bm, const float *__restrict__ Cm, const float *__restrict__ Dv,
                   const float *__restrict__ dbias, const float *__restrict__ dout,
                   const float *__restrict__ cstate, const float *__restrict__ cq, float *__restrict__ du,
                   float *__restrict__ ddelta, float *__restrict__ dB, float *__restrict__ dC,
                   float *__restrict__ part, ScanGeom gm, int softplus, int atomic_bc)
{
    extern __shared__ float4 smem4[];
    float *su = reinterpret_cast<float *>(smem4);
    float *sd = su + gm.CB * UP;
    float *sg = sd + gm.CB * UP;
    float *sB = sg + gm.CB * UP;
    float *sC = sB + ST * BP;
    float *aB = sC + ST * BP;      // [t][n] accumulators, pitch NS
    float *aC = aB + ST * NS;

    const LaneId id = lane_id(gm);
    const int tid = threadIdx.x, L = gm.L;
    const bool vec = (L & 3) == 0;
    const size_t rowoff = ((size_t)id.b * gm.dim + id.d) * L;
    const float *urow = u + rowoff, *drow = delta + rowoff, *grow = dout + rowoff;
    const float *bcrow = nullptr;
    if (tid < 64)
        bcrow = Bm + (((size_t)id.b * gm.G + id.g) * NS + (tid >> 2)) * L;
    else if (tid < 128)
        bcrow = Cm + (((size_t)id.b * gm.G + id.g) * NS + ((tid - 64) >> 2)) * L;
    const float bias = (dbias && id.act) ? dbias[id.d] : 0.f;
    const float Dd = (Dv && id.act) ? Dv[id.d] : 0.f;
    const size_t srow = ((size_t)id.b * gm.nchunks + id.chunk) * gm.dim + id.d;
    const int tc0 = id.chunk * TC;

    float A2[4], Araw[4], h[4], ck[NSUB][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Araw[i] = id.act ? A[id.d * NS + 4 * id.s + i] : 0.f;
        A2[i] = Araw[i] * LOG2E;
        h[i] = 0.f;
    }
    if (id.act) {
        const float4 h0 = *reinterpret_cast<const float4 *>(cstate + srow * NS + 4 * id.s);
        h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w;
    }

    // ---- phase F: forward sweep over sub-tiles 0 .. NSUB-2, checkpointing entry states ----
    float4 pu, pd, pbc;
    pu = pd = pbc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id.act) { pu = load4(urow, tc0 + 4 * id.s, L, vec); pd = load4(drow, tc0 + 4 * id.s, L, vec); }
    if (tid < 64) pbc = load4(bcrow, tc0 + 4 * (tid & 3), L, vec);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ck[sub][i] = h[i];
        if (sub == NSUB - 1) break;
        const int t0 = tc0 + sub * ST;
        __syncthreads();
        if (id.act) {
            *reinterpret_cast<float4 *>(su + id.cl * UP + 4 * id.s) = pu;
            *reinterpret_cast<float4 *>(sd + id.cl * UP + 4 * id.s) =
                activate_delta(pd, bias, softplus, t0 + 4 * id.s, L);
        }
        if (tid < 64) {
            float *dst = sB + (4 * (tid & 3)) * BP + (tid >> 2);
            dst[0] = pbc.x; dst[BP] = pbc.y; dst[2 * BP] = pbc.z; dst[3 * BP] = pbc.w;
        }
        __syncthreads();
        if (sub + 2 < NSUB) {
            const int t = t0 + ST;
            if (id.act) { pu = load4(urow, t + 4 * id.s, L, vec); pd = load4(drow, t + 4 * id.s, L, vec); }
            if (tid < 64) pbc = load4(bcrow, t + 4 * (tid & 3), L, vec);
        }
        if (id.act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 dv = *reinterpret_cast<const float4 *>(sd + id.cl * UP + 4 * q);
                const float4 uv = *reinterpret_cast<const float4 *>(su + id.cl * UP + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dl = f4get(dv, j), dlu = dl * f4get(uv, j);
                    const float4 Bq = *reinterpret_cast<const float4 *>(sB + (4 * q + j) * BP + 4 * id.s);
                    h[0] = fast_exp2(dl * A2[0]) * h[0] + dlu * Bq.x;
                    h[1] = fast_exp2(dl * A2[1]) * h[1] + dlu * Bq.y;
                    h[2] = fast_exp2(dl * A2[2]) * h[2] + dlu * Bq.z;
                    h[3] = fast_exp2(dl * A2[3]) * h[3] + dlu * Bq.w;
                }
            }
        }
    }

    // ---- phase R: sub-tiles in reverse ----
    // Inactive lanes (padding of a partial wave) run the arithmetic on zeros so that the
    // cross-lane butterflies below always see all 64 lanes.
    const int clr = id.act ? id.cl : 0;
    float qc[4] = {0.f, 0.f, 0.f, 0.f};
    if (id.act) {
        const float4 q0 = *reinterpret_cast<const float4 *>(cq + srow * NS + 4 * id.s);
        qc[0] = q0.x; qc[1] = q0.y; qc[2] = q0.z; qc[3] = q0.w;
    }
    float dAacc[4] = {0.f, 0.f, 0.f, 0.f};
    float dDacc = 0.f, dbacc = 0.f;

#pragma unroll 1
    for (int sub = NSUB - 1; sub >= 0; --sub) {
        const int t0 = tc0 + sub * ST;
        float4 ru = make_float4(0.f, 0.f, 0.f, 0.f), rd = ru, rg = ru, rbc = ru;
        if (id.act) {
            ru = load4(urow, t0 + 4 * id.s, L, vec);
            rd = load4(drow, t0 + 4 * id.s, L, vec);
            rg = load4(grow, t0 + 4 * id.s, L, vec);
        }
        if (bcrow) rbc = load4(bcrow, t0 + 4 * (tid & 3), L, vec);
        __syncthreads();
        if (id.act) {
            *reinterpret_cast<float4 *>(su + id.cl * UP + 4 * id.s) = ru;
            *reinterpret_cast<float4 *>(sd + id.cl * UP + 4 * id.s) =
                activate_delta(rd, bias, softplus, t0 + 4 * id.s, L);
            *reinterpret_cast<float4 *>(sg + id.cl * UP + 4 * id.s) = rg;
        }
        if (bcrow)      // phase R keeps B / C as [n][t] rows (pitch BP): a state's 16 steps are 4 b128 reads
            *reinterpret_cast<float4 *>((tid < 64 ? sB : sC) + ((tid & 63) >> 2) * BP + 4 * (tid & 3)) = rbc;
        for (int i = tid; i < 2 * ST * NS; i += blockDim.x) aB[i] = 0.f;   // aB and aC are adjacent
        __syncthreads();

        float ddl[ST], duu[ST];
#pragma unroll
        for (int k = 0; k < ST; ++k) { ddl[k] = 0.f; duu[k] = 0.f; }
        const float actf = id.act ? 1.f : 0.f;
        const float *sdr = sd + clr * UP, *sur = su + clr * UP, *sgr = sg + clr * UP;
        // The lane's 4 states, one at a time.  The loop is NOT unrolled (register budget); the
        // per-state register arrays are rotated so that index 0 is always the current state.
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            // v[k] : a_k, later the dB term ; v[ST + k] : h_{k-1}, later the dC term
            float v[2 * ST];
            float hh = sub == 3 ? ck[3][0] : (sub == 2 ? ck[2][0] : (sub == 1 ? ck[1][0] : ck[0][0]));
            const float *sBn = sB + (4 * id.s + i) * BP, *sCn = sC + (4 * id.s + i) * BP;   // this state's rows
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 dv = *reinterpret_cast<const float4 *>(sdr + 4 * q);
                const float4 uv = *reinterpret_cast<const float4 *>(sur + 4 * q);
                const float4 bb = *reinterpret_cast<const float4 *>(sBn + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * q + j;
                    const float dlk = f4get(dv, j) * actf;
                    v[ST + k] = hh;
                    v[k] = fast_exp2(dlk * A2[0]);
                    hh = v[k] * hh + dlk * f4get(uv, j) * f4get(bb, j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the three phases of a state apart (register pressure)
            float qq = qc[0], dAi = dAacc[0];
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const float4 cc = *reinterpret_cast<const float4 *>(sCn + 4 * q);
                const float4 bb = *reinterpret_cast<const float4 *>(sBn + 4 * q);
                const float4 dv = *reinterpret_cast<const float4 *>(sdr + 4 * q);
                const float4 uv = *reinterpret_cast<const float4 *>(sur + 4 * q);
                const float4 gv = *reinterpret_cast<const float4 *>(sgr + 4 * q);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int k = 4 * q + j;
                    const float dlk = f4get(dv, j) * actf, uk = f4get(uv, j), gyk = f4get(gv, j) * actf;
                    const float Bv = f4get(bb, j), Cv = f4get(cc, j);
                    const float ak = v[k], hp = v[ST + k];
                    const float dlu = dlk * uk;
                    const float hk = ak * hp + dlu * Bv;                       // h_k
                    const float gh = qq + gyk * Cv;                             // dL/dh_k
                    const float t1 = gh * hp * ak;                              // dL/da_k * a_k
                    ddl[k] += t1 * Araw[0] + gh * Bv * uk;
                    duu[k] += gh * dlk * Bv;
                    dAi += t1 * dlk;
                    qq = ak * gh;
                    v[k] = gh * dlu;                                            // dB[k][n] term of this channel
                    v[ST + k] = gyk * hk;                                       // dC[k][n] term of this channel
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            qc[0] = qq;
            dAacc[0] = dAi;
            __builtin_amdgcn_sched_barrier(0);
            // sum over the wave's 16 channels (32 values -> 8, replicated over the 4 low channel lanes);
            // the lane with low channel bits j adds values 2j and 2j+1 to the workgroup tile
            channel_reduce32(v);
            {
                const int c = id.cl & 15;                       // real lane position, also for padding lanes
                const int j = c & 3;
                const float v0 = j == 0 ? v[0] : (j == 1 ? v[2] : (j == 2 ? v[4] : v[6]));
                const float v1 = j == 0 ? v[1] : (j == 1 ? v[3] : (j == 2 ? v[5] : v[7]));
                const int base = 2 * j + 8 * ((c >> 2) & 1) + 16 * ((c >> 3) & 1);     // [0,16): dB steps, [16,32): dC
                float *acc0 = (base < ST ? aB : aC) + (base & (ST - 1)) * NS + 4 * id.s + i;
                atomicAdd(acc0, v0);
                atomicAdd(acc0 + NS, v1);                       // base is even: base + 1 stays in the same half
            }
            rot4(A2); rot4(Araw); rot4(qc); rot4(dAacc);
            rot4(ck[0]); rot4(ck[1]); rot4(ck[2]); rot4(ck[3]);
        }
        // quad-reduce d(delta') and du, apply softplus', D skip; lane s keeps steps 4s..4s+3
        {
            float odd[4] = {0.f, 0.f, 0.f, 0.f}, odu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = quad_sum(ddl[4 * q + j]);
                    const float b2 = quad_sum(duu[4 * q + j]);
                    if (id.s == q) { odd[j] = a; odu[j] = b2; }
                }
            }
            if (id.act) {
                const float4 dv = *reinterpret_cast<const float4 *>(sd + id.cl * UP + 4 * id.s);
                const float4 uv = *reinterpret_cast<const float4 *>(su + id.cl * UP + 4 * id.s);
                const float4 gv = *reinterpret_cast<const float4 *>(sg + id.cl * UP + 4 * id.s);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dlj = f4get(dv, j), gyj = f4get(gv, j), uj = f4get(uv, j);
                    // d softplus(x)/dx = sigmoid(x) = 1 - exp(-softplus(x)); 1 when softplus is off
                    const float sp = softplus ? (1.f - __expf(-dlj)) : 1.f;
                    const bool inr = (t0 + 4 * id.s + j) < L;
                    odd[j] = inr ? odd[j] * sp : 0.f;
                    odu[j] = odu[j] + Dd * gyj;
                    dbacc += odd[j];
                    dDacc += gyj * uj;
                }
                store4(ddelta + rowoff, t0 + 4 * id.s, L, vec, make_float4(odd[0], odd[1], odd[2], odd[3]));
                store4(du + rowoff, t0 + 4 * id.s, L, vec, make_float4(odu[0], odu[1], odu[2], odu[3]));
            }
        }
        __syncthreads();
        // flush the [t][n] accumulators; consecutive threads -> consecutive l of one state row
        for (int i = tid; i < 2 * ST * NS; i += blockDim.x) {
            const int which = i / (ST * NS);           // 0: dB, 1: dC
            const int r = i - which * ST * NS;
            const int n = r / ST, k = r - n * ST;
            const int t = t0 + k;
            if (t < L) {
                float *dst = (which ? dC : dB) + (((size_t)id.b * gm.G + id.g) * NS + n) * L + t;
                const float val = (which ? aC : aB)[k * NS + n];
                if (atomic_bc) atomicAdd(dst, val); else *dst = val;
            }
        }
    }
    // per-chunk partial sums of dA (16 per channel), dD, ddelta_bias -> selscan_reduce_partials
    {
        const float dDs = quad_sum(dDacc), dbs = quad_sum(dbacc);
        if (id.act) {
            float *prow = part + srow * PP;
            *reinterpret_cast<float4 *>(prow + 4 * id.s) = make_float4(dAacc[0], dAacc[1], dAacc[2], dAacc[3]);
            if (id.s == 0) { prow[NS] = dDs; prow[NS + 1] = dbs; }
        }
    }
}

// part[b][chunk][d][PP] -> dA[d][16], dD[d], ddbias[d]; one workgroup of 256 per channel
__global__ void selscan_reduce_partials(const float *__restrict__ part, float *__restrict__ dA,
                                        float *__restrict__ dD, float *__restrict__ ddbias, ScanGeom gm)
{
    const int d = blockIdx.x;
    const int j = threadIdx.x % 18, r0 = threadIdx.x / 18;      // 14 row-lanes x 18 columns = 252 threads
    __shared__ float red[14][18];
    float acc = 0.f;
    const int rows = gm.batch * gm.nchunks;
    if (r0 < 14)
        for (int r = r0; r < rows; r += 14) acc += part[((size_t)r * gm.dim + d) * PP + j];
    if (r0 < 14) red[r0][j] = acc;
    __syncthreads();
    if (threadIdx.x < 18) {
        float s = 0.f;
        for (int r = 0; r < 14; ++r) s += red[r][threadIdx.x];
        if (threadIdx.x < NS) dA[d * NS + threadIdx.x] = s;
        else if (threadIdx.x == NS) { if (dD) dD[d] = s; }
        else { if (ddbias) ddbias[d] = s; }
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

inline int block_threads(const ScanGeom &gm)
{
    int t = ((4 * gm.CB + 63) / 64) * 64;
    return t < 128 ? 128 : t;
}

}  // namespace

extern "C" size_t mlagg_selscan_state_floats(int batch, int dim, int L, int N)
{
    const size_t nchunks = (L + TC - 1) / TC;
    return (size_t)batch * nchunks * dim * (N + 1);
}

extern "C" size_t mlagg_selscan_bwd_workspace_floats(int batch, int dim, int L, int N)
{
    const size_t nchunks = (L + TC - 1) / TC;
    return (size_t)batch * nchunks * dim * (N + PP);
}

extern "C" int mlagg_selscan_fwd(const float *u, const float *delta, const float *A, const float *B,
                                 const float *C, const float *D, const float *delta_bias, float *out,
                                 float *chunk_state, int batch, int dim, int L, int N, int G,
                                 int delta_softplus, void *stream)
{
    if (!u || !delta || !A || !B || !C || !out || !chunk_state) return MLAGG_E_NULLPTR;
    ScanGeom gm;
    if (int rc = make_geom(gm, batch, dim, L, N, G, SCAN_CB)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *cstate = chunk_state;
    float *cdsum = chunk_state + (size_t)batch * gm.nchunks * dim * NS;
    const dim3 grid(gm.nchunks, G * gm.nblk, batch), block(block_threads(gm));
    const size_t lds = (size_t)(2 * gm.CB * UP + 2 * ST * BP) * sizeof(float);
    { MLAGG_TIMED(K_SELSCAN_FWD_LOCAL, st); hipLaunchKernelGGL(selscan_fwd_kernel<false>, grid, block, lds, st, u, delta, A, B, C, D, delta_bias, out,
                       cstate, cdsum, gm, delta_softplus); }
    { MLAGG_TIMED(K_SELSCAN_PREFIX, st); hipLaunchKernelGGL(selscan_chunk_prefix, dim3((dim * NS + 255) / 256, batch), dim3(256), 0, st, A, cstate,
                       cdsum, gm, 0); }
    { MLAGG_TIMED(K_SELSCAN_FWD_FINAL, st); hipLaunchKernelGGL(selscan_fwd_kernel<true>, grid, block, lds, st, u, delta, A, B, C, D, delta_bias, out,
                       cstate, cdsum, gm, delta_softplus); }
    return (int)hipGetLastError();
}

extern "C" int mlagg_selscan_bwd(const float *u, const float *delta, const float *A, const float *B,
                                 const float *C, const float *D, const float *delta_bias, const float *dout,
                                 const float *chunk_state, float *du, float *ddelta, float *dA, float *dB,
                                 float *dC, float *dD, float *ddelta_bias, float *workspace, int batch, int dim,
                                 int L, int N, int G, int delta_softplus, void *stream)
{
    if (!u || !delta || !A || !B || !C || !dout || !chunk_state || !du || !ddelta || !dA || !dB || !dC ||
        !workspace)
        return MLAGG_E_NULLPTR;
    ScanGeom gm;
    if (int rc = make_geom(gm, batch, dim, L, N, G, SCAN_CB)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *cstate = chunk_state;
    const float *cdsum = chunk_state + (size_t)batch * gm.nchunks * dim * NS;
    float *cq = workspace;
    float *part = workspace + (size_t)batch * gm.nchunks * dim * NS;
    const dim3 grid(gm.nchunks, G * gm.nblk, batch), block(block_threads(gm));
    // The main kernel runs at 2 waves per SIMD (247 VGPRs).  Workgroups of 32 channels (2 waves) let a CU
    // hold 4 independent workgroups = 8 waves, 2 per SIMD, instead of one 6-wave workgroup spread 2/2/1/1;
    // the price is a 3-way float-atomic accumulation of dB/dC across the workgroups of a group.
    ScanGeom gb;
    if (int rc = make_geom(gb, batch, dim, L, N, G, BWD_CB)) return rc;
    const dim3 gridb(gb.nchunks, G * gb.nblk, batch), blockb(block_threads(gb));
    const int atomic_bc = gb.nblk > 1;
    if (atomic_bc) {
        const size_t bytes = (size_t)batch * G * NS * L * sizeof(float);
        (void)hipMemsetAsync(dB, 0, bytes, st);
        (void)hipMemsetAsync(dC, 0, bytes, st);
    }
    const size_t lds1 = (size_t)(2 * gm.CB * UP + ST * BP) * sizeof(float);
    { MLAGG_TIMED(K_SELSCAN_BWD_LOCAL, st); hipLaunchKernelGGL(selscan_bwd_local_kernel, grid, block, lds1, st, delta, A, C, delta_bias, dout, cq, gm,
                       delta_softplus); }
    { MLAGG_TIMED(K_SELSCAN_PREFIX, st); hipLaunchKernelGGL(selscan_chunk_prefix, dim3((dim * NS + 255) / 256, batch), dim3(256), 0, st, A, cq, cdsum,
                       gm, 1); }
    const size_t lds3 = (size_t)(3 * gb.CB * UP + 2 * ST * BP + 2 * ST * NS) * sizeof(float);
    { MLAGG_TIMED(K_SELSCAN_BWD, st); hipLaunchKernelGGL(selscan_bwd_kernel, gridb, blockb, lds3, st, u, delta, A, B, C, D, delta_bias, dout, cstate,
                       cq, du, ddelta, dB, dC, part, gb, delta_softplus, atomic_bc); }
    { MLAGG_TIMED(K_SELSCAN_REDUCE, st); hipLaunchKernelGGL(selscan_reduce_partials, dim3(dim), dim3(256), 0, st, part, dA, dD, ddelta_bias, gm); }
    return (int)hipGetLastError();
}
