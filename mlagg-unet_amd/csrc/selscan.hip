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
#include "selscan_common.h"

namespace {



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


// Low-rank delta (LR kernels): the fused form of SS2D_skip's `einsum("b k r l, k d r -> b k d l", dts, dt_projs_weight)`
// (reference MambaSkip.py:430-436).  The R rank rows dtr (B, G, R, L) of the workgroup's group are fetched per
// 16-step sub-tile by the first 4*R threads (one float4 each), parked in LDS as sR[r][t], and every lane forms
// raw delta = sum_r Wdt[d][r] * dtr[r][t] for its channel and its 4 steps.  delta itself never exists in memory.
__device__ __forceinline__ float4 load_rank_rows(const float *__restrict__ dtr_rows, int R, int tid, int t0, int L, bool vec)
{
    return tid < 4 * R ? load4(dtr_rows + (size_t)(tid >> 2) * L, t0 + 4 * (tid & 3), L, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void stage_rank_rows(float *__restrict__ sR, int R, int tid, const float4 &v)
{
    if (tid < 4 * R) *reinterpret_cast<float4 *>(sR + (tid >> 2) * ST + 4 * (tid & 3)) = v;
}
__device__ __forceinline__ float4 lowrank_delta(const float *__restrict__ sR, int R, int s, const float (&w)[RMAX])
{
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        if (i < R) {
            const float4 rv = *reinterpret_cast<const float4 *>(sR + i * ST + 4 * s);
            a.x += w[i] * rv.x; a.y += w[i] * rv.y; a.z += w[i] * rv.z; a.w += w[i] * rv.w;
        }
    }
    return a;
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
// (launch bounds: at least 4 waves per SIMD, i.e. at most 128 VGPRs -- without the cap the branch-free
// FAST bodies are scheduled with every LDS read of a sub-tile hoisted: 319 / 512 registers, one wave per SIMD.)
// FAST: L % 4 == 0, every group a whole number of 32-channel blocks, 128 threads: every lane is active and -- in every chunk
// but a ragged last one -- every access in range, so all loads and stores are UNCONDITIONAL.  In the generic form each guarded access is
// its own exec-masked block, the compiler cannot count the loads and drains them (`s_waitcnt vmcnt(0)`) at every join: the
// "12 float4 in flight" of the prologue completed one after the other, and each store waited for the one before it.
template <bool FINAL, bool LR, bool FAST>
__global__ void __launch_bounds__(128, 4) selscan_fwd_kernel(const float *__restrict__ u, const float *__restrict__ delta,
                                   const float *__restrict__ Wdt, int R,
                                   const float *__restrict__ A, const float *__restrict__ Bm,
                                   const float *__restrict__ Cm, const float *__restrict__ Dv,
                                   const float *__restrict__ dbias, float *__restrict__ out,
                                   float *__restrict__ cstate, float *__restrict__ cdsum, float *__restrict__ csub,
                                   ScanGeom gm, int softplus)
{
    extern __shared__ float4 smem4[];
    float *su = reinterpret_cast<float *>(smem4);
    float *sd = su + gm.CB * UP;
    float *sB = sd + gm.CB * UP;
    float *sC = sB + ST * BP;
    float *sR = sC + ST * BP;          // LR: two buffers of RMAX x ST rank rows

    LaneId id = lane_id(gm);
    if (FAST) id.act = true;
    const int tid = threadIdx.x, L = gm.L;
    const bool vec = (L & 3) == 0;
    // FAST kernels serve every L % 4 == 0: the (at most one) ragged chunk at the end of the sequence takes the guarded
    // forms below, decided per workgroup (uniform)
    const bool whole = FAST && (int)(blockIdx.x + 1) * TC <= L;
    const float *urow = u + ((size_t)id.b * gm.dim + id.d) * L;
    const float *drow = LR ? delta + (((size_t)id.b * gm.G + id.g) * R) * L : delta + ((size_t)id.b * gm.dim + id.d) * L;
    float wdt[RMAX] = {0.f, 0.f, 0.f, 0.f};
    if (LR && id.act)
        for (int i = 0; i < R; ++i) wdt[i] = Wdt[(size_t)id.d * R + i];
    const float *bcrow = nullptr;
    if (tid < 64)
        bcrow = Bm + (((size_t)id.b * gm.G + id.g) * NS + (tid >> 2)) * L;
    else if (FINAL && tid < 128)
        bcrow = Cm + (((size_t)id.b * gm.G + id.g) * NS + ((tid - 64) >> 2)) * L;
    // FAST: a valid row for every thread (the second wave of pass 1 re-reads B rows and drops them)
    const float *bcfast = (FINAL && tid >= 64 ? Cm : Bm) + (((size_t)id.b * gm.G + id.g) * NS + ((tid & 63) >> 2)) * L;
    const float *rkfast = LR ? drow + (size_t)min(tid >> 2, R - 1) * L : drow;     // rank row of this thread (clamped)
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
    float4 pu[NSUB], pd[NSUB], pbc[NSUB];         // pd: the lane's delta segment, or (LR) one rank-row segment
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t0 = tc0 + sub * ST;
        if (whole) {
            pu[sub] = *reinterpret_cast<const float4 *>(urow + t0 + 4 * id.s);
            pd[sub] = LR ? *reinterpret_cast<const float4 *>(rkfast + t0 + 4 * (tid & 3))
                         : *reinterpret_cast<const float4 *>(drow + t0 + 4 * id.s);
            pbc[sub] = *reinterpret_cast<const float4 *>(bcfast + t0 + 4 * (tid & 3));
            continue;
        }
        pu[sub] = pd[sub] = pbc[sub] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LR) pd[sub] = load_rank_rows(drow, R, tid, t0, L, vec);
        if (id.act) {
            pu[sub] = load4(urow, t0 + 4 * id.s, L, vec);
            if (!LR) pd[sub] = load4(drow, t0 + 4 * id.s, L, vec);
        }
        if (bcrow) pbc[sub] = load4(bcrow, t0 + 4 * (tid & 3), L, vec);
    }
    if (FAST) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) { pin4(pu[sub]); pin4(pd[sub]); pin4(pbc[sub]); }
    }
    if (LR) stage_rank_rows(sR, R, tid, pd[0]);       // rank rows run one sub-tile ahead of the barriers below

#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t0 = tc0 + sub * ST;
        if (FINAL && sub > 0 && id.act)       // state entering 8-step tile 2 * sub: saved so that backward does not re-sweep the chunk
            *reinterpret_cast<float4 *>(csub + ((srow / gm.dim * NT8 + (2 * sub - 1)) * gm.dim + id.d) * NS + 4 * id.s) =
                make_float4(h[0], h[1], h[2], h[3]);
        lds_barrier();
        if (id.act) {
            *reinterpret_cast<float4 *>(su + id.cl * UP + 4 * id.s) = pu[sub];
            *reinterpret_cast<float4 *>(sd + id.cl * UP + 4 * id.s) =
                activate_delta(LR ? lowrank_delta(sR + (sub & 1) * RMAX * ST, R, id.s, wdt) : pd[sub], bias, softplus,
                               whole ? 0 : t0 + 4 * id.s, whole ? 4 : L);
        }
        if (LR && sub + 1 < NSUB) stage_rank_rows(sR + ((sub + 1) & 1) * RMAX * ST, R, tid, pd[sub + 1]);
        if (bcrow) {
            float *dst = (tid < 64 ? sB : sC) + (4 * (tid & 3)) * BP + ((tid & 63) >> 2);
            dst[0] = pbc[sub].x; dst[BP] = pbc[sub].y; dst[2 * BP] = pbc[sub].z; dst[3 * BP] = pbc[sub].w;
        }
        lds_barrier();
        if (id.act) {
            float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of a 4-step group with the group (register pressure)
                if (FINAL && q == 2)            // ... and the state entering the odd 8-step tile 2 * sub + 1
                    *reinterpret_cast<float4 *>(csub + ((srow / gm.dim * NT8 + 2 * sub) * gm.dim + id.d) * NS + 4 * id.s) =
                        make_float4(h[0], h[1], h[2], h[3]);
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
            if (FINAL) {
                if (whole) *reinterpret_cast<float4 *>(out + ((size_t)id.b * gm.dim + id.d) * L + t0 + 4 * id.s) = yv;
                else store4(out + ((size_t)id.b * gm.dim + id.d) * L, t0 + 4 * id.s, L, vec, yv);
            }
        }
        // the sub-tile's recurrence is FINISHED here: without this anchor the branch-free pass-1 body has no side effect
        // between the barriers, and the compiler parks the LDS reads of all four sub-tiles (in scratch) to run the whole
        // chain after the last barrier
        if (FAST) asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(dsum));
    }
    if (!FINAL && id.act) {
        *reinterpret_cast<float4 *>(cstate + srow * NS + 4 * id.s) = make_float4(h[0], h[1], h[2], h[3]);
        if (id.s == 0) cdsum[srow] = dsum;
    }
}


// ------------------------------------------------------------------------------------------
// backward pass 1: reverse-local summaries.  q_l = a_l (q_{l+1} + dy_l C_l) run from q = 0 at the
// chunk end; the value at the chunk start is the affine offset of the chunk (slope is the same
// exp(A * dsum_c) as forward).
// ------------------------------------------------------------------------------------------
template <bool LR, bool FAST>
__global__ void __launch_bounds__(128, 4) selscan_bwd_local_kernel(const float *__restrict__ delta, const float *__restrict__ Wdt, int R,
                                         const float *__restrict__ A,
                                         const float *__restrict__ Cm, const float *__restrict__ dbias,
                                         const float *__restrict__ dout, float *__restrict__ cq, ScanGeom gm,
                                         int softplus)
{
    extern __shared__ float4 smem4[];
    float *sg = reinterpret_cast<float *>(smem4);
    float *sd = sg + gm.CB * UP;
    float *sC = sd + gm.CB * UP;
    float *sR = sC + ST * BP;          // LR: two buffers of RMAX x ST rank rows

    LaneId id = lane_id(gm);
    if (FAST) id.act = true;                      // see selscan_fwd_kernel
    const int tid = threadIdx.x, L = gm.L;
    const bool vec = (L & 3) == 0;
    const bool whole = FAST && (int)(blockIdx.x + 1) * TC <= L;
    const float *grow = dout + ((size_t)id.b * gm.dim + id.d) * L;
    const float *drow = LR ? delta + (((size_t)id.b * gm.G + id.g) * R) * L : delta + ((size_t)id.b * gm.dim + id.d) * L;
    float wdt[RMAX] = {0.f, 0.f, 0.f, 0.f};
    if (LR && id.act)
        for (int i = 0; i < R; ++i) wdt[i] = Wdt[(size_t)id.d * R + i];
    const float *crow = tid < 64 ? Cm + (((size_t)id.b * gm.G + id.g) * NS + (tid >> 2)) * L : nullptr;
    const float *cfast = Cm + (((size_t)id.b * gm.G + id.g) * NS + ((tid & 63) >> 2)) * L;
    const float *rkfast = LR ? drow + (size_t)min(tid >> 2, R - 1) * L : drow;
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
        if (whole) {
            pg[sub] = *reinterpret_cast<const float4 *>(grow + t0 + 4 * id.s);
            pd[sub] = LR ? *reinterpret_cast<const float4 *>(rkfast + t0 + 4 * (tid & 3))
                         : *reinterpret_cast<const float4 *>(drow + t0 + 4 * id.s);
            pc[sub] = *reinterpret_cast<const float4 *>(cfast + t0 + 4 * (tid & 3));
            continue;
        }
        pg[sub] = pd[sub] = pc[sub] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LR) pd[sub] = load_rank_rows(drow, R, tid, t0, L, vec);
        if (id.act) {
            pg[sub] = load4(grow, t0 + 4 * id.s, L, vec);
            if (!LR) pd[sub] = load4(drow, t0 + 4 * id.s, L, vec);
        }
        if (crow) pc[sub] = load4(crow, t0 + 4 * (tid & 3), L, vec);
    }
    if (FAST) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) { pin4(pg[sub]); pin4(pd[sub]); pin4(pc[sub]); }
    }
    if (LR) stage_rank_rows(sR + ((NSUB - 1) & 1) * RMAX * ST, R, tid, pd[NSUB - 1]);
#pragma unroll
    for (int sub = NSUB - 1; sub >= 0; --sub) {
        const int t0 = tc0 + sub * ST;
        lds_barrier();
        if (id.act) {
            *reinterpret_cast<float4 *>(sg + id.cl * UP + 4 * id.s) = pg[sub];
            *reinterpret_cast<float4 *>(sd + id.cl * UP + 4 * id.s) =
                activate_delta(LR ? lowrank_delta(sR + (sub & 1) * RMAX * ST, R, id.s, wdt) : pd[sub], bias, softplus,
                               whole ? 0 : t0 + 4 * id.s, whole ? 4 : L);
        }
        if (LR && sub > 0) stage_rank_rows(sR + ((sub - 1) & 1) * RMAX * ST, R, tid, pd[sub - 1]);
        if (crow) {
            float *dst = sC + (4 * (tid & 3)) * BP + (tid >> 2);
            dst[0] = pc[sub].x; dst[BP] = pc[sub].y; dst[2 * BP] = pc[sub].z; dst[3 * BP] = pc[sub].w;
        }
        lds_barrier();
        if (id.act) {
#pragma unroll
            for (int qq = 3; qq >= 0; --qq) {
                __builtin_amdgcn_sched_barrier(0);      // as in the forward kernel
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
        if (FAST) asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));      // see selscan_fwd_kernel
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

// Sum 32 per-lane values over the 16 channels of a wave (lane = 4 * channel + state quad, so the channel is
// lane bits 2..5).  Lane bits 5 and 4: reduce-scatter with v_permlane32_swap / v_permlane16_swap (the swap
// hands each half exactly the operand it needs: no selects, no LDS crossbar); lane bits 3 and 2: the four
// lanes {i, i+4, i+8, i+12} of a DPP row are one orbit of row_ror:4, so two fused DPP adds all-reduce them.
// Result: v[0..7] hold the channel sums of original values  j + 8 * bit4 + 16 * bit5,  identical in the
// four lanes of an orbit.
__device__ __forceinline__ void channel_reduce32(float (&v)[33])      // dB terms v[0..15], dC terms v[17..32]
{
#pragma unroll
    for (int i = 0; i < 16; ++i) { swap32(v[i], v[i + 17]); v[i] += v[i + 17]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { swap16(v[i], v[i + 8]); v[i] += v[i + 8]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] += row_ror4(v[i]); v[i] += row_ror8(v[i]); }
}

// Same reduce-scatter for 16 values per lane: afterwards v[c & 3] of the lane with wave-channel c (lane bits 2..5)
// is the sum over the wave's 16 channels of original value index c, for the lane's own state quad / step quad s.
__device__ __forceinline__ void channel_reduce16(float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) { swap32(v[i], v[i + 8]); v[i] += v[i + 8]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { swap16(v[i], v[i + 4]); v[i] += v[i + 4]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] += row_ror4(v[i]); v[i] += row_ror8(v[i]); }
}

// ------------------------------------------------------------------------------------------
// backward pass 3: per chunk, sub-tiles in reverse (their entry states were saved by the forward pass):
// per state, re-run 16 steps forward keeping
// h_{k-1} and a_k in registers, run the reverse recurrence and form every gradient.
// dB/dC: butterfly over the wave's 16 channels, then ds_add_f32 into the workgroup's [t][n]
// accumulators, stored once per sub-tile (plain stores when one workgroup covers the group).
// ------------------------------------------------------------------------------------------
template <bool LR>
__global__ void __launch_bounds__(128, 3)
selscan_bwd_kernel(const float *__restrict__ u, const float *__restrict__ delta, const float *__restrict__ Wdt, int R,
                   const float *__restrict__ A,
                   const float *__restrict__ Bm, const float *__restrict__ Cm, const float *__restrict__ Dv,
                   const float *__restrict__ dbias, const float *__restrict__ dout,
                   const float *__restrict__ cstate, const float *__restrict__ csub, const float *__restrict__ cq,
                   float *__restrict__ du,
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
    float *aR = aC + ST * NS;      // LR: [r][t] accumulators of d(dtr), pitch ST (adjacent to aB / aC: zeroed together)
    float *sR = aR + (LR ? RMAX * ST : 0);    // LR: the sub-tile's dtr rows [r][t]
    float *sW = sR + (LR ? RMAX * ST : 0);    // LR: per-thread {Wdt[d][0..3], dWdt accumulators[0..3]} -- kept out of the register budget
    float *sZ = sW + (LR ? 8 * 128 : 0);   // one all-zero row: what the padding lanes of a partial wave read as u / delta / dy
    float *sx = sZ + UP;                   // phase R: delta' * u per (channel, step) -- state-independent, formed once

    const LaneId id = lane_id(gm);
    const int tid = threadIdx.x, L = gm.L;
    const bool vec = (L & 3) == 0;
    const size_t rowoff = ((size_t)id.b * gm.dim + id.d) * L;
    const float *urow = u + rowoff, *grow = dout + rowoff;
    const float *drow = LR ? delta + (((size_t)id.b * gm.G + id.g) * R) * L : delta + rowoff;
    if (LR) {
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id.act) {
            w4.x = Wdt[(size_t)id.d * R];
            if (R > 1) w4.y = Wdt[(size_t)id.d * R + 1];
            if (R > 2) w4.z = Wdt[(size_t)id.d * R + 2];
            if (R > 3) w4.w = Wdt[(size_t)id.d * R + 3];
        }
        *reinterpret_cast<float4 *>(sW + 8 * tid) = w4;                       // private slot: no barrier needed
        *reinterpret_cast<float4 *>(sW + 8 * tid + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *bcrow = nullptr;
    if (tid < 64)
        bcrow = Bm + (((size_t)id.b * gm.G + id.g) * NS + (tid >> 2)) * L;
    else if (tid < 128)
        bcrow = Cm + (((size_t)id.b * gm.G + id.g) * NS + ((tid - 64) >> 2)) * L;
    const float bias = (dbias && id.act) ? dbias[id.d] : 0.f;
    const float Dd = (Dv && id.act) ? Dv[id.d] : 0.f;
    const size_t srow = ((size_t)id.b * gm.nchunks + id.chunk) * gm.dim + id.d;
    const int tc0 = id.chunk * TC;

    float A2[4], Araw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Araw[i] = id.act ? A[id.d * NS + 4 * id.s + i] : 0.f;
        A2[i] = Araw[i] * LOG2E;
    }
    if (tid < UP) sZ[tid] = 0.f;           // visible after the first barrier below

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
        float hent[4] = {0.f, 0.f, 0.f, 0.f};      // the 4 states entering this sub-tile (saved by the forward pass)
        if (id.act) {
            const float *src = sub == 0 ? cstate + srow * NS
                                        : csub + ((srow / gm.dim * NT8 + (2 * sub - 1)) * gm.dim + id.d) * NS;
            const float4 h0 = *reinterpret_cast<const float4 *>(src + 4 * id.s);
            hent[0] = h0.x; hent[1] = h0.y; hent[2] = h0.z; hent[3] = h0.w;
        }
        if (LR) rd = load_rank_rows(drow, R, tid, t0, L, vec);
        if (id.act) {
            ru = load4(urow, t0 + 4 * id.s, L, vec);
            if (!LR) rd = load4(drow, t0 + 4 * id.s, L, vec);
            rg = load4(grow, t0 + 4 * id.s, L, vec);
        }
        if (bcrow) rbc = load4(bcrow, t0 + 4 * (tid & 3), L, vec);
        __syncthreads();
        if (LR) {                                   // rank rows first: the activation below (and the dWdt products
            stage_rank_rows(sR, R, tid, rd);        // at the end of the sub-tile) read them
            __syncthreads();
        }
        if (id.act) {
            *reinterpret_cast<float4 *>(su + id.cl * UP + 4 * id.s) = ru;
            float4 raw = rd;
            if (LR) {
                const float4 w4 = *reinterpret_cast<const float4 *>(sW + 8 * tid);
                const float wv[RMAX] = {w4.x, w4.y, w4.z, w4.w};
                raw = lowrank_delta(sR, R, id.s, wv);
            }
            const float4 da = activate_delta(raw, bias, softplus, t0 + 4 * id.s, L);
            *reinterpret_cast<float4 *>(sd + id.cl * UP + 4 * id.s) = da;
            *reinterpret_cast<float4 *>(sx + id.cl * UP + 4 * id.s) = make_float4(da.x * ru.x, da.y * ru.y, da.z * ru.z, da.w * ru.w);
            *reinterpret_cast<float4 *>(sg + id.cl * UP + 4 * id.s) = rg;
        }
        if (bcrow)      // phase R keeps B / C as [n][t] rows (pitch BP): a state's 16 steps are 4 b128 reads
            *reinterpret_cast<float4 *>((tid < 64 ? sB : sC) + ((tid & 63) >> 2) * BP + 4 * (tid & 3)) = rbc;
        for (int i = tid; i < 2 * ST * NS + (LR ? RMAX * ST : 0); i += blockDim.x) aB[i] = 0.f;   // aB, aC, aR are adjacent
        __syncthreads();

        // per step, summed over the lane's 4 states:  sT = sum t1 * A  and  sG = sum gh * B;  then
        // d(delta')[k] = sT + u_k * sG  and  du[k] = delta'_k * sG  (u, delta' do not depend on the state)
        float sT[ST], sG[ST];
#pragma unroll
        for (int k = 0; k < ST; ++k) { sT[k] = 0.f; sG[k] = 0.f; }
        // padding lanes read the zero row: their arithmetic runs on zeros without any masking multiplies
        const float *sdr = id.act ? sd + clr * UP : sZ, *sgr = id.act ? sg + clr * UP : sZ, *sxr = id.act ? sx + clr * UP : sZ;
        // The lane's 4 states, one at a time.  The loop is NOT unrolled (register budget); the
        // per-state register arrays are rotated so that index 0 is always the current state.
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            // v[k] : a_k, later the dB term ;  v[ST + k], k = 0..16 : h_{k-1} (v[2 ST] = h_15); the dC term of step k
            // overwrites v[ST + 1 + k] (= h_k, not needed below step k)
            float v[2 * ST + 1];
            float hh = hent[0];
            const float *sBn = sB + (4 * id.s + i) * BP, *sCn = sC + (4 * id.s + i) * BP;   // this state's rows
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 dv = *reinterpret_cast<const float4 *>(sdr + 4 * q);
                const float4 xv = *reinterpret_cast<const float4 *>(sxr + 4 * q);
                const float4 bb = *reinterpret_cast<const float4 *>(sBn + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * q + j;
                    v[ST + k] = hh;
                    v[k] = fast_exp2(f4get(dv, j) * A2[0]);
                    hh = v[k] * hh + f4get(xv, j) * f4get(bb, j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            v[2 * ST] = hh;
            __builtin_amdgcn_sched_barrier(0);     // keep the three phases of a state apart (register pressure)
            float qq = qc[0], dAi = dAacc[0];
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const float4 cc = *reinterpret_cast<const float4 *>(sCn + 4 * q);
                const float4 bb = *reinterpret_cast<const float4 *>(sBn + 4 * q);
                const float4 dv = *reinterpret_cast<const float4 *>(sdr + 4 * q);
                const float4 xv = *reinterpret_cast<const float4 *>(sxr + 4 * q);
                const float4 gv = *reinterpret_cast<const float4 *>(sgr + 4 * q);
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int k = 4 * q + j;
                    const float gyk = f4get(gv, j);
                    const float ak = v[k], hp = v[ST + k], hk = v[ST + k + 1];
                    const float gh = qq + gyk * f4get(cc, j);                   // dL/dh_k
                    qq = ak * gh;                                               // carried to step k-1
                    const float t1 = qq * hp;                                   // dL/da_k * a_k
                    sT[k] += t1 * Araw[0];
                    sG[k] += gh * f4get(bb, j);
                    dAi += t1 * f4get(dv, j);
                    v[k] = gh * f4get(xv, j);                                   // dB[k][n] term of this channel
                    v[ST + k + 1] = gyk * hk;                                   // dC[k][n] term of this channel
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
            rot4(A2); rot4(Araw); rot4(qc); rot4(dAacc); rot4(hent);
        }
        // quad-reduce d(delta') and du, apply softplus', D skip; lane s keeps steps 4s..4s+3
        {
            float odd[4] = {0.f, 0.f, 0.f, 0.f}, odu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = quad_sum(sT[4 * q + j]);
                    const float b2 = quad_sum(sG[4 * q + j]);
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
                    odd[j] = inr ? (odd[j] + uj * odu[j]) * sp : 0.f;           // (sT + u * sG) * softplus'
                    odu[j] = dlj * odu[j] + Dd * gyj;                           // delta' * sG + D * dy
                    dbacc += odd[j];
                    dDacc += gyj * uj;
                }
                if (!LR) store4(ddelta + rowoff, t0 + 4 * id.s, L, vec, make_float4(odd[0], odd[1], odd[2], odd[3]));
                store4(du + rowoff, t0 + 4 * id.s, L, vec, make_float4(odu[0], odu[1], odu[2], odu[3]));
            }
            if (LR) {
                __builtin_amdgcn_sched_barrier(0);
                // d(raw delta) of (channel, steps 4s..4s+3) is in odd[] (zero in padding lanes).
                //   dWdt[d][r]  += sum_t odd * dtr[r][t]          -> per-lane accumulators, reduced with dA
                //   d(dtr)[r][t] = sum_d odd * Wdt[d][r]          -> sum over the wave's 16 channels, then LDS / global adds
                float pv[16];
                const float4 w4 = *reinterpret_cast<const float4 *>(sW + 8 * tid);
                float4 dw = *reinterpret_cast<const float4 *>(sW + 8 * tid + 4);
                const float wv[RMAX] = {w4.x, w4.y, w4.z, w4.w};
                float dwv[RMAX] = {dw.x, dw.y, dw.z, dw.w};
#pragma unroll
                for (int i = 0; i < RMAX; ++i) {
                    if (i < R) {
                        const float4 rv = *reinterpret_cast<const float4 *>(sR + i * ST + 4 * id.s);
                        dwv[i] += odd[0] * rv.x + odd[1] * rv.y + odd[2] * rv.z + odd[3] * rv.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) pv[4 * i + j] = odd[j] * wv[i];
                }
                *reinterpret_cast<float4 *>(sW + 8 * tid + 4) = make_float4(dwv[0], dwv[1], dwv[2], dwv[3]);
                channel_reduce16(pv);
                const int c = id.cl & 15, j = c & 3;                 // this lane owns value index c = 4 * r + step
                const float val = j == 0 ? pv[0] : (j == 1 ? pv[1] : (j == 2 ? pv[2] : pv[3]));
                if ((c >> 2) < R) atomicAdd(aR + (c >> 2) * ST + 4 * id.s + j, val);
                __builtin_amdgcn_sched_barrier(0);
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
        if (LR) {
            for (int i = tid; i < R * ST; i += blockDim.x) {
                const int r = i / ST, k = i - r * ST;
                if (t0 + k < L) {
                    float *dst = ddelta + (((size_t)id.b * gm.G + id.g) * R + r) * L + t0 + k;   // ddelta = d(dtr) here
                    if (atomic_bc) atomicAdd(dst, aR[i]); else *dst = aR[i];
                }
            }
        }
    }
    // per-chunk partial sums of dA (16 per channel), dD, ddelta_bias -> selscan_reduce_partials
    {
        const float dDs = quad_sum(dDacc), dbs = quad_sum(dbacc);
        if (id.act) {
            float *prow = part + srow * PP;
            *reinterpret_cast<float4 *>(prow + 4 * id.s) = make_float4(dAacc[0], dAacc[1], dAacc[2], dAacc[3]);
            float dWs[RMAX] = {0.f, 0.f, 0.f, 0.f};
            if (LR) {
                const float4 dw = *reinterpret_cast<const float4 *>(sW + 8 * tid + 4);
                dWs[0] = quad_sum(dw.x); dWs[1] = quad_sum(dw.y); dWs[2] = quad_sum(dw.z); dWs[3] = quad_sum(dw.w);
            }
            if (id.s == 0) {
                *reinterpret_cast<float4 *>(prow + NS) = make_float4(dDs, dbs, dWs[0], dWs[1]);
                *reinterpret_cast<float4 *>(prow + NS + 4) = make_float4(dWs[2], dWs[3], 0.f, 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward pass 3, group-per-wave form (round 2): ONE wave owns (batch b, group g, chunk) and ALL channels of the
// group.  lane = 4 * cl + s: channel lane cl (16) x state quad s (states 4s .. 4s+3); the lane walks J = ceil(Hc / 16)
// channels cl + 16 j one after the other, so the sums over channels that dB / dC / d(dtr) need are first taken IN the
// lane (the dB / dC products are the accumulating operand of an FMA: free) and cross lanes once per 8-step tile for all
// J channels -- 0.7 instead of 4 cross-lane instructions per (step, state) pair.  Nothing is shared with another wave:
// no barrier, no LDS / global atomics, no memset of dB / dC, one staging of the B / C tile per chunk and group.
//   * time tile: 8 steps (accumulators: 4 states x 8 steps x {dB, dC} = 64 VGPRs); the forward pass saves the state
//     entering every 8-step tile (a first version kept the 16-step saves and re-ran 8 forward steps for the odd tiles:
//     8 % more VALU work, u and the saved state fetched twice).
//   * the channel loop is ROLLED (unrolled, the compiler keeps ~150 VGPRs of every channel body alive: 700 spills at
//     J = 6); what a channel carries from tile to tile sits in LDS: reverse carry q[4] and dA[4] per lane, dD / d(bias) /
//     dWdt[R] per channel (quad-reduced).
//   * per (tile, channel): lane (cl, s) loads / activates delta for its steps, parks delta', delta' u, u, dy rows in LDS
//     (quad-private rows, b128 broadcast reads), runs its 4 states (forward 8 steps keeping a_k, h_{k-1}; reverse 8 steps),
//     quad-reduces the per-step sums and finishes du / d(delta) for steps 2s, 2s + 1.
// 18.9 KB of LDS per wave: 8 waves per CU (2 per SIMD, which is also what its ~234 VGPRs allow).
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ float2 load2(const float *__restrict__ row, int t, int L, bool vec2)
{
    float2 v = make_float2(0.f, 0.f);
    if (vec2) {
        if (t < L) v = *reinterpret_cast<const float2 *>(row + t);
    } else {
        if (t < L) v.x = row[t];
        if (t + 1 < L) v.y = row[t + 1];
    }
    return v;
}
__device__ __forceinline__ void store2(float *__restrict__ row, int t, int L, bool vec2, float2 v)
{
    if (vec2) {
        if (t < L) *reinterpret_cast<float2 *>(row + t) = v;
    } else {
        if (t < L) row[t] = v.x;
        if (t + 1 < L) row[t + 1] = v.y;
    }
}
// NOCHK: the caller knows [t, t+n) lies inside the row (sequence length a multiple of the chunk): no clamp, no select
template <bool VEC, bool NOCHK = false>
__device__ __forceinline__ float4 gload4(const float *__restrict__ base, unsigned t, unsigned end)   // elements [t, t+4) of base, zero at / beyond `end`
{
    if (NOCHK) return ldg_at<float4>(base, t);
    if (VEC) {
        const bool ok = t < end;
        const float4 v = ldg_at<float4>(base, ok ? t : 0u);
        return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < end) v.x = base[t];
    if (t + 1 < end) v.y = base[t + 1];
    if (t + 2 < end) v.z = base[t + 2];
    if (t + 3 < end) v.w = base[t + 3];
    return v;
}
template <bool VEC, bool NOCHK = false>
__device__ __forceinline__ float2 gload2(const float *__restrict__ base, unsigned t, unsigned end)
{
    if (NOCHK) return ldg_at<float2>(base, t);
    if (VEC) {
        const bool ok = t < end;
        const float2 v = ldg_at<float2>(base, ok ? t : 0u);
        return make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
    }
    float2 v = make_float2(0.f, 0.f);
    if (t < end) v.x = base[t];
    if (t + 1 < end) v.y = base[t + 1];
    return v;
}
template <bool VEC, bool NOCHK = false>
__device__ __forceinline__ void gstore4(float *__restrict__ base, unsigned t, unsigned end, float4 v)
{
    if (NOCHK) {
        stg_at<float4>(base, t, v);
    } else if (VEC) {
        if (t < end) stg_at<float4>(base, t, v);
    } else {
        if (t < end) base[t] = v.x;
        if (t + 1 < end) base[t + 1] = v.y;
        if (t + 2 < end) base[t + 2] = v.z;
        if (t + 3 < end) base[t + 3] = v.w;
    }
}
template <bool VEC, bool NOCHK = false>
__device__ __forceinline__ void gstore2(float *__restrict__ base, unsigned t, unsigned end, float2 v)
{
    if (NOCHK) {
        stg_at<float2>(base, t, v);
    } else if (VEC) {
        if (t < end) stg_at<float2>(base, t, v);
    } else {
        if (t < end) base[t] = v.x;
        if (t + 1 < end) base[t + 1] = v.y;
    }
}

// FULL: the group is a whole number of 16-channel slots (Hc % 16 == 0, every MLAgg-UNet shape): no padding lanes, so the
// ~30 selects per (tile, channel) that zero them disappear.  WHOLE (needs VEC and FULL): L is a multiple of the 64-step
// chunk (the headline 256^2 shape: 21760 = 340 * 64), every access is in range: no clamps / range selects at all.
// RT: 0 = delta is a tensor; RMAX = low-rank form with a run-time rank R <= RMAX; 1 .. RMAX-1 = rank known (R == RT: the
// model's R = 3 skips the fourth, padding, rank row everywhere).
template <int RT, bool VEC, bool FULL, bool WHOLE>
__global__ void __launch_bounds__(64, 2)
selscan_bwd_group_kernel(const float *__restrict__ u, const float *__restrict__ delta, const float *__restrict__ Wdt, int R,
                         const float *__restrict__ A, const float *__restrict__ Bm, const float *__restrict__ Cm,
                         const float *__restrict__ Dv, const float *__restrict__ dbias, const float *__restrict__ dout,
                         const float *__restrict__ cstate, const float *__restrict__ csub, const float *__restrict__ cq,
                         float *__restrict__ du, float *__restrict__ ddelta, float *__restrict__ dB, float *__restrict__ dC,
                         float *__restrict__ part, ScanGeom gm, int softplus)
{
    __shared__ float sB[NS * BP], sC[NS * BP];          // [n][16 steps] of the current 16-step tile
    __shared__ float sR[RMAX * ST];                     // LR: rank rows [r][16 steps]
    __shared__ float sD[16 * SP8], sX[16 * SP8];        // per channel lane: delta', delta' u of the current 8 steps
    __shared__ float sY[16 * SP8];                      // dy of the current 8 steps
    __shared__ float4 sQ[JMAX * 64], sA[JMAX * 64];     // per (channel slot, lane): reverse carry q[4], dA[4]
    __shared__ float2 sE[JMAX * 16 * 3];                // per channel: {dD, d(bias)}, {dW0, dW1}, {dW2, dW3}
    // 18.9 KB in all: 8 waves per CU (2 per SIMD, what ~240 VGPRs allow)

    constexpr bool LR = RT > 0;
    constexpr bool RDYN = RT == RMAX;                   // rank rows beyond the run-time R are skipped one by one
    static_assert(!WHOLE || (VEC && FULL), "WHOLE implies VEC and FULL");
    const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int L = gm.L, Hc = gm.Hc, dim = gm.dim;
    const int J = (Hc + 15) >> 4;
    const int tc0 = chunk * TC;
    const size_t crow = (size_t)b * gm.nchunks + chunk;            // row of the per-chunk tensors
    const size_t bg = (size_t)b * gm.G + g;

    {
        const int lane = threadIdx.x, cl = lane >> 2, s = lane & 3;
        for (int j = 0; j < J; ++j) {
            const bool act = cl + 16 * j < Hc;
            float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) q0 = *reinterpret_cast<const float4 *>(cq + (crow * dim + g * Hc + cl + 16 * j) * NS + 4 * s);
            sQ[j * 64 + lane] = q0;
            sA[j * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int i = lane; i < J * 48; i += 64) sE[i] = make_float2(0.f, 0.f);
    }
    __syncthreads();

    // Tiles run in reverse, channels inside a tile forward: ONE flat loop over (tile, channel) so that the HBM streams of
    // the NEXT (tile, channel) -- u, dy, the saved entry state -- are in flight while this one computes (with ~2 waves
    // per SIMD and no prefetch, half of all wave cycles were s_waitcnt: PMC SQ_WAIT_ANY, profiles/round2_a).
    const int m_first = WHOLE ? TC / T8 - 1 : min(TC / T8 - 1, (L - 1 - tc0) / T8);     // last tile that starts inside the sequence
    const int n_it = (m_first + 1) * J;

    const size_t grow = ((size_t)b * dim + g * Hc) * L;            // first row of the group in the (B, D, L) tensors
    const float *ubase = u + grow, *gbase = dout + grow, *dbase = LR ? u : delta + grow;
    float *dubase = du + grow, *ddbase = LR ? du : ddelta + grow;
    struct Stream { float2 uv; float2 gy; float4 hv; float2 dv; };   // per lane: steps 2s, 2s+1 of the tile; dv: raw delta of the plain form
    // branch-free: a load inside a conditional block makes the compiler's vmcnt bookkeeping fall back to vmcnt(0) at the
    // next use of ANY loaded value, which would drain the prefetch at once.  Out-of-range iterations / padding lanes read
    // a valid address and the value is dropped.
    auto fetch = [&](int m, int j, bool live, int ln) -> Stream {
        Stream f;
        const int cl = ln >> 2, s = ln & 3;
        const bool act = WHOLE || (live && (FULL || cl + 16 * j < Hc));    // WHOLE: the (tile, slot) passed in is always a real one
        const int tm = tc0 + m * T8;
        // uniform 64-bit bases + 32-bit lane offsets: the loads take the SGPR-base form (no 64-bit VALU address arithmetic)
        const unsigned c = act ? cl + 16 * j : 0;
        const unsigned roff = c * (unsigned)L;
        // the state entering 8-step tile m: the chunk's entry state (m = 0) or the forward pass's per-tile save
        const float *sbase = m == 0 ? cstate + (crow * dim + g * Hc) * NS : csub + ((crow * NT8 + (m - 1)) * dim + g * Hc) * NS;
        const float4 hv = ldg_at<float4>(sbase, c * NS + 4 * s);
        const float2 gy = gload2<VEC, WHOLE>(gbase, roff + tm + 2 * s, roff + L);
        const float2 uv = gload2<VEC, WHOLE>(ubase, roff + tm + 2 * s, roff + L);
        f.dv = make_float2(0.f, 0.f);
        if (!LR) f.dv = gload2<VEC, WHOLE>(dbase, roff + tm + 2 * s, roff + L);
        f.hv = hv; f.uv = uv; f.gy = gy;        // raw: the consumer zeroes padding lanes (a select here would wait for the load)
        return f;
    };

    float accB[4][T8], accC[4][T8], accR[RMAX][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < T8; ++k) { accB[i][k] = 0.f; accC[i][k] = 0.f; }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) { accR[r][0] = 0.f; accR[r][1] = 0.f; }

    Stream nx = fetch(m_first, 0, true, threadIdx.x);
    int m = m_first, j = 0;                                        // (tile, channel slot) of the iteration: counters, no division
#pragma unroll 1
    for (int it = 0; it < n_it; ++it) {
        // The lane index is made opaque once per iteration: every LDS / global address below is then a value of THIS
        // iteration.  Left to itself the optimiser hoists the ~100 loop-invariant address variants (per state row, ...)
        // out of the loop and the register allocator spills them again.
        int ln = threadIdx.x;
        asm volatile("" : "+v"(ln));
        const int lane = ln, cl = ln >> 2, s = ln & 3;
        const int sub = m >> 1, odd = m & 1, ho = odd * T8;
        const int t16 = tc0 + sub * ST;                            // first step of the 16-step tile
        const bool last_slot = j == J - 1;
        const int mn = last_slot ? max(m - 1, 0) : m, jn = last_slot ? 0 : j + 1;   // the next iteration's (tile, slot)
        const int tm = t16 + ho;                                   // first step of this 8-step tile
        if (j == 0 && (odd || m == m_first)) {
            // first visit of the 16-step tile (tiles run in reverse): stage B / C rows and the rank rows of all 16 steps
            // (requesting them an iteration ahead was measured: no gain, 12 more VGPRs)
            const unsigned no = (unsigned)(lane >> 2) * (unsigned)L;
            const float4 rb = gload4<VEC, WHOLE>(Bm + bg * NS * L, no + t16 + 4 * (lane & 3), no + L);
            const float4 rc = gload4<VEC, WHOLE>(Cm + bg * NS * L, no + t16 + 4 * (lane & 3), no + L);
            float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
            if (LR && lane < 4 * R) rr = gload4<VEC, WHOLE>(delta + bg * R * L, no + t16 + 4 * (lane & 3), no + L);
            wave_lds_fence();
            *reinterpret_cast<float4 *>(sB + (lane >> 2) * BP + 4 * (lane & 3)) = rb;
            *reinterpret_cast<float4 *>(sC + (lane >> 2) * BP + 4 * (lane & 3)) = rc;
            if (LR) stage_rank_rows(sR, R, lane, rr);
            wave_lds_fence();
        }
        const bool act = FULL || cl + 16 * j < Hc;
        Stream cur;
        cur.hv = keep4(act, nx.hv);
        cur.uv = make_float2(act ? nx.uv.x : 0.f, act ? nx.uv.y : 0.f);
        cur.dv = make_float2(act ? nx.dv.x : 0.f, act ? nx.dv.y : 0.f);
        cur.gy = make_float2(act ? nx.gy.x : 0.f, act ? nx.gy.y : 0.f);
        const unsigned c = act ? cl + 16 * j : 0;
        const unsigned roff = c * (unsigned)L;
        // this channel's small operands (L1 / L2 hits) are requested BEFORE the next iteration's streams: vmcnt retires in
        // order, so a wait for them must not sit behind the prefetch
        float4 Av = make_float4(0.f, 0.f, 0.f, 0.f);
        float wv[RMAX] = {0.f, 0.f, 0.f, 0.f};
        float bias = 0.f, Dd = 0.f;
        // unconditional loads from clamped addresses (see fetch); the selects come after the prefetch has been issued
        const float4 av_raw = ldg_at<float4>(A + (size_t)g * Hc * NS, c * NS + 4 * s);
        float w_raw[RMAX] = {0.f, 0.f, 0.f, 0.f};
        if (LR) {
#pragma unroll
            for (int r = 0; r < RT; ++r) w_raw[r] = ldg_at<float>(Wdt + (size_t)g * Hc * R, c * R + (RDYN ? min(r, R - 1) : r));
        }
        const float b_raw = ldg_at<float>((dbias ? dbias : A) + g * Hc, c), d_raw = ldg_at<float>((Dv ? Dv : A) + g * Hc, c);
        __builtin_amdgcn_sched_barrier(0);
        nx = fetch(mn, jn, it + 1 < n_it, ln);                     // in flight until the next iteration's staging
        __builtin_amdgcn_sched_barrier(0);
        Av = keep4(act, av_raw);
#pragma unroll
        for (int r = 0; r < RT; ++r) wv[r] = (act && (!RDYN || r < R)) ? w_raw[r] : 0.f;
        bias = (act && dbias) ? b_raw : 0.f;
        Dd = (act && Dv) ? d_raw : 0.f;
        asm volatile("" : "+v"(Dd));                               // taken here (vmcnt leaves the prefetch alone), not lazily at its use
        float2 uf;                                                 // u of the two steps this lane finishes
        {
            // ---- activation and staging of this (tile, channel) ----
            wave_lds_fence();                                      // the previous channel's readers of the staging rows are done
            {
                // lane (cl, s) activates steps 2s, 2s+1 of this 8-step tile
                float2 raw = cur.dv;
                if (LR) {
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        if (!RDYN || r < R) {                                  // rows beyond the rank are never staged
                            const float2 rv = *reinterpret_cast<const float2 *>(sR + r * ST + ho + 2 * s);
                            raw.x += wv[r] * rv.x; raw.y += wv[r] * rv.y;
                        }
                }
                float2 da;
                da.x = act ? act_delta(raw.x, bias, softplus, WHOLE || tm + 2 * s < L) : 0.f;
                da.y = act ? act_delta(raw.y, bias, softplus, WHOLE || tm + 2 * s + 1 < L) : 0.f;
                *reinterpret_cast<float2 *>(sD + cl * SP8 + 2 * s) = da;
                *reinterpret_cast<float2 *>(sX + cl * SP8 + 2 * s) = make_float2(da.x * cur.uv.x, da.y * cur.uv.y);
                uf = cur.uv;
            }
            *reinterpret_cast<float2 *>(sY + cl * SP8 + 2 * s) = cur.gy;
            wave_lds_fence();
        }

        const float Ar[4] = {Av.x, Av.y, Av.z, Av.w};
        float hent[4] = {cur.hv.x, cur.hv.y, cur.hv.z, cur.hv.w};
        // this tile's 8 steps of delta', delta' u, dy (shared by the lane's 4 states)
        float dk[T8], xk[T8], yk[T8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 a4 = *reinterpret_cast<const float4 *>(sD + cl * SP8 + 4 * q);
            const float4 x4 = *reinterpret_cast<const float4 *>(sX + cl * SP8 + 4 * q);
            const float4 y4 = *reinterpret_cast<const float4 *>(sY + cl * SP8 + 4 * q);
            dk[4 * q] = a4.x; dk[4 * q + 1] = a4.y; dk[4 * q + 2] = a4.z; dk[4 * q + 3] = a4.w;
            xk[4 * q] = x4.x; xk[4 * q + 1] = x4.y; xk[4 * q + 2] = x4.z; xk[4 * q + 3] = x4.w;
            yk[4 * q] = y4.x; yk[4 * q + 1] = y4.y; yk[4 * q + 2] = y4.z; yk[4 * q + 3] = y4.w;
        }
        const float4 q4 = sQ[j * 64 + lane], a4c = sA[j * 64 + lane];
        float qc[4] = {q4.x, q4.y, q4.z, q4.w}, dAc[4] = {a4c.x, a4c.y, a4c.z, a4c.w};
        float sT[T8], sG[T8];
#pragma unroll
        for (int k = 0; k < T8; ++k) { sT[k] = 0.f; sG[k] = 0.f; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *sBn = sB + (4 * s + i) * BP + ho, *sCn = sC + (4 * s + i) * BP + ho;
            const float4 b0 = *reinterpret_cast<const float4 *>(sBn), b1 = *reinterpret_cast<const float4 *>(sBn + 4);
            const float bk[T8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            const float A2 = Ar[i] * LOG2E;
            float ak[T8], hp[T8 + 1];
            float hh = hent[i];
#pragma unroll
            for (int k = 0; k < T8; ++k) {
                hp[k] = hh;
                ak[k] = fast_exp2(dk[k] * A2);
                hh = ak[k] * hh + xk[k] * bk[k];
            }
            hp[T8] = hh;
            const float4 c0 = *reinterpret_cast<const float4 *>(sCn), c1 = *reinterpret_cast<const float4 *>(sCn + 4);
            const float ck[T8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            float qq = qc[i], dAi = dAc[i];
#pragma unroll
            for (int k = T8 - 1; k >= 0; --k) {
                const float gh = qq + yk[k] * ck[k];                        // dL/dh_k
                qq = ak[k] * gh;                                            // carried to step k-1
                const float t1 = qq * hp[k];                                // dL/da_k * a_k
                sT[k] += t1 * Ar[i];
                sG[k] += gh * bk[k];
                dAi += t1 * dk[k];
                accB[i][k] += gh * xk[k];                                   // this channel's dB[k][n] term, summed in the lane
                accC[i][k] += yk[k] * hp[k + 1];                            // dC[k][n] term
            }
            qc[i] = qq;
            dAc[i] = dAi;
        }
        __builtin_amdgcn_sched_barrier(0);
        sQ[j * 64 + lane] = make_float4(qc[0], qc[1], qc[2], qc[3]);
        sA[j * 64 + lane] = make_float4(dAc[0], dAc[1], dAc[2], dAc[3]);
        // ---- finish d(delta') and du: quad sums, lane s keeps steps 2s, 2s+1 ----
        {
            // reduce-scatter over the quad: 8 per-step sums in, lane s leaves with steps 2s, 2s+1 (18 instead of the 24
            // instructions of 8 all-reduces + selects)
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
            const float2 u2 = uf;
            const float2 g2 = *reinterpret_cast<const float2 *>(sY + cl * SP8 + 2 * s);
            const float dl2[2] = {d2.x, d2.y}, uu2[2] = {u2.x, u2.y}, gg2[2] = {g2.x, g2.y};
            float odd2[2], odu2[2];
            float ev[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                  // this lane's part of dD, d(bias), dW[0..3]
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // d softplus(x)/dx = sigmoid(x) = 1 - exp(-softplus(x)); 1 when softplus is off
                const float sp = softplus ? (1.f - __expf(-dl2[e])) : 1.f;
                const bool inr = WHOLE || ((tm + 2 * s + e) < L && act);
                odd2[e] = inr ? (Ts[e] + uu2[e] * Gs[e]) * sp : 0.f;        // (sT + u * sG) * softplus'
                odu2[e] = dl2[e] * Gs[e] + Dd * gg2[e];                     // delta' * sG + D * dy
                ev[1] += odd2[e];
                ev[0] += gg2[e] * uu2[e];
            }
            if (act) {
                gstore2<VEC, WHOLE>(dubase, roff + tm + 2 * s, roff + L, make_float2(odu2[0], odu2[1]));
                if (!LR) gstore2<VEC, WHOLE>(ddbase, roff + tm + 2 * s, roff + L, make_float2(odd2[0], odd2[1]));
            }
            if (LR) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    if (!RDYN || r < R) {
                        const float w = wv[r];
                        const float2 rv = *reinterpret_cast<const float2 *>(sR + r * ST + ho + 2 * s);
                        ev[2 + r] = odd2[0] * rv.x + odd2[1] * rv.y;        // dWdt[d][r] += sum_t d(raw delta) dtr[r][t]
                        accR[r][0] += odd2[0] * w;                          // d(dtr)[r][t] += d(raw delta) Wdt[d][r]
                        accR[r][1] += odd2[1] * w;
                    }
            }
#pragma unroll
            for (int x = 0; x < 2 + RT; ++x) ev[x] = quad_sum(ev[x]);
            if (s == 0) {
                float2 *e = sE + (j * 16 + cl) * 3;
                float2 e0 = e[0], e1 = e[1], e2 = e[2];
                e0.x += ev[0]; e0.y += ev[1]; e1.x += ev[2]; e1.y += ev[3]; e2.x += ev[4]; e2.y += ev[5];
                e[0] = e0; e[1] = e1; e[2] = e2;
            }
        }

        if (last_slot) {
            // ---- sums over the 16 channel lanes (lane bits 2..5), once per tile for all J channels ----
            // value index = kind * 32 + i * 8 + k  (kind 0: dB, 1: dC).  permlane32 swap: lane bit 5 keeps its kind;
            // permlane16 swap: lane bit 4 keeps states i = 2 * bit4 + {0, 1}; the two low channel bits are an orbit of
            // row_ror:4, all-reduced.  Lane (bit5 = p, bit4 = q, cl & 3 = r) then stores 4 steps of one state row.
            float v[32];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < T8; ++k) {
                    float a = accB[i][k], c = accC[i][k];
                    swap32(a, c);
                    v[i * 8 + k] = a + c;
                    accB[i][k] = 0.f; accC[i][k] = 0.f;
                }
#pragma unroll
            for (int x = 0; x < 16; ++x) { swap16(v[x], v[x + 16]); v[x] += v[x + 16]; }
#pragma unroll
            for (int x = 0; x < 16; ++x) { v[x] += row_ror4(v[x]); v[x] += row_ror8(v[x]); }
            const int p = lane >> 5, q = (lane >> 4) & 1, r4 = cl & 3;
            float4 o;
            o.x = r4 == 0 ? v[0] : (r4 == 1 ? v[4] : (r4 == 2 ? v[8] : v[12]));
            o.y = r4 == 0 ? v[1] : (r4 == 1 ? v[5] : (r4 == 2 ? v[9] : v[13]));
            o.z = r4 == 0 ? v[2] : (r4 == 1 ? v[6] : (r4 == 2 ? v[10] : v[14]));
            o.w = r4 == 0 ? v[3] : (r4 == 1 ? v[7] : (r4 == 2 ? v[11] : v[15]));
            const int n = 4 * s + 2 * q + (r4 >> 1);
            const unsigned no = (unsigned)n * (unsigned)L;
            gstore4<VEC, WHOLE>((p ? dC : dB) + bg * NS * L, no + tm + 4 * (r4 & 1), no + L, o);
            if (LR) {
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
                const int r = 2 * (lane >> 5) + ((lane >> 4) & 1); // the rank row this lane ends up with
                if (r < R && (cl & 3) == 0)
                    gstore2<VEC, WHOLE>(ddelta + bg * R * L, r * (unsigned)L + tm + 2 * s, (r + 1) * (unsigned)L, make_float2(w[0], w[1]));   // ddelta = d(dtr) here
            }
        }
        m = mn; j = jn;
    }

    // per-chunk partial sums of dA (16 per channel), dD, ddelta_bias, dWdt -> selscan_reduce_partials
    __syncthreads();
    {
        const int lane = threadIdx.x, cl = lane >> 2, s = lane & 3;
        for (int j = 0; j < J; ++j) {
            if (cl + 16 * j < Hc) {
                float *prow = part + (crow * dim + g * Hc + cl + 16 * j) * PP;
                *reinterpret_cast<float4 *>(prow + 4 * s) = sA[j * 64 + lane];
                if (s == 0) {
                    const float2 *e = sE + (j * 16 + cl) * 3;
                    *reinterpret_cast<float4 *>(prow + NS) = make_float4(e[0].x, e[0].y, e[1].x, e[1].y);
                    *reinterpret_cast<float4 *>(prow + NS + 4) = make_float4(e[2].x, e[2].y, 0.f, 0.f);
                }
            }
        }
    }
}


inline bool getenv_flag(const char *name)
{
    const char *v = getenv(name);
    return v && v[0] == '1';
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
    return (size_t)batch * nchunks * dim * (N + 1 + NT8 * N);     // chunk entry states, chunk delta sums, 8-step tile entry states
}

extern "C" size_t mlagg_selscan_bwd_workspace_floats(int batch, int dim, int L, int N)
{
    const size_t nchunks = (L + TC - 1) / TC;
    return (size_t)batch * nchunks * dim * (N + PP);
}

namespace {

template <bool LR>
int scan_forward(const float *u, const float *delta, const float *Wdt, int R, const float *A, const float *B,
                 const float *C, const float *D, const float *delta_bias, float *out, float *chunk_state, int batch,
                 int dim, int L, int N, int G, int delta_softplus, void *stream)
{
    ScanGeom gm;
    if (int rc = make_geom(gm, batch, dim, L, N, G, SCAN_CB)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *cstate = chunk_state;
    float *cdsum = chunk_state + (size_t)batch * gm.nchunks * dim * NS;
    float *csub = cdsum + (size_t)batch * gm.nchunks * dim;
    const dim3 grid(gm.nchunks, G * gm.nblk, batch), block(block_threads(gm));
    const size_t lds = (size_t)(2 * gm.CB * UP + 2 * ST * BP + (LR ? 2 * RMAX * ST : 0)) * sizeof(float);
    // every lane active, every access in range (see the kernel): the model's shapes at 256^2
    const bool fast = (L & 3) == 0 && gm.Hc % gm.CB == 0 && block.x == 128 && 4 * gm.CB == 128 && (!LR || R >= 1);
    { MLAGG_TIMED(K_SELSCAN_FWD_LOCAL, st);
      if (fast) hipLaunchKernelGGL((selscan_fwd_kernel<false, LR, true>), grid, block, lds, st, u, delta, Wdt, R, A, B, C, D,
                                   delta_bias, out, cstate, cdsum, csub, gm, delta_softplus);
      else hipLaunchKernelGGL((selscan_fwd_kernel<false, LR, false>), grid, block, lds, st, u, delta, Wdt, R, A, B, C, D,
                              delta_bias, out, cstate, cdsum, csub, gm, delta_softplus); }
    { MLAGG_TIMED(K_SELSCAN_PREFIX, st); hipLaunchKernelGGL(selscan_chunk_prefix, dim3((dim * NS + 255) / 256, batch), dim3(256), 0, st, A, cstate,
                       cdsum, gm, 0); }
    { MLAGG_TIMED(K_SELSCAN_FWD_FINAL, st);
      if (fast) hipLaunchKernelGGL((selscan_fwd_kernel<true, LR, true>), grid, block, lds, st, u, delta, Wdt, R, A, B, C, D,
                                   delta_bias, out, cstate, cdsum, csub, gm, delta_softplus);
      else hipLaunchKernelGGL((selscan_fwd_kernel<true, LR, false>), grid, block, lds, st, u, delta, Wdt, R, A, B, C, D,
                              delta_bias, out, cstate, cdsum, csub, gm, delta_softplus); }
    return (int)hipGetLastError();
}

// ddelta: (B, dim, L) gradient of delta, or -- LR -- (B, G, R, L) gradient of the rank-R rows
template <bool LR>
int scan_backward(const float *u, const float *delta, const float *Wdt, int R, const float *A, const float *B,
                  const float *C, const float *D, const float *delta_bias, const float *dout, const float *chunk_state,
                  float *du, float *ddelta, float *dWdt, float *dA, float *dB, float *dC, float *dD, float *ddelta_bias,
                  float *workspace, int batch, int dim, int L, int N, int G, int delta_softplus, void *stream)
{
    ScanGeom gm;
    if (int rc = make_geom(gm, batch, dim, L, N, G, SCAN_CB)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *cstate = chunk_state;
    const float *cdsum = chunk_state + (size_t)batch * gm.nchunks * dim * NS;
    const float *csub = cdsum + (size_t)batch * gm.nchunks * dim;
    float *cq = workspace;
    float *part = workspace + (size_t)batch * gm.nchunks * dim * NS;
    const dim3 grid(gm.nchunks, G * gm.nblk, batch), block(block_threads(gm));
    // The main kernel runs at 3 waves per SIMD.  Workgroups of 32 channels (2 waves) let a CU hold independent
    // workgroups instead of one 6-wave workgroup spread 2/2/1/1; the price is a 3-way float-atomic accumulation
    // of dB / dC (and d(dtr)) across the workgroups of a group.
    ScanGeom gb;
    if (int rc = make_geom(gb, batch, dim, L, N, G, BWD_CB)) return rc;
    const dim3 gridb(gb.nchunks, G * gb.nblk, batch), blockb(block_threads(gb));
    // groups of up to 96 channels (every MLAgg-UNet shape: 96): the group-per-wave kernel; wider groups: the channel-block
    // kernel with float-atomic accumulation of dB / dC across the blocks of a group
    const bool group_form = gm.Hc <= 96 && !getenv_flag("MLAGG_SELSCAN_BWD_BLOCKED");
    const int atomic_bc = !group_form && gb.nblk > 1;
    if (atomic_bc) {
        const size_t bytes = (size_t)batch * G * NS * L * sizeof(float);
        (void)hipMemsetAsync(dB, 0, bytes, st);
        (void)hipMemsetAsync(dC, 0, bytes, st);
        if (LR) (void)hipMemsetAsync(ddelta, 0, (size_t)batch * G * R * L * sizeof(float), st);
    }
    const size_t lds1 = (size_t)(2 * gm.CB * UP + ST * BP + (LR ? 2 * RMAX * ST : 0)) * sizeof(float);
    const bool fast = (L & 3) == 0 && gm.Hc % gm.CB == 0 && block.x == 128 && 4 * gm.CB == 128 && (!LR || R >= 1);
    { MLAGG_TIMED(K_SELSCAN_BWD_LOCAL, st);
      if (fast) hipLaunchKernelGGL((selscan_bwd_local_kernel<LR, true>), grid, block, lds1, st, delta, Wdt, R, A, C, delta_bias,
                                   dout, cq, gm, delta_softplus);
      else hipLaunchKernelGGL((selscan_bwd_local_kernel<LR, false>), grid, block, lds1, st, delta, Wdt, R, A, C, delta_bias,
                              dout, cq, gm, delta_softplus); }
    { MLAGG_TIMED(K_SELSCAN_PREFIX, st); hipLaunchKernelGGL(selscan_chunk_prefix, dim3((dim * NS + 255) / 256, batch), dim3(256), 0, st, A, cq, cdsum,
                       gm, 1); }
    if (group_form) {
        // one wave per (batch, group, chunk) walks all channels of the group: dB / dC / d(dtr) leave with plain stores
        const dim3 gridg(gm.nchunks, G, batch);
        MLAGG_TIMED(K_SELSCAN_BWD, st);
#define MLAGG_GROUP_LAUNCH(RT, VEC, FULL, WHOLE) hipLaunchKernelGGL((selscan_bwd_group_kernel<RT, VEC, FULL, WHOLE>), gridg, dim3(64), 0, st, u, delta, \
            Wdt, R, A, B, C, D, delta_bias, dout, cstate, csub, cq, du, ddelta, dB, dC, part, gm, delta_softplus)
        constexpr int RGEN = LR ? RMAX : 0;
        const bool vecL = (L & 3) == 0, full = (gm.Hc & 15) == 0, whole = L % TC == 0;
        constexpr int RFAST = LR ? 3 : 0;
        if ((!LR || R == 3) && vecL && full) {                    // the model's shapes: rank 3, 96-channel groups
            if (whole) MLAGG_GROUP_LAUNCH(RFAST, true, true, true);
            else MLAGG_GROUP_LAUNCH(RFAST, true, true, false);
        }
        else if (vecL && full) MLAGG_GROUP_LAUNCH(RGEN, true, true, false);
        else if (vecL) MLAGG_GROUP_LAUNCH(RGEN, true, false, false);
        else MLAGG_GROUP_LAUNCH(RGEN, false, false, false);
#undef MLAGG_GROUP_LAUNCH
    } else {
        const size_t lds3 = (size_t)(3 * gb.CB * UP + 2 * ST * BP + 2 * ST * NS + (LR ? 2 * RMAX * ST + 8 * 128 : 0) + UP + gb.CB * UP) * sizeof(float);
        MLAGG_TIMED(K_SELSCAN_BWD, st);
        hipLaunchKernelGGL(selscan_bwd_kernel<LR>, gridb, blockb, lds3, st, u, delta, Wdt, R, A, B, C, D, delta_bias,
                           dout, cstate, csub, cq, du, ddelta, dB, dC, part, gb, delta_softplus, atomic_bc);
    }
    { MLAGG_TIMED(K_SELSCAN_REDUCE, st); hipLaunchKernelGGL(selscan_reduce_partials, dim3((dim * PP + 63) / 64), dim3(1024), 0, st, part, dA, dD, ddelta_bias,
                       LR ? dWdt : nullptr, R, gm); }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mlagg_selscan_fwd(const float *u, const float *delta, const float *A, const float *B,
                                 const float *C, const float *D, const float *delta_bias, float *out,
                                 float *chunk_state, int batch, int dim, int L, int N, int G,
                                 int delta_softplus, void *stream)
{
    if (!u || !delta || !A || !B || !C || !out || !chunk_state) return MLAGG_E_NULLPTR;
    return scan_forward<false>(u, delta, nullptr, 0, A, B, C, D, delta_bias, out, chunk_state, batch, dim, L, N, G,
                               delta_softplus, stream);
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
    return scan_backward<false>(u, delta, nullptr, 0, A, B, C, D, delta_bias, dout, chunk_state, du, ddelta, nullptr, dA,
                                dB, dC, dD, ddelta_bias, workspace, batch, dim, L, N, G, delta_softplus, stream);
}

extern "C" int mlagg_selscan_lowrank_fwd(const float *u, const float *dtr, const float *Wdt, int R, const float *A,
                                         const float *B, const float *C, const float *D, const float *delta_bias,
                                         float *out, float *chunk_state, int batch, int dim, int L, int N, int G,
                                         int delta_softplus, void *stream)
{
    if (!u || !dtr || !Wdt || !A || !B || !C || !out || !chunk_state) return MLAGG_E_NULLPTR;
    if (R < 1 || R > RMAX) return MLAGG_E_UNSUPPORTED;
    return scan_forward<true>(u, dtr, Wdt, R, A, B, C, D, delta_bias, out, chunk_state, batch, dim, L, N, G,
                              delta_softplus, stream);
}

extern "C" int mlagg_selscan_lowrank_bwd(const float *u, const float *dtr, const float *Wdt, int R, const float *A,
                                         const float *B, const float *C, const float *D, const float *delta_bias,
                                         const float *dout, const float *chunk_state, float *du, float *ddtr,
                                         float *dWdt, float *dA, float *dB, float *dC, float *dD, float *ddelta_bias,
                                         float *workspace, int batch, int dim, int L, int N, int G, int delta_softplus,
                                         void *stream)
{
    if (!u || !dtr || !Wdt || !A || !B || !C || !dout || !chunk_state || !du || !ddtr || !dWdt || !dA || !dB || !dC ||
        !workspace)
        return MLAGG_E_NULLPTR;
    if (R < 1 || R > RMAX) return MLAGG_E_UNSUPPORTED;
    return scan_backward<true>(u, dtr, Wdt, R, A, B, C, D, delta_bias, dout, chunk_state, du, ddtr, dWdt, dA, dB, dC, dD,
                               ddelta_bias, workspace, batch, dim, L, N, G, delta_softplus, stream);
}
