// K9 -- statistics and gradient of the Dice + cross-entropy deep-supervision loss, one pass each over the logits.
//
// Replaces, per deep-supervision level, the eager chain of DC_and_CE_loss (reference loss/compound_losses.py:31-57):
// softmax, one-hot scatter, three masked products + spatial sums (MemoryEfficientSoftDiceLoss, loss/dice.py:73-117),
// log_softmax + nll (RobustCrossEntropyLoss, loss/robust_ce_loss.py:12-16) and the backward of all of them --
// 310 kernel launches and 2.5 ms per train step at config 2 (5 levels, 10 x 14 x 256^2 logits at level 0).
//
//   stats : per (sample b, class c)   I[b][c] = sum_p softmax(z)_c [y_p == c]      ("intersect")
//                                      P[b][c] = sum_p softmax(z)_c                 ("sum_pred")
//                                      G[b][c] = sum_p [y_p == c]                   ("sum_gt")
//           and the scalar            CE      = sum_{b,p} (logsumexp(z) - z_y)
//   grad  : dz_k = p_k (g_k - sum_c p_c g_c) + gCE (p_k - [y == k]),   g_c = gI[b][c] [y == c] + gP[b][c]
// The few-element algebra between them (dice ratio, means, level weights, the data-parallel all-reduce of the
// batch-dice statistics) stays in torch on (levels, classes)-sized tensors: trainer.deep_supervision_loss.
//
// Layout: logits (B, C, HW) fp32 NCHW planes, target (B, HW) float labels (as the nnU-Net pipeline delivers them).
// A thread owns PPT consecutive-by-256 pixels of one sample; every class plane is read as 1 KiB coalesced runs; the
// softmax is recomputed in the gradient pass (one more read of z instead of a write + read of the probabilities).
// HBM-bound: 4 (C + 1) bytes per pixel forward, 4 (2C + 1) backward.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int MAXC = 16;
constexpr int PPT = 4;            // pixels per thread
constexpr int TPB = 256;

struct LossGeom {
    int B, C;
    long HW;
    int ignore;                   // label value whose pixels take no part in the loss (DC_and_CE_loss(ignore_label=...)); -1: none
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Partial sums leave the workgroup as ONE row [I(C) | P(C) | G(C) | ce] of `part` and are added up in a fixed order by
// dice_ce_reduce_kernel: the first version accumulated them with float atomics (LDS, then global), which made the LOSS VALUE -- and
// through the dice gradient every gradient of the step -- differ in the last bits between two runs on identical inputs
// (tools/find_nondeterminism.py; profiles/round3_nondeterminism.log).
__global__ void __launch_bounds__(TPB)
dice_ce_stats_kernel(const float *__restrict__ logits, const float *__restrict__ target, float *__restrict__ part, LossGeom g)
{
    __shared__ float red[TPB / 64][3 * MAXC + 1];
    const int b = blockIdx.y, C = g.C;
    const float *zb = logits + (size_t)b * C * g.HW;
    const float *tb = target + (size_t)b * g.HW;
    float aI[MAXC], aP[MAXC], aG[MAXC], ace = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) aI[c] = aP[c] = aG[c] = 0.f;

    const long p0 = (long)blockIdx.x * (TPB * PPT) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const long p = p0 + (long)j * TPB;
        if (p >= g.HW) break;
        float z[MAXC];
        float m = -3.0e38f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            z[c] = c < C ? zb[(size_t)c * g.HW + p] : -3.0e38f;
            m = fmaxf(m, z[c]);
        }
        const int y = (int)tb[p];
        float s = 0.f, zy = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const float sh = z[c] - m;                     // shifted logit; zy keeps the one of the label
            zy = c == y ? sh : zy;
            z[c] = c < C ? expf(sh) : 0.f;                 // z[] now holds exp(z - m)
            s += z[c];
        }
        const float inv = 1.f / s;
        const bool valid = y != g.ignore;                  // the loss mask of the reference (compound_losses.py:38-46)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const float pc = z[c] * inv;
            const bool hit = valid && c == y;
            aP[c] += valid ? pc : 0.f;
            aI[c] += hit ? pc : 0.f;
            aG[c] += hit ? 1.f : 0.f;
        }
        ace += valid ? logf(s) - zy : 0.f;                 // -log softmax(z)_y (CrossEntropyLoss(ignore_index))
    }
    // block reduction: wave butterflies, one LDS row per wave, the four rows added in a fixed order, one partial row per workgroup
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < C) {
            const float vI = wave_sum(aI[c]), vP = wave_sum(aP[c]), vG = wave_sum(aG[c]);
            if ((threadIdx.x & 63) == 0) { red[wv][c] = vI; red[wv][MAXC + c] = vP; red[wv][2 * MAXC + c] = vG; }
        }
    }
    ace = wave_sum(ace);
    if ((threadIdx.x & 63) == 0) red[wv][3 * MAXC] = ace;
    __syncthreads();
    float *row = part + ((size_t)b * gridDim.x + blockIdx.x) * (3 * C + 1);
    if (threadIdx.x < 3 * C) {
        const int which = threadIdx.x / C, c = threadIdx.x - which * C, i = which * MAXC + c;
        row[threadIdx.x] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
    if (threadIdx.x == 3 * C) row[3 * C] = (red[0][3 * MAXC] + red[1][3 * MAXC]) + (red[2][3 * MAXC] + red[3][3 * MAXC]);
}

// stats_ip (B, 2, C), stats_g (B, C) = column sums of the nblk partial rows of every sample; ce_sum = sum of all ce partials.
// One workgroup, fixed assignment and order: deterministic.
__global__ void __launch_bounds__(256)
dice_ce_reduce_kernel(const float *__restrict__ part, int nblk, float *__restrict__ stats_ip, float *__restrict__ stats_g,
                      float *__restrict__ ce_sum, int B, int C)
{
    __shared__ float red[256];
    const int W = 3 * C + 1;
    for (int i = threadIdx.x; i < B * 3 * C; i += 256) {
        const int b = i / (3 * C), v = i - b * 3 * C;
        const float *p = part + (size_t)b * nblk * W + v;
        float s0 = 0.f, s1 = 0.f;
        int r = 0;
        for (; r + 1 < nblk; r += 2) { s0 += p[(size_t)r * W]; s1 += p[(size_t)(r + 1) * W]; }
        if (r < nblk) s0 += p[(size_t)r * W];
        const float s = s0 + s1;
        const int which = v / C, c = v - which * C;
        if (which < 2) stats_ip[((size_t)b * 2 + which) * C + c] = s;
        else stats_g[(size_t)b * C + c] = s;
    }
    float a = 0.f;
    for (long i = threadIdx.x; i < (long)B * nblk; i += 256) a += part[(size_t)i * W + 3 * C];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *ce_sum = red[0];
}

__global__ void __launch_bounds__(TPB)
dice_ce_grad_kernel(const float *__restrict__ logits, const float *__restrict__ target, const float *__restrict__ g_ip,
                    const float *__restrict__ g_ce, float *__restrict__ dlogits, LossGeom g)
{
    const int b = blockIdx.y, C = g.C;
    const float *zb = logits + (size_t)b * C * g.HW;
    const float *tb = target + (size_t)b * g.HW;
    float *db = dlogits + (size_t)b * C * g.HW;
    float gI[MAXC], gP[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        gI[c] = c < C ? g_ip[((size_t)b * 2 + 0) * C + c] : 0.f;
        gP[c] = c < C ? g_ip[((size_t)b * 2 + 1) * C + c] : 0.f;
    }
    const float gce = g_ce[0];
    const long p0 = (long)blockIdx.x * (TPB * PPT) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const long p = p0 + (long)j * TPB;
        if (p >= g.HW) break;
        float z[MAXC];
        float m = -3.0e38f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            z[c] = c < C ? zb[(size_t)c * g.HW + p] : -3.0e38f;
            m = fmaxf(m, z[c]);
        }
        const int y = (int)tb[p];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            z[c] = c < C ? expf(z[c] - m) : 0.f;
            s += z[c];
        }
        const float inv = 1.f / s;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            z[c] *= inv;                                               // p_c
            dot += z[c] * (gP[c] + (c == y ? gI[c] : 0.f));
        }
        const bool valid = y != g.ignore;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < C) {
                const float hit = c == y ? 1.f : 0.f;
                const float dz = z[c] * (gP[c] + hit * gI[c] - dot) + gce * (z[c] - hit);
                db[(size_t)c * g.HW + p] = valid ? dz : 0.f;
            }
        }
    }
}

int check(int B, int C, long HW)
{
    if (B <= 0 || B > 65535 || C < 2 || C > MAXC || HW <= 0) return MLAGG_E_UNSUPPORTED;
    return 0;
}

}  // namespace

extern "C" int mlagg_dice_ce_max_classes(void) { return MAXC; }

extern "C" size_t mlagg_dice_ce_stats_workspace_floats(int B, int C, long HW)
{
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    return (size_t)B * ((HW + TPB * PPT - 1) / (TPB * PPT)) * (3 * C + 1);
}

extern "C" int mlagg_dice_ce_stats(const float *logits, const float *target, float *stats_ip, float *stats_g, float *ce_sum,
                                   float *workspace, int B, int C, long HW, int ignore_label, void *stream)
{
    if (!logits || !target || !stats_ip || !stats_g || !ce_sum || !workspace) return MLAGG_E_NULLPTR;
    if (int rc = check(B, C, HW)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    LossGeom g{B, C, HW, ignore_label};
    const unsigned nblk = (unsigned)((HW + TPB * PPT - 1) / (TPB * PPT));
    MLAGG_TIMED(K_LOSS_STATS, st);
    hipLaunchKernelGGL(dice_ce_stats_kernel, dim3(nblk, B), dim3(TPB), 0, st, logits, target, workspace, g);
    hipLaunchKernelGGL(dice_ce_reduce_kernel, dim3(1), dim3(256), 0, st, workspace, (int)nblk, stats_ip, stats_g, ce_sum, B, C);
    return (int)hipGetLastError();
}

extern "C" int mlagg_dice_ce_grad(const float *logits, const float *target, const float *g_ip, const float *g_ce,
                                  float *dlogits, int B, int C, long HW, int ignore_label, void *stream)
{
    if (!logits || !target || !g_ip || !g_ce || !dlogits) return MLAGG_E_NULLPTR;
    if (int rc = check(B, C, HW)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    LossGeom g{B, C, HW, ignore_label};
    MLAGG_TIMED(K_LOSS_GRAD, st);
    hipLaunchKernelGGL(dice_ce_grad_kernel, dim3((unsigned)((HW + TPB * PPT - 1) / (TPB * PPT)), B), dim3(TPB), 0, st, logits,
                       target, g_ip, g_ce, dlogits, g);
    return (int)hipGetLastError();
}
