// K11 -- gradient-norm clipping + AdamW for all parameters of the network in two launches.
//
// Replaces `torch.nn.utils.clip_grad_norm_(parameters, 12)` + `AdamW.step()` of the train step (reference
// nnUNetTrainer.py:855-857 with the optimizer of nnUNetTrainer_MLAgg_2D_dt_MS.py:137-147): ATen runs a multi-tensor norm,
// a multi-tensor scale of every gradient and 15 multi-tensor AdamW launches at ~1.2 TB/s (0.9 ms per step for the 27 M
// parameters).  Here
//   1. sumsq : every work item (a <= 64 Ki-element chunk of one gradient) adds its sum of squares into one fp32 scalar
//              (deterministic order is not needed for a norm that only feeds a clamp; the double accumulator removes
//              the sensitivity of fp32 atomics to arrival order beyond 1e-7 relative);
//   2. update: the clip coefficient min(1, max_norm / (norm + 1e-6)) is formed ON THE DEVICE from that scalar -- no host
//              synchronisation -- and applied to the gradient as it is read; decoupled weight decay, moments, bias-corrected
//              step exactly as torch.optim.AdamW (amsgrad off).  The gradients themselves are left unscaled in memory
//              (torch rewrites them: 216 MB of extra traffic).
// Tensors are addressed through a device table of (param, grad, exp_avg, exp_avg_sq, numel) rows and a work list of
// (tensor, chunk) pairs, both built by the host wrapper.  HBM-bound: 28 bytes per parameter for the update, 4 for the norm.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int CHUNK = 65536;          // elements per work item
constexpr int OPT_TPB = 256;

struct TensorRow {                    // mirrors the int64 x 5 rows of the host table
    float *p;
    const float *g;
    float *m;
    float *v;
    long long n;
};

__global__ void __launch_bounds__(OPT_TPB)
adamw_sumsq_kernel(const TensorRow *__restrict__ table, const int2 *__restrict__ work, double *__restrict__ sumsq)
{
    __shared__ float red[OPT_TPB / 64];
    const int2 w = work[blockIdx.x];
    const TensorRow t = table[w.x];
    const long long lo = (long long)w.y * CHUNK, hi = min(lo + CHUNK, t.n);
    const float *g = t.g + lo;
    const long long n = hi - lo;
    float s = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const long long n4 = n >> 2;
        for (long long i = threadIdx.x; i < n4; i += OPT_TPB) {
            const float4 a = reinterpret_cast<const float4 *>(g)[i];
            s += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        }
        for (long long i = 4 * n4 + threadIdx.x; i < n; i += OPT_TPB) s += g[i] * g[i];
    } else {
        for (long long i = threadIdx.x; i < n; i += OPT_TPB) s += g[i] * g[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    // one partial per work item, added up in a fixed order by adamw_sumsq_reduce_kernel (a double atomic add made the clip
    // coefficient -- hence every parameter -- depend on the arrival order in the last bit)
    if (threadIdx.x == 0) sumsq[1 + blockIdx.x] = (double)((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void __launch_bounds__(OPT_TPB)
adamw_sumsq_reduce_kernel(double *__restrict__ sumsq, int n_work, int *__restrict__ step_dev)
{
    // device-resident step counter (the hipGraph form of the step): advanced here, read by the update kernel behind this one
    if (step_dev && threadIdx.x == 0) *step_dev += 1;
    __shared__ double red[OPT_TPB];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_work; i += OPT_TPB) s += sumsq[1 + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = OPT_TPB / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sumsq[0] = red[0];
}

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, max_norm, bias_c1, bias_c2_sqrt;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamArgs &a, float coef)
{
    g *= coef;
    p *= 1.f - a.lr * a.weight_decay;
    m = a.beta1 * m + (1.f - a.beta1) * g;            // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bias_c2_sqrt + a.eps;
    p -= (a.lr / a.bias_c1) * (m / denom);
}

__global__ void __launch_bounds__(OPT_TPB)
adamw_update_kernel(const TensorRow *__restrict__ table, const int2 *__restrict__ work, const double *__restrict__ sumsq,
                    AdamArgs a, const float *__restrict__ lr_dev, const int *__restrict__ step_dev)
{
    if (lr_dev) {
        // learning rate and step live on the device (a captured launch cannot carry them as arguments): the same double-precision
        // bias corrections the host form passes in
        const double step = (double)*step_dev;
        a.lr = *lr_dev;
        a.bias_c1 = (float)(1.0 - pow((double)a.beta1, step));
        a.bias_c2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, step));
    }
    const int2 w = work[blockIdx.x];
    const TensorRow t = table[w.x];
    const long long lo = (long long)w.y * CHUNK, hi = min(lo + CHUNK, t.n);
    const long long n = hi - lo;
    float coef = 1.f;
    if (a.max_norm > 0.f) {
        const float norm = (float)sqrt(*sumsq);
        // a non-finite norm poisons EVERY parameter, as clip_grad_norm_'s NaN coefficient does (fminf(NaN, 1) = 1 would
        // keep stepping the finite gradients and hide the failure)
        coef = (norm == norm && norm < __builtin_huge_valf()) ? fminf(a.max_norm / (norm + 1e-6f), 1.f) : __builtin_nanf("");
    }
    float *p = t.p + lo, *m = t.m + lo, *v = t.v + lo;
    const float *g = t.g + lo;
    const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    const long long n4 = al ? n >> 2 : 0;
    for (long long i = threadIdx.x; i < n4; i += OPT_TPB) {
        float4 p4 = reinterpret_cast<float4 *>(p)[i], m4 = reinterpret_cast<float4 *>(m)[i], v4 = reinterpret_cast<float4 *>(v)[i];
        const float4 g4 = reinterpret_cast<const float4 *>(g)[i];
        adam_one(p4.x, g4.x, m4.x, v4.x, a, coef);
        adam_one(p4.y, g4.y, m4.y, v4.y, a, coef);
        adam_one(p4.z, g4.z, m4.z, v4.z, a, coef);
        adam_one(p4.w, g4.w, m4.w, v4.w, a, coef);
        reinterpret_cast<float4 *>(p)[i] = p4;
        reinterpret_cast<float4 *>(m)[i] = m4;
        reinterpret_cast<float4 *>(v)[i] = v4;
    }
    for (long long i = 4 * n4 + threadIdx.x; i < n; i += OPT_TPB) adam_one(p[i], g[i], m[i], v[i], a, coef);
}

}  // namespace

extern "C" int mlagg_adamw_chunk_elements(void) { return CHUNK; }

extern "C" int mlagg_adamw_clip_step(const void *tensor_table, const void *work_list, int n_work, double *sumsq, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, float max_norm, int step,
                                     void *stream)
{
    if (!tensor_table || !work_list || !sumsq) return MLAGG_E_NULLPTR;
    if (n_work <= 0 || step < 1) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const TensorRow *table = static_cast<const TensorRow *>(tensor_table);
    const int2 *work = static_cast<const int2 *>(work_list);
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, max_norm, 0.f, 0.f};
    a.bias_c1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bias_c2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    MLAGG_TIMED(K_ADAMW, st);
    if (max_norm > 0.f) {
        hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(n_work), dim3(OPT_TPB), 0, st, table, work, sumsq);
        hipLaunchKernelGGL(adamw_sumsq_reduce_kernel, dim3(1), dim3(OPT_TPB), 0, st, sumsq, n_work, (int *)nullptr);
    }
    hipLaunchKernelGGL(adamw_update_kernel, dim3(n_work), dim3(OPT_TPB), 0, st, table, work, sumsq, a, (const float *)nullptr,
                       (const int *)nullptr);
    return (int)hipGetLastError();
}

// The same step with the learning rate and the step counter RESIDENT ON THE DEVICE (lr_dev: one float the schedule writes between
// steps; step_dev: one int32, advanced by this call before it is used), so that the call can be captured into a hipGraph and
// replayed: nothing that changes from step to step is a launch argument.
extern "C" int mlagg_adamw_clip_step_dev(const void *tensor_table, const void *work_list, int n_work, double *sumsq,
                                         const float *lr_dev, int *step_dev, float beta1, float beta2, float eps,
                                         float weight_decay, float max_norm, void *stream)
{
    if (!tensor_table || !work_list || !sumsq || !lr_dev || !step_dev) return MLAGG_E_NULLPTR;
    if (n_work <= 0) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const TensorRow *table = static_cast<const TensorRow *>(tensor_table);
    const int2 *work = static_cast<const int2 *>(work_list);
    AdamArgs a{0.f, beta1, beta2, eps, weight_decay, max_norm, 1.f, 1.f};
    MLAGG_TIMED(K_ADAMW, st);
    if (max_norm > 0.f) hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(n_work), dim3(OPT_TPB), 0, st, table, work, sumsq);
    // without clipping the reduce launch still runs (over zero partials) to advance the step counter
    hipLaunchKernelGGL(adamw_sumsq_reduce_kernel, dim3(1), dim3(OPT_TPB), 0, st, sumsq, max_norm > 0.f ? n_work : 0, step_dev);
    hipLaunchKernelGGL(adamw_update_kernel, dim3(n_work), dim3(OPT_TPB), 0, st, table, work, sumsq, a, lr_dev,
                       (const int *)step_dev);
    return (int)hipGetLastError();
}
