// K17: the key/value reduction of the pooled attention branch in one pass each way,
//   pooled[b][p][c] = mean over the r x r window p of GELU(s[b][tok][c])
// (nnUNetTrainer_MLAgg_2D_dt_MS.py:722  self.pool(self.act(self.sr(x_))): nn.GELU (erf form, T:671) followed by
// nn.AdaptiveAvgPool2d((H / r, W / r)) (T:668), which for H % r == W % r == 0 is the plain window mean).
// s is the third column block of the stacked q | v | sr projection (row stride s_stride), token-major; pooled is (B, P, d).
// Forward: one lane per (window, 4 channels), r * r independent 16-byte loads, nothing but `pooled` is written (the GELU map never
// exists in memory).  Backward: one lane per (token, 4 channels): ds = d(pooled)[window] / r^2 * GELU'(s), written at row stride
// ds_stride (the gradient buffer of the stacked projection).  HBM-bound: 4 d bytes read per token forward, 4 d read + 4 d written
// backward; d(pooled) is read r^2 times from L2.
#include <hip/hip_runtime.h>

#include "../../include/mlagg_hip.h"
#include "prof.h"

namespace {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x)
{
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

struct PoolGeom {
    int batch, H, W, d, r, PH, PW, q;      // q = d / 4
    float inv;
};

// A workgroup owns `cells` windows; a thread sums ONE window row (r pixels, 16-byte loads) of 4 channels, the r row sums of a
// window meet in LDS and are added in a fixed order.  (One thread per window -- r * r = 64 dependent-address loads each at stage 0,
// 480 waves -- took 159 us on the 128 x 128 map.)
__global__ void __launch_bounds__(256)
gelu_pool_fwd_kernel(const float *__restrict__ s, int s_stride, float *__restrict__ pooled, PoolGeom g, int cells)
{
    __shared__ float4 part[256];
    const int per = g.r * g.q;                                  // threads per window
    const int tid = threadIdx.x;
    const int cl = tid / per, rem = tid - cl * per, yr = rem / g.q, c4 = rem - yr * g.q;
    const long ncell = (long)g.batch * g.PH * g.PW;
    const long cell = (long)blockIdx.x * cells + cl;
    const bool live = cl < cells && cell < ncell;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const int px = (int)(cell % g.PW);
        const long t = cell / g.PW;
        const int py = (int)(t % g.PH), b = (int)(t / g.PH);
        const float *row = s + ((long)b * g.H * g.W + (long)(py * g.r + yr) * g.W + (long)px * g.r) * s_stride + 4 * c4;
#pragma unroll 4
        for (int x = 0; x < g.r; ++x) {
            const float4 v = *reinterpret_cast<const float4 *>(row + (long)x * s_stride);
            acc.x += gelu_f(v.x);
            acc.y += gelu_f(v.y);
            acc.z += gelu_f(v.z);
            acc.w += gelu_f(v.w);
        }
    }
    part[tid] = acc;
    __syncthreads();
    if (live && yr == 0) {
        float4 sum = part[tid];
        for (int k = 1; k < g.r; ++k) {
            const float4 v = part[tid + k * g.q];
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        *reinterpret_cast<float4 *>(pooled + (cell * g.q + c4) * 4) = make_float4(sum.x * g.inv, sum.y * g.inv, sum.z * g.inv, sum.w * g.inv);
    }
}

__global__ void __launch_bounds__(256)
gelu_pool_bwd_kernel(const float *__restrict__ s, int s_stride, const float *__restrict__ dpooled, float *__restrict__ ds,
                     int ds_stride, PoolGeom g)
{
    const long n = (long)g.batch * g.H * g.W * g.q;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c4 = (int)(i % g.q);
    const long tok = i / g.q;                     // b * H * W + y * W + x
    const int x = (int)(tok % g.W);
    const long t = tok / g.W;
    const int y = (int)(t % g.H);
    const int b = (int)(t / g.H);
    const long p = ((long)b * g.PH + y / g.r) * g.PW + x / g.r;
    const float4 v = *reinterpret_cast<const float4 *>(s + tok * s_stride + 4 * c4);
    const float4 dp = *reinterpret_cast<const float4 *>(dpooled + (p * g.q + c4) * 4);
    *reinterpret_cast<float4 *>(ds + tok * ds_stride + 4 * c4) =
        make_float4(dp.x * g.inv * gelu_grad_f(v.x), dp.y * g.inv * gelu_grad_f(v.y), dp.z * g.inv * gelu_grad_f(v.z),
                    dp.w * g.inv * gelu_grad_f(v.w));
}

int make_geom(PoolGeom &g, int batch, int H, int W, int d, int r, int s_stride)
{
    if (batch <= 0 || H <= 0 || W <= 0 || d <= 0 || r <= 0) return MLAGG_E_UNSUPPORTED;
    if ((d & 3) || s_stride < d || (s_stride & 3) || H % r || W % r) return MLAGG_E_UNSUPPORTED;
    g = PoolGeom{batch, H, W, d, r, H / r, W / r, d / 4, 1.f / (float)(r * r)};
    return 0;
}

}  // namespace

extern "C" int mlagg_gelu_pool_fwd(const float *s, int s_stride, float *pooled, int batch, int H, int W, int d, int r, void *stream)
{
    if (!s || !pooled) return MLAGG_E_NULLPTR;
    PoolGeom g;
    if (int rc = make_geom(g, batch, H, W, d, r, s_stride)) return rc;
    if (reinterpret_cast<uintptr_t>(s) & 15) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_GELU_POOL, st);
    const int per = g.r * g.q;
    if (per > 256) return MLAGG_E_UNSUPPORTED;
    const int cells = 256 / per;
    const long ncell = (long)batch * g.PH * g.PW;
    hipLaunchKernelGGL(gelu_pool_fwd_kernel, dim3((unsigned)((ncell + cells - 1) / cells)), dim3(256), 0, st, s, s_stride, pooled, g, cells);
    return (int)hipGetLastError();
}

extern "C" int mlagg_gelu_pool_bwd(const float *s, int s_stride, const float *dpooled, float *ds, int ds_stride, int batch, int H,
                                   int W, int d, int r, void *stream)
{
    if (!s || !dpooled || !ds) return MLAGG_E_NULLPTR;
    PoolGeom g;
    if (int rc = make_geom(g, batch, H, W, d, r, s_stride)) return rc;
    if (ds_stride < d || (ds_stride & 3) || ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(ds)) & 15))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_GELU_POOL, st);
    const long n = (long)batch * H * W * g.q;
    hipLaunchKernelGGL(gelu_pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, s_stride, dpooled, ds, ds_stride,
                       g);
    return (int)hipGetLastError();
}
