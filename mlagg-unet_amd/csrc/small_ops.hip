// K8 -- the launch-count killers of the MLLA / MSMM blocks: tiny or purely streaming ops that the eager
// reference spends 6-15 kernel launches on each.
//
//  * diff_lambda: lambda = exp(<lq1, lk1>) - exp(<lq2, lk2>) + lambda_init of the differential attention
//    (reference nnUNetTrainer_MLAgg_2D_dt_MS.py:709-711, 770-772): 2 mul + 2 sum + 2 exp + sub + add forward and
//    ~14 kernels backward per attention module, 16 modules -> ~350 launches per train step, here 1 + 1.
//  * scaled_residual: out = skip + branch * scale[sample] -- the residual connection under stochastic depth
//    (timm DropPath, T:903, 907; MambaSkip.py:741, 745).  ATen's addcmul with a (B, 1, 1) operand runs its
//    generic broadcast kernel (103 us for 3 x 63 MB at stage 0); this is a float4 stream.
//    row_scale: out = x * scale[sample], the branch gradient of the same op.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

__global__ void __launch_bounds__(64)
diff_lambda_fwd_kernel(const float *__restrict__ q1, const float *__restrict__ k1, const float *__restrict__ q2,
                       const float *__restrict__ k2, float init, int n, float *__restrict__ lam, float *__restrict__ e)
{
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) {
        s1 += q1[i] * k1[i];
        s2 += q2[i] * k2[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
    }
    if (threadIdx.x == 0) {
        const float e1 = expf(s1), e2 = expf(s2);
        lam[0] = e1 - e2 + init;
        e[0] = e1;
        e[1] = e2;
    }
}

__global__ void __launch_bounds__(64)
diff_lambda_bwd_kernel(const float *__restrict__ dlam, const float *__restrict__ q1, const float *__restrict__ k1,
                       const float *__restrict__ q2, const float *__restrict__ k2, const float *__restrict__ e, int n,
                       float *__restrict__ dq1, float *__restrict__ dk1, float *__restrict__ dq2, float *__restrict__ dk2)
{
    const float g1 = dlam[0] * e[0], g2 = -dlam[0] * e[1];
    for (int i = threadIdx.x; i < n; i += 64) {
        dq1[i] = g1 * k1[i];
        dk1[i] = g1 * q1[i];
        dq2[i] = g2 * k2[i];
        dk2[i] = g2 * q2[i];
    }
}

// per4 = floats per sample / 4; RESIDUAL: out = skip + x * scale[b], else out = x * scale[b]
template <bool RESIDUAL>
__global__ void __launch_bounds__(256)
row_scale_kernel(const float *__restrict__ skip, const float *__restrict__ x, const float *__restrict__ scale,
                 float *__restrict__ out, long per4, long total4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const float s = scale[i / per4];
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        float4 o = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
        if (RESIDUAL) {
            const float4 k = reinterpret_cast<const float4 *>(skip)[i];
            o.x += k.x; o.y += k.y; o.z += k.z; o.w += k.w;
        }
        reinterpret_cast<float4 *>(out)[i] = o;
    }
}

int stream_grid(long total4)
{
    long blocks = (total4 + 255) / 256;
    return (int)(blocks < 1 ? 1 : blocks > 8192 ? 8192 : blocks);
}

}  // namespace

extern "C" int mlagg_diff_lambda_fwd(const float *q1, const float *k1, const float *q2, const float *k2, float lambda_init,
                                     int n, float *lam, float *saved_exp, void *stream)
{
    if (!q1 || !k1 || !q2 || !k2 || !lam || !saved_exp) return MLAGG_E_NULLPTR;
    if (n <= 0) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(diff_lambda_fwd_kernel, dim3(1), dim3(64), 0, st, q1, k1, q2, k2, lambda_init, n, lam, saved_exp);
    return (int)hipGetLastError();
}

extern "C" int mlagg_diff_lambda_bwd(const float *dlam, const float *q1, const float *k1, const float *q2, const float *k2,
                                     const float *saved_exp, int n, float *dq1, float *dk1, float *dq2, float *dk2,
                                     void *stream)
{
    if (!dlam || !q1 || !k1 || !q2 || !k2 || !saved_exp || !dq1 || !dk1 || !dq2 || !dk2) return MLAGG_E_NULLPTR;
    if (n <= 0) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(diff_lambda_bwd_kernel, dim3(1), dim3(64), 0, st, dlam, q1, k1, q2, k2, saved_exp, n, dq1, dk1, dq2,
                       dk2);
    return (int)hipGetLastError();
}

extern "C" int mlagg_scaled_residual(const float *skip, const float *branch, const float *scale, float *out, int batch,
                                     long per_sample, void *stream)
{
    if (!branch || !scale || !out) return MLAGG_E_NULLPTR;
    if (batch <= 0 || per_sample <= 0 || (per_sample & 3)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long per4 = per_sample >> 2, total4 = per4 * batch;
    MLAGG_TIMED(K_ROW_SCALE, st);
    if (skip)
        hipLaunchKernelGGL(row_scale_kernel<true>, dim3(stream_grid(total4)), dim3(256), 0, st, skip, branch, scale, out,
                           per4, total4);
    else
        hipLaunchKernelGGL(row_scale_kernel<false>, dim3(stream_grid(total4)), dim3(256), 0, st, skip, branch, scale, out,
                           per4, total4);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Batched 2-D transpose dst[b][c][r] = src[b][r][c] (rows R, columns Cc, both contiguous): the NCHW <-> token-major
// flips at the stage boundaries of the encoder, the MSMM block and PatchEmbed's LayerNorms.  ATen's generic strided
// copy runs them at ~2 TB/s (62 us for 63 MB in + 63 MB out); a 64x64 LDS tile with 16-byte global accesses on both
// sides streams.
// ------------------------------------------------------------------------------------------------------------
namespace {

constexpr int TT = 64;             // tile side
constexpr int TTP = TT + 1;        // LDS pitch (conflict-free column reads)

__global__ void __launch_bounds__(256)
transpose_tile_kernel(const float *__restrict__ src, float *__restrict__ dst, int R, int Cc, long src_bstride, long dst_bstride)
{
    __shared__ float tile[TT * TTP];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * TT, c0 = blockIdx.x * TT;
    const float *s = src + (size_t)b * src_bstride;
    float *d = dst + (size_t)b * dst_bstride;
    const int tid = threadIdx.x;
    const bool v_in = (Cc & 3) == 0, v_out = (R & 3) == 0;
    // load: 64 rows x 16 float4
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = (tid >> 4) + 16 * it, c4 = (tid & 15) * 4;
        const int r = r0 + row, c = c0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) {
            const float *p = s + (size_t)r * Cc + c;
            if (v_in && c + 3 < Cc) v = *reinterpret_cast<const float4 *>(p);
            else {
                if (c < Cc) v.x = p[0];
                if (c + 1 < Cc) v.y = p[1];
                if (c + 2 < Cc) v.z = p[2];
                if (c + 3 < Cc) v.w = p[3];
            }
        }
        float *t = tile + row * TTP + c4;
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
    // store: 64 output rows (= source columns) x 16 float4 along the source-row axis
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int col = (tid >> 4) + 16 * it, r4 = (tid & 15) * 4;
        const int c = c0 + col, r = r0 + r4;
        if (c >= Cc) continue;
        const float4 v = make_float4(tile[r4 * TTP + col], tile[(r4 + 1) * TTP + col], tile[(r4 + 2) * TTP + col],
                                     tile[(r4 + 3) * TTP + col]);
        float *p = d + (size_t)c * R + r;
        if (v_out && r + 3 < R) *reinterpret_cast<float4 *>(p) = v;
        else {
            if (r < R) p[0] = v.x;
            if (r + 1 < R) p[1] = v.y;
            if (r + 2 < R) p[2] = v.z;
            if (r + 3 < R) p[3] = v.w;
        }
    }
}

}  // namespace

extern "C" int mlagg_transpose_2d(const float *src, long src_batch_stride, float *dst, int batch, int R, int C, void *stream)
{
    return mlagg_transpose_2d_into(src, src_batch_stride, dst, 0, batch, R, C, stream);
}

// ... with the (C, R) results dst_batch_stride floats apart (0 = R * C): the transposed matrices land in a channel slice of a wider map
extern "C" int mlagg_transpose_2d_into(const float *src, long src_batch_stride, float *dst, long dst_batch_stride, int batch, int R, int C,
                                       void *stream)
{
    if (!src || !dst) return MLAGG_E_NULLPTR;
    if (batch <= 0 || R <= 0 || C <= 0 || batch > 65535 || (R + TT - 1) / TT > 65535) return MLAGG_E_UNSUPPORTED;
    if (src_batch_stride == 0) src_batch_stride = (long)R * C;
    if (dst_batch_stride == 0) dst_batch_stride = (long)R * C;
    if (dst_batch_stride < (long)R * C || ((R & 3) == 0 && (dst_batch_stride & 3))) return MLAGG_E_UNSUPPORTED;
    // 16-byte row loads are used when C % 4 == 0: the batch stride must then keep rows 16-byte aligned
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) != 0 || src_batch_stride < (long)R * C ||
        ((C & 3) == 0 && (src_batch_stride & 3)))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_TRANSPOSE, st);
    hipLaunchKernelGGL(transpose_tile_kernel, dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, st, src, dst, R, C,
                       src_batch_stride, dst_batch_stride);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Pixel shuffle by 2 (round 4): y[b][o][2 i + a][2 j + c] = z[b][(2 a + c) O + o][i][j] and its inverse -- what turns the
// kernel-2 / stride-2 transposed convolution of the decoder (`get_conv_layer(..., kernel_size=2, stride=2, is_transposed=True)` in
// UnetrUpBlock, nnUNetTrainer_MLAgg_2D_dt_MS.py:1340-1368 via MONAI) into ONE pointwise product with 4 O output channels on K18:
// the taps of that convolution do not overlap, so output pixel (2 i + a, 2 j + c) is tap (a, c) applied to input pixel (i, j).
// A thread owns two input pixels: four 8-byte reads (one per tap plane), two 16-byte writes (output rows 2 i and 2 i + 1).
// ------------------------------------------------------------------------------------------------------------
namespace {
template <bool INVERSE>
__global__ void __launch_bounds__(256)
pixel_shuffle2_kernel(const float *__restrict__ src, float *__restrict__ dst, int O, int H, int W, long y_batch)
{
    const int w2 = W >> 1;                                          // pixel pairs per input row
    const long n = (long)O * H * w2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int b = blockIdx.y;
    const int jp = (int)(idx % w2), i = (int)((idx / w2) % H), o = (int)(idx / ((long)w2 * H));
    const size_t plane = (size_t)H * W;
    // z (B, 4 O, H, W), y (B, O, 2 H, 2 W)
    const size_t zoff = ((size_t)b * 4 * O + o) * plane + (size_t)i * W + 2 * jp;
    const size_t yoff = (size_t)b * y_batch + ((size_t)o * 2 * H + 2 * i) * (2 * (size_t)W) + 4 * jp;      // y_batch: floats between samples of y
    if (!INVERSE) {
        float2 t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const float2 *>(src + zoff + (size_t)q * O * plane);
        *reinterpret_cast<float4 *>(dst + yoff) = make_float4(t[0].x, t[1].x, t[0].y, t[1].y);
        *reinterpret_cast<float4 *>(dst + yoff + 2 * (size_t)W) = make_float4(t[2].x, t[3].x, t[2].y, t[3].y);
    } else {
        const float4 r0 = *reinterpret_cast<const float4 *>(src + yoff), r1 = *reinterpret_cast<const float4 *>(src + yoff + 2 * (size_t)W);
        *reinterpret_cast<float2 *>(dst + zoff) = make_float2(r0.x, r0.z);
        *reinterpret_cast<float2 *>(dst + zoff + (size_t)O * plane) = make_float2(r0.y, r0.w);
        *reinterpret_cast<float2 *>(dst + zoff + 2 * (size_t)O * plane) = make_float2(r1.x, r1.z);
        *reinterpret_cast<float2 *>(dst + zoff + 3 * (size_t)O * plane) = make_float2(r1.y, r1.w);
    }
}
}  // namespace

extern "C" int mlagg_pixel_shuffle2(const float *src, float *dst, int B, int O, int H, int W, int inverse, void *stream)
{
    if (!src || !dst) return MLAGG_E_NULLPTR;
    if (B <= 0 || B > 65535 || O <= 0 || H <= 0 || W <= 0 || (W & 1)) return MLAGG_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_TRANSPOSE, st);
    const long n = (long)O * H * (W >> 1);
    const dim3 grid((unsigned)((n + 255) / 256), B);
    const long yb = (long)O * 4 * H * W;
    if (inverse)
        hipLaunchKernelGGL(pixel_shuffle2_kernel<true>, grid, dim3(256), 0, st, src, dst, O, H, W, yb);
    else
        hipLaunchKernelGGL(pixel_shuffle2_kernel<false>, grid, dim3(256), 0, st, src, dst, O, H, W, yb);
    return (int)hipGetLastError();
}

extern "C" int mlagg_pixel_unshuffle2_strided(const float *src, long src_batch, float *dst, int B, int O, int H, int W, void *stream)
{
    if (!src || !dst) return MLAGG_E_NULLPTR;
    const long yb = (long)O * 4 * H * W;
    if (src_batch == 0) src_batch = yb;
    if (B <= 0 || B > 65535 || O <= 0 || H <= 0 || W <= 0 || (W & 1) || src_batch < yb || (src_batch & 3)) return MLAGG_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_TRANSPOSE, st);
    const long n = (long)O * H * (W >> 1);
    hipLaunchKernelGGL(pixel_shuffle2_kernel<true>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, src, dst, O, H, W, src_batch);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Bias gradients.  channel_sum: out[c] = sum_{b, p} g[b][c][p] of an NCHW map (what convolution backward needs for its bias;
// ATen's generic reduction runs these at 1.3-2 TB/s: 49 us for 63 MB).  column_sum: out[c] = sum_r x[r][c] of a row-major
// matrix (the bias gradient of the small-M Linear layers that go to the library GEMM).
// ------------------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256)
plane_sum_kernel(const float *__restrict__ g, float *__restrict__ part, int C, long HW)
{
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float *p = g + ((size_t)b * C + c) * HW;
    float s0 = 0.f, s1 = 0.f;
    const long n4 = ((HW & 3) == 0 && (((uintptr_t)p) & 15) == 0) ? HW >> 2 : 0;
    for (long i = threadIdx.x; i < n4; i += 512) {
        const float4 a = reinterpret_cast<const float4 *>(p)[i];
        s0 += (a.x + a.y) + (a.z + a.w);
        if (i + 256 < n4) {
            const float4 b4 = reinterpret_cast<const float4 *>(p)[i + 256];
            s1 += (b4.x + b4.y) + (b4.z + b4.w);
        }
    }
    for (long i = 4 * n4 + threadIdx.x; i < HW; i += 256) s0 += p[i];
    float s = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)b * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

#include "internal.h"

extern "C" size_t mlagg_channel_sum_workspace_floats(int B, int C) { return (size_t)(B > 0 ? B : 0) * (C > 0 ? C : 0); }

extern "C" int mlagg_channel_sum(const float *g, float *out, float *workspace, int B, int C, long HW, void *stream)
{
    if (!g || !out || !workspace) return MLAGG_E_NULLPTR;
    if (B <= 0 || C <= 0 || HW <= 0 || B > 65535) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_BIAS_GRAD, st);
    hipLaunchKernelGGL(plane_sum_kernel, dim3(C, B), dim3(256), 0, st, g, workspace, C, HW);
    hipLaunchKernelGGL(mlagg_internal::column_sum_kernel<false>, dim3((C + 63) / 64), dim3(1024), 0, st, workspace, B, C, C, out);
    return (int)hipGetLastError();
}

// rows are cut into slabs when one 1024-thread workgroup per 64 columns would leave most of the 256 CUs idle (7840 x 384: 6
// workgroups, 79 us); slab s sums rows [s * per, (s + 1) * per) into workspace row s, a second launch sums the slabs
namespace {
int column_sum_slabs(int rows, int cols)
{
    const int groups = (cols + 63) / 64;
    if (rows < 2048 || groups >= 128) return 1;
    int s = 256 / groups;
    if (s > rows / 512) s = rows / 512;
    return s < 2 ? 1 : s;
}

__global__ void __launch_bounds__(1024)
column_sum_slab_kernel(const float *__restrict__ x, int rows, int pitch, int cols, int per, float *__restrict__ part)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int r = r0 + rg;
        for (; r + 16 < r1; r += 32) {
            s0 += x[(size_t)r * pitch + c];
            s1 += x[(size_t)(r + 16) * pitch + c];
        }
        if (r < r1) s0 += x[(size_t)r * pitch + c];
    }
    red[rg][cx] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cx];
        part[(size_t)blockIdx.y * cols + c] = s;
    }
}
}  // namespace

extern "C" size_t mlagg_column_sum_workspace_floats(int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    const int s = column_sum_slabs(rows, cols);
    return s > 1 ? (size_t)s * cols : 0;
}

extern "C" int mlagg_column_sum(const float *x, int x_stride, float *out, float *workspace, int rows, int cols, void *stream)
{
    if (!x || !out) return MLAGG_E_NULLPTR;
    if (rows <= 0 || cols <= 0 || x_stride < cols) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_BIAS_GRAD, st);
    const int groups = (cols + 63) / 64;
    const int slabs = workspace ? column_sum_slabs(rows, cols) : 1;
    if (slabs > 1) {
        const int per = (rows + slabs - 1) / slabs;
        hipLaunchKernelGGL(column_sum_slab_kernel, dim3(groups, slabs), dim3(1024), 0, st, x, rows, x_stride, cols, per, workspace);
        hipLaunchKernelGGL(mlagg_internal::column_sum_kernel<false>, dim3(groups), dim3(1024), 0, st, workspace, slabs, cols, cols, out);
    } else {
        hipLaunchKernelGGL(mlagg_internal::column_sum_kernel<false>, dim3(groups), dim3(1024), 0, st, x, rows, x_stride, cols, out);
    }
    return (int)hipGetLastError();
}
