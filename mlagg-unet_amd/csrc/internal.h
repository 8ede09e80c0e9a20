// Launchers shared between translation units of libmlagg_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mlagg_internal {

// out[c] (+)= sum_r part[r * pitch + c], c < cols: column sums of per-workgroup partial rows.
// Workgroup = 64 columns x 16 row-groups (1024 threads): coalesced 256-byte row reads, 16-way row
// parallelism, LDS combine.  Deterministic (fixed summation order), no atomics.
template <bool ACCUMULATE>
__global__ void __launch_bounds__(1024)
column_sum_kernel(const float *__restrict__ part, int rows, int pitch, int cols, float *__restrict__ out)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int r = rg;
        for (; r + 16 < rows; r += 32) {
            s0 += part[(size_t)r * pitch + c];
            s1 += part[(size_t)(r + 16) * pitch + c];
        }
        if (r < rows) s0 += part[(size_t)r * pitch + c];
    }
    red[rg][cx] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cx];
        if (ACCUMULATE) out[c] += s; else out[c] = s;
    }
}

// Same, columns [0, split) to out0 and [split, cols) to out1 (may be NULL): two reductions of one partial matrix in one launch.
template <int UNUSED = 0>              // a template only so that the header can hold the definition
__global__ void __launch_bounds__(1024)
column_sum_split_kernel(const float *__restrict__ part, int rows, int pitch, int cols, int split, float *__restrict__ out0,
                        float *__restrict__ out1)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int r = rg;
        for (; r + 16 < rows; r += 32) {
            s0 += part[(size_t)r * pitch + c];
            s1 += part[(size_t)(r + 16) * pitch + c];
        }
        if (r < rows) s0 += part[(size_t)r * pitch + c];
    }
    red[rg][cx] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cx];
        if (c < split) out0[c] = s;
        else if (out1) out1[c - split] = s;
    }
}

// Column sums of a (rows x cols) matrix whose columns alternate between two outputs: even columns -> out0[col / 2], odd -> out1.
template <int UNUSED = 0>
__global__ void __launch_bounds__(1024)
column_sum_interleaved_kernel(const float *__restrict__ part, int rows, int cols, float *__restrict__ out0,
                              float *__restrict__ out1)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (c < cols)
        for (int r = rg; r < rows; r += 16) s += part[(size_t)r * cols + c];
    red[rg][cx] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cx];
        float *o = (c & 1) ? out1 : out0;
        if (o) o[c >> 1] = t;
    }
}

size_t dwconv_wgrad_workspace_floats(int batch, int H, int W, int C);
// dw (C, 9) and dbias (C, may be NULL) are written (accumulate: added to); part = workspace of the size above
void dwconv_wgrad_launch(const float *x, int x_stride, const float *dy, int dy_stride, const float *pre, float *dw,
                         float *dbias, float *part, int batch, int H, int W, int C, int silu, hipStream_t st,
                         float *gbuf = nullptr, bool accumulate = false, const float *gate = nullptr, int gate_stride = 0,
                         float *dgate = nullptr, int dgate_stride = 0);
}  // namespace mlagg_internal
