// FETCH_SIZE calibration for the access patterns of the group-per-wave scan backward (MI355X_MICROARCH.md, section HBM:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel reads every byte of a 512 MiB buffer exactly once; run under `rocprofv3 --pmc FETCH_SIZE` and compare.
//   coalesced16 : lane reads 16 B, a wave 1 KiB contiguous            (the guide's calibrated case: counter = bytes / 2)
//   rows64      : lane = 4 * row + s reads 16 B: a wave touches 16 rows x 64 B, rows 87040 B apart   (u, saved states)
//   rows32      : lane = 4 * row + s reads  8 B: a wave touches 16 rows x 32 B                        (dy)
// hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr size_t BYTES = 512ull << 20;
constexpr int L = 21760;                       // row length in floats (the scan's L at 256 x 256)

__global__ void coalesced16(const float4 *p, float *out, size_t n4)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// rows of L floats; a wave handles 16 consecutive rows, walks them in 16-float (64 B) steps
template <int W>     // W = 4: float4 per lane (64 B per row per step), W = 2: float2 (32 B per row per step)
__global__ void rows(const float *p, float *out, int nrows)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int row = wave * 16 + (lane >> 2), s = lane & 3;
    float acc = 0.f;
    if (row < nrows) {
        const float *r = p + (size_t)row * L;
        for (int t = 0; t + 4 * W <= L; t += 4 * W) {
            if (W == 4) { const float4 v = *reinterpret_cast<const float4 *>(r + t + 4 * s); acc += v.x + v.y + v.z + v.w; }
            else { const float2 v = *reinterpret_cast<const float2 *>(r + t + 2 * s); acc += v.x + v.y; }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    float *buf, *out;
    hipMalloc(&buf, BYTES); hipMalloc(&out, 4);
    hipMemset(buf, 0, BYTES);
    const int nrows = (int)(BYTES / 4 / L);
    printf("buffer %zu bytes; rows kernels read %zu bytes (%d rows x %d floats, tail of each row %d floats skipped)\n", BYTES,
           (size_t)nrows * (L / 16 * 16) * 4, nrows, L, L % 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(coalesced16, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, out, BYTES / 16);
        hipLaunchKernelGGL(rows<4>, dim3((nrows / 16 + 3) / 4), dim3(256), 0, 0, buf, out, nrows);
        hipLaunchKernelGGL(rows<2>, dim3((nrows / 16 + 3) / 4), dim3(256), 0, 0, buf, out, nrows);
    }
    hipDeviceSynchronize();
    return 0;
}
