// K16 -- forward and data gradient of the dense stride-1 convolutions of the 3-D network as tap GEMMs on the fp32 matrix cores.
//
// What it replaces: MIOpen's immediate-mode pick for `nn.Conv3d` forward / backward-data in fp32 (reference
// variants/mamba/UMambaEnc_SS3D.py:477-513 BasicResBlock, :589-637 stages, :744-777 decoder): CK grouped-convolution kernels that
// run the 32 -> 32 channel 3x3x3 convolution at 2 x 96x160x160 voxels in 16.7 ms forward and 12 ms backward-data = 16 / 23 TFLOP/s
// (profiles/round3_b_config4_kernel_trace_timed_region.md: 114 + 106 ms per step, plus 35 ms of layout kernels).
//
// Formulation (the mirror of K15, csrc/conv_wgrad.hip): with the input in the zero-padded box of mlagg_volume_pad (wide form: data
// at x = 4, so aligned groups of 4 padded voxels are aligned groups of 4 data voxels),
//     y[o][q] = sum_t sum_i W[o][i][t] * xp[i][q + off_t]                q = flat padded voxel index
// is per tap a GEMM (32 output channels) x (voxels) with the input channel as the contraction.  For v_mfma_f32_32x32x2_f32 lane l
// supplies A[o = l & 31][k = l >> 5] = W (from an LDS image [tap][channel][32 o] of the workgroup's weight slice) and
// B[k][j = l & 31]: it loads ONE float4 = 4 consecutive voxels of its channel row (unaligned by the tap shift) and feeds 4 MFMAs
// with its components -- accumulator tile g holds the voxels 4 j + g, so for each output channel the four tiles give back 4
// consecutive voxels and the result leaves as 16-byte stores into the UNPADDED output map.  A wave owns 32 channels x 128 voxels
// (64 accumulator registers), a workgroup 4 waves; input channels are consumed in chunks of 32 (108 KiB of LDS for 27 taps).
// The data gradient is the same kernel on the padded OUTPUT gradient with the weight read transposed and the taps flipped
// (dx[i][q] = sum_t sum_o W[o][i][T - 1 - t] * dyp[o][q + off_t]); K15 then reuses both padded copies for the weight gradient.
//
// Roofline: fp32 MFMA (157 TFLOP/s dense): 2 * taps * O * I flop per voxel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MAXTAPS = 27;
constexpr int CCH = 32;                 // contraction channels per LDS weight image
constexpr int WAVE_VOX = 128;           // voxels per wave: 32 lanes x float4
constexpr int WG_VOX = 4 * WAVE_VOX;

struct CGeom {
    int O, I, ntaps, D, H, W, Dq, Hq, Wq, orgz;
    long Q, x_row, x_batch;
    long w_so, w_si;                    // element strides of the weight tensor for the output / contraction channel
    int flip;                           // taps read in reverse order (data gradient)
    long off[MAXTAPS];
};

__device__ __forceinline__ float4 ld4u(const float *p)
{
    float4 v;
    v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
    return v;
}

__global__ void __launch_bounds__(256)
conv_taps_kernel(const float *__restrict__ xp, const float *__restrict__ Wt, float *__restrict__ y, CGeom g)
{
    extern __shared__ float sW[];                       // [ntaps][CCH][32]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z, o0 = blockIdx.y * 32;
    const long qw = (long)blockIdx.x * WG_VOX + wave * WAVE_VOX;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // lanes of the last workgroup that lie past the box load from its last group instead (their results are never stored): no read
    // ever leaves [box start - guard, box end + guard)
    const float *xb = xp + (long)b * g.x_batch + min(qw + 4 * col, g.Q - 4);
    for (int ic = 0; ic < g.I; ic += CCH) {
        __syncthreads();                                // the previous chunk's weight reads are done
        for (int e = threadIdx.x; e < g.ntaps * CCH * 32; e += 256) {
            const int co = e & 31, ci = (e >> 5) & (CCH - 1), t = e / (CCH * 32);
            const int oo = o0 + co, ii = ic + ci;
            const int tt = g.flip ? g.ntaps - 1 - t : t;
            sW[e] = (oo < g.O && ii < g.I) ? Wt[(long)oo * g.w_so + (long)ii * g.w_si + tt] : 0.f;
        }
        __syncthreads();
        // 16 channel pairs per tap (channels past I are clamped: their weight rows are zero); the float4s of the NEXT tap are in
        // flight behind the 64 MFMAs of the current one (one wave per SIMD: the loop itself has to hide the load latency)
        float4 v0[CCH / 2], v1[CCH / 2];
        auto fetch = [&](float4 (&v)[CCH / 2], int t) {
            const float *p = xb + g.off[t];
#pragma unroll
            for (int m = 0; m < CCH / 2; ++m) v[m] = ld4u(p + (long)min(ic + 2 * m + kh, g.I - 1) * g.x_row);
        };
        auto consume = [&](const float4 (&v)[CCH / 2], int t) {
            const float *swt = sW + (t * CCH + kh) * 32 + col;
#pragma unroll
            for (int m = 0; m < CCH / 2; ++m) {
                const float a = swt[2 * m * 32];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[m].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[m].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[m].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[m].w, acc[3], 0, 0, 0);
            }
        };
        fetch(v0, 0);
        int t = 0;
        for (; t + 1 < g.ntaps; t += 2) {
            fetch(v1, t + 1);
            consume(v0, t);
            if (t + 2 < g.ntaps) fetch(v0, t + 2);
            consume(v1, t + 1);
        }
        if (t < g.ntaps) consume(v0, t);
    }
    // the lane's 4 voxels: padded flat index qg .. qg + 3 (qg % 4 == 0, one row, all inside or all outside the data box)
    const long qg = qw + 4 * col;
    const int xq = (int)(qg % g.Wq);
    const long r1 = qg / g.Wq;
    const int yq = (int)(r1 % g.Hq), zq = (int)(r1 / g.Hq);
    const bool inside = qg < g.Q && xq >= 4 && xq < g.W + 4 && yq >= 1 && yq <= g.H && zq >= g.orgz && zq < g.orgz + g.D;
    if (inside) {
        const long pos = ((long)(zq - g.orgz) * g.H + (yq - 1)) * g.W + (xq - 4), plane = (long)g.D * g.H * g.W;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oo = o0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (oo < g.O)
                *reinterpret_cast<float4 *>(y + ((long)b * g.O + oo) * plane + pos) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
    }
}

}  // namespace

// y (B, O, D, H, W) = sum_t sum_i W[o * w_so + i * w_si + t'] * xp[b][i][q + tap_off[t]], t' = t or ntaps - 1 - t (flip), over the
// padded box (Dq, Hq, Wq) of mlagg_volume_pad's wide stride-1 form holding a (D, H, W) map; xp points at the first box element of
// sample 0, channel 0 (rows of x_row floats, samples x_batch apart).  W % 4 == 0; tap_off: HOST array of ntaps <= 27 offsets.
extern "C" int mlagg_conv_taps(const float *xp, long x_batch, long x_row, const float *weight, long w_so, long w_si, int flip,
                               const long *tap_off, int ntaps, float *y, int B, int O, int I, int D, int H, int W, void *stream)
{
    if (!xp || !weight || !tap_off || !y) return MLAGG_E_NULLPTR;
    if (B <= 0 || O <= 0 || I <= 0 || D <= 0 || H <= 0 || W <= 0 || (W & 3) || ntaps <= 0 || ntaps > MAXTAPS || B > 65535) return MLAGG_E_UNSUPPORTED;
    CGeom g;
    g.O = O; g.I = I; g.ntaps = ntaps; g.D = D; g.H = H; g.W = W;
    g.Dq = D == 1 ? 1 : D + 2; g.Hq = H + 2; g.Wq = W + 8; g.orgz = D == 1 ? 0 : 1;
    g.Q = (long)g.Dq * g.Hq * g.Wq; g.x_row = x_row; g.x_batch = x_batch; g.w_so = w_so; g.w_si = w_si; g.flip = flip;
    for (int t = 0; t < MAXTAPS; ++t) g.off[t] = tap_off[t < ntaps ? t : 0];
    const size_t lds = (size_t)ntaps * CCH * 32 * sizeof(float);
    if (lds > 160 * 1024) return MLAGG_E_UNSUPPORTED;
    if (lds > 48 * 1024)
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_taps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            return (int)e;
    const long nwg = (g.Q + WG_VOX - 1) / WG_VOX;
    if (nwg > 2147483647L || (O + 31) / 32 > 65535) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CONV_TAPS, st);
    hipLaunchKernelGGL(conv_taps_kernel, dim3((unsigned)nwg, (O + 31) / 32, B), dim3(256), lds, st, xp, weight, y, g);
    return (int)hipGetLastError();
}
