// K15 -- weight gradient of the full convolutions on channel-major (NCHW / NCDHW) maps as "tap GEMMs" on the fp32 matrix cores.
//
// What it replaces: MIOpen's weight-gradient solvers behind `torch.nn.Conv2d / Conv3d` backward on the path -- for the 3-D network
// (reference variants/mamba/UMambaEnc_SS3D.py:477-513 BasicResBlock, :589-637 stages, :744-777 decoder) MIOpen's immediate mode picks
// a CK batched-GEMM weight-gradient kernel that runs 11..770 ms PER CONVOLUTION at 2 x 96x160x160 voxels (8.2 s of a 9.1 s train
// step, profiles/round3_a_config4_kernel_trace_timed_region.md) or its naive reference kernel; for the 2-D network its implicit-GEMM
// solvers want NHWC and pay two layout transposes per call.
//
// Formulation.  With the input AND the output gradient copied into zero-padded volumes of the SAME padded geometry (pad ring of
// the output gradient = 0), a k x k x k convolution's weight gradient is, per tap t = (kz, ky, kx),
//     dW[o][i][t] = sum_q dyp[o][q] * xp[i][q + off_t]             q = flat padded voxel index, off_t = a constant 1-D shift
// because the zero ring of dyp cancels every product whose input voxel would wrap around a row / plane.  A stride-2 convolution is
// the same sum over the 8 parity phases of the padded input (tap kz reads phase kz & 1 at shift kz >> 1).  So every tap is a GEMM
// "O x I, contraction over ~10^5..10^6 voxels, both operands contiguous along the contraction" -- pixel index as the MFMA k, as K5w
// does with tokens: for v_mfma_f32_32x32x2_f32 lane l supplies A[o = l & 31][k = l >> 5] and B[k][i = l & 31]; a lane loads ONE
// float4 of its channel's row per operand (lanes < 32: voxels q..q+3, lanes >= 32: q+4..q+7) and issues 4 MFMAs per tap with it.
// A wave owns (32 output channels) x (32 input channels) x (up to 9 taps: 144 accumulator registers) x (a slab of voxels); the A
// float4 is shared by its taps: 10 loads feed 36 MFMAs, the loop is matrix-core-bound.  Partial blocks per (slab, sample) are summed
// by a second kernel in a fixed order (no atomics) straight into the (O, I, taps) weight layout.
//
// Roofline: fp32 MFMA (157 TFLOP/s dense on gfx950): 2 * taps * O * I flop per voxel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 9;                   // taps per wave
constexpr int MAXTAPS = 27;

struct TapGeom {
    int O, I, ntaps, ngroups, otiles, itiles, slab, nslabs;
    long Q, a_row, a_batch, b_row, b_batch;
    long off[MAXTAPS];
};

// unaligned 16-byte global load (4-byte aligned address): the tap shifts are arbitrary
__device__ __forceinline__ float4 ld4u(const float *p)
{
    float4 v;
    v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
    return v;
}

__global__ void __launch_bounds__(64)
conv_wgrad_taps_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ part, TapGeom g)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    const int slab = blockIdx.x, b = blockIdx.z;
    int w = blockIdx.y;
    const int grp = w % g.ngroups; w /= g.ngroups;
    const int it = w % g.itiles, ot = w / g.itiles;
    const int t0 = grp * NT, nt = min(NT, g.ntaps - t0);
    const int o = min(ot * 32 + col, g.O - 1), i = min(it * 32 + col, g.I - 1);        // clamped: rows past O / I are never stored
    const long q0 = (long)slab * g.slab, q1 = min(g.Q, q0 + g.slab);
    const float *ap = A + (long)b * g.a_batch + (long)o * g.a_row + q0 + 4 * kh;
    const float *bp = B + (long)b * g.b_batch + (long)i * g.b_row + q0 + 4 * kh;
    long off[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) off[t] = g.off[t0 + min(t, nt - 1)];
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // Q and the slab are multiples of 8: whole steps only.  Double-buffered: the operands of the next 8 voxels are in flight behind
    // the 4 * NT MFMAs of the current ones
    float4 a4[2], b4[2][NT];
    const long nstep = (q1 - q0) >> 3;
    auto fetch = [&](int s) {
        a4[s] = *reinterpret_cast<const float4 *>(ap);
#pragma unroll
        for (int t = 0; t < NT; ++t) b4[s][t] = ld4u(bp + off[t]);
        ap += 8; bp += 8;
    };
    auto consume = [&](int s) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].x, b4[s][t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].y, b4[s][t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].z, b4[s][t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].w, b4[s][t].w, acc[t], 0, 0, 0);
        }
    };
    if (nstep > 0) {
        fetch(0);
        long s = 0;
        for (; s + 2 < nstep; s += 2) {
            fetch(1);
            consume(0);
            fetch(0);
            consume(1);
        }
        if (s + 1 < nstep) {
            fetch(1);
            consume(0);
            consume(1);
        } else {
            consume(0);
        }
    }
    // partial block: part[(b * nslabs + slab)][t][o][i]; D layout: column (i) = lane & 31, row (o) = (r & 3) + 8 * (r >> 2) + 4 * kh
    float *prow = part + ((long)b * g.nslabs + slab) * ((long)g.ntaps * g.O * g.I);
    const int ii = it * 32 + col;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt && ii < g.I) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oo = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (oo < g.O) prow[((long)(t0 + t) * g.O + oo) * g.I + ii] = acc[t][r];
            }
        }
    }
}

// dW[(o * I + i) * ntaps + t] = sum over rows of part[row][t][o][i]: a workgroup owns 64 consecutive partial columns and splits the
// rows over 16 row-groups, finished through LDS (fixed order)
__global__ void __launch_bounds__(1024)
conv_wgrad_reduce_kernel(const float *__restrict__ part, int rows, int ntaps, int O, int I, float *__restrict__ dW, int accumulate)
{
    __shared__ float red[16][65];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long n = (long)ntaps * O * I, c = (long)blockIdx.x * 64 + cx;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {
        int r = rg;
        for (; r + 16 < rows; r += 32) {
            s0 += part[(long)r * n + c];
            s1 += part[(long)(r + 16) * n + c];
        }
        if (r < rows) s0 += part[(long)r * n + c];
    }
    red[rg][cx] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cx];
        const int t = (int)(c / ((long)O * I)), oi = (int)(c - (long)t * O * I);
        float *dst = dW + (long)oi * ntaps + t;
        *dst = accumulate ? *dst + s : s;
    }
}

// Padded / phase-split copy of a channel-major volume for the tap GEMMs.
//   src (B, C, D, H, W) contiguous -> dst (B, [phases], C, guard + Dq*Hq*Wq + guard) where
//   stride 1: one phase, dst[z + 1][y + 1][x + 1] = src[z][y][x] inside a (D + 2, H + 2, Wq) box, zero elsewhere
//   stride 2: 8 parity phases of the zero-padded source: phase (pz, py, px)[z'][y'][x'] = srcpad[2 z' + pz][2 y' + py][2 x' + px],
//             srcpad[a] = src[a - 1], inside a (Dq, Hq, Wq) box
//   as_output: the same box geometry filled with an OUTPUT-sized map at origin `org` (1 for stride 1, 0 for the phase grid) -- the
//             zero ring that makes the flat 1-D shifts exact.
// One thread per 4 destination voxels of a row.
struct PadGeom {
    int C, D, H, W, Dq, Hq, Wq, stride, org, orgx, orgz, nphase;
    long guard, row;                   // row = 2 * guard + Dq * Hq * Wq
};

__global__ void __launch_bounds__(256)
volume_pad_kernel(const float *__restrict__ src, float *__restrict__ dst, PadGeom g, long total4)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int w4 = g.Wq >> 2;
    long r = idx;
    const int xq = (int)(r % w4) * 4; r /= w4;
    const int yq = (int)(r % g.Hq); r /= g.Hq;
    const int zq = (int)(r % g.Dq); r /= g.Dq;
    const int c = (int)(r % g.C); r /= g.C;
    const int ph = (int)(r % g.nphase);
    const long b = r / g.nphase;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int z, y, x;
        if (g.stride == 1 || g.nphase == 1) {
            z = zq - g.orgz; y = yq - g.org; x = xq + j - g.orgx;
        } else {
            z = 2 * zq + ((ph >> 2) & 1) - 1; y = 2 * yq + ((ph >> 1) & 1) - 1; x = 2 * (xq + j) + (ph & 1) - 1;
        }
        const bool ok = z >= 0 && z < g.D && y >= 0 && y < g.H && x >= 0 && x < g.W;
        v[j] = ok ? src[(((long)b * g.C + c) * g.D + z) * (long)g.H * g.W + (long)y * g.W + x] : 0.f;
    }
    float *p = dst + (((long)b * g.nphase + ph) * g.C + c) * g.row + g.guard + ((long)zq * g.Hq + yq) * g.Wq + xq;
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void __launch_bounds__(256)
guard_zero_kernel(float *__restrict__ dst, long rows, long row, long guard)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * 2 * guard) return;
    const long r = idx / (2 * guard), j = idx - r * 2 * guard;
    dst[r * row + (j < guard ? j : row - 2 * guard + j)] = 0.f;
}

}  // namespace

// geometry of the padded box for an input of (D, H, W) under a k = 3 (pad 1) or k = 1 (pad 0) convolution of stride 1 or 2.
// wide (stride 1, W % 4 == 0): the data starts at x = 4 instead of 1 (Wq = W + 8), so that aligned groups of 4 padded voxels are
// aligned groups of 4 data voxels -- the geometry of the forward / data-gradient tap kernels (conv_taps.hip), shared with K15.
extern "C" int mlagg_conv_pad_geometry(int D, int H, int W, int stride, int wide, int *Dq, int *Hq, int *Wq, long *guard)
{
    if (D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || !Dq || !Hq || !Wq || !guard) return MLAGG_E_UNSUPPORTED;
    if (stride == 2 && D == 1) return MLAGG_E_UNSUPPORTED;                  // 2-D maps (D = 1): stride 1 only
    if (wide && (stride != 1 || (W & 3))) return MLAGG_E_UNSUPPORTED;
    if (stride == 1) { *Dq = D == 1 ? 1 : D + 2; *Hq = H + 2; *Wq = wide ? W + 8 : (W + 2 + 3) & ~3; }      // a 2-D map gets no z ring
    else { *Dq = (D - 1) / 2 + 2; *Hq = (H - 1) / 2 + 2; *Wq = ((W - 1) / 2 + 2 + 3) & ~3; }
    // the box is walked in whole 8-voxel steps; shifts reach one plane + one row + one voxel either way
    *guard = (((long)*Hq * *Wq + *Wq + 1 + 8) + 7) & ~7L;
    return 0;
}

extern "C" int mlagg_volume_pad(const float *src, float *dst, int B, int C, int D, int H, int W, int stride, int wide, int as_output,
                                int out_D, int out_H, int out_W, void *stream)
{
    if (!src || !dst) return MLAGG_E_NULLPTR;
    int Dq, Hq, Wq;
    long guard;
    // `src` is either the convolution input (D, H, W) or, as_output, an output-sized map (out_D, out_H, out_W) laid into the box of
    // the input geometry (D, H, W)
    if (int rc = mlagg_conv_pad_geometry(D, H, W, stride, wide, &Dq, &Hq, &Wq, &guard)) return rc;
    if (B <= 0 || C <= 0) return MLAGG_E_UNSUPPORTED;
    PadGeom g;
    g.C = C; g.Dq = Dq; g.Hq = Hq; g.Wq = Wq; g.guard = guard;
    g.row = 2 * guard + (long)Dq * Hq * Wq;
    if (as_output) {
        g.D = out_D; g.H = out_H; g.W = out_W; g.stride = 1; g.nphase = 1; g.org = stride == 1 ? 1 : 0;
        g.orgz = (stride == 1 && D > 1) ? 1 : 0;
        g.orgx = wide ? 4 : g.org;
        if (out_D + g.orgz > Dq || out_H + g.org > Hq || out_W + g.orgx > Wq) return MLAGG_E_UNSUPPORTED;
    } else {
        g.D = D; g.H = H; g.W = W; g.stride = stride; g.nphase = stride == 1 ? 1 : 8; g.org = 1;
        g.orgz = D > 1 ? 1 : 0;
        g.orgx = wide ? 4 : 1;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long rows = (long)B * g.nphase * C, total4 = rows * Dq * Hq * (Wq >> 2);
    MLAGG_TIMED(K_CONV_PAD, st);
    hipLaunchKernelGGL(guard_zero_kernel, dim3((unsigned)((rows * 2 * guard + 255) / 256)), dim3(256), 0, st, dst, rows, g.row, guard);
    hipLaunchKernelGGL(volume_pad_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, src, dst, g, total4);
    return (int)hipGetLastError();
}

static int slab_len(long Q, long waves_per_slab)
{
    int slab = 8192;
    while (slab > 256 && ((Q + slab - 1) / slab) * waves_per_slab < 2048) slab >>= 1;
    return slab;
}

extern "C" size_t mlagg_conv_wgrad_taps_workspace_floats(int batch, long Q, int O, int I, int ntaps)
{
    if (batch <= 0 || Q <= 0 || O <= 0 || I <= 0 || ntaps <= 0) return 0;
    const long tiles = (long)((O + 31) / 32) * ((I + 31) / 32) * ((ntaps + NT - 1) / NT);
    const int slab = slab_len(Q, tiles * batch);
    return (size_t)batch * ((Q + slab - 1) / slab) * (size_t)ntaps * O * I;
}

// dW (O, I, ntaps) (+)= sum over batch and q of A[b][o][q] * B[b][i][q + tap_off[t]]: rows of a_row / b_row floats, samples
// a_batch / b_batch floats apart; Q, a_row, b_row and the pointers multiples of 4 floats; every B access must stay inside the
// caller's buffer (mlagg_volume_pad's guards).
extern "C" int mlagg_conv_wgrad_taps(const float *A, long a_batch, long a_row, const float *B, long b_batch, long b_row,
                                     const long *tap_off, int ntaps, long Q, int O, int I, int batch, float *dW, int accumulate,
                                     float *workspace, void *stream)
{
    if (!A || !B || !tap_off || !dW || !workspace) return MLAGG_E_NULLPTR;
    if (ntaps <= 0 || ntaps > MAXTAPS || Q <= 0 || (Q & 7) || O <= 0 || I <= 0 || batch <= 0 || batch > 65535) return MLAGG_E_UNSUPPORTED;
    if ((a_row & 3) || (a_batch & 3) || (((uintptr_t)A) & 15)) return MLAGG_E_UNSUPPORTED;
    TapGeom g;
    g.O = O; g.I = I; g.ntaps = ntaps; g.ngroups = (ntaps + NT - 1) / NT;
    g.otiles = (O + 31) / 32; g.itiles = (I + 31) / 32;
    g.Q = Q; g.a_row = a_row; g.a_batch = a_batch; g.b_row = b_row; g.b_batch = b_batch;
    for (int t = 0; t < MAXTAPS; ++t) g.off[t] = tap_off[t < ntaps ? t : 0];
    const long tiles = (long)g.otiles * g.itiles * g.ngroups;
    if (tiles > 65535) return MLAGG_E_UNSUPPORTED;
    g.slab = slab_len(Q, tiles * batch);
    g.nslabs = (int)((Q + g.slab - 1) / g.slab);
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        MLAGG_TIMED(K_CONV_WGRAD, st);
        hipLaunchKernelGGL(conv_wgrad_taps_kernel, dim3(g.nslabs, (unsigned)tiles, batch), dim3(64), 0, st, A, B, workspace, g);
    }
    const long n = (long)ntaps * O * I;
    MLAGG_TIMED(K_CONV_WGRAD_REDUCE, st);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, workspace, batch * g.nslabs, ntaps, O,
                       I, dW, accumulate);
    return (int)hipGetLastError();
}
