// K1' for volumes -- cross-scan / cross-merge by index table.
//
// The 3-D selective scan (reference variants/mamba/UMambaEnc_SS3D.py:244-296, `SS3D.forward_corev0`: the design source for
// the 3-D MLAgg variant, SURVEY finding 6) runs K = 12 directions over the D*H*W tokens of a volume: the six axis orders
// (d,h,w) (d,w,h) (h,d,w) (h,w,d) (w,d,h) (w,h,d) and their reversals, built in the reference with stack / permute /
// contiguous / flip / cat chains (14 full copies of the (B, C, L) tensor) and undone with another 11.  Here a direction is
// an int32 permutation table idx[k][l] (position of the l-th scan step in natural (d,h,w) order; host-built once per
// volume shape) and two kernels do every re-ordering, both directions of autograd included:
//   scan   seq[b][k * CB + c][l]  = tok[b][idx[k][l]][k * blk_stride + c]          token-major -> scan rows
//   merge  tok[b][idx[k][l]][k * blk_stride + c] (+)= seq[b][k * CB + c][l]        scan rows -> token-major
//          (blk_stride = 0: all K directions land on the same columns and are SUMMED, SS3D.forward's torch.sum(y, dim=1))
// A workgroup moves a 64-step x CB-channel tile through LDS so that both sides are touched in runs: scan rows as 256-byte
// runs along l, token rows as CB-float runs.  The merge with blk_stride = 0 accumulates with float atomics (12 directions,
// different workgroups): the sum order is not fixed.  HBM-bound: 4 * (K + K) * CB bytes per token.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int TL = 64;                 // scan steps per tile
constexpr int UN = 8;                  // loads in flight per thread

template <bool MERGE>
__global__ void __launch_bounds__(256)
index_scan_kernel(float *__restrict__ tok, long tok_stride, int blk_stride, const int *__restrict__ idx, float *__restrict__ seq,
                  int L, int K, int CB, int atomic)
{
    extern __shared__ float sT[];                        // [CB][TL + 1]
    __shared__ int sI[TL];
    const int l0 = blockIdx.x * TL, k = blockIdx.y, b = blockIdx.z;
    const int n = min(TL, L - l0);
    if (threadIdx.x < TL) sI[threadIdx.x] = threadIdx.x < n ? idx[(size_t)k * L + l0 + threadIdx.x] : 0;
    float *srow = seq + ((size_t)b * K + k) * CB * L;
    float *tb = tok + (size_t)b * L * tok_stride + (size_t)k * blk_stride;
    if (!MERGE) {
        __syncthreads();
        // UN gathers in flight per thread, unconditional (elements past the tile read its first element and are dropped): one
        // guarded load per trip is waited for before the next is issued ("Guarded memory operations")
        for (int e0 = threadIdx.x; e0 < n * CB; e0 += 256 * UN) {
            float v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + 256 * u;
                const bool ok = e < n * CB;
                const int i = ok ? e / CB : 0, c = ok ? e - i * CB : 0;
                v[u] = tb[(size_t)sI[i] * tok_stride + c];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + 256 * u;
                if (e < n * CB) { const int i = e / CB, c = e - i * CB; sT[c * (TL + 1) + i] = v[u]; }
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < CB * TL; e += 256) {
            const int c = e / TL, i = e - c * TL;
            if (i < n) srow[(size_t)c * L + l0 + i] = sT[c * (TL + 1) + i];
        }
    } else {
        for (int e0 = threadIdx.x; e0 < CB * TL; e0 += 256 * UN) {
            float v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + 256 * u;
                const int c = min(e / TL, CB - 1), i = e % TL;
                v[u] = srow[(size_t)c * L + l0 + (i < n ? i : 0)];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + 256 * u;
                const int c = e / TL, i = e - c * TL;
                if (e < CB * TL && i < n) sT[c * (TL + 1) + i] = v[u];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n * CB; e += 256) {
            const int i = e / CB, c = e - i * CB;
            float *dst = tb + (size_t)sI[i] * tok_stride + c;
            const float v = sT[c * (TL + 1) + i];
            if (atomic) atomicAdd(dst, v); else *dst = v;
        }
    }
}

// out[row][c] = sum_k wide[row][k * CB + c]: the K column blocks of the direction-separated merge, summed in a FIXED order
// (the atomic form of the summed merge is order-dependent and paid 12 read-modify-writes per output in L2)
__global__ void __launch_bounds__(256)
block_sum_kernel(const float *__restrict__ wide, float *__restrict__ out, long rows, int K, int CB)
{
    const int q4 = CB >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * q4) return;
    const long row = i / q4;
    const int c = 4 * (int)(i - row * q4);
    const float *p = wide + row * (long)K * CB + c;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int k = 0;
    for (; k + 1 < K; k += 2) {
        const float4 a = *reinterpret_cast<const float4 *>(p + (long)k * CB), b = *reinterpret_cast<const float4 *>(p + (long)(k + 1) * CB);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    }
    if (k < K) {
        const float4 a = *reinterpret_cast<const float4 *>(p + (long)k * CB);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    *reinterpret_cast<float4 *>(out + row * CB + c) = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
}

int check(int B, int L, int K, int CB, long tok_stride, int blk_stride)
{
    if (B <= 0 || L <= 0 || K <= 0 || CB <= 0 || B > 65535 || K > 65535) return MLAGG_E_UNSUPPORTED;
    if (CB > 512 || tok_stride < CB || blk_stride < 0) return MLAGG_E_UNSUPPORTED;       // LDS tile: CB x 65 floats <= 130 KiB
    return 0;
}

template <typename Kn>
int allow_lds(Kn kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return 0;
}

}  // namespace

extern "C" int mlagg_index_scan(const float *tok, long tok_stride, int blk_stride, const int *idx, float *seq, int B, int L, int K,
                                int CB, void *stream)
{
    if (!tok || !idx || !seq) return MLAGG_E_NULLPTR;
    if (int rc = check(B, L, K, CB, tok_stride, blk_stride)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)CB * (TL + 1) * sizeof(float);
    if (int rc = allow_lds(index_scan_kernel<false>, lds)) return rc;
    MLAGG_TIMED(K_CROSS_SCAN, st);
    hipLaunchKernelGGL(index_scan_kernel<false>, dim3((L + TL - 1) / TL, K, B), dim3(256), lds, st, const_cast<float *>(tok), tok_stride,
                       blk_stride, idx, seq, L, K, CB, 0);
    return (int)hipGetLastError();
}

// blk_stride = 0: the K directions are summed into the same CB columns (tok is zero-filled by the call first)
extern "C" int mlagg_index_merge(const float *seq, const int *idx, float *tok, long tok_stride, int blk_stride, int B, int L, int K,
                                 int CB, void *stream)
{
    if (!tok || !idx || !seq) return MLAGG_E_NULLPTR;
    if (int rc = check(B, L, K, CB, tok_stride, blk_stride)) return rc;
    if (blk_stride == 0 && tok_stride != CB) return MLAGG_E_UNSUPPORTED;      // the summed form owns whole rows (it zero-fills them)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)CB * (TL + 1) * sizeof(float);
    if (int rc = allow_lds(index_scan_kernel<true>, lds)) return rc;
    const int atomic = blk_stride == 0 && K > 1;
    if (atomic) (void)hipMemsetAsync(tok, 0, (size_t)B * L * tok_stride * sizeof(float), st);
    MLAGG_TIMED(K_CROSS_MERGE, st);
    hipLaunchKernelGGL(index_scan_kernel<true>, dim3((L + TL - 1) / TL, K, B), dim3(256), lds, st, tok, tok_stride, blk_stride, idx,
                       const_cast<float *>(seq), L, K, CB, atomic);
    return (int)hipGetLastError();
}

// (rows, K * CB) -> (rows, CB): sum of the K column blocks (CB % 4 == 0), deterministic
extern "C" int mlagg_block_sum(const float *wide, float *out, long rows, int K, int CB, void *stream)
{
    if (!wide || !out) return MLAGG_E_NULLPTR;
    if (rows <= 0 || K <= 0 || CB <= 0 || (CB & 3) || rows * (CB >> 2) > 2147483647L * 256) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CROSS_MERGE, st);
    const long n = rows * (CB >> 2);
    hipLaunchKernelGGL(block_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wide, out, rows, K, CB);
    return (int)hipGetLastError();
}
