// 16-bit tensor I/O for the HBM-bound kernels of the convolutional chains (K10, K13) in the bf16 / fp16 modes: a map that only a
// 16-bit library convolution reads -- or that one wrote -- is kept in that type in memory (what the reference's autocast step holds
// there, nnUNetTrainer.py:848), arithmetic stays fp32 in registers.  The element type is a runtime code (uniform branch per access:
// these kernels wait on memory): MLAGG_DTYPE_F32 / _BF16 / _F16 of include/mlagg_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace mlagg_lpio {

__host__ __device__ inline int elem_bytes(int dt) { return dt == 0 ? 4 : 2; }

__device__ __forceinline__ float from16(unsigned short u, int dt)
{
    if (dt == 1) return __uint_as_float(((unsigned)u) << 16);
    return __half2float(*reinterpret_cast<const __half *>(&u));
}

__device__ __forceinline__ unsigned pack2(float a, float b, int dt)
{
    if (dt == 1) {
        const __hip_bfloat162 v = __float22bfloat162_rn(make_float2(a, b));
        return *reinterpret_cast<const unsigned *>(&v);
    }
    const __half2 v = __floats2half2_rn(a, b);
    return *reinterpret_cast<const unsigned *>(&v);
}

// 4 consecutive elements starting at element 4 * i4 of a plane whose first element is 8-byte (16-bit) / 16-byte (fp32) aligned
__device__ __forceinline__ float4 ld4(const void *base, long i4, int dt)
{
    if (dt == 0) return reinterpret_cast<const float4 *>(base)[i4];
    const uint2 w = reinterpret_cast<const uint2 *>(base)[i4];
    return make_float4(from16((unsigned short)(w.x & 0xffff), dt), from16((unsigned short)(w.x >> 16), dt),
                       from16((unsigned short)(w.y & 0xffff), dt), from16((unsigned short)(w.y >> 16), dt));
}

__device__ __forceinline__ void st4(void *base, long i4, int dt, const float4 &v)
{
    if (dt == 0) reinterpret_cast<float4 *>(base)[i4] = v;
    else reinterpret_cast<uint2 *>(base)[i4] = make_uint2(pack2(v.x, v.y, dt), pack2(v.z, v.w, dt));
}

__device__ __forceinline__ float ld1(const void *base, long i, int dt)
{
    if (dt == 0) return reinterpret_cast<const float *>(base)[i];
    return from16(reinterpret_cast<const unsigned short *>(base)[i], dt);
}

__device__ __forceinline__ void st1(void *base, long i, int dt, float v)
{
    if (dt == 0) reinterpret_cast<float *>(base)[i] = v;
    else reinterpret_cast<unsigned short *>(base)[i] = (unsigned short)(pack2(v, 0.f, dt) & 0xffff);
}

// plane `plane` of HW elements
__device__ __forceinline__ const void *plane_ptr(const void *base, long plane, long HW, int dt)
{
    return reinterpret_cast<const char *>(base) + plane * HW * elem_bytes(dt);
}
__device__ __forceinline__ void *plane_ptr(void *base, long plane, long HW, int dt)
{
    return reinterpret_cast<char *>(base) + plane * HW * elem_bytes(dt);
}

}  // namespace mlagg_lpio
