// fp32 products on the 16-bit matrix instructions: x = hi + mid + lo with three bf16 pieces (8 + 8 + 8 significand bits; the split
// is exact up to 2^-24 |x|), and a product is the six partial products of weight >= 2^-16 -- hi.hi, hi.mid, mid.hi, mid.mid,
// hi.lo, lo.hi.  A bf16 x bf16 product is exact in fp32 and the matrix core accumulates in fp32, so the result carries the
// error of an fp32 FMA chain (the dropped terms are 2^-24 of a product); six v_mfma_f32_32x32x16_bf16 stand for eight
// v_mfma_f32_32x32x2_f32 at 1/16 of their cost each.  Used by K5 (linear_lp.hip MODE 2) and K5w (linear_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

namespace bf16x3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// two fp32 -> one dword of two bf16 (round to nearest even), low half = first value
__device__ __forceinline__ unsigned pack2(float a, float b)
{
    const __hip_bfloat162 v = __float22bfloat162_rn(make_float2(a, b));
    return *reinterpret_cast<const unsigned *>(&v);
}

__device__ __forceinline__ void split3(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo)
{
    hi = pack2(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = pack2(ra, rb);
    lo = pack2(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}

__device__ __forceinline__ f32x16 mfma(const uint4 &a, const uint4 &b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0,
                                                   0);
}

// c += a . b for operands held as (hi, mid, lo) images of 8 consecutive k each; smallest terms first
__device__ __forceinline__ f32x16 mfma6(const uint4 (&a)[3], const uint4 (&b)[3], f32x16 c)
{
    c = mfma(a[2], b[0], c);
    c = mfma(a[0], b[2], c);
    c = mfma(a[1], b[1], c);
    c = mfma(a[1], b[0], c);
    c = mfma(a[0], b[1], c);
    return mfma(a[0], b[0], c);
}

}  // namespace bf16x3
