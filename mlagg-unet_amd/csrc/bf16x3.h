// fp32 products on the 16-bit matrix instructions: x = hi + mid + lo with three bf16 pieces (8 + 8 + 8 significand bits; the split
// is exact), and a product is the six partial products of weight >= 2^-16 -- hi.hi, hi.mid, mid.hi, mid.mid,
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

// (a, b) -> three dwords, each holding one bf16 piece of a (low half) and of b (high half).  Pieces by TRUNCATION: hi = the top 16
// bits of x, r = x - hi (exact), mid = the top 16 bits of r, lo = the top 16 bits of r - mid.  A 24-bit significand splits into
// 8 + 8 + 8 bits without remainder (truncation never borrows), so hi + mid + lo == x exactly for normal numbers.  Non-finite
// operands: a NaN stays a NaN; an infinity gives hi = inf and a NaN residual, i.e. the product is NaN where the fp32 instruction
// would return +-inf (both poison the step the same way; the fp32 train step has no scaler that relies on telling them apart).  11 full-rate
// instructions per pair (3 v_perm_b32, 4 v_and, 4 v_sub); rounding each piece with v_cvt_pk_bf16_f32 cost 18 (the compiler
// converts the two values separately and re-packs them with SDWA ors).
__device__ __forceinline__ void split3(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo)
{
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    mid = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

__device__ __forceinline__ f32x16 mfma(const uint4 &a, const uint4 &b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0,
                                                   0);
}

// The six partial products, smallest first, as (piece of a, piece of b).  Kernels issue them TERM-MAJOR over their accumulator
// tiles -- for each term, every tile -- so that consecutive MFMAs never depend on each other: tile-major (the six terms of a tile
// back to back) made every instruction wait for its predecessor's result (SQ_WAIT_INST_ANY 44 % of the wave cycles of K19).
__device__ constexpr int kTermA[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int kTermB[6] = {0, 2, 1, 0, 1, 0};

// acc[a][b] += A[a] . B[b] for NA x NB tiles, operands as (hi, mid, lo) images of 8 consecutive k each
template <int NA, int NB>
__device__ __forceinline__ void mfma_tiles(const uint4 (&A)[NA][3], const uint4 (&B)[NB][3], f32x16 (&acc)[NA][NB])
{
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = mfma(A[a][kTermA[term]], B[b][kTermB[term]], acc[a][b]);
}

}  // namespace bf16x3
