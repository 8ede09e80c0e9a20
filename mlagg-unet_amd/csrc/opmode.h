// Operand forms of a product on the 16-bit matrix instructions, selected by the MLAGG_DTYPE_* code DT of the call:
//   MLAGG_DTYPE_BF16X3   fp32 layers: every fp32 operand as three bf16 pieces, six partial products (bf16x3.h; fp32-accurate)
//   MLAGG_DTYPE_BF16     the reference's autocast(bf16) step (BASELINE configs[2]): operands rounded ONCE to bf16 (nearest even), one product
//   MLAGG_DTYPE_F16      its default autocast(fp16) + GradScaler step (nnUNetTrainer.py:848): the same with fp16 operands
// Sums are fp32 in every form; the maps stay fp32 in memory (the rounding happens in registers on the way to the matrix core), so the
// 16-bit modes run the SAME kernels as the fp32 step at a sixth of its matrix work and a third of its operand-preparation work -- no
// cast kernels, no 16-bit copies of the maps.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "mlagg_hip.h"
#include "bf16x3.h"

namespace opmode {

using bf16x3::f32x16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int DT>
struct Form {
    static_assert(DT == MLAGG_DTYPE_BF16 || DT == MLAGG_DTYPE_F16 || DT == MLAGG_DTYPE_BF16X3, "operand form");
    static constexpr int NQ = DT == MLAGG_DTYPE_BF16X3 ? 3 : 1;       // operand images (pieces)
    static constexpr int NT = DT == MLAGG_DTYPE_BF16X3 ? 6 : 1;       // partial products
    __device__ static constexpr int termA(int t) { return DT == MLAGG_DTYPE_BF16X3 ? bf16x3::kTermA[t] : 0; }
    __device__ static constexpr int termB(int t) { return DT == MLAGG_DTYPE_BF16X3 ? bf16x3::kTermB[t] : 0; }
};

inline bool valid(int dt) { return dt == MLAGG_DTYPE_BF16 || dt == MLAGG_DTYPE_F16 || dt == MLAGG_DTYPE_BF16X3; }

// (a, b) -> one dword per piece, a in the low half.  One-piece forms leave mid / lo untouched (never read: loops run over Form::NQ).
template <int DT>
__device__ __forceinline__ void split(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo)
{
    if constexpr (DT == MLAGG_DTYPE_BF16X3) {
        bf16x3::split3(a, b, hi, mid, lo);
    } else if constexpr (DT == MLAGG_DTYPE_BF16) {
        hi = bf16x3::pack2(a, b);
    } else {
        const __half2 v = __floats2half2_rn(a, b);
        hi = *reinterpret_cast<const unsigned *>(&v);
    }
}

template <int DT>
__device__ __forceinline__ void split8(const float (&f)[8], uint4 (&q)[3])
{
    split<DT>(f[0], f[1], q[0].x, q[1].x, q[2].x);
    split<DT>(f[2], f[3], q[0].y, q[1].y, q[2].y);
    split<DT>(f[4], f[5], q[0].z, q[1].z, q[2].z);
    split<DT>(f[6], f[7], q[0].w, q[1].w, q[2].w);
}

template <int DT>
__device__ __forceinline__ f32x16 mfma(const uint4 &a, const uint4 &b, f32x16 c)
{
    if constexpr (DT == MLAGG_DTYPE_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a), *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
    else
        return bf16x3::mfma(a, b, c);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// the 16 x 16 x 32 shape: lane l holds A[row l & 15][k = 8 (l >> 4) ..+7], B[k = 8 (l >> 4) ..+7][col l & 15]; D[row 4 (l >> 4) + r][col l & 15]
template <int DT>
__device__ __forceinline__ f32x4 mfma16(const uint4 &a, const uint4 &b, f32x4 c)
{
    if constexpr (DT == MLAGG_DTYPE_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8 *>(&a), *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x3::bf16x8 *>(&a),
                                                       *reinterpret_cast<const bf16x3::bf16x8 *>(&b), c, 0, 0, 0);
}

// acc[a][b] += A[a] . B[b], term-major (bf16x3.h: consecutive MFMAs never depend on each other)
template <int DT, int NA, int NB>
__device__ __forceinline__ void mfma_tiles(const uint4 (&A)[NA][3], const uint4 (&B)[NB][3], f32x16 (&acc)[NA][NB])
{
#pragma unroll
    for (int term = 0; term < Form<DT>::NT; ++term)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = mfma<DT>(A[a][Form<DT>::termA(term)], B[b][Form<DT>::termB(term)], acc[a][b]);
}

// one fp32 -> the 16-bit pattern of its piece q (weight images)
template <int DT>
__device__ __forceinline__ void pieces(float v, unsigned short (&p)[3])
{
    unsigned hi = 0, mid = 0, lo = 0;
    split<DT>(v, 0.f, hi, mid, lo);
    p[0] = (unsigned short)(hi & 0xffff);
    p[1] = (unsigned short)(mid & 0xffff);
    p[2] = (unsigned short)(lo & 0xffff);
}

}  // namespace opmode
