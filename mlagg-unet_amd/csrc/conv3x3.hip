// K19 -- dense 3 x 3 (and 3 x 3 x 3) convolutions (stride 1, zero padding 1) on channel-major (NCHW / NCDHW) maps as nine (27) shifted
// GEMMs on the 16-bit matrix instructions with fp32 accuracy (bf16x3.h):
//   y[b][o][p] = sum_t sum_i w[o][i][t] x[b][i][p + off_t],   off_t = (ty - 1) W + (tx - 1),  zero outside the image
// -- the forward of the convolutional stem / decoder blocks (nnUNetTrainer_MLAgg_2D_dt_MS.py:1340-1368 UnetrBasicBlock / UnetrUpBlock
// -> UnetResBlock conv1 / conv2, :972-1001 Project, MambaSkip.py:706-712 conv branches) and, on the transposed weight with the taps
// flipped, their data gradient.  MIOpen runs these as fp32 Winograd kernels at 64-104 TFLOP/s (2.8 ms forward + 2.5 ms data
// gradient per 256 x 256 step); six bf16 MFMAs per 16-deep block cost 0.375 of the eight fp32 ones.
//
// No padded copy and no layout change.  Forward / data gradient: D rows = output channels, D columns = pixels, contraction = input
// channel.  A lane owns TP consecutive pixels of ONE image row; per 16-channel block and kernel ROW (ky; (kz, ky) for 3 x 3 x 3
// volumes: nine rows) it loads that run and its two neighbours for eight channels, zeroes what is padding (a row flag and two edge
// flags made once per lane), splits these TP + 2 source elements ONCE into bf16 pieces and serves the row's three taps from them
// (tap kx, pixel j reads element j + kx).  The weights are pre-split once per launch by a tiny kernel into three bf16 images
// [tap][o][i] -- for the data gradient that kernel also transposes (o <-> i) and flips the taps -- and a 4-wave workgroup shares the
// three taps of the current kernel row in LDS (double-buffered).  Weight gradient: the pixel is the contraction (K18's weight
// gradient with nine shifted B operands); a wave owns 32 x 32 channels x 9 taps (one kernel slice kz for volumes); per 16-pixel
// block and kernel row it loads the aligned 8-pixel run of its x row and the two neighbours, splits the ten values once -- the
// pairs (p-1, p) .. (p+7, p+8) are the dx = -1 / +1 operands, the centre tap's pairs are one v_alignbit_b32 each.  Every load is
// unconditional (clamped address, masked value); partial blocks are summed in a fixed order (no atomics).  History of the forms
// and their measurements: DESIGN.md section 4i.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mlagg_hip.h"
#include "prof.h"
#include "bf16x3.h"
#include "opmode.h"

namespace {

using bf16x3::f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // native vector: HIP's uint4 (a struct) kept register arrays on the stack

template <int T>
struct __attribute__((packed, aligned(4))) FVec {
    float v[T];
};

struct C3Geom {
    int B, O, I, D, H, W, P;        // D = 1: 2-D (three kernel rows); D > 1: 3 x 3 x 3 (nine kernel rows (kz, ky))
    long x_batch, y_batch;
};

// wimg[q][t][o][i] (bf16 piece q of the weight of tap t); flip: the data gradient's weight, w'[i][o][t] = w[o][i][8 - t] with the
// roles of o and i exchanged (O, I are the OUTPUT / CONTRACTION extents of the product the image serves)
// (DT: operand form, opmode.h -- one image per piece: three for the fp32 layers, one rounded image in the 16-bit modes)
template <int DT>
__global__ void __launch_bounds__(256)
conv3x3_weight_image_kernel(const float *__restrict__ w, unsigned short *__restrict__ img, int O, int I, int ntaps, int flip)
{
    const int n = ntaps * O * I;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int i = idx % I, o = (idx / I) % O, t = idx / (I * O);
    const float v = flip ? w[((size_t)i * O + o) * ntaps + (ntaps - 1 - t)] : w[((size_t)o * I + i) * ntaps + t];
    unsigned short p[3];
    opmode::pieces<DT>(v, p);
#pragma unroll
    for (int q = 0; q < opmode::Form<DT>::NQ; ++q) img[(size_t)q * n + idx] = p[q];
}

#ifndef K19_INTERLEAVE
#define K19_INTERLEAVE 0
#endif
constexpr int WAVES = 4;            // pixel groups per workgroup: they share the weight stage in LDS

#ifndef K19_OCC3
#define K19_OCC3 1
#endif
// three workgroups per CU (three waves per SIMD, 168 registers) for the tiles that fit: the 64-row x 2-pixel tile sat at 169
template <int TO, int TP, int NR, int DT>
__global__ void __launch_bounds__(64 * WAVES, (K19_OCC3 && NR == 3 && TO <= 2 && TO * TP <= 4) ? 3 : 2)
conv3x3_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Wimg, const float *__restrict__ bias,
               float *__restrict__ Y, C3Geom g)
{
    // Weight stage: the three taps of one kernel row for one 16-channel block, all three bf16 pieces, the workgroup's 32 TO output
    // channels: [tap][piece][row][k half] x 16 bytes, double-buffered (2 x 9 x 32 TO x 32 B = 54 KB at TO = 3).  The four waves read
    // their A operands from it (one conflict-free ds_read_b128 per tile and piece) -- from global memory every wave fetched the same
    // 9 KB per tap, more than the x rows, and the first form of this kernel was bound by L1 bandwidth (its rate followed bytes per
    // MFMA across tile shapes: tools/bench_conv3x3.py with MLAGG_K19_TILE).
    constexpr int ROWS = 32 * TO;
    constexpr int NQ = opmode::Form<DT>::NQ, NT = opmode::Form<DT>::NT;       // pieces per operand, partial products (opmode.h)
    constexpr int STAGE = 3 * NQ * ROWS * 2;                // uint4 per stage
    constexpr int WL = (STAGE + 64 * WAVES - 1) / (64 * WAVES);
    __shared__ u32x4 sW[2][STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, kh = lane >> 5;
    const int p0 = (blockIdx.x * WAVES + wave) * (32 * TP), o0 = blockIdx.y * ROWS, b = blockIdx.z;
    f32x16 acc[TO][TP];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;
    // a pixel group past the end of the plane (grid rounding) works on the last run again and stores nothing
    const int pnat = p0 + TP * col, pc = max(min(pnat, g.P - TP), 0);
    // The lane's TP pixels lie in one image row (W % TP == 0, checked by the launcher).  Per kernel row ky the lane needs the source
    // pixels pc + (ky - 1) W - 1 .. + TP: its own run (one TP-float load per channel row, inside the plane whenever the source row
    // is) and the two neighbours (scalar loads).  They are split ONCE and serve the three taps of the row: tap kx, pixel j reads
    // source element j + kx.  Zero padding: a source row outside the image zeroes everything; the left neighbour is padding iff the
    // run starts the row, the right one iff it ends it -- and only then can their clamped addresses be displaced.
    const int x0 = pc % g.W, yz = pc / g.W, y0 = yz % g.H, z0 = yz / g.H;
    bool row_ok[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int kz = NR == 9 ? r / 3 : 1, ky = NR == 9 ? r % 3 : r;
        row_ok[r] = (unsigned)(z0 + kz - 1) < (unsigned)g.D && (unsigned)(y0 + ky - 1) < (unsigned)g.H;
    }
    const bool left_ok = x0 > 0, right_ok = x0 + TP < g.W;
    // addresses: BUFFER loads (round 4) -- resource = the sample (base X + b x_batch, 2 GB of range), scalar offset = the channel row
    // (16 blk + r) P (uniform), vector offset = the lane's 8 kh P + pixel part, fixed per kernel row.  Zero padding costs nothing in the
    // loop: a lane whose source element is padding (source row outside the image, the neighbour beyond the row's end) carries an
    // out-of-range vector offset for that kernel row, and the buffer load returns 0 for it.  (With global loads the loop spent 29 64-bit
    // address additions and 24 selects per stage beside the 132 instructions of the split: 5.2 vector instructions per MFMA, matrix pipe
    // 47 % busy on 96 -> 96 @ 128 x 128: profiles/round4_k_pmc_conv3x3_forward_96.md.)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + (size_t)b * g.x_batch), 0, 0x7fffffff, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned offc[NR], offl[NR], offr[NR];                                            // run / left / right neighbour per kernel row
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int kz = NR == 9 ? r / 3 : 1, ky = NR == 9 ? r % 3 : r;
        const int q = pc + (kz - 1) * g.H * g.W + (ky - 1) * g.W;
        offc[r] = row_ok[r] ? 4u * (unsigned)(8 * kh * g.P + q) : OOB;
        offl[r] = (row_ok[r] && left_ok) ? 4u * (unsigned)(8 * kh * g.P + q - 1) : OOB;
        offr[r] = (row_ok[r] && right_ok) ? 4u * (unsigned)(8 * kh * g.P + q + TP) : OOB;
    }
    const size_t img = (size_t)(3 * NR) * g.O * g.I;                                  // elements per weight image (piece)
    const size_t tstride = (size_t)g.O * g.I;
    const int nblk = g.I / 16, nstage = NR * nblk;
    // weight stage loader: element e = ((tt * NQ + q) * ROWS + row) * 2 + h  <-  Wimg[q][3 srow + tt][o0 + row][16 blk + 8 h ..]
    unsigned wsrc[WL];                                                               // bytes within a (blk, kernel row) slice
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int e = min(tid + 64 * WAVES * i, STAGE - 1);
        const int h = e & 1, row = (e >> 1) % ROWS, q = ((e >> 1) / ROWS) % NQ, tt = (e >> 1) / (NQ * ROWS);
        wsrc[i] = 2u * (unsigned)(q * img + tt * tstride + (size_t)min(o0 + row, g.O - 1) * g.I + 8 * h);
    }
    u32x4 wreg[WL];
    // stage s = 3 blk + kernel row (clamped: past the end the last stage again); macros, not lambdas (see u32x4)
#define K19_WFETCH(S)                                                                                                         \
    {                                                                                                                         \
        const int sc_ = min((S), nstage - 1), blk_ = sc_ / NR, srow_ = sc_ - NR * blk_;                                       \
        const char *base_ = reinterpret_cast<const char *>(Wimg + (size_t)(3 * srow_) * tstride + 16 * blk_);                \
        _Pragma("unroll") for (int i_ = 0; i_ < WL; ++i_) wreg[i_] = *reinterpret_cast<const u32x4 *>(base_ + (size_t)wsrc[i_]); \
    }
    // unconditional: the tail threads rewrite the last element with the same value (their load was clamped alike)
#define K19_WSTORE(BUF)                                                                                                       \
    {                                                                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < WL; ++i_) sW[(BUF)][min(tid + 64 * WAVES * i_, STAGE - 1)] = wreg[i_];        \
    }
    FVec<TP> xc[2][8];
    float xl[2][8], xr[2][8];
    // x rows of stage s (= iteration s): buffer s & 1; the fetch of s + 1 is in flight while s is consumed
    auto xfetch = [&](FVec<TP> (&C)[8], float (&L)[8], float (&R)[8], int blk_, int ky) __attribute__((always_inline)) {
        const int blk = min(blk_, nblk - 1);                    // past the end: the last block again, dropped
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned so = 4u * (unsigned)((16 * blk + r) * g.P);               // bytes, uniform
            if constexpr (TP == 1) {
                C[r].v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, offc[ky], so, 0));
            } else {
                static_assert(TP == 2, "pixel runs of one or two");
                // two dword loads (the second at instruction offset 4): this toolchain's __builtin_amdgcn_raw_buffer_load_b64 emits ONE
                // buffer_load_dword and copies it into both halves (ROCm 7.2 hipcc; found by a structured-input test, round 4)
                C[r].v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, offc[ky], so, 0));
                C[r].v[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, offc[ky] + 4u, so, 0));
            }
            L[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, offl[ky], so, 0));
            R[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, offr[ky], so, 0));
        }
    };
    auto consume = [&](const u32x4 *wst, const FVec<TP> (&C)[8], const float (&L)[8], const float (&R)[8], int ky)
                       __attribute__((always_inline)) {
        // source elements m = 0 .. TP + 1 (left neighbour, the run, right neighbour), three bf16 pieces each
        uint4 src[TP + 2][3];
#pragma unroll
        for (int m = 0; m < TP + 2; ++m) {
            float f[8];                                          // padding arrives as 0 (out-of-range buffer offsets)
#pragma unroll
            for (int r = 0; r < 8; ++r) f[r] = m == 0 ? L[r] : (m == TP + 1 ? R[r] : C[r].v[(m >= 1 && m <= TP) ? m - 1 : 0]);
            opmode::split8<DT>(f, src[m]);
        }
#if K19_INTERLEAVE
        // ask the scheduler for MFMA / VALU alternation over the stage: the split of the later source elements and the LDS reads of
        // the next tap's weights then issue in the matrix instructions' shadows instead of in a phase of their own
#pragma unroll
        for (int i = 0; i < 3 * NT * TO * TP; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, K19_INTERLEAVE, 0);   // VALU
        }
#endif
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {                         // tap (ky, tt): pixel j reads source element j + tt
            uint4 aq[TO][3];
#pragma unroll
            for (int a = 0; a < TO; ++a)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const u32x4 v = wst[(tt * NQ + q) * ROWS * 2 + (32 * a + col) * 2 + kh];
                    aq[a][q] = make_uint4(v.x, v.y, v.z, v.w);
                }
#pragma unroll
            for (int term = 0; term < NT; ++term)
#pragma unroll
                for (int a = 0; a < TO; ++a)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
                        acc[a][j] = opmode::mfma<DT>(aq[a][opmode::Form<DT>::termA(term)], src[j + tt][opmode::Form<DT>::termB(term)], acc[a][j]);
        }
    };
    // one stage: the three taps of a kernel row on weight buffer s & 1 while the next stage's weights travel global -> registers
    // -> LDS and the next stage's x rows are in flight
    auto stage = [&](int blk_, int ky, int par) __attribute__((always_inline)) {          // stage s = NR blk + ky, par = s & 1
        K19_WFETCH(NR * blk_ + ky + 1)
        xfetch(xc[par ^ 1], xl[par ^ 1], xr[par ^ 1], ky == NR - 1 ? blk_ + 1 : blk_, ky == NR - 1 ? 0 : ky + 1);
        consume(sW[par], xc[par], xl[par], xr[par], ky);
        K19_WSTORE(par ^ 1)
        __syncthreads();                                          // next stage's weights visible; this stage's buffer free for s + 2
    };
    K19_WFETCH(0)
    K19_WSTORE(0)
    xfetch(xc[0], xl[0], xr[0], 0, 0);
    __syncthreads();
    int blk = 0;
#pragma unroll 1
    for (; blk + 2 <= nblk; blk += 2) {                    // 2 NR stages (NR is odd): kernel row and buffer parity both repeat
#pragma unroll
        for (int r = 0; r < NR; ++r) stage(blk, r, r & 1);
#pragma unroll
        for (int r = 0; r < NR; ++r) stage(blk + 1, r, (r + 1) & 1);
    }
    if (blk < nblk) {
#pragma unroll
        for (int r = 0; r < NR; ++r) stage(blk, r, r & 1);
    }
    float *yb = Y + (size_t)b * g.y_batch + pc;
#pragma unroll
    for (int a = 0; a < TO; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (o >= g.O) continue;
            const float bv = bias ? bias[o] : 0.f;
            float *dst = yb + (size_t)o * g.P;
            if (pc == pnat) {
                FVec<TP> v;
#pragma unroll
                for (int j = 0; j < TP; ++j) v.v[j] = acc[a][j][r] + bv;
                *reinterpret_cast<FVec<TP> *>(dst) = v;
            } else {
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    if (pc + j >= pnat && pnat < g.P) dst[j] = acc[a][j][r] + bv;
            }
        }
    }
}

#undef K19_WFETCH
#undef K19_WSTORE

// ---- weight gradient: dW[o][i][t] = sum_b sum_p dy[b][o][p] x[b][i][p + off_t]  (x = 0 outside the image) ----------------------
// The pixel is the contraction and it is contiguous in both operands (K18's weight gradient with nine shifted B operands): a wave
// owns 32 output x 32 input channels x 9 taps (144 accumulators) and a slab of pixels of one sample; per 16-pixel block a lane reads
// 8 consecutive pixels of its dy row once (A) and, per tap, 8 consecutive pixels of its x row shifted by the tap (B: two 16-byte
// loads at a 4-byte-aligned address; the nine taps of a block overlap in L1).  W % 16 == 0, so a 16-pixel block lies in ONE image
// row: the row test of a tap is wave-uniform (a kernel row whose source row is outside the image is skipped: no loads sit under
// that branch, they were issued unconditionally with a clamped address), and the only per-element cases are the first pixel of a
// row for dx = -1 and the last one for dx = +1.  Partials [slab][tap][o][i] (coalesced stores), summed in a fixed order into the
// (O, I, 3, 3) layout by a second launch.
struct W3Geom {
    int B, O, I, D, H, W, P;        // D = 1: 3 x 3 (nz = 1 kernel slice); D > 1: 3 x 3 x 3 (nz = 3: one wave per kernel slice kz)
    int nz;
    long dy_batch, x_batch;
    int slab, nslabs;
    long x_last;                    // last float offset from x at which an 8-float load stays inside the operand
};

template <int DT>
__global__ void __launch_bounds__(64)
conv3x3_wgrad_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, W3Geom g)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    const int kz = (int)(blockIdx.x % g.nz) + (g.nz == 1 ? 1 : 0);          // kernel slice of this wave (2-D: the middle one)
    const int bs = blockIdx.x / g.nz, b = bs / g.nslabs, s = bs % g.nslabs;
    const int o0 = blockIdx.y * 32, i0 = blockIdx.z * 32;
    f32x16 acc[1][9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][t][r] = 0.f;
    const int pb = s * g.slab, pe = min(pb + g.slab, g.P);
    const int nblk = (pe - pb) / 16;
    const int oc = min(o0 + col, g.O - 1), ic = min(i0 + col, g.I - 1);
    const float *ap = dY + (size_t)b * g.dy_batch + (size_t)oc * g.P + pb + 8 * kh;          // + 16 blk
    const long xrow = (long)b * g.x_batch + (long)ic * g.P + pb + 8 * kh;                      // float offset of the lane's run, + 16 blk
    // iteration = (blk, kernel row): A (row 0 only) + the ten pixels p - 1 .. p + 8 of the source row (the aligned run as two
    // 16-byte loads + its two neighbours); double-buffered.  The ten values are split ONCE into five dwords per bf16 piece --
    // pairs (p-1, p), (p+1, p+2) .. (p+7, p+8) -- which ARE the operands of the taps dx = -1 (dwords 0-3) and dx = +1 (dwords 1-4);
    // the centre tap's pairs (p, p+1) .. are one v_alignbit_b32 each.  (Splitting every tap's run separately made the kernel
    // VALU-bound: 680 instead of 250 vector instructions per block, 45-85 TFLOP/s.)
    float4 av[2][2], bv[2][2];
    float be[2][2];                                             // x[p - 1], x[p + 8]
    auto fetch = [&](float4 (&A)[2], float4 (&Bv)[2], float (&E)[2], int blk, int ky) __attribute__((always_inline)) {
        const int bc = min(blk, nblk - 1);
        if (ky == 0) {
            A[0] = *reinterpret_cast<const float4 *>(ap + 16 * bc);
            A[1] = *reinterpret_cast<const float4 *>(ap + 16 * bc + 4);
        }
        // clamped into the operand: a displaced address only occurs for a source row outside the image (skipped) or for the two
        // neighbours at the operand's very first / last float, which are the zero-padded ones
        const long want = xrow + 16 * bc + (long)(ky - 1) * g.W + (long)(kz - 1) * g.H * g.W;
        const long at = min(max(want, 0L), g.x_last);
        Bv[0] = *reinterpret_cast<const float4 *>(X + at);
        Bv[1] = *reinterpret_cast<const float4 *>(X + at + 4);
        E[0] = X[min(max(want - 1, 0L), g.x_last + 7)];
        E[1] = X[min(max(want + 8, 0L), g.x_last + 7)];
    };
    uint4 aq[1][3];
    auto consume = [&](const float4 (&A)[2], const float4 (&Bv)[2], const float (&E)[2], int blk, int ky) __attribute__((always_inline)) {
        if (ky == 0) {
            const float f[8] = {A[0].x, A[0].y, A[0].z, A[0].w, A[1].x, A[1].y, A[1].z, A[1].w};
            opmode::split8<DT>(f, aq[0]);
        }
        // the lane's 8-pixel run lies in one image row (W % 8 == 0); its two halves of a block may lie in different rows
        const int p = pb + 16 * blk + 8 * kh;
        const int x0 = p % g.W, yz = p / g.W, y = yz % g.H, z = yz / g.H;
        const bool rok = (unsigned)(y + ky - 1) < (unsigned)g.H && (unsigned)(z + kz - 1) < (unsigned)g.D;
        if (!__any(rok)) return;                                 // source rows of the whole block outside the image
        const float left = (!rok || x0 == 0) ? 0.f : E[0];                                  // left neighbour of the row's first pixel
        const float right = (!rok || x0 + 8 == g.W) ? 0.f : E[1];                           // right neighbour of its last pixel
        float4 c0 = Bv[0], c1 = Bv[1];
        if (!rok) c0 = c1 = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned d[3][5];                                        // [piece][pair]: (p-1, p), (p+1, p+2), (p+3, p+4), (p+5, p+6), (p+7, p+8)
        opmode::split<DT>(left, c0.x, d[0][0], d[1][0], d[2][0]);
        opmode::split<DT>(c0.y, c0.z, d[0][1], d[1][1], d[2][1]);
        opmode::split<DT>(c0.w, c1.x, d[0][2], d[1][2], d[2][2]);
        opmode::split<DT>(c1.y, c1.z, d[0][3], d[1][3], d[2][3]);
        opmode::split<DT>(c1.w, right, d[0][4], d[1][4], d[2][4]);
        uint4 bq[3][3];
#pragma unroll
        for (int q = 0; q < opmode::Form<DT>::NQ; ++q) {
            bq[0][q] = make_uint4(d[q][0], d[q][1], d[q][2], d[q][3]);                                          // dx = -1
            bq[2][q] = make_uint4(d[q][1], d[q][2], d[q][3], d[q][4]);                                          // dx = +1
            bq[1][q] = make_uint4(__builtin_amdgcn_alignbit(d[q][1], d[q][0], 16), __builtin_amdgcn_alignbit(d[q][2], d[q][1], 16),
                                  __builtin_amdgcn_alignbit(d[q][3], d[q][2], 16), __builtin_amdgcn_alignbit(d[q][4], d[q][3], 16));
        }
#pragma unroll
        for (int term = 0; term < opmode::Form<DT>::NT; ++term)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                acc[0][3 * ky + kx] = opmode::mfma<DT>(aq[0][opmode::Form<DT>::termA(term)], bq[kx][opmode::Form<DT>::termB(term)],
                                                       acc[0][3 * ky + kx]);
    };
    if (nblk > 0) {
        fetch(av[0], bv[0], be[0], 0, 0);
#pragma unroll 1
        for (int blk = 0; blk < nblk; blk += 2) {              // six iterations: buffer parity repeats every two blocks
            fetch(av[1], bv[1], be[1], blk, 1);
            consume(av[0], bv[0], be[0], blk, 0);
            fetch(av[0], bv[0], be[0], blk, 2);
            consume(av[0], bv[1], be[1], blk, 1);
            fetch(av[1], bv[1], be[1], blk + 1, 0);
            consume(av[0], bv[0], be[0], blk, 2);
            if (blk + 1 < nblk) {
                fetch(av[0], bv[0], be[0], blk + 1, 1);
                consume(av[1], bv[1], be[1], blk + 1, 0);
                fetch(av[1], bv[1], be[1], blk + 1, 2);
                consume(av[1], bv[0], be[0], blk + 1, 1);
                fetch(av[0], bv[0], be[0], blk + 2, 0);
                consume(av[1], bv[1], be[1], blk + 1, 2);
            }
        }
    }
    // partial [slab][tap][O][I]: D row R -> output channel o0 + R, column = lane -> input channel i0 + col
    const int ntaps = 9 * g.nz, t0 = g.nz == 1 ? 0 : 9 * kz;
    float *prow = part + (size_t)bs * ((size_t)ntaps * g.O * g.I);
    const int i = i0 + col;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (o < g.O && i < g.I) prow[((size_t)(t0 + t) * g.O + o) * g.I + i] = acc[0][t][r];
        }
}

// dW[(o, i, t)] = sum over partial blocks of part[s][t][o][i], fixed order.  Workgroup = 64 elements x 16 groups of partial blocks
// (coalesced 256-byte reads, 16 partial blocks in flight per element, LDS combine): one thread per element walked its hundreds of
// partial blocks serially and cost 40-130 us per layer (profiles/round4_h_*).
__global__ void __launch_bounds__(1024)
conv3x3_wgrad_reduce_kernel(const float *__restrict__ part, int nparts, int O, int I, int ntaps, float *__restrict__ dW)
{
    __shared__ float red[16][65];
    const int n = ntaps * O * I;
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cx;                         // (t, o, i): the partial layout
    float s0 = 0.f, s1 = 0.f;
    if (idx < n) {
        int r = rg;
        for (; r + 16 < nparts; r += 32) {
            s0 += part[(size_t)r * n + idx];
            s1 += part[(size_t)(r + 16) * n + idx];
        }
        if (r < nparts) s0 += part[(size_t)r * n + idx];
    }
    red[rg][cx] = s0 + s1;
    __syncthreads();
    if (rg == 0 && idx < n) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][cx];
        const int i = idx % I, o = (idx / I) % O, t = idx / (I * O);
        dW[((size_t)o * I + i) * ntaps + t] = v;
    }
}

// ---- weight gradient, second form (round 4): 16 x 16 x 32 matrix instructions, a wave owns 16 TO output x 16 TI input channels x the
// THREE taps of one kernel row ky (TO = TI = 3: 48 x 48 channels -- every dense 3 x 3 layer of the network has 48 k channels -- in 108
// accumulator registers, two waves per SIMD) and a slab of pixels of one sample; the three kernel rows of a slab are three
// single-wave workgroups that the grid mapping puts on ONE XCD next to each other in its dispatch order (they read the same dy rows
// and neighbouring x rows: one L2).  Per 32-pixel block a lane reads 8 consecutive pixels of its dy rows and of its x rows in image
// row y + ky - 1 plus their two neighbours, splits them once, and the dx = -1 / 0 / +1 operands are dword windows of the same five
// pairs (see the first form).  The first form (32 x 32 tiles, nine taps per wave: 144 accumulators + 167 registers = one wave per
// SIMD with a one-step prefetch) ran at 24 % of the matrix rate on the step's shapes, bound by memory latency, and filled 56 % of its
// tiles on 48 x 48 channels.
struct W16Geom {
    int B, O, I, H, W, P;
    long dy_batch, x_batch;
    int slab, nslabs, nbs;          // pixels per slab (multiple of 32), slabs per sample, B * nslabs
    long x_last;                    // last float offset from x at which an 8-float load stays inside the operand
    int q32, r32;                   // 32 = q32 * W + r32
};

template <int TO, int TI, int DT>
__global__ void __launch_bounds__(64)
conv3x3_wgrad16_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, W16Geom g)
{
    using opmode::f32x4;
    constexpr int NQ = opmode::Form<DT>::NQ, NT = opmode::Form<DT>::NT;
    const int lane = threadIdx.x, r16 = lane & 15, kg = lane >> 4;
    // workgroup id -> (XCD, its dispatch slot): slots 3 n .. 3 n + 2 of an XCD are the kernel rows of slab 8 n + xcd
    const int id = blockIdx.x, slot = id >> 3;
    const int ky = slot % 3, bs = (slot / 3) * 8 + (id & 7);
    if (bs >= g.nbs) return;
    const int b = bs / g.nslabs, s = bs % g.nslabs;
    const int o0 = blockIdx.y * (16 * TO), i0 = blockIdx.z * (16 * TI);
    f32x4 acc[TO][TI][3];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[a][j][kx] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pb = s * g.slab, pe = min(pb + g.slab, g.P);
    const int nblk = (pe - pb + 31) / 32;
    const float *ap[TO];
    long xrow[TI];
#pragma unroll
    for (int a = 0; a < TO; ++a) ap[a] = dY + (size_t)b * g.dy_batch + (size_t)min(o0 + 16 * a + r16, g.O - 1) * g.P;
#pragma unroll
    for (int j = 0; j < TI; ++j) xrow[j] = (long)b * g.x_batch + (long)min(i0 + 16 * j + r16, g.I - 1) * g.P + (long)(ky - 1) * g.W;
    struct Raw {
        float4 a[TO][2], b[TI][2];
        float e[TI][2];
    };
    // every load is unconditional: clamped into the operand; a displaced address only occurs where the value is masked (tail pixels
    // of the slab, a source row outside the image, the neighbours across an image edge)
    auto fetch = [&](Raw &R, int blk) __attribute__((always_inline)) {
        const int p = min(pb + 32 * min(blk, nblk - 1) + 8 * kg, g.P - 8);
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            R.a[a][0] = *reinterpret_cast<const float4 *>(ap[a] + p);
            R.a[a][1] = *reinterpret_cast<const float4 *>(ap[a] + p + 4);
        }
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const long want = xrow[j] + p;
            const long at = min(max(want, 0L), g.x_last);
            R.b[j][0] = *reinterpret_cast<const float4 *>(X + at);
            R.b[j][1] = *reinterpret_cast<const float4 *>(X + at + 4);
            R.e[j][0] = X[min(max(want - 1, 0L), g.x_last + 7)];
            R.e[j][1] = X[min(max(want + 8, 0L), g.x_last + 7)];
        }
    };
    // position of the lane's run in block 0; advanced by 32 pixels per block without divisions
    int x0 = (pb + 8 * kg) % g.W, y = (pb + 8 * kg) / g.W, pl = pb + 8 * kg;
    auto consume = [&](const Raw &R) __attribute__((always_inline)) {
        const bool live = pl < pe;                               // tail of the slab: the dy values are dropped
        const bool rok = (unsigned)(y + ky - 1) < (unsigned)g.H;
        if (__any(live && rok)) {
            uint4 aq[TO][3];
#pragma unroll
            for (int a = 0; a < TO; ++a) {
                float f[8] = {R.a[a][0].x, R.a[a][0].y, R.a[a][0].z, R.a[a][0].w, R.a[a][1].x, R.a[a][1].y, R.a[a][1].z, R.a[a][1].w};
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = live ? f[k] : 0.f;
                opmode::split8<DT>(f, aq[a]);
            }
#pragma unroll
            for (int j = 0; j < TI; ++j) {
                float4 c0 = R.b[j][0], c1 = R.b[j][1];
                if (!rok) c0 = c1 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float left = (!rok || x0 == 0) ? 0.f : R.e[j][0];
                const float right = (!rok || x0 + 8 == g.W) ? 0.f : R.e[j][1];
                unsigned d[3][5];                                // [piece][pair]: (p-1, p), (p+1, p+2), (p+3, p+4), (p+5, p+6), (p+7, p+8)
                opmode::split<DT>(left, c0.x, d[0][0], d[1][0], d[2][0]);
                opmode::split<DT>(c0.y, c0.z, d[0][1], d[1][1], d[2][1]);
                opmode::split<DT>(c0.w, c1.x, d[0][2], d[1][2], d[2][2]);
                opmode::split<DT>(c1.y, c1.z, d[0][3], d[1][3], d[2][3]);
                opmode::split<DT>(c1.w, right, d[0][4], d[1][4], d[2][4]);
                uint4 bq[3][3];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    bq[0][q] = make_uint4(d[q][0], d[q][1], d[q][2], d[q][3]);                                          // dx = -1
                    bq[2][q] = make_uint4(d[q][1], d[q][2], d[q][3], d[q][4]);                                          // dx = +1
                    bq[1][q] = make_uint4(__builtin_amdgcn_alignbit(d[q][1], d[q][0], 16), __builtin_amdgcn_alignbit(d[q][2], d[q][1], 16),
                                          __builtin_amdgcn_alignbit(d[q][3], d[q][2], 16), __builtin_amdgcn_alignbit(d[q][4], d[q][3], 16));
                }
#pragma unroll
                for (int term = 0; term < NT; ++term)
#pragma unroll
                    for (int a = 0; a < TO; ++a)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[a][j][kx] = opmode::mfma16<DT>(aq[a][opmode::Form<DT>::termA(term)], bq[kx][opmode::Form<DT>::termB(term)],
                                                               acc[a][j][kx]);
            }
        }
        pl += 32;
        x0 += g.r32;
        y += g.q32;
        if (x0 >= g.W) {
            x0 -= g.W;
            ++y;
        }
    };
    Raw r0, r1;
    fetch(r0, 0);
    int blk = 0;
#pragma unroll 1
    for (; blk + 1 < nblk; blk += 2) {                            // pairs; the odd last block is peeled: a conditional consume inside the
        fetch(r1, blk + 1);                                      // loop made the compiler shuffle all 108 accumulators per iteration
        consume(r0);
        fetch(r0, blk + 2);
        consume(r1);
    }
    if (blk < nblk) consume(r0);
    // partial [bs][tap][O][I]: D row 4 kg + r -> output channel, column r16 -> input channel (64-byte runs)
    float *prow = part + (size_t)bs * ((size_t)9 * g.O * g.I) + (size_t)(3 * ky) * g.O * g.I;
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int i = i0 + 16 * j + r16;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = o0 + 16 * a + 4 * kg + r;
                    if (o < g.O && i < g.I) prow[((size_t)kx * g.O + o) * g.I + i] = acc[a][j][kx][r];
                }
        }
}

// ---- the same tile with the operands staged through LDS in FULL LINES (round 4, second step).  The fragment-shaped loads of the
// kernel above -- 16 bytes per lane, neighbouring lanes in different channel rows -- cost the texture addresser one access per LANE
// (65 L1 accesses per load instruction measured, 1150 addresser cycles per block and wave against 2600 matrix cycles, four waves per
// CU: addresser-bound at 47 % matrix utilisation).  Here a load instruction covers 8 channel rows x 128 bytes (dy) or 6.4 rows x
// 160 bytes (x: the 32 pixels of the block and 4 on either side, so the two neighbours of every run come with the same lines), 8
// lanes per row and line; the tile goes registers -> LDS (row pitches 36 / 44 floats: the 16 rows of a fragment read hit 64 distinct
// banks) and the matrix operands are read back as ds_read_b128 fragments.  One wave per workgroup: no barrier, the wave's LDS
// operations complete in order.  Stage s + 1 is in flight (registers) while stage s is consumed from LDS.
template <int TO, int TI, int DT>
__global__ void __launch_bounds__(64)
conv3x3_wgrad16_lds_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, W16Geom g, long dy_last)
{
    using opmode::f32x4;
    constexpr int NQ = opmode::Form<DT>::NQ, NT = opmode::Form<DT>::NT;
    constexpr int AP = 36, BP = 44;                              // LDS row pitches, floats
    constexpr int AROWS = 16 * TO, BROWS = 16 * TI;
    constexpr int NLA = AROWS / 8;                               // dy: 8 rows x 8 chunks of 16 bytes per load instruction
    constexpr int NLB = (BROWS * 10 + 63) / 64;                  // x: 10 chunks per row
    __shared__ float sA[2][AROWS * AP];
    __shared__ float sB[2][BROWS * BP];
    const int lane = threadIdx.x, r16 = lane & 15, kg = lane >> 4;
    const int id = blockIdx.x, slot = id >> 3;
    const int ky = slot % 3, bs = (slot / 3) * 8 + (id & 7);
    if (bs >= g.nbs) return;
    const int b = bs / g.nslabs, s = bs % g.nslabs;
    const int o0 = blockIdx.y * (16 * TO), i0 = blockIdx.z * (16 * TI);
    f32x4 acc[TO][TI][3];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[a][j][kx] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pb = s * g.slab, pe = min(pb + g.slab, g.P);
    const int nblk = (pe - pb + 31) / 32;
    // load lanes: float offsets relative to the block's uniform base (row * P + 4 * chunk), LDS float offsets of the chunk
    int offA[NLA], offB[NLB];
    int ldsA[NLA], ldsB[NLB];
#pragma unroll
    for (int m = 0; m < NLA; ++m) {
        const int row = 8 * m + (lane >> 3), c = lane & 7;
        offA[m] = min(o0 + row, g.O - 1) * g.P + 4 * c;
        ldsA[m] = row * AP + 4 * c;
    }
#pragma unroll
    for (int m = 0; m < NLB; ++m) {
        const int idx = 64 * m + lane, row = min(idx / 10, BROWS - 1), c = idx % 10;      // past the tile: the last row again (same bytes)
        offB[m] = min(i0 + row, g.I - 1) * g.P + 4 * c;
        ldsB[m] = row * BP + 4 * c;
    }
    const long baseA = (long)b * g.dy_batch + pb;                                          // float index of (row 0, pixel pb)
    const long baseB = (long)b * g.x_batch + pb + (long)(ky - 1) * g.W - 4;                // ... of the x window's first float
    opmode::f32x4 ga[NLA], gb[NLB];                              // native vectors: arrays of HIP's float4 struct went to scratch
    // every load is unconditional; a lane's offset is clamped into the operand (uniform bounds per block, 32-bit per-lane arithmetic):
    // a displaced chunk only holds values that are masked (tail pixels, a source row outside the image, neighbours across an edge)
    auto fetch = [&](int blk) __attribute__((always_inline)) {
        const long ua = baseA + 32L * min(blk, nblk - 1), ub = baseB + 32L * min(blk, nblk - 1);
        const int hiA = (int)min(dy_last - ua, 0x7fffffffL);
        const int loB = (int)min(max(-ub, 0L), 0x7fffffffL), hiB = (int)min(g.x_last + 4 - ub, 0x7fffffffL);
        const float *pa = dY + ua, *pbx = X + ub;
#pragma unroll
        for (int m = 0; m < NLA; ++m) ga[m] = *reinterpret_cast<const opmode::f32x4 *>(pa + min(offA[m], hiA));
#pragma unroll
        for (int m = 0; m < NLB; ++m) gb[m] = *reinterpret_cast<const opmode::f32x4 *>(pbx + min(max(offB[m], loB), hiB));
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < NLA; ++m) *reinterpret_cast<opmode::f32x4 *>(&sA[buf][ldsA[m]]) = ga[m];
#pragma unroll
        for (int m = 0; m < NLB; ++m) *reinterpret_cast<opmode::f32x4 *>(&sB[buf][ldsB[m]]) = gb[m];
    };
    int x0 = (pb + 8 * kg) % g.W, y = (pb + 8 * kg) / g.W, pl = pb + 8 * kg;
    auto consume = [&](int buf) __attribute__((always_inline)) {
        const bool live = pl < pe;
        const bool rok = (unsigned)(y + ky - 1) < (unsigned)g.H;
        if (__any(live && rok)) {
            uint4 aq[TO][3];
#pragma unroll
            for (int a = 0; a < TO; ++a) {
                const float *src = &sA[buf][(16 * a + r16) * AP + 8 * kg];
                const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
                float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = live ? f[k] : 0.f;
                opmode::split8<DT>(f, aq[a]);
            }
#pragma unroll
            for (int j = 0; j < TI; ++j) {
                const float *src = &sB[buf][(16 * j + r16) * BP + 8 * kg];        // window float 0 = pixel p - 4
                float4 c0 = *reinterpret_cast<const float4 *>(src + 4), c1 = *reinterpret_cast<const float4 *>(src + 8);
                if (!rok) c0 = c1 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float left = (!rok || x0 == 0) ? 0.f : src[3];
                const float right = (!rok || x0 + 8 == g.W) ? 0.f : src[12];
                unsigned d[3][5];
                opmode::split<DT>(left, c0.x, d[0][0], d[1][0], d[2][0]);
                opmode::split<DT>(c0.y, c0.z, d[0][1], d[1][1], d[2][1]);
                opmode::split<DT>(c0.w, c1.x, d[0][2], d[1][2], d[2][2]);
                opmode::split<DT>(c1.y, c1.z, d[0][3], d[1][3], d[2][3]);
                opmode::split<DT>(c1.w, right, d[0][4], d[1][4], d[2][4]);
                uint4 bq[3][3];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    bq[0][q] = make_uint4(d[q][0], d[q][1], d[q][2], d[q][3]);
                    bq[2][q] = make_uint4(d[q][1], d[q][2], d[q][3], d[q][4]);
                    bq[1][q] = make_uint4(__builtin_amdgcn_alignbit(d[q][1], d[q][0], 16), __builtin_amdgcn_alignbit(d[q][2], d[q][1], 16),
                                          __builtin_amdgcn_alignbit(d[q][3], d[q][2], 16), __builtin_amdgcn_alignbit(d[q][4], d[q][3], 16));
                }
#pragma unroll
                for (int term = 0; term < NT; ++term)
#pragma unroll
                    for (int a = 0; a < TO; ++a)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[a][j][kx] = opmode::mfma16<DT>(aq[a][opmode::Form<DT>::termA(term)], bq[kx][opmode::Form<DT>::termB(term)],
                                                               acc[a][j][kx]);
            }
        }
        pl += 32;
        x0 += g.r32;
        y += g.q32;
        if (x0 >= g.W) {
            x0 -= g.W;
            ++y;
        }
    };
    fetch(0);
    stage(0);
    fetch(1);
    int blk = 0;
#pragma unroll 1
    for (; blk + 1 < nblk; blk += 2) {                            // pairs (buffer parity is static); the odd last block is peeled
        consume(0);                                              // block blk; block blk + 1 is in flight
        stage(1);
        fetch(blk + 2);
        consume(1);
        stage(0);
        fetch(blk + 3);
    }
    if (blk < nblk) consume(0);
    float *prow = part + (size_t)bs * ((size_t)9 * g.O * g.I) + (size_t)(3 * ky) * g.O * g.I;
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int i = i0 + 16 * j + r16;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = o0 + 16 * a + 4 * kg + r;
                    if (o < g.O && i < g.I) prow[((size_t)kx * g.O + o) * g.I + i] = acc[a][j][kx][r];
                }
        }
}

int make_w16geom(W16Geom &g, int B, int O, int I, int H, int W, long dy_batch, long x_batch, int to, int ti)
{
    if (B <= 0 || O <= 0 || I <= 0 || H <= 0 || W <= 0 || (W & 7)) return MLAGG_E_UNSUPPORTED;
    const long P = (long)H * W;
    if (P >= (1L << 28) || dy_batch < (long)O * P || x_batch < (long)I * P || ((dy_batch | x_batch) & 3)) return MLAGG_E_UNSUPPORTED;
    g = W16Geom{B, O, I, H, W, (int)P, dy_batch, x_batch, 0, 0, 0, (long)(B - 1) * x_batch + (long)I * P - 8, 32 / W, 32 % W};
    const int og = (O + 16 * to - 1) / (16 * to), ig = (I + 16 * ti - 1) / (16 * ti);
    // two waves per SIMD over (sample, slab, kernel row, channel blocks); at least eight 32-pixel blocks per slab
    static const int target = [] { const char *e = getenv("MLAGG_K19W16_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2048; }();
    int per_sample = (target + B * og * ig * 3 - 1) / (B * og * ig * 3);
    const long cap = (256L << 20) / (4L * 9 * O * I) / B;   // partial blocks: under 256 MB in total
    if (per_sample > cap) per_sample = (int)cap;
    if (per_sample < 1) per_sample = 1;
    int slab = (int)((P + per_sample - 1) / per_sample);
    slab = ((slab + 31) / 32) * 32;
    if (slab < 256) slab = 256;
    g.slab = slab;
    g.nslabs = (int)((P + slab - 1) / slab);
    if ((long)B * g.nslabs > (1L << 26) || og > 65535 || ig > 65535) return MLAGG_E_UNSUPPORTED;
    g.nbs = B * g.nslabs;
    return 0;
}

// which form serves a 2-D layer: the 16-wide tiles unless a channel extent is small enough that the 32 x 32 x nine-tap form wastes less
int w16_tiles(int O, int I, int dt, int &to, int &ti)
{
    // MLAGG_K19W16: 0 = the 32 x 32 form, 1 = fragment-shaped loads, 2 = LDS-staged lines, unset = by operand form.  Measured on the step's
    // shapes (profiles/round4_h_conv3x3_wgrad_forms.log): the six-product form is bound by vector-instruction issue (operand
    // splitting) and loses 5-10 % to the staging instructions; the one-product forms are bound by the loads and gain 20-35 % from them.
    static const int mode = [] { const char *e = getenv("MLAGG_K19W16"); return e ? atoi(e) : -1; }();
    if (!mode) return 0;
    to = 3;
    ti = I <= 16 ? 1 : 3;
    return mode < 0 ? (dt == MLAGG_DTYPE_BF16X3 ? 1 : 2) : mode;
}

int make_w3geom(W3Geom &g, int B, int O, int I, int D, int H, int W, long dy_batch, long x_batch)
{
    if (B <= 0 || O <= 0 || I <= 0 || D <= 0 || H <= 0 || W <= 0 || (W & 7)) return MLAGG_E_UNSUPPORTED;
    const long P = (long)D * H * W;
    if ((P & 15) || P >= (1L << 28) || dy_batch < (long)O * P || x_batch < (long)I * P || ((dy_batch | x_batch) & 3)) return MLAGG_E_UNSUPPORTED;
    g = W3Geom{B, O, I, D, H, W, (int)P, D > 1 ? 3 : 1, dy_batch, x_batch, 0, 0, (long)(B - 1) * x_batch + (long)I * P - 8};
    const int og = (O + 31) / 32, ig = (I + 31) / 32;
    static const int target = [] { const char *e = getenv("MLAGG_K19W_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2048; }();
    int per_sample = (target + B * og * ig * g.nz - 1) / (B * og * ig * g.nz);
    // partial blocks are (taps x O x I) floats each: keep their total under 256 MB (wide layers have many channel tiles per slab anyway)
    const long cap = (256L << 20) / (4L * 9 * g.nz * O * I) / B;
    if (per_sample > cap) per_sample = (int)cap;
    if (per_sample < 1) per_sample = 1;
    int slab = (int)((P + per_sample - 1) / per_sample);
    slab = ((slab + 15) / 16) * 16;
    if (slab < 64) slab = 64;
    g.slab = slab;
    g.nslabs = (int)((P + slab - 1) / slab);
    if ((long)B * g.nslabs > 2147483647L || og > 65535 || ig > 65535) return MLAGG_E_UNSUPPORTED;
    return 0;
}


template <int TO, int TP>
void launch(const float *x, const unsigned short *wimg, const float *bias, float *y, const C3Geom &g, int dt, hipStream_t st)
{
    const int groups = (g.P + 32 * TP - 1) / (32 * TP);
    const dim3 grid((groups + WAVES - 1) / WAVES, (g.O + 32 * TO - 1) / (32 * TO), g.B);
    if (g.D > 1)                                            // volumes: the fp32 form only (conv_fwd checks)
        hipLaunchKernelGGL((conv3x3_kernel<TO, TP, 9, MLAGG_DTYPE_BF16X3>), grid, dim3(64 * WAVES), 0, st, x, wimg, bias, y, g);
    else if (dt == MLAGG_DTYPE_BF16)
        hipLaunchKernelGGL((conv3x3_kernel<TO, TP, 3, MLAGG_DTYPE_BF16>), grid, dim3(64 * WAVES), 0, st, x, wimg, bias, y, g);
    else if (dt == MLAGG_DTYPE_F16)
        hipLaunchKernelGGL((conv3x3_kernel<TO, TP, 3, MLAGG_DTYPE_F16>), grid, dim3(64 * WAVES), 0, st, x, wimg, bias, y, g);
    else
        hipLaunchKernelGGL((conv3x3_kernel<TO, TP, 3, MLAGG_DTYPE_BF16X3>), grid, dim3(64 * WAVES), 0, st, x, wimg, bias, y, g);
}

}  // namespace

namespace {
// y = conv(x, w) for a 3 x 3 (D == 1) or 3 x 3 x 3 kernel, stride 1, zero padding 1
// the kernel's per-lane byte offsets are 32-bit: 4 * (8 * P + q) with q < P, i.e. up to 36 P bytes -- planes (volumes) of up to
// 2^32 / 36 - 1 elements (about 492^3); larger ones are refused, not wrapped
constexpr long K19_MAX_PLANE = (1LL << 32) / 36 - 1;

int conv_fwd(const float *x, long x_batch, const float *w, int transposed, const float *bias, float *y, long y_batch, void *workspace,
             int B, int O, int I, int D, int H, int W, int dt, void *stream)
{
    if (!x || !w || !y || !workspace) return MLAGG_E_NULLPTR;
    if (!opmode::valid(dt) || (D > 1 && dt != MLAGG_DTYPE_BF16X3)) return MLAGG_E_UNSUPPORTED;
    const long P = (long)D * H * W;
    if (B <= 0 || B > 65535 || O <= 0 || I <= 0 || (I % 16) || D <= 0 || H <= 0 || W <= 0 || P < 96 || P > K19_MAX_PLANE)
        return MLAGG_E_UNSUPPORTED;
    if (x_batch < (long)I * P || y_batch < (long)O * P || (reinterpret_cast<uintptr_t>(workspace) & 15)) return MLAGG_E_UNSUPPORTED;
    if ((long)I * P * 4 >= (1L << 31) - 64) return MLAGG_E_UNSUPPORTED;      // a sample is one buffer resource: 2 GB of range
    C3Geom g{B, O, I, D, H, W, (int)P, x_batch, y_batch};
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CONV3X3, st);
    unsigned short *img = static_cast<unsigned short *>(workspace);
    const int ntaps = D > 1 ? 27 : 9;
    const int n = ntaps * O * I;
    const dim3 igrid((n + 255) / 256);
    if (dt == MLAGG_DTYPE_BF16)
        hipLaunchKernelGGL(conv3x3_weight_image_kernel<MLAGG_DTYPE_BF16>, igrid, dim3(256), 0, st, w, img, O, I, ntaps, transposed ? 1 : 0);
    else if (dt == MLAGG_DTYPE_F16)
        hipLaunchKernelGGL(conv3x3_weight_image_kernel<MLAGG_DTYPE_F16>, igrid, dim3(256), 0, st, w, img, O, I, ntaps, transposed ? 1 : 0);
    else
        hipLaunchKernelGGL(conv3x3_weight_image_kernel<MLAGG_DTYPE_BF16X3>, igrid, dim3(256), 0, st, w, img, O, I, ntaps, transposed ? 1 : 0);
    // tile per wave = (32 TO output channels) x (32 TP pixels); measured on the step's shapes (tools/bench_conv3x3.py with
    // MLAGG_K19_TILE, profiles/round3_h_conv3x3_k19_vs_miopen_tiles.log): 2 x 2 (two workgroups per CU, most waves) everywhere
    // except outputs that fill 96-channel groups exactly on large maps
    int to = O <= 32 ? 1 : 2, tp = 2;
    if (O % 96 == 0 && P >= 16384) {
        to = 3;
        tp = P >= 65536 ? 2 : 1;
    }
    if (W % tp) tp = 1;                                     // the kernel wants a lane's pixel run inside one image row
    static const char *env = getenv("MLAGG_K19_TILE");
    if (env && env[0] >= '1' && env[0] <= '3' && env[1] == ',' && env[2] >= '1' && env[2] <= '3') {
        to = env[0] - '0';
        tp = env[2] - '0';
        if (32 * (to - 1) >= O) to = (O + 31) / 32;
        if (W % tp) tp = 1;
    }
    if (tp == 3) tp = 2 - (W & 1);                          // three-pixel runs are not instantiated
    switch (to * 4 + tp) {
    case 1 * 4 + 1: launch<1, 1>(x, img, bias, y, g, dt, st); break;
    case 1 * 4 + 2: launch<1, 2>(x, img, bias, y, g, dt, st); break;
    case 2 * 4 + 1: launch<2, 1>(x, img, bias, y, g, dt, st); break;
    case 2 * 4 + 2: launch<2, 2>(x, img, bias, y, g, dt, st); break;
    case 3 * 4 + 1: launch<3, 1>(x, img, bias, y, g, dt, st); break;
    default: launch<3, 2>(x, img, bias, y, g, dt, st); break;
    }
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mlagg_conv3x3_supported(int O, int I, int H, int W)
{
    // a sample of the input (I planes) is one buffer resource: under 2 GB
    return O > 0 && I > 0 && (I % 16) == 0 && H > 0 && W > 0 && (long)H * W >= 96 && (long)H * W <= (1LL << 32) / 36 - 1 &&
           (long)I * H * W * 4 < (1L << 31) - 64;
}

// bytes of the weight image: 3 pieces x 9 taps x O x I bf16
extern "C" size_t mlagg_conv3x3_workspace_bytes(int O, int I) { return O > 0 && I > 0 ? (size_t)3 * 9 * O * I * 2 : 0; }

// y (B, O, H, W) = conv3x3(x (B, I, H, W), w) (+ bias).  transposed == 0: w is (O, I, 3, 3), the forward.  transposed != 0: w is the
// forward weight (I, O, 3, 3) of the layer whose DATA GRADIENT this is (x = dy of that layer, O = its input channels).
extern "C" int mlagg_conv3x3_fwd(const float *x, long x_batch, const float *w, int transposed, const float *bias, float *y,
                                 long y_batch, void *workspace, int B, int O, int I, int H, int W, void *stream)
{
    return conv_fwd(x, x_batch, w, transposed, bias, y, y_batch, workspace, B, O, I, 1, H, W, MLAGG_DTYPE_BF16X3, stream);
}

// the same product in the operand form `dtype` (MLAGG_DTYPE_BF16X3: the call above; MLAGG_DTYPE_BF16 / _F16: the 16-bit modes -- operands
// rounded once, one product, fp32 sums; x, w, y stay fp32 in memory)
extern "C" int mlagg_conv3x3_fwd_lp(const float *x, long x_batch, const float *w, int transposed, const float *bias, float *y,
                                    long y_batch, void *workspace, int B, int O, int I, int H, int W, int dtype, void *stream)
{
    return conv_fwd(x, x_batch, w, transposed, bias, y, y_batch, workspace, B, O, I, 1, H, W, dtype, stream);
}

// the same for 3 x 3 x 3 kernels on (B, C, D, H, W) volumes (nine kernel rows (kz, ky) of three taps each)
extern "C" int mlagg_conv3x3x3_supported(int O, int I, int D, int H, int W)
{
    return O > 0 && I > 0 && (I % 16) == 0 && D > 0 && H > 0 && W > 0 && (long)D * H * W >= 96 && (long)D * H * W <= (1LL << 32) / 36 - 1 &&
           (long)I * D * H * W * 4 < (1L << 31) - 64;
}

extern "C" size_t mlagg_conv3x3x3_workspace_bytes(int O, int I) { return O > 0 && I > 0 ? (size_t)3 * 27 * O * I * 2 : 0; }

extern "C" int mlagg_conv3x3x3_fwd(const float *x, long x_batch, const float *w, int transposed, const float *bias, float *y,
                                   long y_batch, void *workspace, int B, int O, int I, int D, int H, int W, void *stream)
{
    if (D <= 1) return MLAGG_E_UNSUPPORTED;
    return conv_fwd(x, x_batch, w, transposed, bias, y, y_batch, workspace, B, O, I, D, H, W, MLAGG_DTYPE_BF16X3, stream);
}

extern "C" int mlagg_conv3x3_wgrad_supported(int O, int I, int H, int W)
{
    return O > 0 && I > 0 && H > 0 && W >= 8 && (W % 8) == 0 && ((long)H * W) % 16 == 0;
}

extern "C" int mlagg_conv3x3x3_wgrad_supported(int O, int I, int D, int H, int W)
{
    return O > 0 && I > 0 && D > 1 && H > 0 && W >= 8 && (W % 8) == 0 && ((long)D * H * W) % 16 == 0;
}

namespace {
size_t wgrad_ws(int B, int O, int I, int D, int H, int W)
{
    int to = 0, ti = 0;
    if (D == 1 && w16_tiles(O, I, MLAGG_DTYPE_BF16X3, to, ti)) {
        W16Geom g16;
        if (make_w16geom(g16, B, O, I, H, W, (long)O * H * W, (long)I * H * W, to, ti)) return 0;
        return (size_t)g16.nbs * 9 * O * I;
    }
    W3Geom g;
    if (make_w3geom(g, B, O, I, D, H, W, (long)O * D * H * W, (long)I * D * H * W)) return 0;
    return (size_t)B * g.nslabs * 9 * g.nz * O * I;
}

int wgrad3(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B, int O, int I, int D, int H,
           int W, int dt, void *stream)
{
    if (!dy || !x || !dW || !workspace) return MLAGG_E_NULLPTR;
    if (!opmode::valid(dt)) return MLAGG_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(x) & 3)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int to = 0, ti = 0;
    const int w16 = D == 1 ? w16_tiles(O, I, dt, to, ti) : 0;
    // the LDS-staged form loads 16-byte chunks of x and addresses rows with 32-bit float offsets
    const bool lds_ok = !(reinterpret_cast<uintptr_t>(x) & 15) && (long)max(O, I) * H * W < (1L << 30);
    if (w16) {
        W16Geom g;
        if (int rc = make_w16geom(g, B, O, I, H, W, dy_batch, x_batch, to, ti)) return rc;
        MLAGG_TIMED(K_CONV3X3, st);
        const dim3 grid(24 * ((g.nbs + 7) / 8), (O + 16 * to - 1) / (16 * to), (I + 16 * ti - 1) / (16 * ti));
        const long dy_last = (long)(B - 1) * dy_batch + (long)O * H * W - 4;
        const bool lds = w16 == 2 && lds_ok;
#define K19W16_LAUNCH(TI_, DT_)                                                                                                       \
    if (lds) hipLaunchKernelGGL((conv3x3_wgrad16_lds_kernel<3, TI_, DT_>), grid, dim3(64), 0, st, dy, x, workspace, g, dy_last);        \
    else hipLaunchKernelGGL((conv3x3_wgrad16_kernel<3, TI_, DT_>), grid, dim3(64), 0, st, dy, x, workspace, g)
        if (ti == 1) {
            if (dt == MLAGG_DTYPE_BF16) { K19W16_LAUNCH(1, MLAGG_DTYPE_BF16); }
            else if (dt == MLAGG_DTYPE_F16) { K19W16_LAUNCH(1, MLAGG_DTYPE_F16); }
            else { K19W16_LAUNCH(1, MLAGG_DTYPE_BF16X3); }
        } else {
            if (dt == MLAGG_DTYPE_BF16) { K19W16_LAUNCH(3, MLAGG_DTYPE_BF16); }
            else if (dt == MLAGG_DTYPE_F16) { K19W16_LAUNCH(3, MLAGG_DTYPE_F16); }
            else { K19W16_LAUNCH(3, MLAGG_DTYPE_BF16X3); }
        }
#undef K19W16_LAUNCH
        const int n = 9 * O * I;
        hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, workspace, g.nbs, O, I, 9, dW);
        return (int)hipGetLastError();
    }
    W3Geom g;
    if (int rc = make_w3geom(g, B, O, I, D, H, W, dy_batch, x_batch)) return rc;
    MLAGG_TIMED(K_CONV3X3, st);
    const dim3 grid(B * g.nslabs * g.nz, (O + 31) / 32, (I + 31) / 32);
    if (dt == MLAGG_DTYPE_BF16)
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<MLAGG_DTYPE_BF16>, grid, dim3(64), 0, st, dy, x, workspace, g);
    else if (dt == MLAGG_DTYPE_F16)
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<MLAGG_DTYPE_F16>, grid, dim3(64), 0, st, dy, x, workspace, g);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<MLAGG_DTYPE_BF16X3>, grid, dim3(64), 0, st, dy, x, workspace, g);
    const int ntaps = 9 * g.nz, n = ntaps * O * I;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, workspace, B * g.nslabs, O, I, ntaps, dW);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" size_t mlagg_conv3x3_wgrad_workspace_floats(int B, int O, int I, int H, int W) { return wgrad_ws(B, O, I, 1, H, W); }
extern "C" size_t mlagg_conv3x3x3_wgrad_workspace_floats(int B, int O, int I, int D, int H, int W) { return D > 1 ? wgrad_ws(B, O, I, D, H, W) : 0; }

// dW (O, I, 3, 3) = weight gradient of y = conv3x3(x, w, padding 1) from dy (B, O, H, W) and x (B, I, H, W); overwritten
extern "C" int mlagg_conv3x3_wgrad(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B,
                                   int O, int I, int H, int W, void *stream)
{
    return wgrad3(dy, dy_batch, x, x_batch, dW, workspace, B, O, I, 1, H, W, MLAGG_DTYPE_BF16X3, stream);
}

// ... in the operand form `dtype` (see mlagg_conv3x3_fwd_lp)
extern "C" int mlagg_conv3x3_wgrad_lp(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B,
                                      int O, int I, int H, int W, int dtype, void *stream)
{
    return wgrad3(dy, dy_batch, x, x_batch, dW, workspace, B, O, I, 1, H, W, dtype, stream);
}

// dW (O, I, 3, 3, 3) of the 3 x 3 x 3 convolution on (B, C, D, H, W) volumes
extern "C" int mlagg_conv3x3x3_wgrad(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B,
                                     int O, int I, int D, int H, int W, void *stream)
{
    if (D <= 1) return MLAGG_E_UNSUPPORTED;
    return wgrad3(dy, dy_batch, x, x_batch, dW, workspace, B, O, I, D, H, W, MLAGG_DTYPE_BF16X3, stream);
}
