// K1' -- cross-scan / cross-merge of the MSMM skip module: the four-direction, multi-scale re-ordering
// between token-major maps and the (B, 4*C, L_cat) channel-major scan sequences.
//
// Replaces the eager copy chains of SS2D_skip.forward_corev0 (reference MambaSkip.py:414-422: view /
// transpose / stack / flip / cat per scale; M:455-471: split / flip / transpose / cat per direction, and the
// 4-way sum at M:534) -- 2.3 ms forward + 3.0 ms backward of copy, cat, flip, index and add kernels per step in
// the round-1 profile -- by ONE data-movement kernel pair:
//   scatter: src (B, L_cat, ...) token-major  ->  dst (B, 4*CB, L_cat), row k*CB + c holds direction k
//   gather : the adjoint (sum over directions when the four directions share one source block).
// Directions (scan position p inside a scale of H x W tokens, token (y, x)):
//   k=0: p = y*W + x   k=1: p = x*H + y   k=2: p = L-1-(y*W+x)   k=3: p = L-1-(x*H+y)     (M:419-421)
// and scales are concatenated in the same order for every direction (M:422).
// The same pair moves x (CB = 96, one shared block), the per-direction delta (CB = 96, four blocks) and
// the per-direction B / C rows (CB = 16, four blocks at stride 35 inside the x_proj output), so the
// projections themselves stay token-major GEMMs.
//
// A workgroup owns a 16x16 token tile of one scale and 32 channels: token rows are read/written as
// 128-byte channel runs, sequence rows as 64-byte position runs (both traversal orders of the tile are
// contiguous in one of its two axes), transposed through a padded LDS tile.  Pure HBM traffic:
// 4 * (1 or 4 + 4) * CB bytes per token.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"

namespace {

constexpr int TS = 16;                 // tile side (tokens)
constexpr int CH = 32;                 // channels per workgroup
constexpr int LP = 33;                 // LDS channel pitch
constexpr int ROWP = TS + 1;           // LDS tile row pitch (tokens): both traversal orders conflict-free
constexpr int UN = 8;                  // token loads in flight per thread

struct XGeom {
    int B, Lc, nscale;
    int H[4], W[4], off[4], tile0[5];  // tile0[i]: first tile index of scale i
    int CB, nblk, tok_stride, blk_stride;
};

__device__ __forceinline__ int seq_pos(int k, int y, int x, int H, int W)
{
    const int p = (k & 1) ? x * H + y : y * W + x;
    return (k & 2) ? H * W - 1 - p : p;
}

// GATHER = false: tok -> seq (scatter);  GATHER = true: seq -> tok (adjoint; sums directions if nblk == 1)
template <bool GATHER>
__global__ void __launch_bounds__(256)
cross_scan_kernel(const float *__restrict__ tok_c, float *__restrict__ tok_m, const float *__restrict__ seq_c,
                  float *__restrict__ seq_m, XGeom g)
{
    __shared__ float tile[TS * ROWP * LP];
    int sc = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.nscale && (int)blockIdx.x >= g.tile0[i]) sc = i;
    const int H = g.H[sc], W = g.W[sc], off = g.off[sc];
    const int tl = blockIdx.x - g.tile0[sc];
    const int tiles_x = (W + TS - 1) / TS;
    const int y0 = (tl / tiles_x) * TS, x0 = (tl % tiles_x) * TS;
    const int c0 = blockIdx.y * CH, b = blockIdx.z;
    const int nc = min(CH, g.CB - c0);
    const int tid = threadIdx.x;
    // 16-byte sequence accesses need H, W, the scale offset and L_cat to be multiples of 4 (all true for power-of-two maps)
    const bool vec4 = ((H | W | off | g.Lc) & 3) == 0;
    const size_t tokbase = (size_t)b * g.Lc + off;

    auto tok_ptr = [&](int ly, int lx, int blk, int c) -> size_t {
        return (tokbase + (size_t)(y0 + ly) * W + x0 + lx) * g.tok_stride + (size_t)blk * g.blk_stride + c0 + c;
    };
    auto lds_idx = [&](int ly, int lx, int c) { return (ly * ROWP + lx) * LP + c; };

    if (!GATHER) {
        for (int k = 0; k < 4; ++k) {
            if (k == 0 || g.nblk > 1) {                 // (re)load the source block of this direction
                __syncthreads();
                // UN loads in flight per thread, all unconditional (clamped to the tile's first element, value dropped): a
                // guarded load per iteration is waited for before the next one is issued, i.e. ONE 4-byte load in flight
                // per lane -- the kernel then runs at the memory latency, not the bandwidth
                const size_t safe = tok_ptr(0, 0, g.nblk > 1 ? k : 0, 0);
                for (int e0 = tid; e0 < TS * TS * CH; e0 += 256 * UN) {
                    float v[UN];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = e0 + 256 * u;
                        const int c = e & (CH - 1), t = e >> 5, ly = t >> 4, lx = t & 15;
                        const bool ok = c < nc && y0 + ly < H && x0 + lx < W;
                        v[u] = tok_c[ok ? tok_ptr(ly, lx, g.nblk > 1 ? k : 0, c) : safe];
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = e0 + 256 * u;
                        const int c = e & (CH - 1), t = e >> 5, ly = t >> 4, lx = t & 15;
                        const bool ok = c < nc && y0 + ly < H && x0 + lx < W;
                        tile[lds_idx(ly, lx, c)] = ok ? v[u] : 0.f;
                    }
                }
                __syncthreads();
            }
            if (vec4) {
                // 4 consecutive sequence positions per thread: one 16-byte store instead of four 4-byte ones
                for (int e = tid; e < TS * (TS / 4) * CH; e += 256) {
                    const int q = e & 63, c = e >> 6;
                    const int a = q >> 2, f4 = (q & 3) * 4;      // f4..f4+3 along the direction's own axis
                    const int ly0 = (k & 1) ? f4 : a, lx0 = (k & 1) ? a : f4;
                    if (c < nc && y0 + ly0 < H && x0 + lx0 < W) {
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = tile[lds_idx((k & 1) ? ly0 + j : ly0, (k & 1) ? lx0 : lx0 + j, c)];
                        const int pf = (k & 1) ? (x0 + lx0) * H + y0 + ly0 : (y0 + ly0) * W + x0 + lx0;   // forward position
                        float *dst = seq_m + ((size_t)b * 4 * g.CB + (size_t)k * g.CB + c0 + c) * g.Lc + off;
                        if (k & 2) *reinterpret_cast<float4 *>(dst + (H * W - 4 - pf)) = make_float4(v[3], v[2], v[1], v[0]);
                        else *reinterpret_cast<float4 *>(dst + pf) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            } else
            for (int e = tid; e < TS * TS * CH; e += 256) {
                const int q = e & 255, c = e >> 8;
                const int a = q >> 4, f = q & 15;            // f runs fastest along the direction's own axis
                const int ly = (k & 1) ? f : a, lx = (k & 1) ? a : f;
                if (c < nc && y0 + ly < H && x0 + lx < W) {
                    const int p = seq_pos(k, y0 + ly, x0 + lx, H, W);
                    seq_m[((size_t)b * 4 * g.CB + (size_t)k * g.CB + c0 + c) * g.Lc + off + p] = tile[lds_idx(ly, lx, c)];
                }
            }
        }
    } else {
        for (int k = 0; k < 4; ++k) {
            const bool fresh = (k == 0 || g.nblk > 1);
            __syncthreads();
            if (vec4) {
                // the 8 float4 of a thread for this direction in flight together (see the scatter branch)
                constexpr int NV = TS * (TS / 4) * CH / 256;
                float4 r[NV];
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int e = tid + 256 * u;
                    const int q = e & 63, c = e >> 6;
                    const int a = q >> 2, f4 = (q & 3) * 4;
                    const int ly0 = (k & 1) ? f4 : a, lx0 = (k & 1) ? a : f4;
                    const bool ok = c < nc && y0 + ly0 < H && x0 + lx0 < W;
                    const int pf = (k & 1) ? (x0 + lx0) * H + y0 + ly0 : (y0 + ly0) * W + x0 + lx0;
                    const float *src = seq_c + ((size_t)b * 4 * g.CB + (size_t)k * g.CB + c0 + (ok ? c : 0)) * g.Lc + off;
                    const int pos = ok ? ((k & 2) ? H * W - 4 - pf : pf) : 0;
                    r[u] = *reinterpret_cast<const float4 *>(src + pos);
                }
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int e = tid + 256 * u;
                    const int q = e & 63, c = e >> 6;
                    const int a = q >> 2, f4 = (q & 3) * 4;
                    const int ly0 = (k & 1) ? f4 : a, lx0 = (k & 1) ? a : f4;
                    const bool ok = c < nc && y0 + ly0 < H && x0 + lx0 < W;
                    const float v[4] = {ok ? ((k & 2) ? r[u].w : r[u].x) : 0.f, ok ? ((k & 2) ? r[u].z : r[u].y) : 0.f,
                                        ok ? ((k & 2) ? r[u].y : r[u].z) : 0.f, ok ? ((k & 2) ? r[u].x : r[u].w) : 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int li = lds_idx((k & 1) ? ly0 + j : ly0, (k & 1) ? lx0 : lx0 + j, c);
                        if (fresh) tile[li] = v[j]; else tile[li] += v[j];
                    }
                }
            } else
            for (int e = tid; e < TS * TS * CH; e += 256) {
                const int q = e & 255, c = e >> 8;
                const int a = q >> 4, f = q & 15;
                const int ly = (k & 1) ? f : a, lx = (k & 1) ? a : f;
                float v = 0.f;
                if (c < nc && y0 + ly < H && x0 + lx < W) {
                    const int p = seq_pos(k, y0 + ly, x0 + lx, H, W);
                    v = seq_c[((size_t)b * 4 * g.CB + (size_t)k * g.CB + c0 + c) * g.Lc + off + p];
                }
                // every (token, channel) of the tile is owned by exactly one thread per direction
                if (fresh) tile[lds_idx(ly, lx, c)] = v; else tile[lds_idx(ly, lx, c)] += v;
            }
            if (k == 3 || g.nblk > 1) {                 // flush the finished block token-major
                __syncthreads();
                for (int e = tid; e < TS * TS * CH; e += 256) {
                    const int c = e & (CH - 1), t = e >> 5, ly = t >> 4, lx = t & 15;
                    if (c < nc && y0 + ly < H && x0 + lx < W)
                        tok_m[tok_ptr(ly, lx, g.nblk > 1 ? k : 0, c)] = tile[lds_idx(ly, lx, c)];
                }
            }
        }
    }
}

int make_geom(XGeom &g, int B, int nscale, const int *H, const int *W, int CB, int nblk, int tok_stride, int blk_stride)
{
    if (B <= 0 || B > 65535 || nscale < 1 || nscale > 4 || CB <= 0 || (nblk != 1 && nblk != 4) || !H || !W)
        return MLAGG_E_UNSUPPORTED;
    g.B = B; g.nscale = nscale; g.CB = CB; g.nblk = nblk; g.tok_stride = tok_stride; g.blk_stride = blk_stride;
    int off = 0, t0 = 0;
    for (int i = 0; i < 4; ++i) {
        g.H[i] = i < nscale ? H[i] : 1;
        g.W[i] = i < nscale ? W[i] : 1;
        g.off[i] = off;
        g.tile0[i] = t0;
        if (i < nscale) {
            if (H[i] <= 0 || W[i] <= 0) return MLAGG_E_UNSUPPORTED;
            off += H[i] * W[i];
            t0 += ((H[i] + TS - 1) / TS) * ((W[i] + TS - 1) / TS);
        }
    }
    g.tile0[4] = t0;
    g.Lc = off;
    if (tok_stride < (nblk > 1 ? 3 * blk_stride + CB : CB)) return MLAGG_E_UNSUPPORTED;
    return 0;
}

}  // namespace

extern "C" int mlagg_cross_scan(const float *tok, int tok_stride, int blk_stride, float *seq, int B, int nscale,
                                const int *H, const int *W, int CB, int nblk, void *stream)
{
    if (!tok || !seq) return MLAGG_E_NULLPTR;
    XGeom g;
    if (int rc = make_geom(g, B, nscale, H, W, CB, nblk, tok_stride, blk_stride)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CROSS_SCAN, st);
    hipLaunchKernelGGL(cross_scan_kernel<false>, dim3(g.tile0[4], (CB + CH - 1) / CH, B), dim3(256), 0, st, tok, nullptr,
                       nullptr, seq, g);
    return (int)hipGetLastError();
}

extern "C" int mlagg_cross_merge(const float *seq, float *tok, int tok_stride, int blk_stride, int B, int nscale,
                                 const int *H, const int *W, int CB, int nblk, void *stream)
{
    if (!tok || !seq) return MLAGG_E_NULLPTR;
    XGeom g;
    if (int rc = make_geom(g, B, nscale, H, W, CB, nblk, tok_stride, blk_stride)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MLAGG_TIMED(K_CROSS_MERGE, st);
    hipLaunchKernelGGL(cross_scan_kernel<true>, dim3(g.tile0[4], (CB + CH - 1) / CH, B), dim3(256), 0, st, nullptr, tok,
                       seq, nullptr, g);
    return (int)hipGetLastError();
}
