// K3 -- fused 3x3-window differential attention + RMSNorm + LePE for gfx950.
//
// Replaces the eager chain of the `local=True` branch of AggregatedAttention.forward (reference
// nnUNetTrainer_MLAgg_2D_dt_MS.py:693-717, 779-782): two nn.Unfold calls that write 9x copies of K
// and V, a batched 1x24 @ 24x9 matmul, masked softmax, head-pair subtraction, 1x9 @ 9x48 matmul,
// RMSNorm, scaling and the depthwise-3x3 LePE on V.  Here one pass reads q, k, v once from HBM (the
// 3x3 halo is served by L1/L2: a workgroup owns an 8x8 token tile of one head) and writes the
// branch output; nothing of size 9x is ever materialised.
//
// Work decomposition: a "unit" is (batch, token, head); a DPP quad of 4 lanes owns a unit.  Lane r
// of the quad holds channels [12r, 12r+12) of the head's 48: lanes 0,1 carry q1/k1 (map "+"), lanes
// 2,3 carry q2/k2 (map "-"), and all four carry a quarter of v / out.  Dot products are finished
// with one quad_perm add, the two softmaxes run redundantly in each lane pair, the map exchange is
// one quad_perm, the RMS statistic is a quad sum.
//
// Backward is gather-form in two launches (no atomics on dq/dk/dv):
//   A) per unit: recompute the window softmaxes, form d(o) through the RMSNorm, dA, dlogits, dq and
//      park {A_j, dlogit1_j, dlogit2_j, d(o)} (75 floats per unit) in the workspace;
//   B) per unit: dk and dv of token t are the sums over the <= 9 tokens whose window contains t.
// Parameter gradients (lambda, subln weight, LePE weight/bias) are block-reduced, then atomically
// accumulated (a few hundred floats per launch).
//
// Roofline: HBM-bound; algorithmic bytes forward 16*d*N per image and module.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"
#include "prof.h"
#include "internal.h"

namespace {

constexpr int HD2 = 48;       // 2 * head_dim: channels per head in q / k / v / out
constexpr int PER = 12;       // channels per lane
constexpr int TILE = 8;       // 8x8 tokens per workgroup
constexpr float RMS_EPS = 1e-5f;
constexpr float OUT_GAIN = 0.2f;          // 1 - lambda_init (reference T:717)
constexpr int WS_PER_UNIT = 76;           // 9 A + 9 dl1 + 9 dl2 + 48 d(o), padded to a float4 multiple

struct Geom {
    int batch, H, W, nh, d;
    int q_stride, kv_stride, out_stride;
    float scale;
};

__device__ __forceinline__ float dpp_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    return v;
}

__device__ __forceinline__ void load12(const float *__restrict__ p, float (&x)[PER])
{
    const float4 a = *reinterpret_cast<const float4 *>(p);
    const float4 b = *reinterpret_cast<const float4 *>(p + 4);
    const float4 c = *reinterpret_cast<const float4 *>(p + 8);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w;
}

__device__ __forceinline__ void store12(float *__restrict__ p, const float (&x)[PER])
{
    *reinterpret_cast<float4 *>(p) = make_float4(x[0], x[1], x[2], x[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(x[4], x[5], x[6], x[7]);
    *reinterpret_cast<float4 *>(p + 8) = make_float4(x[8], x[9], x[10], x[11]);
}

struct Unit {
    int b, h, y, x, r;
    bool act;
};

__device__ __forceinline__ Unit unit_id(const Geom &g)
{
    Unit u;
    const int tiles_x = (g.W + TILE - 1) / TILE;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int lu = threadIdx.x >> 2;
    u.r = threadIdx.x & 3;
    u.y = ty * TILE + (lu >> 3);
    u.x = tx * TILE + (lu & 7);
    u.h = blockIdx.y;
    u.b = blockIdx.z;
    u.act = u.y < g.H && u.x < g.W;
    return u;
}

// masked softmax over the 9 window logits of this lane's map; invalid entries -> 0
__device__ __forceinline__ void window_softmax(const float (&l)[9], unsigned valid, float (&s)[9])
{
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 9; ++j)
        if ((valid >> j) & 1) m = fmaxf(m, l[j]);
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        s[j] = ((valid >> j) & 1) ? __expf(l[j] - m) : 0.f;
        z += s[j];
    }
    const float iz = 1.f / z;
#pragma unroll
    for (int j = 0; j < 9; ++j) s[j] *= iz;
}

// Shared front half of forward and backward-A: logits, the two softmaxes, A_j, o and its rstd.
struct Front {
    float s[9];       // this lane pair's own softmax map (map 1 for r < 2, map 2 otherwise)
    float other[9];   // the other map
    float Aw[9];      // s1 - lam * s2
    float o[PER];     // unnormalised output channels of this lane
    float rstd;
    unsigned valid;
};

__device__ __forceinline__ void front_half(const Geom &g, const Unit &u, const float *__restrict__ q,
                                           const float *__restrict__ kv, float lam, Front &f)
{
    const int N = g.H * g.W;
    const size_t tok = (size_t)u.b * N + (size_t)u.y * g.W + u.x;
    float qv[PER];
    load12(q + tok * g.q_stride + u.h * HD2 + PER * u.r, qv);
    float lg[9];
    f.valid = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int yy = u.y + j / 3 - 1, xx = u.x + j % 3 - 1;
        const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        float p = 0.f;
        if (in) {
            f.valid |= 1u << j;
            float kx[PER];
            load12(kv + ((size_t)u.b * N + (size_t)yy * g.W + xx) * g.kv_stride + u.h * HD2 + PER * u.r, kx);
#pragma unroll
            for (int e = 0; e < PER; ++e) p += qv[e] * kx[e];
        }
        lg[j] = (p + dpp_xor1(p)) * g.scale;
    }
    window_softmax(lg, f.valid, f.s);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        f.other[j] = dpp_xor2(f.s[j]);
        f.Aw[j] = u.r < 2 ? f.s[j] - lam * f.other[j] : f.other[j] - lam * f.s[j];
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) f.o[e] = 0.f;
}

// LePE weights of the workgroup's head, transposed to [tap][channel] in LDS: a lane's 12 channels of one tap are
// three ds_read_b128 (the untransposed global layout cost 108 scattered dword loads per lane).
__device__ __forceinline__ void stage_lepe(float (*swt)[HD2], const float *__restrict__ lepe_w, int head)
{
    for (int i = threadIdx.x; i < 9 * HD2; i += blockDim.x) {
        const int c = i / 9, j = i - 9 * c;
        swt[j][c] = lepe_w[(head * HD2 + c) * 9 + j];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256)
local_attn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ lamp,
                      const float *__restrict__ subln_w, const float *__restrict__ lepe_w,
                      const float *__restrict__ lepe_b, float *__restrict__ out, Geom g)
{
    __shared__ float swt[9][HD2];
    stage_lepe(swt, lepe_w, blockIdx.y);
    const Unit u = unit_id(g);
    if (!u.act) return;            // whole quads leave together; no block-level sync below
    const float lam = lamp[0];
    const int N = g.H * g.W;
    Front f;
    front_half(g, u, q, kv, lam, f);
    const int cbase = u.h * HD2 + PER * u.r;          // channel of this lane inside d
    float lp[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) lp[e] = lepe_b[cbase + e];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        if (!((f.valid >> j) & 1)) continue;
        const int yy = u.y + j / 3 - 1, xx = u.x + j % 3 - 1;
        float vx[PER], wj[PER];
        load12(kv + ((size_t)u.b * N + (size_t)yy * g.W + xx) * g.kv_stride + g.d + cbase, vx);
        load12(&swt[j][PER * u.r], wj);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            f.o[e] += f.Aw[j] * vx[e];
            lp[e] += wj[e] * vx[e];
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < PER; ++e) ss += f.o[e] * f.o[e];
    const float rstd = rsqrtf(quad_sum(ss) * (1.f / HD2) + RMS_EPS);
    float res[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) res[e] = OUT_GAIN * subln_w[PER * u.r + e] * f.o[e] * rstd + lp[e];
    store12(out + ((size_t)u.b * N + (size_t)u.y * g.W + u.x) * g.out_stride + cbase, res);
}

// backward A: everything that is local to the unit
__global__ void __launch_bounds__(256)
local_attn_bwd_a_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ lamp,
                        const float *__restrict__ subln_w, const float *__restrict__ dout, int dout_stride,
                        float *__restrict__ dq, int dq_stride, float *__restrict__ ws, float *__restrict__ pgrad, Geom g)
{
    __shared__ float red[4][4][PER + 1];   // per wave and quad-lane r: 12 subln-weight partials (+ dlam in [w][0][12])
    __syncthreads();
    const Unit u = unit_id(g);
    const float lam = lamp[0];
    const int N = g.H * g.W;
    float dw[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) dw[e] = 0.f;
    float dl_acc = 0.f;
    if (u.act) {
        Front f;
        front_half(g, u, q, kv, lam, f);
        const int cbase = u.h * HD2 + PER * u.r;
        const size_t tok = (size_t)u.b * N + (size_t)u.y * g.W + u.x;
        float vj[9][PER];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int yy = u.y + j / 3 - 1, xx = u.x + j % 3 - 1;
            if ((f.valid >> j) & 1) {
                load12(kv + ((size_t)u.b * N + (size_t)yy * g.W + xx) * g.kv_stride + g.d + cbase, vj[j]);
            } else {
#pragma unroll
                for (int e = 0; e < PER; ++e) vj[j][e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < PER; ++e) f.o[e] += f.Aw[j] * vj[j][e];
        }
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < PER; ++e) ss += f.o[e] * f.o[e];
        const float rstd = rsqrtf(quad_sum(ss) * (1.f / HD2) + RMS_EPS);
        float dy[PER], dO[PER];
        load12(dout + tok * dout_stride + cbase, dy);
        // y_e = G w_e o_e rstd  ->  d(o)_e = G rstd (w_e dy_e - o_e rstd^2 mean_e'(w dy o))
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const float wdy = subln_w[PER * u.r + e] * dy[e];
            dot += wdy * f.o[e];
            dw[e] = OUT_GAIN * dy[e] * f.o[e] * rstd;
        }
        dot = quad_sum(dot) * (1.f / HD2) * rstd * rstd;
#pragma unroll
        for (int e = 0; e < PER; ++e)
            dO[e] = OUT_GAIN * rstd * (subln_w[PER * u.r + e] * dy[e] - f.o[e] * dot);
        // dA_j = d(o) . v_j over the head's 48 channels
        float dA[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float p = 0.f;
#pragma unroll
            for (int e = 0; e < PER; ++e) p += dO[e] * vj[j][e];
            dA[j] = quad_sum(p);
        }
        // map 1: ds = dA ; map 2: ds = -lam dA ; d(lam) = -sum_j dA_j s2_j (count once per unit)
        float dsum = 0.f, dlm = 0.f;
        float ds[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            ds[j] = u.r < 2 ? dA[j] : -lam * dA[j];
            dsum += ds[j] * f.s[j];
            if (u.r == 2) dlm -= dA[j] * f.s[j];
        }
        dl_acc = dlm;
        float dlg[9];      // d(logit) of this lane pair's map, already times scale
#pragma unroll
        for (int j = 0; j < 9; ++j) dlg[j] = f.s[j] * (ds[j] - dsum) * g.scale;
        // dq = sum_j dlogit_j k_j  (this lane's 12 channels of q1 or q2)
        float dqv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) dqv[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (!((f.valid >> j) & 1)) continue;
            const int yy = u.y + j / 3 - 1, xx = u.x + j % 3 - 1;
            float kx[PER];
            load12(kv + ((size_t)u.b * N + (size_t)yy * g.W + xx) * g.kv_stride + u.h * HD2 + PER * u.r, kx);
#pragma unroll
            for (int e = 0; e < PER; ++e) dqv[e] += dlg[j] * kx[e];
        }
        store12(dq + tok * dq_stride + cbase, dqv);
        // workspace row of the unit: [0,9) A, [9,18) dlogit map1, [18,27) dlogit map2, [28,76) d(o)
        float *wrow = ws + (tok * g.nh + u.h) * WS_PER_UNIT;
        if (u.r == 0) {
#pragma unroll
            for (int j = 0; j < 9; ++j) { wrow[j] = f.Aw[j]; wrow[9 + j] = dlg[j]; }
        }
        if (u.r == 2) {
#pragma unroll
            for (int j = 0; j < 9; ++j) wrow[18 + j] = dlg[j];
        }
        store12(wrow + 28 + PER * u.r, dO);
    }
    // block reduction of the parameter gradients: first over the 16 units of a wave (lanes with equal r: xor 4, 8, 16, 32),
    // then one LDS add per (wave, r, e) -- 64 lanes adding to one LDS word serialise
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        float v = dw[e];
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) < 4) red[threadIdx.x >> 6][threadIdx.x & 3][e] = v;
    }
    {
        float v = dl_acc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][0][PER] = v;
    }
    __syncthreads();
    // one partial row [d(subln_w) (48) | d(lambda)] per workgroup, summed over workgroups in a fixed order by the launcher's column
    // sum (the first version added them with float atomics: run-to-run differences in the last bits)
    float *prow = pgrad + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (4 * PER + 1);
    if (threadIdx.x < 4 * PER) {
        const int r = threadIdx.x / PER, e = threadIdx.x % PER;
        prow[threadIdx.x] = (red[0][r][e] + red[1][r][e]) + (red[2][r][e] + red[3][r][e]);
    }
    if (threadIdx.x == 63) prow[4 * PER] = (red[0][0][PER] + red[1][0][PER]) + (red[2][0][PER] + red[3][0][PER]);
}

// backward B: gather dk, dv for token t from the <= 9 windows that contain it
__global__ void __launch_bounds__(256)
local_attn_bwd_b_kernel(const float *__restrict__ q, const float *__restrict__ lepe_w,
                        const float *__restrict__ dout, int dout_stride, const float *__restrict__ ws,
                        float *__restrict__ dkv, int dkv_stride, Geom g)
{
    __shared__ float swt[9][HD2];
    stage_lepe(swt, lepe_w, blockIdx.y);
    const Unit u = unit_id(g);
    if (!u.act) return;
    const int N = g.H * g.W;
    const int cbase = u.h * HD2 + PER * u.r;
    const size_t tok = (size_t)u.b * N + (size_t)u.y * g.W + u.x;
    float dk[PER], dv[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
#pragma unroll
    for (int jj = 0; jj < 9; ++jj) {
        // token i = t + offset(jj) sees t at window index 8 - jj
        const int yy = u.y + jj / 3 - 1, xx = u.x + jj % 3 - 1;
        if (!(yy >= 0 && yy < g.H && xx >= 0 && xx < g.W)) continue;
        const int jw = 8 - jj;
        const size_t ti = (size_t)u.b * N + (size_t)yy * g.W + xx;
        const float *wrow = ws + (ti * g.nh + u.h) * WS_PER_UNIT;
        const float Ai = wrow[jw];
        const float dli = wrow[(u.r < 2 ? 9 : 18) + jw];
        float dOi[PER], qi[PER], dyi[PER], wj[PER];
        load12(wrow + 28 + PER * u.r, dOi);
        load12(q + ti * g.q_stride + cbase, qi);
        load12(dout + ti * dout_stride + cbase, dyi);
        load12(&swt[jw][PER * u.r], wj);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            dv[e] += Ai * dOi[e] + wj[e] * dyi[e];
            dk[e] += dli * qi[e];
        }
    }
    store12(dkv + tok * dkv_stride + cbase, dk);
    store12(dkv + tok * dkv_stride + g.d + cbase, dv);
}

int make_geom(Geom &g, int batch, int H, int W, int nh, int qs, int kvs, int outs, float scale)
{
    if (batch <= 0 || H <= 0 || W <= 0 || nh <= 0 || batch > 65535 || nh > 65535) return MLAGG_E_UNSUPPORTED;
    g.batch = batch; g.H = H; g.W = W; g.nh = nh; g.d = nh * HD2;
    g.q_stride = qs; g.kv_stride = kvs; g.out_stride = outs; g.scale = scale;
    if (qs < g.d || kvs < 2 * g.d || (qs & 3) || (kvs & 3)) return MLAGG_E_UNSUPPORTED;
    return 0;
}

dim3 tile_grid(const Geom &g)
{
    return dim3(((g.W + TILE - 1) / TILE) * ((g.H + TILE - 1) / TILE), g.nh, g.batch);
}

}  // namespace

extern "C" int mlagg_local_attn_fwd(const float *q, int q_stride, const float *kv, int kv_stride,
                                    const float *lam, const float *subln_w, const float *lepe_w,
                                    const float *lepe_b, float *out, int out_stride, int batch, int H, int W,
                                    int nh, float scale, void *stream)
{
    if (!q || !kv || !lam || !subln_w || !lepe_w || !lepe_b || !out) return MLAGG_E_NULLPTR;
    Geom g;
    if (int rc = make_geom(g, batch, H, W, nh, q_stride, kv_stride, out_stride, scale)) return rc;
    if (out_stride < g.d || (out_stride & 3)) return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    { MLAGG_TIMED(K_LOCAL_FWD, st); hipLaunchKernelGGL(local_attn_fwd_kernel, tile_grid(g), dim3(256), 0, st, q, kv, lam, subln_w, lepe_w, lepe_b,
                       out, g); }
    return (int)hipGetLastError();
}

extern "C" size_t mlagg_local_attn_bwd_workspace_floats(int batch, int H, int W, int nh)
{
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE) * nh * batch;
    return (size_t)batch * H * W * nh * WS_PER_UNIT + mlagg_internal::dwconv_wgrad_workspace_floats(batch, H, W, nh * HD2) +
           tiles * (4 * PER + 1);
}

extern "C" int mlagg_local_attn_bwd(const float *q, int q_stride, const float *kv, int kv_stride,
                                    const float *lam, const float *subln_w, const float *lepe_w,
                                    const float *dout, int dout_stride, float *dq, int dq_stride, float *dkv,
                                    int dkv_stride, float *dlam, float *dsubln_w, float *dlepe_w,
                                    float *dlepe_b, float *workspace, int batch, int H, int W, int nh,
                                    float scale, void *stream)
{
    if (!q || !kv || !lam || !subln_w || !lepe_w || !dout || !dq || !dkv || !dlam || !dsubln_w || !dlepe_w ||
        !dlepe_b || !workspace)
        return MLAGG_E_NULLPTR;
    Geom g;
    if (int rc = make_geom(g, batch, H, W, nh, q_stride, kv_stride, dout_stride, scale)) return rc;
    if (dout_stride < g.d || (dout_stride & 3) || dq_stride < g.d || (dq_stride & 3) || dkv_stride < 2 * g.d ||
        (dkv_stride & 3))
        return MLAGG_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 tg = tile_grid(g);
    const int nwg = (int)(tg.x * tg.y * tg.z);
    float *pgrad = workspace + (size_t)batch * H * W * nh * WS_PER_UNIT + mlagg_internal::dwconv_wgrad_workspace_floats(batch, H, W, g.d);
    { MLAGG_TIMED(K_LOCAL_BWD_A, st); hipLaunchKernelGGL(local_attn_bwd_a_kernel, tg, dim3(256), 0, st, q, kv, lam, subln_w, dout,
                       dout_stride, dq, dq_stride, workspace, pgrad, g);
      hipLaunchKernelGGL(mlagg_internal::column_sum_split_kernel<0>, dim3(1), dim3(1024), 0, st, pgrad, nwg, 4 * PER + 1, 4 * PER + 1,
                         4 * PER, dsubln_w, dlam); }
    { MLAGG_TIMED(K_LOCAL_BWD_B, st); hipLaunchKernelGGL(local_attn_bwd_b_kernel, tile_grid(g), dim3(256), 0, st, q, lepe_w, dout, dout_stride,
                       workspace, dkv, dkv_stride, g); }
    // LePE is a depthwise 3x3 on v: its weight/bias gradients are K2's weight-gradient kernel on (v, dout)
    mlagg_internal::dwconv_wgrad_launch(kv + g.d, kv_stride, dout, dout_stride, nullptr, dlepe_w, dlepe_b,
                                        workspace + (size_t)batch * H * W * nh * WS_PER_UNIT, batch, H, W, g.d, 0, st);
    return (int)hipGetLastError();
}
