/*
 * mlagg_hip.h -- C ABI of libmlagg_hip.so (MI355X / gfx950 kernels for the MLAgg-UNet 2D hot path).
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes, no torch types; every buffer is owned by the caller, the
 *     library never allocates, frees or synchronises (safe inside hipGraph capture);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - return 0 on success, a negative MLAGG_E_* code for a rejected argument, or a positive
 *     hipError_t from the launch.  The Python host turns any non-zero return into RuntimeError,
 *     the only exception type nnU-Net handles
 *     (mlagg/nnunetv2/training/nnUNetTrainer/variants/benchmarking/nnUNetTrainerBenchmark_5epochs.py:25-29);
 *   - all tensors fp32, contiguous in the stated layout; re-entrant; callable from autograd's
 *     backward thread.
 */
#ifndef MLAGG_HIP_H
#define MLAGG_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLAGG_E_UNSUPPORTED (-1) /* shape outside what the kernels are built for */
#define MLAGG_E_NULLPTR     (-2)
#define MLAGG_E_WORKSPACE   (-3)

/* arithmetic type of the mixed-precision entry points (tensors stay fp32 in memory; see mlagg_linear_lp_*) */
#define MLAGG_DTYPE_F32  0
#define MLAGG_DTYPE_BF16 1
#define MLAGG_DTYPE_F16  2
#define MLAGG_DTYPE_BF16X3 3 /* GEMM arithmetic only: fp32 operands as three bf16 pieces each, six partial products (fp32-accurate) */

const char *mlagg_version(void);
const char *mlagg_error_string(int code);

/* Per-kernel HIP-event timing for bench.py's roofline line (diagnostics; off by default, zero cost).
 * select: -1 off, -2 every kernel, otherwise one kernel id in [0, mlagg_profile_kernel_count()).
 * collect: synchronises with the recorded events (so NOT graph-capturable), writes the summed
 * milliseconds and launch counts per kernel id into ms[count] / counts[count] and resets. */
int mlagg_profile_kernel_count(void);
const char *mlagg_profile_kernel_name(int id);
int mlagg_profile_select(int id);
int mlagg_profile_collect(double *ms, int *counts);

/* ------------------------------------------------------------------------------------------
 * K1: selective scan.  Replaces mamba-ssm's `selective_scan_cuda.fwd/bwd` behind
 * `selective_scan_fn(u, delta, A, B, C, D, z=None, delta_bias, delta_softplus=True)` as called at
 * mlagg/nnunetv2/training/nnUNetTrainer/variants/mamba/MambaSkip.py:445-451 (ABI mirrored at
 * .../variants/mamba/vmamba/csms6s.py:224,235-238).
 *   u, delta, out      (batch, dim, L)        dim = G * H, channel d uses group d / H
 *   A                  (dim, N)               N must be 16
 *   B, C               (batch, G, N, L)
 *   D, delta_bias      (dim) or NULL
 *   chunk_state        workspace AND saved-for-backward tensor, mlagg_selscan_state_floats() floats:
 *                      [batch][nchunks][dim][N] states entering each 64-step chunk, followed by
 *                      [batch][nchunks][dim] per-chunk sums of softplus'd delta and
 *                      [batch][nchunks][7][dim][N] states entering the 2nd..8th 8-step tile of each chunk.
 * ------------------------------------------------------------------------------------------ */
size_t mlagg_selscan_state_floats(int batch, int dim, int L, int N);
int mlagg_selscan_fwd(const float *u, const float *delta, const float *A, const float *B, const float *C,
                      const float *D, const float *delta_bias, float *out, float *chunk_state,
                      int batch, int dim, int L, int N, int G, int delta_softplus, void *stream);

/* Backward.  `workspace` needs mlagg_selscan_bwd_workspace_floats() floats (reverse chunk carries and
 * per-chunk partial sums of dA / dD / ddelta_bias, reduced deterministically inside the call).
 * dD / ddelta_bias may be NULL when D / delta_bias are NULL. */
size_t mlagg_selscan_bwd_workspace_floats(int batch, int dim, int L, int N);
int mlagg_selscan_bwd(const float *u, const float *delta, const float *A, const float *B, const float *C,
                      const float *D, const float *delta_bias, const float *dout, const float *chunk_state,
                      float *du, float *ddelta, float *dA, float *dB, float *dC, float *dD,
                      float *ddelta_bias, float *workspace,
                      int batch, int dim, int L, int N, int G, int delta_softplus, void *stream);

/* K1, low-rank delta form: the scan with SS2D_skip's dt projection folded in.  Replaces the pair
 *   dts = einsum("b k r l, k d r -> b k d l", dts_r, dt_projs_weight)            (MambaSkip.py:430-436)
 *   selective_scan_fn(xs, dts, As, Bs, Cs, Ds, delta_bias=..., delta_softplus=True)  (MambaSkip.py:445-451)
 * dtr (batch, G, R, L): the rank-R rows of group g (scan order); Wdt (dim, R): row d = projection of channel d
 * (dt_projs_weight viewed as (K*d_inner, R)); 1 <= R <= 4.  delta (batch, dim, L) never exists in memory.
 * Backward overwrites du (batch, dim, L), ddtr (batch, G, R, L), dWdt (dim, R), dA, dB, dC and, when non-NULL,
 * dD, ddelta_bias; chunk_state / workspace sizes as for mlagg_selscan_fwd / _bwd. */
int mlagg_selscan_lowrank_fwd(const float *u, const float *dtr, const float *Wdt, int R, const float *A, const float *B,
                              const float *C, const float *D, const float *delta_bias, float *out, float *chunk_state,
                              int batch, int dim, int L, int N, int G, int delta_softplus, void *stream);
int mlagg_selscan_lowrank_bwd(const float *u, const float *dtr, const float *Wdt, int R, const float *A, const float *B,
                              const float *C, const float *D, const float *delta_bias, const float *dout,
                              const float *chunk_state, float *du, float *ddtr, float *dWdt, float *dA, float *dB,
                              float *dC, float *dD, float *ddelta_bias, float *workspace, int batch, int dim, int L,
                              int N, int G, int delta_softplus, void *stream);

/* ------------------------------------------------------------------------------------------
 * K3: 3x3-window differential attention + RMSNorm + LePE, the `local=True` branch of
 * AggregatedAttention.forward (mlagg/.../nnUNetTrainer_MLAgg_2D_dt_MS.py:693-717, 779-782).
 * Token-major inputs: q (batch, H*W, d) and kv (batch, H*W, 2d) straight out of the q / kv
 * Linear layers (row strides in floats given explicitly, so strided views need no copy).
 *   d = nh * 48, head h: q1 = q[.., 48h .. 48h+23], q2 = q[.., 48h+24 .. 48h+47]   (T:687, 712-713)
 *   k likewise in kv[.., 0 .. d), v = kv[.., d + 48h .. d + 48h + 47]               (T:690, 702-703)
 *   out[t, 48h+e] = 0.2 * w_subln[e] * rmsnorm_e( sum_j (s1_j - lam * s2_j) v_j[e] )
 *                   + lepe_b[48h+e] + sum_j lepe_w[48h+e][j] v_j[48h+e]
 *   s1 = softmax_j(scale * q1.k1_j), s2 = softmax_j(scale * q2.k2_j) over the in-image 3x3 window.
 *   `lam` is read from device memory (1 float) so the call stays graph-capturable.
 * ------------------------------------------------------------------------------------------ */
int mlagg_local_attn_fwd(const float *q, int q_stride, const float *kv, int kv_stride,
                         const float *lam, const float *subln_w, const float *lepe_w, const float *lepe_b,
                         float *out, int out_stride, int batch, int H, int W, int nh, float scale,
                         void *stream);
/* Backward: gather form, no atomics on dq/dkv.  `workspace` = mlagg_local_attn_bwd_workspace_floats().
 * dlam / dsubln_w are ACCUMULATED into (caller zero-fills); dlepe_w / dlepe_b are written. */
size_t mlagg_local_attn_bwd_workspace_floats(int batch, int H, int W, int nh);
int mlagg_local_attn_bwd(const float *q, int q_stride, const float *kv, int kv_stride,
                         const float *lam, const float *subln_w, const float *lepe_w,
                         const float *dout, int dout_stride,
                         float *dq, int dq_stride, float *dkv, int dkv_stride,
                         float *dlam, float *dsubln_w, float *dlepe_w, float *dlepe_b, float *workspace,
                         int batch, int H, int W, int nh, float scale, void *stream);

/* ------------------------------------------------------------------------------------------
 * K4: pooled differential attention + RMSNorm, the `local=False` branch (T:733-760 variant A via
 * flash_attn_func, T:762-777 variant B); supersedes the four `flash_attn_func(q_j, k_j, v_i)` calls.
 *   q (batch, N, d) unscaled projection; kp, vp (batch, P, d) pooled keys / values (kv(x_) halves);
 *   `scale` is the total logit scale: head_dim^-0.5 for variant B, 1/head_dim for variant A.
 *   out[t, 48h+e] = 0.2 * w_subln[e] * rmsnorm_e( sum_p (s1_p - lam * s2_p) vp[p, 48h+e] )
 * ------------------------------------------------------------------------------------------ */
int mlagg_pooled_attn_fwd(const float *q, int q_stride, const float *kp, int kp_stride,
                          const float *vp, int vp_stride, const float *lam, const float *subln_w,
                          float *out, int out_stride,
                          float *lse,   /* (batch, N, nh, 2) log-sum-exp of both maps, or NULL (inference) */
                          float *o_pre, /* (batch, N, d) output before the RMSNorm, or NULL (inference) */
                          int batch, int N, int P, int nh, float scale, void *stream);
/* Backward needs the forward's lse / o_pre and mlagg_pooled_attn_bwd_workspace_floats() of scratch.
 * dq / dkp / dvp are overwritten; dlam / dsubln_w are ACCUMULATED into (caller zero-fills). */
size_t mlagg_pooled_attn_bwd_workspace_floats(int batch, int N, int P, int nh);
int mlagg_pooled_attn_bwd(const float *q, int q_stride, const float *kp, int kp_stride,
                          const float *vp, int vp_stride, const float *lam, const float *subln_w,
                          const float *dout, int dout_stride, const float *lse, const float *o_pre,
                          float *dq, int dq_stride, float *dkp, int dkp_stride, float *dvp, int dvp_stride,
                          float *dlam, float *dsubln_w, float *workspace,
                          int batch, int N, int P, int nh, float scale, void *stream);

/* ------------------------------------------------------------------------------------------
 * K2: depthwise 3x3 convolution (zero padding 1, stride 1) + bias (+ SiLU when `silu`), token-major
 * x, y (batch, H*W, C) with row strides in floats; w (C, 9) = Conv2d weight (C, 1, 3, 3) flattened.
 * Replaces nn.Conv2d(groups=C) at nnUNetTrainer_MLAgg_2D_dt_MS.py:890 (dwc + SiLU), T:781-782 (LePE of
 * the pooled branch), MambaSkip.py:521-523 (conv2d + SiLU) and M:553 (ConvolutionalGLU.dwconv).
 *   pre: (batch, H*W, C) contiguous pre-activation saved for backward when silu (NULL otherwise /
 *        inference).  Backward WRITES dw (C, 9) and dbias (C) (no zero-fill needed).
 *   res: NULL, or (batch, H*W, C) contiguous, added to the output (silu == 0 only): x + lepe(v) of T:782 in the same pass.
 * ------------------------------------------------------------------------------------------ */
int mlagg_dwconv3x3_fwd(const float *x, int x_stride, const float *w, const float *bias, const float *res, float *y,
                        int y_stride, float *pre, int batch, int H, int W, int C, int silu, void *stream);
size_t mlagg_dwconv3x3_bwd_workspace_floats(int batch, int H, int W, int C);
/* Gated form: y = SiLU(conv(x) + bias) * gate -- `self.fc2(self.act(self.dwconv(x, H, W)) * v)` of ConvolutionalGLU (MambaSkip.py:559-577)
 * with the product in the convolution's epilogue; gate (batch, H*W, C) with row stride gate_stride (the v half of fc1's output).  Backward
 * also writes dgate = dy * SiLU(pre) (row stride dgate_stride).  Workspace as for mlagg_dwconv3x3_bwd. */
int mlagg_dwconv3x3_gated_fwd(const float *x, int x_stride, const float *w, const float *bias, const float *gate, int gate_stride, float *y,
                              int y_stride, float *pre, int batch, int H, int W, int C, void *stream);
int mlagg_dwconv3x3_gated_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride, const float *pre,
                              const float *gate, int gate_stride, float *dx, int dx_stride, float *dgate, int dgate_stride, float *dw,
                              float *dbias, float *workspace, int batch, int H, int W, int C, void *stream);
int mlagg_dwconv3x3_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride,
                        const float *pre, float *dx, int dx_stride, float *dw, float *dbias, float *workspace,
                        int batch, int H, int W, int C, int silu, void *stream);

/* ------------------------------------------------------------------------------------------
 * K5w: weight / bias gradient of a token-major Linear,  dW (O, I) = dy^T x,  db (O) = column sums of dy,
 * with dy (M, O) and x (M, I) (row strides in floats), M = batch * tokens.  Replaces the library GEMM in the
 * backward of the nn.Linear layers at nnUNetTrainer_MLAgg_2D_dt_MS.py:687-690, 887-907 and
 * MambaSkip.py:518, 538, 572-575 (split-K over tokens on fp32 MFMA).  dW / db are overwritten.
 * ------------------------------------------------------------------------------------------ */
/* K5: the same layers' forward y (M, N) = x (M, K) . w (N, K)^T + bias and input gradient
 * dx (M, I) = dy (M, O) . w (O, I) on fp32 MFMA (w contiguous; x / y / dy / dx with row strides; K and I % 4 == 0). */
int mlagg_linear_fwd(const float *x, int x_stride, const float *w, const float *bias, float *y, int y_stride,
                     int M, int N, int K, void *stream);
int mlagg_linear_dgrad(const float *dy, int dy_stride, const float *w, float *dx, int dx_stride, int M, int O, int I,
                       void *stream);
/* Mixed precision: the same two products with the operands rounded to bf16 / fp16 (dtype = MLAGG_DTYPE_BF16 / _F16) on
 * their way into the matrix cores and fp32 accumulation -- what torch.autocast makes of nn.Linear in the reference's
 * default train step (mlagg/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:848-851; fp16 + GradScaler :152, bf16 in
 * BASELINE configs[2]).  x, w, y / dy, dx stay fp32 in memory, same layouts and strides as above.
 * dtype = MLAGG_DTYPE_BF16X3: the fp32 layers on the 16-bit matrix instructions -- each fp32 operand as three bf16 pieces, six
 * partial products accumulated in fp32 (error 2^-24 of a product: as accurate as the fp32 instruction, 2.7x its rate). */
int mlagg_linear_lp_fwd(const float *x, int x_stride, const float *w, const float *bias, float *y, int y_stride,
                        int M, int N, int K, int dtype, void *stream);
int mlagg_linear_lp_dgrad(const float *dy, int dy_stride, const float *w, float *dx, int dx_stride, int M, int O, int I,
                          int dtype, void *stream);
size_t mlagg_linear_wgrad_workspace_floats(int M, int O, int I);
int mlagg_linear_wgrad(const float *dy, int dy_stride, const float *x, int x_stride, float *dW, float *db,
                       float *workspace, int M, int O, int I, void *stream);
/* The same gradient with every fp32 operand as three bf16 pieces on v_mfma_f32_32x32x16_bf16 (six partial products, fp32
 * accumulation: the error of the fp32 instruction at 2.7x less matrix-pipe time; same workspace, same layouts). */
int mlagg_linear_wgrad_x3(const float *dy, int dy_stride, const float *x, int x_stride, float *dW, float *db, float *workspace,
                          int M, int O, int I, void *stream);

/* ------------------------------------------------------------------------------------------
 * Boundary #3: `flash_attn.flash_attn_func(q, k, v, causal=False)` as called four times per pooled AggregatedAttention
 * (nnUNetTrainer_MLAgg_2D_dt_MS.py:173, 745-750): out = softmax(q k^T * softmax_scale) v per (batch, head), 16-bit
 * tensors (dtype = MLAGG_DTYPE_BF16 / _F16), fp32 arithmetic.  Serves mlagg_unet_amd.shims.flash_attn_func (the
 * reference's own model file running unmodified); the product network uses the fused K4 launch instead.
 *   q, out, dout, dq  (B, N, nh, head_dim)     head_dim must be 24 (every MLAgg-UNet stage), P <= 512
 *   k, v              (B, P, nh, head_dim)
 *   lse               (B, nh, N) fp32, written by fwd when not NULL (needed by bwd)
 *   workspace         mlagg_flash_attn_bwd_workspace_floats() floats; after bwd its LAST B*P*nh*2*head_dim floats hold
 *                     the fp32 gradients (B, P, nh, 2, head_dim): [..., 0, :] = dk, [..., 1, :] = dv
 * ------------------------------------------------------------------------------------------ */
int mlagg_flash_attn_fwd(const void *q, const void *k, const void *v, void *out, float *lse, int B, int N, int P, int nh,
                         int head_dim, float softmax_scale, int dtype, void *stream);
size_t mlagg_flash_attn_bwd_workspace_floats(int B, int N, int P, int nh, int head_dim);
int mlagg_flash_attn_bwd(const void *q, const void *k, const void *v, const void *out, const void *dout, const float *lse,
                         void *dq, float *workspace, int B, int N, int P, int nh, int head_dim, float softmax_scale,
                         int dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * K6: LayerNorm over the last dimension of token-major rows, x (rows, C) with row stride x_stride,
 * y (rows, C) contiguous.  Replaces nn.LayerNorm at nnUNetTrainer_MLAgg_2D_dt_MS.py:723, 887, 907, 984-1001 and
 * MambaSkip.py:536, 741-742.  Built for C in {48, 96, 192, 384, 768} (mlagg_layernorm_supported).
 *   stats: (rows, 2) mean / rstd saved for backward (NULL for inference).
 *   backward overwrites dx (rows, C), dgamma (C), dbeta (C or NULL).
 * ------------------------------------------------------------------------------------------ */
int mlagg_layernorm_supported(int C);
int mlagg_layernorm_fwd(const float *x, int x_stride, const float *gamma, const float *beta, float *y,
                        float *stats, int rows, int C, float eps, void *stream);
size_t mlagg_layernorm_bwd_workspace_floats(int rows, int C);
int mlagg_layernorm_bwd(const float *x, int x_stride, const float *dy, int dy_stride, const float *gamma,
                        const float *stats, float *dx, float *dgamma, float *dbeta, float *workspace,
                        int rows, int C, void *stream);
/* K6 with the residual junction in front of the norm in the same pass (nnUNetTrainer_MLAgg_2D_dt_MS.py:905-907: x = shortcut +
 * drop_path(out_proj(..)); x = x + drop_path(mlp(norm2(x))) -- and norm1 of the next block of the stage, T:887):
 *   xsum = skip + branch * scale[row / rows_per_sample],   y = LayerNorm(xsum) * gamma + beta
 * skip, branch, xsum, y (rows, C) contiguous; scale: one stochastic-depth factor per sample (rows_per_sample rows each), NULL = 1.
 * Backward: dskip = LayerNorm'(dy) + dres (dres: gradient reaching xsum from its other consumers, NULL = none), dbranch = dskip *
 * scale (NULL exactly when scale is: the two gradients coincide); workspace: mlagg_layernorm_bwd_workspace_floats(rows, C). */
int mlagg_residual_layernorm_fwd(const float *skip, const float *branch, const float *scale, const float *gamma, const float *beta,
                                 float *xsum, float *y, float *stats, int rows, int rows_per_sample, int C, float eps, void *stream);
int mlagg_residual_layernorm_bwd(const float *xsum, const float *dy, int dy_stride, const float *dres, const float *scale,
                                 const float *gamma, const float *stats, float *dskip, float *dbranch, float *dgamma, float *dbeta,
                                 float *workspace, int rows, int rows_per_sample, int C, void *stream);

/* ------------------------------------------------------------------------------------------
 * K2v: depthwise 3x3x3 convolution (zero padding 1, stride 1) + bias (+ SiLU when `silu`) on token-major volumes:
 * x, y (batch, D*H*W, C) with row strides in floats (multiples of 4, C % 4 == 0); w (C, 27) = Conv3d weight (C, 1, 3, 3, 3)
 * flattened, tap index kd * 9 + kh * 3 + kw.  Replaces the depthwise nn.Conv3d + SiLU of SS3D
 * (variants/mamba/UMambaEnc_SS3D.py:166-174, 331-333).  pre: (batch, D*H*W, C) contiguous pre-activation saved for
 * backward when silu (NULL otherwise / inference).  Backward writes dx, dw (C, 27) and dbias (C, may be NULL).
 * ------------------------------------------------------------------------------------------ */
int mlagg_dwconv3d_fwd(const float *x, int x_stride, const float *w, const float *bias, float *y, int y_stride,
                       float *pre, int batch, int D, int H, int W, int C, int silu, void *stream);
size_t mlagg_dwconv3d_bwd_workspace_floats(int batch, int D, int H, int W, int C);
int mlagg_dwconv3d_bwd(const float *x, int x_stride, const float *w, const float *dy, int dy_stride, const float *pre,
                       float *dx, int dx_stride, float *dw, float *dbias, float *workspace, int batch, int D, int H,
                       int W, int C, int silu, void *stream);

/* ------------------------------------------------------------------------------------------
 * K2n: depthwise 3x3 convolution on NCHW maps (zero padding 1, stride 1 or 2) + bias.  x (B, C, H, W),
 * y (B, C, Ho, Wo), w (C, 9).  Replaces nn.Conv2d(groups=C) of MedNeXtBlock.conv1 / MedNeXtDownBlock.conv1
 * (nnUNetTrainer_MLAgg_2D_dt_MS.py:256-263, 310, 349-356).  Backward overwrites dx, dw (C, 9), dbias (C or NULL).
 * ------------------------------------------------------------------------------------------ */
int mlagg_dwconv3x3_nchw_fwd(const float *x, const float *w, const float *bias, float *y, int B, int C, int H, int W,
                             int stride, void *stream);
size_t mlagg_dwconv3x3_nchw_bwd_workspace_floats(int B, int C, int H, int W, int stride);
int mlagg_dwconv3x3_nchw_bwd(const float *x, const float *w, const float *dy, float *dx, float *dw, float *dbias,
                             float *workspace, int B, int C, int H, int W, int stride, void *stream);
/* ... dx = data gradient + dres (NULL: none): the gradient of the block's residual connection on the same map (`x1 = x + ...`,
 * T:256-300) summed inside the kernel; needs stride 1 and W % 4 == 0. */
int mlagg_dwconv3x3_nchw_bwd_res(const float *x, const float *w, const float *dy, const float *dres, float *dx, float *dw, float *dbias,
                                 float *workspace, int B, int C, int H, int W, int stride, void *stream);

/* ------------------------------------------------------------------------------------------
 * K1': cross-scan / cross-merge of SS2D_skip.forward_corev0 (MambaSkip.py:414-422 and 455-471 + 534).
 *   tok: token-major (B, L_cat, tok_stride floats per token); direction k reads/writes CB channels at
 *        column k * blk_stride (nblk = 4) or at column 0 for every direction (nblk = 1);
 *   seq: (B, 4 * CB, L_cat), row k * CB + c = direction k in scan order (k=0 row-major, 1 column-major,
 *        2/3 their reversals; scales concatenated in order).  H, W: host arrays of the nscale (<= 4) map sizes.
 * cross_scan: tok -> seq.  cross_merge: the adjoint, seq -> tok (sums the four directions when nblk = 1).
 * ------------------------------------------------------------------------------------------ */
int mlagg_cross_scan(const float *tok, int tok_stride, int blk_stride, float *seq, int B, int nscale,
                     const int *H, const int *W, int CB, int nblk, void *stream);
int mlagg_cross_merge(const float *seq, float *tok, int tok_stride, int blk_stride, int B, int nscale,
                      const int *H, const int *W, int CB, int nblk, void *stream);

/* ------------------------------------------------------------------------------------------
 * K18: 1 x 1 convolutions (stride 1, no padding, groups 1) on channel-major maps, fp32 operands as three bf16 pieces on the 16-bit
 * matrix instructions (six partial products, fp32 accumulation: the accuracy of the fp32 instruction).  Replaces the library
 * convolution behind nn.Conv2d(kernel_size=1) at nnUNetTrainer_MLAgg_2D_dt_MS.py:279-316 (MedNeXtBlock conv2 / conv3), :319-367
 * (down / up blocks) and :972-1001 (Project).
 *   fwd:   y (B, O, P) = w (O, I) . x (B, I, P) (+ bias[o], NULL: none); P = H * W pixels; x_batch / y_batch: floats between
 *          samples (a channel slice of a wider map is a valid operand).  The data gradient is the same call on w^T (I, O) and dy.
 *   wgrad: dW (O, I) = sum_b dy (B, O, P) . x (B, I, P)^T, overwritten; workspace: mlagg_conv1x1_wgrad_workspace_floats floats.
 * Supported (mlagg_conv1x1_supported): contraction I % 16 == 0, P % 16 == 0, P >= 96; anything else MLAGG_E_UNSUPPORTED (the
 * caller keeps the library convolution).
 * ------------------------------------------------------------------------------------------ */
int mlagg_conv1x1_supported(int O, int I, long P);
int mlagg_conv1x1_fwd(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B, int O, int I,
                      long P, void *stream);
size_t mlagg_conv1x1_wgrad_workspace_floats(int B, int O, int I, long P);
int mlagg_conv1x1_wgrad(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B, int O,
                        int I, long P, void *stream);
/* The same two products in the operand form `dtype`: MLAGG_DTYPE_BF16X3 = the calls above; MLAGG_DTYPE_BF16 / MLAGG_DTYPE_F16 = the
 * reference's autocast train step (nnUNetTrainer.py:848; BASELINE configs[2] / [4]), where the convolution's operands are rounded to
 * 16 bits and the sums are fp32: operands rounded ONCE in registers (nearest even), one product; x, w, y / dW stay fp32 in memory. */
int mlagg_conv1x1_fwd_lp(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B, int O, int I,
                         long P, int dtype, void *stream);
int mlagg_conv1x1_wgrad_lp(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B, int O,
                           int I, long P, int dtype, void *stream);
/* Forward form with a ragged contraction: x has I_valid <= I channels, w is (O, I) with I % 16 == 0 and zero columns from I_valid on
 * (the segmentation heads, nnUNetTrainer_MLAgg_2D_dt_MS.py:549-561 OutBlock: the data gradient of a 14-class head contracts over 14). */
int mlagg_conv1x1_fwd_ragged(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B, int O, int I,
                             int I_valid, long P, int dtype, void *stream);
/* ... accumulate != 0: y += w . x.  The data gradient of the second convolution that reads a map (UnetResBlock's conv3 beside conv1 on the
 * same input, MONAI structure behind T:1340-1368) is added to the first one's inside the kernel instead of by an add_ over the map. */
int mlagg_conv1x1_fwd_acc(const float *x, long x_batch, const float *w, const float *bias, float *y, long y_batch, int B, int O, int I,
                          int I_valid, long P, int dtype, int accumulate, void *stream);

/* ------------------------------------------------------------------------------------------
 * K19: dense 3 x 3 convolutions (stride 1, zero padding 1, groups 1) on channel-major maps as nine shifted GEMMs on the 16-bit matrix
 * instructions (fp32 operands as three bf16 pieces, six partial products, fp32 accumulation).  Replaces the library convolution
 * behind nn.Conv2d(kernel_size=3, padding=1) in UnetResBlock conv1 / conv2 (nnUNetTrainer_MLAgg_2D_dt_MS.py:1340-1368 via
 * UnetrBasicBlock / UnetrUpBlock), Project (T:972-1001) and the MSMM conv branches (MambaSkip.py:706-712): forward, and -- with
 * `transposed`, on the layer's forward weight (I_layer = O here) and its output gradient -- the data gradient.
 *   y (B, O, H, W) = conv3x3(x (B, I, H, W), w) (+ bias[o], NULL: none); x_batch / y_batch: floats between samples;
 *   workspace: mlagg_conv3x3_workspace_bytes(O, I) bytes, 16-byte aligned (the pre-split weight image, rebuilt by every call).
 * Supported (mlagg_conv3x3_supported): contraction I % 16 == 0, H * W >= 96.
 * ------------------------------------------------------------------------------------------ */
int mlagg_conv3x3_supported(int O, int I, int H, int W);
size_t mlagg_conv3x3_workspace_bytes(int O, int I);
int mlagg_conv3x3_fwd(const float *x, long x_batch, const float *w, int transposed, const float *bias, float *y, long y_batch,
                      void *workspace, int B, int O, int I, int H, int W, void *stream);
/* The same kernel for 3 x 3 x 3 convolutions (stride 1, padding 1) on (B, C, D, H, W) volumes -- nine kernel rows (kz, ky) of three
 * taps: the forward / data gradient of `nn.Conv3d(kernel_size=3, padding=1)` in the 3-D network (variants/mamba/UMambaEnc_SS3D.py:49-66
 * BasicResBlock / BasicBlockD convolutions, :477-513, 589-637, 744-779), straight on the unpadded volumes; w (O, I, 3, 3, 3).
 * workspace: mlagg_conv3x3x3_workspace_bytes(O, I). */
int mlagg_conv3x3x3_supported(int O, int I, int D, int H, int W);
size_t mlagg_conv3x3x3_workspace_bytes(int O, int I);
int mlagg_conv3x3x3_fwd(const float *x, long x_batch, const float *w, int transposed, const float *bias, float *y, long y_batch,
                        void *workspace, int B, int O, int I, int D, int H, int W, void *stream);
/* Weight gradient of the same convolution: dW (O, I, 3, 3) = sum_b sum_p dy (B, O, H, W) [.] x (B, I, H, W) shifted by the tap,
 * overwritten (inside `convolution_backward` of the layers above); W % 8 == 0 and H W % 16 == 0 (mlagg_conv3x3_wgrad_supported); workspace:
 * mlagg_conv3x3_wgrad_workspace_floats floats.  dy 16-byte aligned. */
int mlagg_conv3x3_wgrad_supported(int O, int I, int H, int W);
/* ... and of the 3 x 3 x 3 convolution: dW (O, I, 3, 3, 3) from dy (B, O, D, H, W) and x (B, I, D, H, W), unpadded volumes (one wave
 * per kernel slice kz); W % 8 == 0 and D H W % 16 == 0 (mlagg_conv3x3x3_wgrad_supported), else the caller keeps K15. */
int mlagg_conv3x3x3_wgrad_supported(int O, int I, int D, int H, int W);
size_t mlagg_conv3x3x3_wgrad_workspace_floats(int B, int O, int I, int D, int H, int W);
int mlagg_conv3x3x3_wgrad(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B, int O,
                          int I, int D, int H, int W, void *stream);
size_t mlagg_conv3x3_wgrad_workspace_floats(int B, int O, int I, int H, int W);
int mlagg_conv3x3_wgrad(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B, int O,
                        int I, int H, int W, void *stream);
/* The 3 x 3 products in the operand form `dtype` (see mlagg_conv1x1_fwd_lp; workspaces as above). */
int mlagg_conv3x3_fwd_lp(const float *x, long x_batch, const float *w, int transposed, const float *bias, float *y, long y_batch,
                         void *workspace, int B, int O, int I, int H, int W, int dtype, void *stream);
int mlagg_conv3x3_wgrad_lp(const float *dy, long dy_batch, const float *x, long x_batch, float *dW, float *workspace, int B, int O,
                           int I, int H, int W, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * K17: key / value reduction of the pooled attention branch, pooled (B, (H/r)(W/r), d) = r x r window mean of GELU(s), s (B, H W, d)
 * token-major at row stride s_stride (a column block of the stacked q | v | sr projection).  Replaces nn.GELU + nn.AdaptiveAvgPool2d
 * at nnUNetTrainer_MLAgg_2D_dt_MS.py:722 (modules at :668, :671) for H % r == W % r == 0 (other sizes: MLAGG_E_UNSUPPORTED, the
 * caller keeps the library's adaptive pooling).  Backward overwrites ds (B, H W, d) at row stride ds_stride.
 * ------------------------------------------------------------------------------------------ */
int mlagg_gelu_pool_fwd(const float *s, int s_stride, float *pooled, int batch, int H, int W, int d, int r, void *stream);
int mlagg_gelu_pool_bwd(const float *s, int s_stride, const float *dpooled, float *ds, int ds_stride, int batch, int H, int W, int d,
                        int r, void *stream);

/* ------------------------------------------------------------------------------------------
 * K7: gate of the MLLA block, out (rows, 2h) = concat(a0, a1) * SiLU(act) with a0, a1 (rows, h) contiguous and
 * act (rows, 2h) at row stride act_stride.  Replaces SiLU + torch.cat + product at
 * nnUNetTrainer_MLAgg_2D_dt_MS.py:888, 899, 902.  Backward overwrites da0, da1 (rows, h) and dact (rows, 2h).
 * ------------------------------------------------------------------------------------------ */
int mlagg_gate_fwd(const float *a0, const float *a1, const float *act, int act_stride, float *out, long rows, int h,
                   void *stream);
int mlagg_gate_bwd(const float *dout, int dout_stride, const float *a0, const float *a1, const float *act, int act_stride,
                   float *da0, float *da1, float *dact, int dact_stride, long rows, int h, void *stream);

/* ------------------------------------------------------------------------------------------
 * K8: small fused ops.
 * diff_lambda: lam[0] = exp(<q1, k1>) - exp(<q2, k2>) + lambda_init over n-element vectors (the differential-attention
 * lambda of nnUNetTrainer_MLAgg_2D_dt_MS.py:709-711 / 770-772); saved_exp[2] receives the two exponentials for the
 * backward, which overwrites dq1, dk1, dq2, dk2 (n each) from dlam[0].
 * scaled_residual: out = skip + branch * scale[b] for b in [0, batch), per_sample floats per sample (multiple of 4),
 * everything contiguous -- the residual under timm DropPath (T:903, 907; MambaSkip.py:741, 745), scale = mask / keep.
 * skip == NULL computes out = branch * scale[b] (the branch gradient).
 * ------------------------------------------------------------------------------------------ */
int mlagg_diff_lambda_fwd(const float *q1, const float *k1, const float *q2, const float *k2, float lambda_init, int n,
                          float *lam, float *saved_exp, void *stream);
int mlagg_diff_lambda_bwd(const float *dlam, const float *q1, const float *k1, const float *q2, const float *k2,
                          const float *saved_exp, int n, float *dq1, float *dk1, float *dq2, float *dk2, void *stream);
int mlagg_scaled_residual(const float *skip, const float *branch, const float *scale, float *out, int batch,
                          long per_sample, void *stream);
/* transpose_2d: dst[b][c][r] = src[b][r][c], (batch, R, C) -> contiguous (batch, C, R); src matrices contiguous, src_batch_stride
 * elements apart (0 = R * C; a channel slice of an NCHW map has a larger one): the NCHW <-> token-major flips
 * (`x.flatten(2).transpose(1, 2)` / its inverse at nnUNetTrainer_MLAgg_2D_dt_MS.py:878-880, 910; MambaSkip.py:727-733, 747-751). */
int mlagg_transpose_2d(const float *src, long src_batch_stride, float *dst, int batch, int R, int C, void *stream);
/* ... with the results dst_batch_stride floats apart (0 = R * C): the gradient of a channel slice of an NCHW map written where the
 * map's gradient lives (the halves of `x.split([48, C - 48], dim=1)` in MambaSkip.py:727-733 share one gradient buffer). */
int mlagg_transpose_2d_into(const float *src, long src_batch_stride, float *dst, long dst_batch_stride, int batch, int R, int C,
                            void *stream);
/* pixel_shuffle2: dst (B, O, 2 H, 2 W)[b][o][2 i + a][2 j + c] = src (B, 4 O, H, W)[b][(2 a + c) O + o][i][j]; inverse != 0: the other
 * way round (src is the (B, O, 2 H, 2 W) map).  Around K18 this is the kernel-2 / stride-2 transposed convolution of UnetrUpBlock
 * (nnUNetTrainer_MLAgg_2D_dt_MS.py:1340-1368) and its backward.  W even, both pointers 16-byte aligned, contiguous maps. */
int mlagg_pixel_shuffle2(const float *src, float *dst, int B, int O, int H, int W, int inverse, void *stream);
/* ... the inverse with the (B, O, 2 H, 2 W) source a channel slice of a wider map: src_batch floats between samples (0: contiguous) */
int mlagg_pixel_unshuffle2_strided(const float *src, long src_batch, float *dst, int B, int O, int H, int W, void *stream);
/* Bias gradients.  channel_sum: out[c] = sum over batch and pixels of an NCHW gradient map g (B, C, HW) -- the bias gradient of
 * the convolutions around the path (torch computes it with a generic reduction inside convolution_backward); workspace:
 * mlagg_channel_sum_workspace_floats(B, C) floats.  column_sum: out[c] = sum_r x[r][c], x (rows, cols) at row stride x_stride --
 * the bias gradient of the Linear layers whose GEMMs go to the library (few tokens); workspace: mlagg_column_sum_workspace_floats(rows,
 * cols) floats (0 for short matrices; NULL: single launch, one workgroup per 64 columns). */
size_t mlagg_channel_sum_workspace_floats(int B, int C);
int mlagg_channel_sum(const float *g, float *out, float *workspace, int B, int C, long HW, void *stream);
size_t mlagg_column_sum_workspace_floats(int rows, int cols);
int mlagg_column_sum(const float *x, int x_stride, float *out, float *workspace, int rows, int cols, void *stream);

/* ------------------------------------------------------------------------------------------
 * K1' for volumes: cross-scan / cross-merge by permutation table, the re-ordering of the 3-D selective scan
 * (`SS3D.forward_corev0`, mlagg/nnunetv2/training/nnUNetTrainer/variants/mamba/UMambaEnc_SS3D.py:244-296: K = 12 directions
 * = six axis orders of (D, H, W) and their reversals; the stack / permute / flip / cat chains at :251-259 and :284-295).
 *   idx  (K, L) int32: natural (d, h, w) position of scan step l of direction k
 *   scan   seq[b][k * CB + c][l] = tok[b][idx[k][l]][k * blk_stride + c]       tok (B, L, *) rows of tok_stride floats
 *   merge  the transpose; blk_stride = 0 sums the K directions into the same CB columns (tok_stride must equal CB; float
 *          atomics: the sum order is not fixed)
 * ------------------------------------------------------------------------------------------ */
int mlagg_index_scan(const float *tok, long tok_stride, int blk_stride, const int *idx, float *seq, int B, int L, int K, int CB,
                     void *stream);
int mlagg_index_merge(const float *seq, const int *idx, float *tok, long tok_stride, int blk_stride, int B, int L, int K, int CB,
                      void *stream);
/* out (rows, CB) = sum of the K column blocks of wide (rows, K * CB), in a fixed order (CB % 4 == 0): with
 * mlagg_index_merge(blk_stride = CB) the deterministic form of the summed merge (torch.sum(y, dim=1), UMambaEnc_SS3D.py:338). */
int mlagg_block_sum(const float *wide, float *out, long rows, int K, int CB, void *stream);

/* ------------------------------------------------------------------------------------------
 * K1f: the whole of SS2D_skip.forward_corev0 behind x_proj (reference MambaSkip.py:405-473) + the four-way sum of M:534 on
 * TOKEN-MAJOR tensors: the four scan orders of every scale (M:414-422), the dt einsum (M:430-436), `selective_scan_fn(xs, dts,
 * As, Bs, Cs, Ds, z=None, delta_bias, delta_softplus=True)` (M:445-451) and the inverse re-orderings (M:455-471) in the scan
 * kernels' own address arithmetic -- no scan-order copy of x, of the projections or of any gradient exists.
 * Fixed shape (every MLAgg-UNet configuration): 4 directions, d_inner 96, d_state 16, dt_rank 3; L % 4 == 0
 * (mlagg_msmm_scan_supported says whether a shape qualifies; others take mlagg_cross_scan + mlagg_selscan_lowrank_*).
 *   xc    (B, L, 96)   conv outputs of all scales concatenated, natural token order (u of all four directions)
 *   xdbl  (B, L, 144)  x_proj output with the weight rows laid out per direction as [dt0 dt1 dt2 0 | B(16) | C(16)]
 *   idx   (4, L) int32 token visited by direction k at scan position t (k = 0: row-major, 1: column-major, 2 / 3: their reversals
 *         inside every scale; scales concatenated in the same order for every direction); every entry must be in [0, L)
 *   Wdt (384, 3) dt_projs_weight; A (384, 16) = -exp(A_logs); D (384) or NULL; delta_bias (384) or NULL
 *   y     (B, L, 96)   sum over the four directions at the token's natural position
 *   state saved for backward (mlagg_msmm_scan_state_floats); workspace: the per-direction outputs before the sum
 * Backward overwrites dxc (B, L, 96), dxdbl (B, L, 144: every column, the pad columns with zeros), dWdt, dA and, when non-NULL,
 * dD, ddelta_bias.  No atomics: bit-reproducible.
 * ------------------------------------------------------------------------------------------ */
int mlagg_msmm_scan_supported(int d_inner, int d_state, int dt_rank, int directions, int L);
size_t mlagg_msmm_scan_state_floats(int batch, int L);
size_t mlagg_msmm_scan_fwd_workspace_floats(int batch, int L);
size_t mlagg_msmm_scan_bwd_workspace_floats(int batch, int L);
int mlagg_msmm_scan_fwd(const float *xc, const float *xdbl, const int *idx, const float *Wdt, const float *A, const float *D,
                        const float *delta_bias, float *y, float *state, float *workspace, int batch, int L, void *stream);
int mlagg_msmm_scan_bwd(const float *xc, const float *xdbl, const int *idx, const float *Wdt, const float *A, const float *D,
                        const float *delta_bias, const float *dy, const float *state, float *dxc, float *dxdbl, float *dWdt,
                        float *dA, float *dD, float *ddelta_bias, float *workspace, int batch, int L, void *stream);

/* ------------------------------------------------------------------------------------------
 * K1s: selective scan with ONE state per channel on token-major volumes, scan orders applied inside the kernels.
 * Replaces, for the reference's 3-D network (variants/mamba/UMambaEnc_SS3D.py: `SS3D` built with d_state = 1 at :640-655),
 * everything of `SS3D.forward_corev0` (:244-296) behind x_proj: the copies that build the K = 12 scan sequences of u
 * (:251-259), the dt einsum (:264), `selective_scan_fn(xs, dts, As, Bs, Cs, Ds, z=None, delta_bias, delta_softplus=True)`
 * (:277-283) and the inverse permutations (:285-296); the caller finishes `torch.sum(y, dim=1)` (:338) with mlagg_block_sum.
 *   tok   (B, L, C) conv output u in natural (d, h, w) token order, rows of tok_stride floats; C % 64 == 0
 *   idx   (K, L) int32: token visited at step l of direction k
 *   dtr   (B, K, R, L), Bs, Cs (B, K, L): x_proj's output in scan order (mlagg_index_scan); R in {1, 2, 3, 4, 8, 16, 20}
 *   Wdt   (K*C, R) dt_projs_weight; A (K*C) = -exp(A_logs); D (K*C); bias (K*C) dt_projs_bias
 *   yk    (B, L, K, C): y of direction k at the natural position of its token
 *   state saved for backward, mlagg_selscan1_state_floats() floats: chunk-entry states, chunk sums of delta, states entering
 *         every 16-step tile
 * Backward: dout (B, L, C) (rows of dout_stride floats) = gradient of the K-way sum; writes duk (B, L, K, C) (the caller sums the
 * K blocks), ddtr / dBs / dCs in scan order, and dparams (K*C, 3 + R) = [dA | dD | dbias | dWdt] per channel.
 * mlagg_selscan1_chunk: the chunk length the kernels cut L into (diagnostics).
 * ------------------------------------------------------------------------------------------ */
int mlagg_selscan1_chunk(int B, int L, int K);
size_t mlagg_selscan1_state_floats(int B, int L, int C, int K);
int mlagg_selscan1_fwd(const float *tok, long tok_stride, const int *idx, const float *dtr, const float *Bs, const float *Cs,
                       const float *Wdt, int R, const float *A, const float *D, const float *bias, float *yk, float *state,
                       int B, int L, int C, int K, void *stream);
size_t mlagg_selscan1_bwd_workspace_floats(int B, int L, int C, int K, int R);
int mlagg_selscan1_bwd(const float *tok, long tok_stride, const int *idx, const float *dtr, const float *Bs, const float *Cs,
                       const float *Wdt, int R, const float *A, const float *D, const float *bias, const float *dout,
                       long dout_stride, const float *state, float *duk, float *ddtr, float *dBs, float *dCs, float *dparams,
                       float *workspace, int B, int L, int C, int K, void *stream);

/* ------------------------------------------------------------------------------------------
 * K13: epilogue of the library convolutions on NCHW maps (B, C, HW): y = act(x + bias[c] + res), act 0 = none (in place on
 * x, y ignored), 1 = GELU (erf form; x is overwritten with the pre-activation, y receives the result).  Replaces the
 * bias add / GELU / residual add chain behind the convolutions of MedNeXtBlock, MedNeXtDownBlock, PatchExpand and project
 * (nnUNetTrainer_MLAgg_2D_dt_MS.py:307-324, 358-366, 498-546, 984-1001).  bias, res may be NULL.
 * Backward of the GELU form: dpre = dy * gelu'(pre), dbias[c] = sum of dpre over batch and pixels (NULL to skip;
 * workspace: mlagg_channel_sum_workspace_floats(B, C) floats).
 * ------------------------------------------------------------------------------------------ */
int mlagg_channel_epilogue_fwd(float *x, const float *bias, const float *res, float *y, int B, int C, long HW, int act,
                               void *stream);
int mlagg_channel_gelu_bwd(const float *pre, const float *dy, float *dpre, float *dbias, float *workspace, int B, int C,
                           long HW, void *stream);
/* The same epilogue in the 16-bit modes, with the element type of every map in memory as an argument (MLAGG_DTYPE_*; HW % 4 == 0):
 * x is the bf16 / fp16 output of the library convolution and is left untouched, y = act(x + bias[c] + res) is written in y_dtype
 * (16-bit when only the next 16-bit convolution reads it, fp32 when it joins the residual stream): no cast kernels on either side
 * of the convolution.  Backward: dx = dy * act'(x + bias + res) in dx_dtype (the convolution's gradient operand) and dbias in one
 * pass (act 0: x / bias / res may be NULL).  Arithmetic fp32. */
int mlagg_channel_epilogue_lp_fwd(const void *x, int x_dtype, const float *bias, const void *res, int res_dtype, void *y, int y_dtype,
                                  int B, int C, long HW, int act, void *stream);
int mlagg_channel_epilogue_lp_bwd(const void *x, int x_dtype, const float *bias, const void *res, int res_dtype, const void *dy,
                                  int dy_dtype, void *dx, int dx_dtype, float *dbias, float *workspace, int B, int C, long HW, int act,
                                  void *stream);

/* ------------------------------------------------------------------------------------------
 * K9: Dice + cross-entropy statistics and gradient of one deep-supervision level.  Replaces softmax, one-hot scatter,
 * masked products, spatial sums, log_softmax + nll and all their backward kernels of DC_and_CE_loss
 * (loss/compound_losses.py:31-57, loss/dice.py:73-117, loss/robust_ce_loss.py:12-16).
 * logits (B, C, HW) fp32, 2 <= C <= mlagg_dice_ce_max_classes(); target (B, HW) float labels in [0, C).
 * stats: stats_ip[b][0][c] = sum_p softmax_c [y == c], stats_ip[b][1][c] = sum_p softmax_c,
 * stats_g[b][c] = sum_p [y == c], ce_sum[0] = sum_{b,p} -log softmax_y.
 * grad: dlogits = d(loss)/d(logits) for upstream gradients g_ip (B, 2, C) of stats_ip and g_ce[0] of ce_sum (device scalars:
 * no host synchronisation).
 * ignore_label >= 0: pixels carrying that label take no part in any sum and get a zero gradient -- the loss mask of
 * DC_and_CE_loss(ignore_label=...) (loss/compound_losses.py:38-50; the number of valid pixels is sum_{b,c} stats_g); -1: none.
 */
int mlagg_dice_ce_max_classes(void);
/*
 * stats OVERWRITES its three outputs: per-workgroup partial rows in `workspace` (mlagg_dice_ce_stats_workspace_floats floats) are
 * summed in a fixed order -- the value of the loss is bit-reproducible from run to run.
 * ------------------------------------------------------------------------------------------ */
size_t mlagg_dice_ce_stats_workspace_floats(int B, int C, long HW);
int mlagg_dice_ce_stats(const float *logits, const float *target, float *stats_ip, float *stats_g, float *ce_sum,
                        float *workspace, int B, int C, long HW, int ignore_label, void *stream);
int mlagg_dice_ce_grad(const float *logits, const float *target, const float *g_ip, const float *g_ce, float *dlogits, int B,
                       int C, long HW, int ignore_label, void *stream);

/* ------------------------------------------------------------------------------------------
 * K10: per-plane normalisation of an NCHW map fused with the activation that follows it:
 *   y = act((x - mean_bc) * rstd_bc * gamma_c + beta_c),  statistics over the HW pixels of each (batch, channel) plane.
 * gamma / beta (C) may be NULL (1 / 0).  act: 0 none, 1 LeakyReLU(slope), 2 SiLU.  res (B, C, HW) or NULL is added before the
 * activation (the residual sum of the UnetResBlock, MambaSkip.py:662-666): y = act(norm(x) + res); backward then also writes
 * dres (may be NULL).  Replaces nn.GroupNorm(C, C)
 * (nnUNetTrainer_MLAgg_2D_dt_MS.py:268-270, 357, 500-502), nn.InstanceNorm2d + LeakyReLU(0.01) of the MONAI UnetResBlock
 * (T:1339-1357; structure at MambaSkip.py:581-667) and nn.InstanceNorm2d(affine) + SiLU (MambaSkip.py:700-706).
 * stats (B*C, 2) receives mean and rstd for the backward, which overwrites dx and, when non-NULL, dgamma / dbeta (C);
 * also nn.InstanceNorm3d(affine) + LeakyReLU of the 3-D network (variants/mamba/UMambaEnc_SS3D.py:477-513): planes of 256 Ki
 * elements and more are cut into 16 Ki-element segments, one workgroup each, with exact pooled statistics.
 * workspace: mlagg_plane_norm_fwd_workspace_floats / _bwd_workspace_floats(B, C, HW) floats (forward: 0 floats -- NULL allowed --
 * unless the planes are cut into segments; backward: needed with dgamma / dbeta or segments).
 * ------------------------------------------------------------------------------------------ */
size_t mlagg_plane_norm_fwd_workspace_floats(int B, int C, long HW);
/* x_dtype / res_dtype / y_dtype (forward), x_dtype / dy_dtype / res_dtype (backward; dx has x's type, dres res's): MLAGG_DTYPE_F32,
 * _BF16 or _F16 -- the element type of each map IN MEMORY (arithmetic and statistics are fp32).  In the 16-bit modes a map that only
 * a 16-bit library convolution reads or wrote is kept 16-bit (the reference's autocast tensors, nnUNetTrainer.py:848): no cast
 * kernels around the convolutions, half the bytes through this kernel. */
int mlagg_plane_norm_fwd(const void *x, const float *gamma, const float *beta, const void *res, void *y, float *stats,
                         float *workspace, int B, int C, long HW, float eps, int act, float slope, int x_dtype, int res_dtype,
                         int y_dtype, void *stream);
size_t mlagg_plane_norm_bwd_workspace_floats(int B, int C, long HW);
int mlagg_plane_norm_bwd(const void *x, const void *dy, const float *gamma, const float *beta, const void *res, const float *stats,
                         void *dx, void *dres, float *dgamma, float *dbeta, float *workspace, int B, int C, long HW, int act,
                         float slope, int x_dtype, int dy_dtype, int res_dtype, void *stream);
/* ... with dy a channel slice of a wider map (the halves of `torch.cat([up, skip], 1)`'s gradient, T:1360-1366): dy_batch elements
 * between samples, a multiple of HW (0: contiguous) -- no copy of the slice in front of the kernel. */
int mlagg_plane_norm_bwd_strided(const void *x, const void *dy, long dy_batch, const float *gamma, const float *beta, const void *res,
                                 const float *stats, void *dx, void *dres, float *dgamma, float *dbeta, float *workspace, int B, int C,
                                 long HW, int act, float slope, int x_dtype, int dy_dtype, int res_dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * K4lp: pooled differential attention on the 16-bit matrix cores (fp32 tensors in memory; q * scale, k, v, the softmax weights and
 * d(o) rounded to bf16 / fp16 operands, fp32 sums / softmax / RMSNorm).  Replaces, under the reference's autocast step, the four
 * `flash_attn_func(q_j, k_j, v_i, causal=False)` launches of the pooled branch, the lambda subtraction, RMSNorm and 0.2 gain
 * (nnUNetTrainer_MLAgg_2D_dt_MS.py:733-760) and their backward.  Layouts and strides as mlagg_pooled_attn_fwd / _bwd; P <= 320.
 * Saved for backward: lse (B, N, nh, 2) and the two maps' own outputs o1 = P1 v, o2 = P2 v (B, N, d) (NULL, NULL, NULL: inference).
 * dtype: MLAGG_DTYPE_BF16 / MLAGG_DTYPE_F16.  Backward overwrites dq, dkp, dvp, dlam[1], dsubln_w[48]; fixed summation order.
 * ------------------------------------------------------------------------------------------ */
int mlagg_pooled_attn_lp_fwd(const float *q, int q_stride, const float *kp, int kp_stride, const float *vp, int vp_stride,
                             const float *lam, const float *subln_w, float *out, int out_stride, float *lse, float *o1, float *o2,
                             int batch, int N, int P, int nh, float scale, int dtype, void *stream);
size_t mlagg_pooled_attn_lp_bwd_workspace_floats(int batch, int N, int P, int nh);
int mlagg_pooled_attn_lp_bwd(const float *q, int q_stride, const float *kp, int kp_stride, const float *vp, int vp_stride,
                             const float *lam, const float *subln_w, const float *dout, int dout_stride, const float *lse,
                             const float *o1, const float *o2, float *dq, int dq_stride, float *dkp, int dkp_stride, float *dvp,
                             int dvp_stride, float *dlam, float *dsubln_w, float *workspace, int batch, int N, int P, int nh,
                             float scale, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * K15: weight gradient of the full convolutions on channel-major maps as tap GEMMs on the fp32 matrix cores.  Replaces MIOpen's
 * weight-gradient solvers behind the backward of `nn.Conv3d` / `nn.Conv2d` (kernel 3 with padding 1, or kernel 1; stride 1 or 2) of
 * the 3-D network's BasicResBlock / BasicBlockD / UpsampleLayer / seg layers (variants/mamba/UMambaEnc_SS3D.py:49-66, 477-513,
 * 589-637, 744-779) and of the 2-D network's 3x3 / 1x1 convolutions (nnUNetTrainer_MLAgg_2D_dt_MS.py:976-977, 278-296; MambaSkip.py:703).
 *   mlagg_conv_pad_geometry: box (Dq, Hq, Wq) and guard (floats in front of and behind every channel row) of the padded copies for an
 *       input of (D, H, W) voxels (D = 1: a 2-D map, stride 1 only).
 *   mlagg_volume_pad: src (B, C, D, H, W) -> dst (B, phases, C, 2 * guard + Dq * Hq * Wq): zero-padded copy (stride 1: one phase, data
 *       at origin 1) or the 8 parity phases of the zero-padded input (stride 2).  as_output = 1: an output-sized map (out_D, out_H,
 *       out_W) laid into the box of the input geometry (D, H, W) with a zero ring (origin 1 for stride 1, 0 for stride 2).
 *   wide = 1 (stride 1, W % 4 == 0): the data starts at x = 4 (Wq = W + 8): aligned groups of 4 padded voxels are aligned groups of 4
 *       data voxels -- the geometry mlagg_conv_taps needs; K15 works on either.
 *   mlagg_conv_wgrad_taps: dW (O, I, ntaps) (+)= sum_{b, q < Q} A[b][o][q] * B[b][i][q + tap_off[t]].  Q and the row strides are
 *       multiples of 4 floats (Q of 8), tap_off is a HOST array of ntaps <= 27 element offsets; workspace of
 *       mlagg_conv_wgrad_taps_workspace_floats() floats.  fp32 in, fp32 MFMA (exact products), fixed summation order.
 * ------------------------------------------------------------------------------------------ */
int mlagg_conv_pad_geometry(int D, int H, int W, int stride, int wide, int *Dq, int *Hq, int *Wq, long *guard);
int mlagg_volume_pad(const float *src, float *dst, int B, int C, int D, int H, int W, int stride, int wide, int as_output, int out_D,
                     int out_H, int out_W, void *stream);
size_t mlagg_conv_wgrad_taps_workspace_floats(int batch, long Q, int O, int I, int ntaps);
/* K16: forward / data gradient of the same convolutions (stride 1; kernel 3 pad 1 or kernel 1) on the padded copy (wide form):
 * y (B, O, D, H, W) = sum_t sum_i weight[o * w_so + i * w_si + t'] * xp[b][i][q + tap_off[t]], t' = t (flip 0) or ntaps - 1 - t (flip 1: the data
 * gradient, with w_so / w_si exchanged by the caller).  Replaces MIOpen's Conv3d forward / backward-data behind BasicResBlock / BasicBlockD /
 * UpsampleLayer / seg layers of the 3-D network (UMambaEnc_SS3D.py:49-66, 477-513, 589-637, 744-779).  fp32 MFMA, exact products. */
int mlagg_conv_taps(const float *xp, long x_batch, long x_row, const float *weight, long w_so, long w_si, int flip, const long *tap_off,
                    int ntaps, float *y, int B, int O, int I, int D, int H, int W, void *stream);
int mlagg_conv_wgrad_taps(const float *A, long a_batch, long a_row, const float *B, long b_batch, long b_row, const long *tap_off,
                          int ntaps, long Q, int O, int I, int batch, float *dW, int accumulate, float *workspace, void *stream);

/* ------------------------------------------------------------------------------------------
 * K5, round-4 form (csrc/linear_x3.hip): y (M, N) = x (M, K) . W (N, K)^T for token-major activations, fp32 in / out, split-bf16
 * products (three exact bf16 pieces per operand, six partial products, fp32 accumulation: the error of an fp32 GEMM).  Replaces the
 * forward and data-gradient GEMMs behind every token-major nn.Linear of the path (nnUNetTrainer_MLAgg_2D_dt_MS.py:176-192, 687-690,
 * 719-723, 887-907; MambaSkip.py:518, 538, 559-577) at every token count.
 * The weight operand is an IMAGE: mlagg_weight_image lays the three bf16 pieces of W (N, K) out as img[piece][n][k] with rows padded
 * with zeros to a multiple of 32 (mlagg_weight_image_bytes(N, K) bytes) and, if img_t is non-NULL, the same for W^T (K, N)
 * (mlagg_weight_image_bytes(K, N) bytes): the operand of the data gradient dx (M, K) = dy (M, N) . W = mlagg_linear_x3(dy, img_t, ...,
 * N_gemm = K, K_gemm = N).  mlagg_weight_images builds the images of a whole network in one launch from a device table of rows
 * {const float *w; void *img; void *img_t; int32 N, K, w_stride, pad} (40 bytes each); max_tiles >= ceil(N / 32) * ceil(K / 32) of
 * every row.  K % 8 == 0 (mlagg_linear_x3_supported).
 * epilogue 0: y = x W^T + bias (bias may be NULL); 1: y = x W^T + bias AND y_act = GELU(y) (exact, erf: Mlp fc1 + nn.GELU, T:188-190);
 * 2: y = (x W^T) * GELU'(pre) (the data gradient that flows back through that GELU; bias ignored).
 * ------------------------------------------------------------------------------------------ */
size_t mlagg_weight_image_bytes(int rows, int cols);
int mlagg_weight_image(const float *w, int w_stride, void *img, void *img_t, int N, int K, void *stream);
int mlagg_weight_images(const void *jobs, int n_jobs, int max_tiles, void *stream);
int mlagg_linear_x3_supported(int M, int N, int K);
int mlagg_linear_x3(const float *x, int x_stride, const void *w_image, const float *bias, float *y, int y_stride, float *y_act,
                    const float *pre, int pre_stride, int M, int N, int K, int epilogue, void *stream);

/* ------------------------------------------------------------------------------------------
 * K11: clip_grad_norm_ + AdamW.step() of the train step (nnUNetTrainer.py:855-857; optimizer of
 * nnUNetTrainer_MLAgg_2D_dt_MS.py:137-147) for every parameter in two launches.
 * tensor_table: device array of rows {param*, grad*, exp_avg*, exp_avg_sq*, int64 numel} (5 x 8 bytes each);
 * work_list: device array of int32 pairs (tensor index, chunk index), one per mlagg_adamw_chunk_elements() elements;
 * sumsq: 1 + n_work device doubles (scratch: [0] the squared global norm, then one partial per work item, summed in a fixed
 * order: bit-reproducible).  max_norm > 0: gradients are scaled by min(1, max_norm / (||g||_2 + 1e-6)) as they are
 * read (the global norm is formed on the device; no host synchronisation; gradients in memory stay unscaled); <= 0: no clipping.
 * step: 1-based step count for the bias corrections.  Arithmetic of torch.optim.AdamW (decoupled decay, amsgrad off).
 * ------------------------------------------------------------------------------------------ */
int mlagg_adamw_chunk_elements(void);
int mlagg_adamw_clip_step(const void *tensor_table, const void *work_list, int n_work, double *sumsq, float lr, float beta1,
                          float beta2, float eps, float weight_decay, float max_norm, int step, void *stream);
/* The same step for a hipGraph capture of the whole train step (B:833-863 replayed as one graph): nothing that changes between steps
 * is a launch argument.  lr_dev: one device float (the cosine schedule writes it between replays, B:825); step_dev: one device int32,
 * advanced by this call before the bias corrections read it.  With max_norm <= 0 the norm pass is skipped, the counter still moves. */
int mlagg_adamw_clip_step_dev(const void *tensor_table, const void *work_list, int n_work, double *sumsq, const float *lr_dev,
                              int *step_dev, float beta1, float beta2, float eps, float weight_decay, float max_norm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MLAGG_HIP_H */
