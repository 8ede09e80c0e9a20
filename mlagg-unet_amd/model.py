"""MLAgg-UNet 2D network for MI355X: the module tree behind
``nnUNetTrainer_MLAgg_2D_dt_MS.build_network_architecture`` (reference
nnUNetTrainer_MLAgg_2D_dt_MS.py:62-92, 1183-1407) with the token mixers on hand-written HIP kernels.

What runs where:
  * HIP (libmlagg_hip.so): K1 selective scan of the MSMM skip module, K2 depthwise 3x3 (+SiLU) on
    token-major maps, K3 3x3-window differential attention (+RMSNorm+LePE), K4 pooled differential
    attention (+RMSNorm);
  * PyTorch-ROCm (rocBLAS / MIOpen / ATen): Linear layers, full convolutions of the stem, the
    down/up blocks, the decoder and the heads, LayerNorm/GroupNorm/InstanceNorm.
Module and parameter names equal the reference's, so checkpoints (525 state_dict keys) load either
way.  There is no eager fallback for the HIP ops: on a machine without the library the forward raises.

Layout choice: the encoder keeps activations token-major (B, N, C) across both MLLA blocks of a
stage (the reference flips NCHW <-> NLC per block and again around every depthwise conv); NCHW is
produced once per stage for the MIOpen convolutions that consume the stage output.
"""
import math
import os
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

LAMBDA_INIT = 0.8


class _DropPathPool:
    """All stochastic-depth masks of one forward pass from ONE bernoulli launch.  The reference draws a fresh
    (B, 1, 1) mask per DropPath call (timm): 19 x (bernoulli_, div_) launches per step at config 2.  The first
    training forward records the order of keep probabilities; later forwards draw an (n_calls, B) table up front
    and hand out rows in that order."""

    active = None

    def __init__(self):
        self.order, self.keep, self.masks, self.pos, self.recording = [], None, None, 0, False
        self.injected, self.forced = None, False

    def inject(self, factors):
        """(n_calls, batch) factors -- keep mask / keep probability, in call order -- that the NEXT training forward uses instead of
        drawing its own (parity tests against a reference run whose DropPath draws were recorded)."""
        self.injected = factors

    def begin(self, batch, device):
        if self.injected is not None:
            self.masks = self.injected.to(device=device, dtype=torch.float32).contiguous()
            if self.masks.dim() != 2 or self.masks.shape[1] != batch:
                raise RuntimeError("injected DropPath factors must be (n_calls, batch)")
            self.injected, self.forced, self.pos = None, True, 0
        elif not self.order:
            self.recording = True
        else:
            if self.keep is None or self.keep.device != device or self.keep.shape[1] != batch:
                self.keep = torch.tensor(self.order, device=device, dtype=torch.float32)[:, None].expand(-1, batch).contiguous()
            self.masks = torch.bernoulli(self.keep).div_(self.keep)
            self.pos = 0
        _DropPathPool.active = self

    def end(self):
        forced, self.forced = self.forced, False
        self.recording = False
        _DropPathPool.active = None
        if forced and self.pos != self.masks.shape[0]:
            raise RuntimeError(f"{self.masks.shape[0]} DropPath factors were injected, the forward made {self.pos} calls")

    def next(self, keep, batch):
        if self.forced:
            if self.pos >= self.masks.shape[0]:
                raise RuntimeError("more DropPath calls than injected factors")
            self.pos += 1
            return self.masks[self.pos - 1]
        if self.recording:
            self.order.append(keep)
            return None
        if self.pos >= len(self.order) or self.order[self.pos] != keep or self.masks.shape[1] != batch:
            raise RuntimeError("DropPath call order changed between forward passes")
        self.pos += 1
        return self.masks[self.pos - 1]


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm semantics; reference T:868, M:688)."""

    def __init__(self, p: float = 0.0):
        super().__init__()
        self.drop_prob = float(p)

    def scale_mask(self, x):
        """Per-sample keep mask (B,) already divided by the keep probability, or None when inactive."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        pool = _DropPathPool.active
        if pool is not None:
            row = pool.next(keep, x.shape[0])
            if row is not None:
                return row
        return torch.empty(x.shape[0], device=x.device, dtype=x.dtype).bernoulli_(keep).div_(keep)

    def forward(self, x):
        m = self.scale_mask(x)
        return x if m is None else x * m.view((-1,) + (1,) * (x.dim() - 1))

    def residual_norm(self, skip, branch, norm):
        """(x, norm(x)) with x = skip + drop_path(branch): the junction and the LayerNorm that follows it in one K6 pass."""
        if branch.is_cuda and ops.FUSED_RESIDUAL_NORM:
            return ops.residual_layer_norm(skip, branch, self.scale_mask(branch), norm.weight, norm.bias, norm.eps)
        x = self.residual(skip, branch)
        return x, norm(x)

    def residual(self, skip, branch):
        """skip + drop_path(branch) in one pass (K8)."""
        m = self.scale_mask(branch)
        if m is None:
            return skip + branch
        if branch.is_cuda and (branch.numel() // branch.shape[0]) % 4 == 0:
            return ops.scaled_residual(skip, branch, m)
        return torch.addcmul(skip, branch, m.view((-1,) + (1,) * (branch.dim() - 1)))


_NO_DROP = DropPath(0.0)


class _ConvMixin:
    """Shared call forms of the library convolutions (same parameter names as the torch modules):
      raw(x)                       the convolution alone, no bias: fp32 in fp32 mode, bf16 / fp16 in the 16-bit modes (operands cast,
                                   what the reference's autocast makes of it, B:848) -- for a consumer that reads that type itself;
      fused(x, res, act, dtype)    act(conv(x) + bias + res) with everything behind the convolution in one K13 pass; in the 16-bit
                                   modes the epilogue reads the 16-bit convolution output and writes ``dtype`` (default fp32: the
                                   map joins the residual stream) -- no cast kernel on either side;
      forward(x)                   conv(x) + bias as an fp32 map.
    In fp32 mode the weight gradient of dense 3x3 / 1x1 stride-1 convolutions may go to K15 (ops.K15_2D, off: MIOpen wins)."""

    def _lib_conv(self, x, w):
        raise NotImplementedError

    def _fp32_conv(self, x):
        return self._lib_conv(x, self.weight)

    def _lp_conv(self, x, form):
        return None

    def raw(self, x):
        cdt = ops.conv_dtype()
        if cdt == torch.float32:
            return self._fp32_conv(x)
        if ops.LP_K and x.dtype == torch.float32:
            y = self._lp_conv(x, ops.conv_form())                    # K18 / K19, one rounded product, fp32 map out (round 4)
            if y is not None:
                return y
        return self._lib_conv(x if x.dtype == cdt else x.to(cdt), self.weight.to(cdt))

    def fused(self, x, res=None, act=ops.EPI_NONE, dtype=None):
        if not x.is_cuda or getattr(self, "padding_mode", "zeros") != "zeros":
            y = self._eager(x)
            y = y if res is None else y + res
            return F.gelu(y) if act == ops.EPI_GELU else y
        y = self.raw(x)
        if y.dtype == torch.float32:
            return ops.channel_epilogue(y, self.bias, res if res is None or res.dtype == torch.float32 else res.float(), act)
        if y.numel() // (y.shape[0] * y.shape[1]) % 4 or not ops.LP_IO:
            return ops.channel_epilogue(y.float(), self.bias, None if res is None else res.float(), act)
        return ops.channel_epilogue_lp(y, self.bias, res, act, dtype or torch.float32)

    def _device_forward(self, x):
        y = self.raw(x)
        if y.dtype == torch.float32:
            return y if self.bias is None else ops.channel_bias(y, self.bias)
        if y.numel() // (y.shape[0] * y.shape[1]) % 4 or not ops.LP_IO:
            y = y.float()
            return y if self.bias is None else ops.channel_bias(y, self.bias)
        return ops.channel_epilogue_lp(y, self.bias, None, ops.EPI_NONE, torch.float32)


class Conv2d(_ConvMixin, nn.Conv2d):
    """nn.Conv2d on MIOpen without its bias; the bias, residual and GELU behind it are K13 (see _ConvMixin)."""

    def _lib_conv(self, x, w):
        return F.conv2d(x, w, None, self.stride, self.padding, self.dilation, self.groups)

    def _fp32_conv(self, x):
        if ops.conv1x1_supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return ops.conv1x1(x, self.weight)                       # K18
        if ops.conv3x3_supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return ops.conv3x3(x, self.weight)                       # K19
        if self.groups == 1 and ops.K15_2D and tuple(self.dilation) == (1, 1):
            return ops.conv_nd(x, self.weight, self.stride, self.padding)
        return self._lib_conv(x, self.weight)

    def _lp_conv(self, x, form):
        if ops.conv1x1_supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups, form):
            return ops.conv1x1(x, self.weight, form)
        if ops.conv3x3_supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups, form):
            return ops.conv3x3(x, self.weight, form)
        return None

    def _eager(self, x):
        return nn.Conv2d.forward(self, x)

    def forward(self, x):
        if not x.is_cuda or self.padding_mode != "zeros":
            return nn.Conv2d.forward(self, x)
        return self._device_forward(x)


class ConvTranspose2d(_ConvMixin, nn.ConvTranspose2d):
    def _lib_conv(self, x, w):
        return F.conv_transpose2d(x, w, None, self.stride, self.padding, self.output_padding, self.groups, self.dilation)

    def _pointwise(self):
        """A 1 x 1, stride-1 transposed convolution IS the 1 x 1 convolution with the weight's first two axes exchanged (the segmentation
        heads, T:549-561): it takes K18 like the other pointwise convolutions."""
        return (tuple(self.kernel_size) == (1, 1) and tuple(self.stride) == (1, 1) and tuple(self.padding) == (0, 0)
                and tuple(self.output_padding) == (0, 0) and self.groups == 1 and tuple(self.dilation) == (1, 1))

    def _k18(self, x, form):
        # only where all three products run on K18 (the heads on the 256 x 256 and 128 x 128 maps): on the small maps the library's
        # transposed-convolution solvers are the faster ones
        I, O = self.weight.shape[:2]
        P = int(x.shape[2] * x.shape[3])
        return (ops.K18_THIN and self._pointwise() and ops.conv1x1_supported(x, self.weight, (1, 1), (0, 0), (1, 1), 1, form)
                and ops._k18_product(O, I, P, form) and ops._k18_product(I, O, P, form))

    def _t2(self, x, form):
        return ops.conv_t2x2_supported(x, self.weight, self.stride, self.padding, self.output_padding, self.dilation, self.groups, form)

    def _fp32_conv(self, x):
        if self._k18(x, ops._DTYPE_BF16X3):
            return ops.conv1x1(x, self.weight.permute(1, 0, 2, 3))
        if self._t2(x, ops._DTYPE_BF16X3):
            return ops.conv_t2x2(x, self.weight)                     # K18 + pixel shuffle
        return self._lib_conv(x, self.weight)

    def _lp_conv(self, x, form):
        if self._k18(x, form):
            return ops.conv1x1(x, self.weight.permute(1, 0, 2, 3), form)
        if self._t2(x, form):
            return ops.conv_t2x2(x, self.weight, form)
        return None

    def _eager(self, x):
        return nn.ConvTranspose2d.forward(self, x)

    def forward(self, x, output_size=None):
        if not x.is_cuda or output_size is not None:
            return nn.ConvTranspose2d.forward(self, x, output_size)
        return self._device_forward(x)


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm (same parameter names); with one channel per group -- every use on the path -- K10."""

    def forward(self, x, out_dtype=None):
        if self.num_groups == self.num_channels and x.is_cuda and x.dim() == 4:
            return ops.plane_norm(x, self.weight, self.bias, self.eps, out_dtype=out_dtype)
        return super().forward(x)


def _chain_dtype(x):
    """Element type of a map that only 16-bit library convolutions read, in the current mode: bf16 / fp16 in the 16-bit modes when
    the kernels' vector path applies (plane size a multiple of 4), else None (= keep fp32)."""
    cdt = ops.conv_dtype()
    pixels = x.numel() // (x.shape[0] * x.shape[1])
    if cdt == torch.float32 or not ops.LP_IO or pixels % 4:
        return None
    if ops.LP_K and pixels >= ops.LP_K_MIN_PIXELS:       # the map's stride-1 convolutions are K18 / K19 here: they read fp32
        return None
    return cdt


def _instance_norm_act(norm, x, act=ops.ACT_NONE, slope=0.0, res=None, out_dtype=None):
    """act(nn.InstanceNorm2d `norm`(x) + res), fused on K10 (plain modules off the device)."""
    if x.is_cuda and not norm.track_running_stats:
        return ops.plane_norm(x, norm.weight, norm.bias, norm.eps, act, slope, res, out_dtype or torch.float32)
    y = norm(x) if res is None else norm(x) + res
    return F.leaky_relu(y, slope) if act == ops.ACT_LEAKY else (F.silu(y) if act == ops.ACT_SILU else y)


class Linear(nn.Linear):
    """nn.Linear (same parameter names) whose weight/bias gradients run on K5w for large token counts."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameter names) over the channel dimension, on K6."""

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = Linear(dim, hidden)
        self.fc2 = Linear(hidden, dim)

    def forward(self, x):
        if ops.mlp_supported(x, self.fc1.weight, self.fc2.weight):
            return ops.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)        # GELU in K5's epilogues
        return self.fc2(F.gelu(self.fc1(x)))


class RMSNormWeight(nn.Module):
    """Holds ``subln.weight`` (the normalisation itself is fused into K3 / K4)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))


WEIGHT_STACKS = os.environ.get("MLAGG_WEIGHT_STACKS", "1") == "1"      # 0: torch.cat per use (the round-2 form)
GLU_FUSED = os.environ.get("MLAGG_GLU_FUSED", "1") == "1"               # 0: the gated MLP's product as a torch multiplication
DWC_MERGED = os.environ.get("MLAGG_DWC_MERGED", "1") == "1"             # 0: one depthwise conv per channel half of the MLLA block


class _StackFn(torch.autograd.Function):
    """The stacked matrix as a function of its sources: forward hands out the (already refreshed) buffer, backward cuts the gradient
    into the sources' row blocks -- a view where a source is stacked whole, a zero-filled copy where only rows [lo, hi) of it are.
    The cutting runs on the leaf-gradient stream (ops._LeafStream): what comes out goes straight to the parameters' AccumulateGrad
    nodes, so the gradient of a stacked projection never has to be waited for inside backward."""

    @staticmethod
    def forward(ctx, buf, ranges, *params):
        ctx.ranges, ctx.shapes, ctx.sources = ranges, [tuple(p.shape) for p in params], list(params)
        ops.note_leaf_use(*params)
        return buf.detach()

    @staticmethod
    def backward(ctx, g):
        out, off = [None, None], 0
        with ops._LeafStream(g, ok=ops.leaf_single_use(ctx.sources)):
            for (lo, hi), shape in zip(ctx.ranges, ctx.shapes):
                piece = g[off:off + hi - lo]
                if lo == 0 and hi == shape[0]:
                    out.append(piece)
                else:
                    full = torch.zeros(shape, device=g.device, dtype=g.dtype)
                    full[lo:hi] = piece
                    out.append(full)
                off += hi - lo
        return tuple(out)


def _capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


class _Stack:
    """Several parameters (or row slices of parameters) stacked along dim 0 for ONE projection.  ``torch.cat`` per use cost 48
    launches per forward of the 256 x 256 network (24 weight + 24 bias stacks: 0.28 ms of device time and as much host time); here
    every stack owns a buffer that is rewritten only when a source CHANGED since the last copy (the sources' version counters, plus
    the epoch ``ops.invalidate_weight_images`` bumps when ClipAdamW rewrites parameters through raw pointers): the network refreshes
    all stale stacks with one multi-tensor copy at the start of a forward (``refresh_stacks``), a stack used outside that refreshes
    itself.  Two forward passes between optimizer steps therefore share one, unmodified, buffer -- the graph of the first stays
    valid for its backward -- and a stack can never serve weights older than its sources.  Inside a hipGraph capture the copy is
    always recorded (a replay must see the parameters of ITS step).  Not parameters, not buffers: state_dict keys are untouched."""

    def __init__(self, sources):
        self.sources, self.buf, self.sig = sources, None, None             # sources: bound method -> list of tensors / (tensor, lo, hi)

    @staticmethod
    def _norm(srcs):
        """[(parameter (or a view of one), lo, hi)]: a bare tensor is stacked whole, a tuple contributes its rows [lo, hi)."""
        return [(t[0], int(t[1]), int(t[2])) if isinstance(t, tuple) else (t, 0, int(t.shape[0])) for t in srcs]

    def _views(self, srcs):
        rows = sum(hi - lo for _, lo, hi in srcs)
        shape = (rows,) + tuple(srcs[0][0].shape[1:])
        dev, dt = srcs[0][0].device, srcs[0][0].dtype
        if self.buf is None or self.buf.device != dev or tuple(self.buf.shape) != shape:
            self.buf = torch.empty(shape, device=dev, dtype=dt)
            self.sig = None
        out, off = [], 0
        for _, lo, hi in srcs:
            out.append(self.buf[off:off + hi - lo])
            off += hi - lo
        return out

    @staticmethod
    def _rows(srcs):
        return [t.detach()[lo:hi] for t, lo, hi in srcs]

    @staticmethod
    def signature(srcs):
        return (ops.image_epoch(),) + tuple(t._version for t, _, _ in srcs) + tuple(t.data_ptr() for t, _, _ in srcs)

    def stale(self, srcs):
        return self.sig != self.signature(srcs) or _capturing(srcs[0][0])

    def get(self):
        srcs = self._norm(self.sources())
        if not WEIGHT_STACKS:
            return torch.cat([t[lo:hi] for t, lo, hi in srcs])
        views = self._views(srcs)
        if self.stale(srcs):
            with torch.no_grad():
                torch._foreach_copy_(views, self._rows(srcs))
            self.sig = self.signature(srcs)
        out = _StackFn.apply(self.buf, [(lo, hi) for _, lo, hi in srcs], *[t for t, _, _ in srcs])
        out._mlagg_buffer = self.buf                            # ops.WeightImageSet keeps THIS (no grad_fn), never the graph-bound view
        out._mlagg_leaf_safe = True                             # its gradient goes to AccumulateGrad nodes only (see _StackFn.backward)
        out._mlagg_sources = [t for t, _, _ in srcs]            # ... of THESE parameters (ops.leaf_single_use)
        return out


def refresh_stacks(stacks):
    """One multi-tensor copy for every stack of a network whose sources changed (called at the start of its forward)."""
    if not WEIGHT_STACKS:
        return
    dst, src, done = [], [], []
    for st in stacks:
        srcs = st._norm(st.sources())
        views = st._views(srcs)
        if st.stale(srcs):
            dst += views
            src += st._rows(srcs)
            done.append((st, srcs))
    if dst:
        with torch.no_grad():
            torch._foreach_copy_(dst, src)
    for st, srcs in done:
        st.sig = st.signature(srcs)


class AggregatedAttention(nn.Module):
    """Reference T:625-784 on token-major input.  ``variant`` "B": logit scale head_dim^-0.5 (fp32 path,
    T:762-777); "A": 1/head_dim, the shipped flash path's double scaling (T:688 + T:745-750)."""

    def __init__(self, dim, input_resolution, num_heads, local, sr_ratio, variant="B"):
        super().__init__()
        self.dim, self.num_heads, self.local, self.variant = dim, num_heads, local, variant
        self.head_dim = dim // num_heads // 2
        if self.head_dim != 24:
            raise RuntimeError("the gfx950 attention kernels are built for head_dim 24 (every MLAgg-UNet stage)")
        self.scale = self.head_dim ** -0.5
        for nm in ("lambda_q1", "lambda_k1", "lambda_q2", "lambda_k2"):
            setattr(self, nm, nn.Parameter(torch.zeros(self.head_dim).normal_(0, 0.1)))
        self.subln = RMSNormWeight(2 * self.head_dim)
        self.H, self.W = input_resolution
        if not local:
            self.sr_ratio = sr_ratio
            self.pool_H, self.pool_W = self.H // sr_ratio, self.W // sr_ratio
            self.sr = Conv2d(dim, dim, 1)
            self.norm = LayerNorm(dim)
        self.q = Linear(dim, dim)
        self.kv = Linear(dim, 2 * dim)
        self.lepe = Conv2d(dim, dim, 3, padding=1, groups=dim)
        self._w_stack, self._b_stack = _Stack(self._stacked_weights), _Stack(self._stacked_biases)

    def _stacked_weights(self):
        if self.local:
            return [self.q.weight, self.kv.weight]
        d = self.q.weight.shape[0]
        return [self.q.weight, (self.kv.weight, d, 2 * d), self.sr.weight.view(d, d)]       # the value half of kv only (T:719)

    def _stacked_biases(self):
        if self.local:
            return [self.q.bias, self.kv.bias]
        d = self.q.bias.shape[0]
        return [self.q.bias, (self.kv.bias, d, 2 * d), self.sr.bias]

    def lambda_full(self):
        return ops.diff_lambda(self.lambda_q1, self.lambda_k1, self.lambda_q2, self.lambda_k2, LAMBDA_INIT)

    def forward(self, x):
        """x: (B, N, dim) (may be a channel slice of a wider row) -> (B, N, dim)."""
        B, N, d = x.shape
        lam = self.lambda_full()
        if self.local:
            # q and kv in ONE GEMM over stacked weights (one read of x, one gradient into x); the kernel
            # takes the q / kv column blocks of the (B, N, 3d) result as strided views
            qkv = ops.linear(x, self._w_stack.get(), self._b_stack.get())
            # split_cols: the q / kv backward kernels write into one (B, N, 3d) gradient buffer (no concatenation)
            q, kv = ops.split_cols(qkv, (d, 2 * d))
            return ops.local_diff_attn(q, kv, lam, self.subln.weight, self.lepe.weight, self.lepe.bias,
                                       self.H, self.W, self.num_heads, self.scale)
        # q, the value half of kv (LePE input; k is discarded at full resolution, T:719) and the 1x1 `sr`
        # conv in ONE GEMM
        qvs = ops.linear(x, self._w_stack.get(), self._b_stack.get())
        q, v_full, s_pre = ops.split_cols(qvs, (d, d, d))
        if self.H % self.sr_ratio == 0 and self.W % self.sr_ratio == 0:
            r = self.sr_ratio
            if s_pre.is_cuda:
                pooled = ops.gelu_pool(s_pre, self.H, self.W, r)             # K17: GELU + window mean, one pass
            else:
                pooled = F.gelu(s_pre).view(B, self.pool_H, r, self.pool_W, r, d).mean(dim=(2, 4)).reshape(B, -1, d)
        else:
            img = F.gelu(s_pre).view(B, self.H, self.W, d).permute(0, 3, 1, 2)
            pooled = F.adaptive_avg_pool2d(img, (self.pool_H, self.pool_W)).flatten(2).transpose(1, 2)
        k_pool, v_pool = self.kv(self.norm(pooled)).split([d, d], dim=-1)
        scale = self.scale if self.variant == "B" else self.scale * self.scale
        o = ops.pooled_diff_attn(q, k_pool, v_pool, lam, self.subln.weight, self.num_heads, scale)
        return ops.dwconv3x3_nlc(v_full, self.lepe.weight, self.lepe.bias, self.H, self.W, silu=False, res=o)      # o + lepe(v), T:782


class MLLABlock(nn.Module):
    """Reference T:824-915.  ``forward`` keeps the reference's NCHW contract; stages call ``forward_tokens``."""

    def __init__(self, dim, input_resolution, num_heads, mlp_ratio, drop_path, sr_ratio, variant="B"):
        super().__init__()
        self.dim, self.input_resolution = dim, tuple(input_resolution)
        self.norm1 = LayerNorm(dim)
        self.in_proj = Linear(dim, dim)
        self.act_proj = Linear(dim, dim)
        self.dwc = Conv2d(dim, dim, 3, padding=1, groups=dim)
        self.attn = nn.ModuleList([
            AggregatedAttention(dim // 2, input_resolution, num_heads // 2, True, sr_ratio, variant),
            AggregatedAttention(dim // 2, input_resolution, num_heads // 2, False, sr_ratio, variant)])
        self.out_proj = Linear(dim, dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self._w_stack, self._b_stack = _Stack(self._stacked_weights), _Stack(self._stacked_biases)

    def _stacked_weights(self):
        return [self.act_proj.weight, self.in_proj.weight]

    def _stacked_biases(self):
        return [self.act_proj.bias, self.in_proj.bias]

    def forward_tokens(self, x, xn=None, next_norm=None):
        """x -> block(x).  ``xn``: norm1(x) when the caller already has it; ``next_norm``: the LayerNorm the caller applies to the
        result next (norm1 of the following block) -- the block then returns (x_out, next_norm(x_out)), formed in the pass of its last
        residual junction."""
        H, W = self.input_resolution
        C = self.dim
        if xn is None:
            xn = self.norm1(x)
        # act_proj and in_proj in ONE GEMM over stacked weights: (B, N, 2C) = [act | in]
        ai = ops.linear(xn, self._w_stack.get(), self._b_stack.get())
        h = C // 2
        if not DWC_MERGED:                                     # round-3 form: one depthwise conv per channel half
            act_pre, xa_in, za_in = ops.split_cols(ai, (C, h, h))
            wa, wz = self.dwc.weight.split([h, h], dim=0)
            ba, bz = self.dwc.bias.split([h, h], dim=0)
            xa = ops.dwconv3x3_nlc(xa_in, wa, ba, H, W, silu=True)
            za = ops.dwconv3x3_nlc(za_in, wz, bz, H, W, silu=True)
            return self._after_dwc(x, xa, za, act_pre, next_norm)
        act_pre, xz_in = ops.split_cols(ai, (C, C))
        # ONE depthwise conv over both channel halves (round 4: one launch instead of two in forward and in each of the three backward
        # kernels, and no split of dwc.weight / dwc.bias); the branches read their halves as column blocks of the result -- the
        # projection kernels take a row stride -- and write their input gradients into one buffer (split_cols)
        xz = ops.dwconv3x3_nlc(xz_in, self.dwc.weight, self.dwc.bias, H, W, silu=True)
        xa, za = ops.split_cols(xz, (h, h))
        return self._after_dwc(x, xa, za, act_pre, next_norm)

    def _after_dwc(self, x, xa, za, act_pre, next_norm):
        gated = ops.gate(self.attn[0](xa), self.attn[1](za), act_pre)       # K7: cat(.) * SiLU(act_proj(.))
        dp = self.drop_path if isinstance(self.drop_path, DropPath) else _NO_DROP
        x, n2 = dp.residual_norm(x, self.out_proj(gated), self.norm2)
        y = self.mlp(n2)
        if next_norm is not None:
            return dp.residual_norm(x, y, next_norm)
        return dp.residual(x, y)

    def forward(self, x):
        B, C, h, w = x.shape
        if (h, w) != self.input_resolution:
            raise RuntimeError("input feature has wrong size")
        return _TokensToMap.apply(self.forward_tokens(_MapToTokens.apply(x)), h, w)


def _flip(t3):
    """(B, R, C) -> contiguous (B, C, R): the library's tiled transpose on the device, a plain copy elsewhere."""
    if t3.is_cuda:
        return ops.transpose_2d(t3)
    return t3.transpose(1, 2).contiguous()


class _ZeroGradParam(torch.autograd.Function):
    """Ties a parameter whose effect on ``y`` cancels exactly (a convolution bias in front of an InstanceNorm) into the graph:
    y passes through, the parameter's gradient is the exact value, zero."""

    @staticmethod
    def forward(ctx, y, p):
        ctx.save_for_backward(p)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return g, torch.zeros_like(p)


class _MapToTokens(torch.autograd.Function):
    """(B, C, H, W) -> contiguous token-major (B, H*W, C); the gradient comes back as a contiguous NCHW map.
    Both directions are REAL transposes: permute + reshape alone is a strided view, and every residual add,
    LayerNorm and Linear of the stage (and their gradients) would then re-transpose it -- 62 us clones and
    114 us adds per use at stage 0."""

    @staticmethod
    def forward(ctx, x, slot=None):
        B, C, h, w = x.shape
        ctx.hw, ctx.slot = (h, w), slot
        return _flip(x.reshape(B, C, h * w))

    @staticmethod
    def backward(ctx, g):
        B, N, C = g.shape
        if ctx.slot is not None and g.is_cuda:
            return ops.transpose_2d_into(g, ctx.slot.view()), None          # a piece of ops.split_planes: into the map's gradient buffer
        return _flip(g).view(B, C, *ctx.hw), None


def _map_to_tokens(x):
    return _MapToTokens.apply(x, ops._claim(x) if x.is_cuda else None)


class _TokensToMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, h, w):
        B, N, C = t.shape
        return _flip(t).view(B, C, h, w)

    @staticmethod
    def backward(ctx, g):
        B, C, h, w = g.shape
        return _flip(g.reshape(B, C, h * w)), None, None


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, mlp_ratio, drop_path, sr_ratio, variant):
        super().__init__()
        self.blocks = nn.ModuleList([
            MLLABlock(dim, input_resolution, num_heads, mlp_ratio, drop_path[i], sr_ratio, variant)
            for i in range(depth)])

    def forward(self, x):
        B, C, h, w = x.shape
        # (B, C, H, W) -> token-major (B, N, C): the one transpose copy of the stage, and back at the end.
        # (Keeping the convolutional parts in torch.channels_last, which would make both free views, was
        # measured 2x slower end to end: MIOpen's fp32 NHWC kernels and the NHWC Group/InstanceNorms.)
        t = _MapToTokens.apply(x)
        tn = None
        for i, blk in enumerate(self.blocks):
            if i + 1 < len(self.blocks):
                t, tn = blk.forward_tokens(t, tn, self.blocks[i + 1].norm1)
            else:
                t = blk.forward_tokens(t, tn)
        return _TokensToMap.apply(t, h, w)


class Project(nn.Module):  # reference T:972-1001
    def __init__(self, cin, cout, stride, last):
        super().__init__()
        self.conv1 = Conv2d(cin, cout, 3, stride=stride, padding=1)
        self.conv2 = Conv2d(cout, cout, 3, stride=1, padding=1)
        self.norm1 = LayerNorm(cout)
        self.last = last
        if not last:
            self.norm2 = LayerNorm(cout)

    @staticmethod
    def _ln(norm, x):
        return _TokensToMap.apply(norm(_MapToTokens.apply(x)), x.shape[2], x.shape[3])

    def forward(self, x):
        x = self._ln(self.norm1, self.conv1.fused(x, act=ops.EPI_GELU))
        return self.conv2(x) if self.last else self._ln(self.norm2, self.conv2.fused(x, act=ops.EPI_GELU))


class PatchEmbed(nn.Module):  # reference T:1004-1043
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.proj1 = Project(in_chans, embed_dim // 2, (2, 2), last=False)
        self.proj2 = Project(embed_dim // 2, embed_dim, (patch_size // 2, patch_size // 2), last=True)

    def forward(self, x):
        return self.proj2(self.proj1(x))


class MedNeXtBlock(nn.Module):  # reference T:230-324
    def __init__(self, cin, cout, exp_r, k=3, do_res=True, stride=1):
        super().__init__()
        self.do_res = do_res
        self.conv1 = Conv2d(cin, cin, k, stride=stride, padding=k // 2, groups=cin)
        self.norm = GroupNorm(cin, cin)
        self.conv2 = Conv2d(cin, exp_r * cin, 1)
        self.conv3 = Conv2d(exp_r * cin, cout, 1)

    def body(self, x, res=None):
        # conv1 is depthwise 3x3 (stride 1, or 2 in the down block): K2n instead of MIOpen's naive fallback; the bias,
        # GELU and residual behind the two 1x1 convolutions are one K13 pass each
        pair = ops.dwconv3x3_nchw_res(x, self.conv1.weight, self.conv1.bias) if (res is x and self.conv1.stride[0] == 1) else None
        if pair is not None:
            x1, res = pair                       # the residual's gradient is summed inside K2n's data-gradient kernel
        else:
            x1 = ops.dwconv3x3_nchw(x, self.conv1.weight, self.conv1.bias, self.conv1.stride[0])
        # 16-bit modes: the maps between the norm and the two 1x1 convolutions are read only by 16-bit convolutions and stay 16-bit
        ct = _chain_dtype(x1)
        return self.conv3.fused(self.conv2.fused(self.norm(x1, ct), act=ops.EPI_GELU, dtype=ct), res=res)

    def forward(self, x):
        return self.body(x, x if self.do_res else None)


class MedNeXtDownBlock(MedNeXtBlock):  # reference T:327-366
    def __init__(self, cin, cout, exp_r, k=3):
        super().__init__(cin, cout, exp_r, k, do_res=False, stride=2)
        self.res_conv = Conv2d(cin, cout, 1, stride=2)

    def forward(self, x):
        # 16-bit modes: the strided 1x1 residual is read once, by conv3's epilogue: it may stay 16-bit
        return self.body(x, self.res_conv.fused(x, dtype=ops.conv_dtype()) if x.is_cuda else self.res_conv(x))


class PatchExpand(nn.Module):  # reference T:479-546
    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.res_conv = ConvTranspose2d(cin, cout, 1, stride=2)
        self.conv1 = ConvTranspose2d(cin, cout, k, stride=2, padding=k // 2)
        self.norm = GroupNorm(cin, cin)

    def forward(self, x):
        # pad(a) + pad(b) = pad(a + b): one padded copy, and the sum rides in conv1's epilogue
        ct = _chain_dtype(x)
        res = self.res_conv.fused(x, dtype=ops.conv_dtype()) if x.is_cuda else self.res_conv(x)
        return F.pad(self.conv1.fused(self.norm(x, ct), res=res), (1, 0, 1, 0))


class OutBlock(nn.Module):  # reference T:549-561
    def __init__(self, cin, n_classes):
        super().__init__()
        self.conv_out = ConvTranspose2d(cin, n_classes, 1)

    def forward(self, x):
        return self.conv_out(x)


class _ConvOnly(nn.Module):
    def __init__(self, cin, cout, k, stride=1, transposed=False):
        super().__init__()
        if transposed:
            self.conv = ConvTranspose2d(cin, cout, k, stride=stride, bias=False)
        else:
            self.conv = Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)

    def forward(self, x):
        return self.conv(x)


class UnetResBlock(nn.Module):
    """MONAI 1.3.0 UnetResBlock semantics (structure vendored in the reference at M:581-667)."""

    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.conv1 = _ConvOnly(cin, cout, k)
        self.conv2 = _ConvOnly(cout, cout, k)
        self.norm1 = nn.InstanceNorm2d(cout)
        self.norm2 = nn.InstanceNorm2d(cout)
        if cin != cout:
            self.conv3 = _ConvOnly(cin, cout, 1)
            self.norm3 = nn.InstanceNorm2d(cout)

    def forward(self, x):
        # 16-bit modes: every map of the block is written by / read by 16-bit convolutions only (the block output feeds the
        # transposed / 1x1 convolutions of decoder0 and the head): all of them stay 16-bit in memory, K10 converts on the fly
        ct = _chain_dtype(x) if x.is_cuda else None
        pair = None
        if hasattr(self, "conv3") and x.is_cuda and x.dtype == torch.float32 and (ops.conv_dtype() == torch.float32 or ops.LP_K):
            form = ops.conv_form()
            if ops.conv_pair_supported(x, self.conv1.conv, self.conv3.conv, form):
                pair = ops.conv_pair(x, self.conv1.conv.weight, self.conv3.conv.weight, form)   # one input gradient from both (K18 adds to K19's)
        c1 = pair[0] if pair is not None else (self.conv1.conv.raw(x) if x.is_cuda else self.conv1(x))
        out = _instance_norm_act(self.norm1, c1, ops.ACT_LEAKY, 0.01, None, ct)
        if hasattr(self, "conv3"):
            c3 = pair[1] if pair is not None else (self.conv3.conv.raw(x) if x.is_cuda else self.conv3(x))
            res = _instance_norm_act(self.norm3, c3, out_dtype=ct)
        else:
            res = x
        c2 = self.conv2.conv.raw(out) if x.is_cuda else self.conv2(out)
        return _instance_norm_act(self.norm2, c2, ops.ACT_LEAKY, 0.01, res, ct)                 # act(norm2(.) + res)


class UnetrBasicBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.layer = UnetResBlock(cin, cout)

    def forward(self, x):
        return self.layer(x)


class UnetrUpBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.transp_conv = _ConvOnly(cin, cout, 2, stride=2, transposed=True)
        self.conv_block = UnetResBlock(2 * cout, cout)

    def forward(self, x, skip):
        up = self.transp_conv.conv.raw(x) if x.is_cuda else self.transp_conv(x)
        if up.dtype != skip.dtype:                            # (a 16-bit chain met an fp32 map: plane size not a multiple of 4)
            up, skip = up.float(), skip.float()
        return self.conv_block(torch.cat([up, skip], dim=1))


class MLLA_Enc(nn.Module):  # reference T:1046-1179
    def __init__(self, img_size, patch_size, in_chans, embed_dim, depths, num_heads, mlp_ratio, drop_path_rate,
                 sr_ratio, variant):
        super().__init__()
        self.num_layers = len(depths)
        res = [s // patch_size for s in img_size]
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList([
            BasicLayer(embed_dim * 2 ** i, (res[0] // 2 ** i, res[1] // 2 ** i), depths[i], num_heads[i], mlp_ratio,
                       dpr[sum(depths[:i]):sum(depths[:i + 1])], sr_ratio[i], variant)
            for i in range(self.num_layers)])
        self.downs = nn.ModuleList([
            MedNeXtDownBlock(embed_dim * 2 ** i, embed_dim * 2 ** (i + 1), mlp_ratio)
            for i in range(self.num_layers - 1)])

    def forward(self, x):
        outs = [x]
        x = self.patch_embed(x)
        for i, layer in enumerate(self.layers):
            x = layer(x)
            outs.append(x)
            if i < self.num_layers - 1:
                x = self.downs[i](x)
        return outs


# ------------------------------------------------------------------------------------------------
# MSMM skip module (reference MambaSkip.py:266-577, 669-804)
# ------------------------------------------------------------------------------------------------
def _dt_init(dt_rank, d_inner, dt_min=0.001, dt_max=0.1, floor=1e-4):
    proj = nn.Linear(dt_rank, d_inner, bias=True)
    std = dt_rank ** -0.5
    nn.init.uniform_(proj.weight, -std, std)
    dt = torch.exp(torch.rand(d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=floor)
    with torch.no_grad():
        proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
    return proj


class SS2D_skip(nn.Module):
    def __init__(self, stage_num, d_model, d_state=16, expand=2):
        super().__init__()
        self.d_model, self.d_state = d_model, d_state
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16)
        self.in_proj = Linear(d_model, self.d_inner, bias=False)
        self.conv2d = nn.ModuleList([
            Conv2d(self.d_inner, self.d_inner, 3, padding=1, groups=self.d_inner) for _ in range(stage_num)])
        xp = [nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False) for _ in range(4)]
        self.x_proj_weight = nn.Parameter(torch.stack([t.weight for t in xp], 0))
        dts = [_dt_init(self.dt_rank, self.d_inner) for _ in range(4)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], 0))
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], 0))
        A = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(4 * self.d_inner, 1)
        self.A_logs = nn.Parameter(torch.log(A))
        self.Ds = nn.Parameter(torch.ones(4 * self.d_inner))
        self.out_norm = LayerNorm(self.d_inner)
        self.out_proj = Linear(self.d_inner, d_model, bias=False)

    def core(self, xc, HW):
        """xc: (B, L_cat, d_inner) token-major conv outputs of all scales -> (B, L_cat, d_inner): sum of the four
        re-ordered scan directions (reference M:405-473 and the 4-way sum at M:534).
        x_proj for all four directions is one token-major 96 -> 140 Linear on the natural token order, K1'
        (cross_scan / cross_merge) does every re-ordering, and the rank-3 dt projection lives inside K1;
        nothing is stacked, flipped or concatenated."""
        B, Lc, dI = xc.shape
        K, R, N = 4, self.dt_rank, self.d_state
        per = R + 2 * N
        if ops.msmm_scan_supported(xc, N, R):
            # K1f: one token-major 96 -> 4 * 36 Linear (a zero row keeps every direction's B / C block 16-byte aligned), then the scan
            # kernels apply the four scan orders in their own address arithmetic: nothing is re-ordered, stacked or summed outside
            xdbl = ops.linear(xc, ops.pad_x_proj(self.x_proj_weight))                     # (B, L, 4 * 36)
            return ops.msmm_scan(xc, xdbl, ops.msmm_scan_index(HW, xc.device), self.dt_projs_weight.reshape(K * dI, R),
                                 -torch.exp(self.A_logs), self.Ds, self.dt_projs_bias.reshape(-1))
        xdbl = ops.linear(xc, self.x_proj_weight.reshape(K * per, dI))                     # (B, L, 4*35)
        dtr, Bs, Cs = ops.cross_scan_bc(xdbl, HW, R, N)               # (B, 4, 3, L) | (B, 4, 16, L) | (B, 4, 16, L)
        xs = ops.cross_scan(xc, HW, dI, 1)                                                  # (B, 4*96, L)
        # delta = softplus(dt_projs_weight . dtr + bias) is formed inside K1: the (B, 4*96, L) delta tensor of
        # reference M:436 (334 MB at config 2) and its gradient never exist
        out = ops.selective_scan_lowrank_fn(xs, dtr, self.dt_projs_weight.reshape(K * dI, R), -torch.exp(self.A_logs),
                                            Bs, Cs, self.Ds, delta_bias=self.dt_projs_bias.reshape(-1),
                                            delta_softplus=True)
        return ops.cross_merge(out, HW, dI)

    def forward(self, x, HW, L_split):
        """x: (B, L_cat, d_model) -> (B, L_cat, d_model)."""
        parts = self.in_proj(x).split(list(L_split), dim=1)
        toks = [ops.dwconv3x3_nlc(xi, self.conv2d[i].weight, self.conv2d[i].bias, H, W, silu=True)
                for i, (xi, (H, W)) in enumerate(zip(parts, HW))]
        return self.out_proj(self.out_norm(self.core(torch.cat(toks, dim=1), HW)))


class _DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = Conv2d(dim, dim, 3, padding=1, groups=dim)


class ConvolutionalGLU(nn.Module):  # reference M:559-577
    def __init__(self, dim, hidden):
        super().__init__()
        hidden = int(2 * hidden / 3)
        self.hidden = hidden
        self.fc1 = Linear(dim, hidden * 2)
        self.dwconv = _DWConv(hidden)
        self.fc2 = Linear(hidden, dim)

    def forward(self, x, H, W):
        xg, vg = ops.split_cols(self.fc1(x), (self.hidden, self.hidden))      # the depthwise convolution's backward writes both halves in place
        if x.is_cuda and GLU_FUSED:
            # SiLU(dwconv(x)) * v in the convolution's epilogue (round 4: the product and its two backward products were ATen kernels)
            return self.fc2(ops.dwconv3x3_gated(xg, vg, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, H, W))
        g = ops.dwconv3x3_nlc(xg, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, H, W, silu=True)
        return self.fc2(g * vg)


class VSS_Conv_Block(nn.Module):  # reference M:669-753
    def __init__(self, feature_dims, hidden_dim, drop_path):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.conv_dims = [c - hidden_dim for c in feature_dims]
        self.ln_1 = LayerNorm(hidden_dim)
        self.self_attention = SS2D_skip(len(feature_dims), hidden_dim)
        self.drop_path = DropPath(drop_path)
        self.norm2 = LayerNorm(hidden_dim)
        self.mlps = nn.ModuleList([ConvolutionalGLU(hidden_dim, hidden_dim * 4) for _ in feature_dims])
        self.conv_branches = nn.ModuleList([
            nn.Sequential(Conv2d(c, c, 3, padding=1), nn.InstanceNorm2d(c, affine=True), nn.SiLU())
            for c in self.conv_dims])

    def forward(self, inputs):
        B = inputs[0].shape[0]
        HW = [(t.shape[2], t.shape[3]) for t in inputs]
        Ls = [h * w for h, w in HW]
        hd = self.hidden_dim
        # (B, L_cat, 48) token-major concatenation of the first 48 channels of every scale
        halves = [ops.split_planes(t, (hd, t.shape[1] - hd)) for t in inputs]  # (mamba 48 | conv rest) per scale; one gradient buffer per map
        m = torch.cat([_map_to_tokens(mh) for mh, _ in halves], dim=1)
        m = self.drop_path.residual(m, self.self_attention(self.ln_1(m), HW, Ls))
        m = self.norm2(m)
        outs = []
        for i, (mi, (H, W)) in enumerate(zip(m.split(Ls, dim=1), HW)):
            mi = self.drop_path.residual(mi, self.mlps[i](mi, H, W))
            mi = _TokensToMap.apply(mi.contiguous(), H, W)      # real transpose: keeps the gradient token-major
            conv, norm = self.conv_branches[i][0], self.conv_branches[i][1]          # [2] is the SiLU fused into K10
            xc = halves[i][1]
            if xc.is_cuda and _chain_dtype(xc) is not None:
                # 16-bit modes: K10 reads the 16-bit convolution output; the convolution's bias cancels exactly in the
                # InstanceNorm that follows (its gradient is exactly zero)
                cv = _ZeroGradParam.apply(conv.raw(xc), conv.bias)
            else:
                cv = conv(xc)
            outs.append(torch.cat([mi, _instance_norm_act(norm, cv, ops.ACT_SILU)], dim=1))
        return outs


class VSS_Conv_Layer(nn.Module):  # reference M:756-804
    def __init__(self, feature_dims, hidden_dim, depth=1, drop_path=0.0):
        super().__init__()
        self.blocks = nn.ModuleList([VSS_Conv_Block(feature_dims, hidden_dim, drop_path) for _ in range(depth)])

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


class MLLA_Uper(nn.Module):  # reference T:1183-1407
    def __init__(self, img_size: Sequence[int], patch_size=2, in_channels=1, out_channels=14, embed_dim=96,
                 depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), mlp_ratio=2, dropout_path_rate=0.1,
                 sr_ratio=(16, 8, 4, 2), deep_supervision=True, variant="B", precision="fp32"):
        super().__init__()
        if precision not in ops.PRECISIONS:
            raise RuntimeError(f"precision {precision!r}: one of {tuple(ops.PRECISIONS)}")
        self.precision = precision             # arithmetic type of the dense products (ops.compute_precision)
        self.deep_supervision = deep_supervision
        E = embed_dim
        self.mlla = MLLA_Enc(list(img_size), patch_size, in_channels, E, list(depths), list(num_heads), mlp_ratio,
                             dropout_path_rate, list(sr_ratio), variant)
        self.mambaskip = VSS_Conv_Layer([E, 2 * E, 4 * E, 8 * E], E // 2, depth=1, drop_path=0.1)
        self.up_2 = PatchExpand(8 * E, 4 * E)
        self.dec_block_2 = nn.Sequential(*[MedNeXtBlock(4 * E, 4 * E, mlp_ratio) for _ in range(depths[-2])])
        self.up_1 = PatchExpand(4 * E, 2 * E)
        self.dec_block_1 = nn.Sequential(*[MedNeXtBlock(2 * E, 2 * E, mlp_ratio) for _ in range(depths[-3])])
        self.up_0 = PatchExpand(2 * E, E)
        self.dec_block_0 = nn.Sequential(*[MedNeXtBlock(E, E, mlp_ratio) for _ in range(depths[-4])])
        self.encoder0 = UnetrBasicBlock(in_channels, E // 2)
        self.decoder0 = UnetrUpBlock(E, E // 2)
        self.out_0 = OutBlock(E // 2, out_channels)
        # unused in forward (reference T:1362): kept as a Parameter for the state_dict key and its slot in the optimizer's
        # parameter list, but frozen -- DistributedDataParallel only registers parameters that require a gradient, so
        # the reference's "finished reduction" hazard (SURVEY finding 7a) cannot arise and no DDP-private API is needed
        self.dummy_tensor = nn.Parameter(torch.tensor([1.0]), requires_grad=False)
        self._dp_pool = _DropPathPool()
        self._images = ops.WeightImageSet()
        if deep_supervision:
            self.out_1 = OutBlock(E, out_channels)
            self.out_2 = OutBlock(2 * E, out_channels)
            self.out_3 = OutBlock(4 * E, out_channels)
            self.out_4 = OutBlock(8 * E, out_channels)

    def forward(self, x_in):
        # The network computes in ITS precision whatever the caller's autocast state: the reference loop runs it under
        # autocast('cuda') (B:848), where the first MIOpen convolution would hand fp16 maps to fp32 HIP kernels.
        with torch.autocast(x_in.device.type, enabled=False), ops.compute_precision(self.precision):
            x_in = x_in.float()
            if not self.training:
                return self._forward(x_in)
            self._dp_pool.begin(x_in.shape[0], x_in.device)
            try:
                return self._forward(x_in)
            finally:
                self._dp_pool.end()

    def _stacks(self):
        return [st for m in self.modules() for st in (getattr(m, "_w_stack", None), getattr(m, "_b_stack", None)) if isinstance(st, _Stack)]

    def _forward(self, x_in):
        refresh_stacks(self._stacks())                      # every stacked projection weight / bias in one multi-tensor copy
        with self._images:                                  # ... and the bf16 images of every projection weight in one launch
            return self._forward_body(x_in)

    def _forward_body(self, x_in):
        hs = self.mlla(x_in)
        hs[1:] = self.mambaskip(hs[1:])
        ds = self.deep_supervision
        if ds:
            y4 = self.out_4(hs[4])
        x = self.dec_block_2(hs[3] + self.up_2(hs[4]))
        if ds:
            y3 = self.out_3(x)
        x = self.dec_block_1(hs[2] + self.up_1(x))
        if ds:
            y2 = self.out_2(x)
        x = self.dec_block_0(hs[1] + self.up_0(x))
        if ds:
            y1 = self.out_1(x)
        x = self.out_0(self.decoder0(x, self.encoder0(hs[0])))
        return [x, y1, y2, y3, y4] if ds else x


def build_network_architecture(patch_size, num_input_channels, num_segmentation_heads, enable_deep_supervision=True,
                               variant="B", precision="fp32"):
    """The hyper-parameters hard-coded at reference T:71-89; arguments are the only values that reach
    the model from the plans (patch_size, #channels, #classes)."""
    return MLLA_Uper(tuple(patch_size), 2, num_input_channels, num_segmentation_heads, 96, (2, 2, 2, 2),
                     (2, 4, 8, 16), 2, 0.1, (16, 8, 4, 2), enable_deep_supervision, variant, precision)
