"""3-D network for MI355X (BASELINE configs[3]): the module tree behind
``nnUNetTrainerUMambaEnc_SS3D.build_network_architecture`` (reference variants/mamba/nnUNetTrainerUMambaEnc_SS3D.py:8-31 ->
``get_umamba_enc_3d_from_plans``, variants/mamba/UMambaEnc_SS3D.py:890-942 -- cited as S below).  The reference ships no 3-D MLAgg
source (SURVEY finding 6: stale bytecode only); this file is the in-tree 3-D design: a residual convolutional U-Net whose every
encoder stage ends in a ``VSSLayer`` of 12-direction selective scans (``SS3D`` with d_state = 1, S:640-655).

Module and parameter names equal the reference's (614 state_dict keys for 6 stages, the ``decoder.encoder.*`` duplicates of S:715
and the ``all_modules.*`` duplicates of dynamic_network_architectures' conv blocks included), so checkpoints load either way.

What runs where:
  * HIP (libmlagg_hip.so): the scan K1s with the 12 scan orders applied in-kernel (csrc/selscan1.hip), K14 gathers of the
    x_proj columns, K2v depthwise 3x3x3 + SiLU, K5 / K5w projections of the token-major volume, K6 LayerNorm, K10
    InstanceNorm3d(affine) + LeakyReLU (+ residual), K8 tiled transposes between NCDHW and token-major, K9 loss;
  * PyTorch-ROCm: the full 3x3x3 / 1x1x1 convolutions (MIOpen), nearest upsampling, GELU, concatenation -- the north star's
    "conv stem/decoder stages live in PyTorch-ROCm".
Arithmetic the reference does and this file skips because it cancels exactly: the bias of a convolution that feeds an
InstanceNorm (the norm subtracts the plane mean; the parameter still receives its -- exactly zero -- gradient); and the 1x1x1
convolution of ``UpsampleLayer`` runs BEFORE the nearest-neighbour upsampling instead of after it (same values, 1/8 of the work).
There is no eager fallback for device tensors; host tensors take plain torch modules (CPU tests of the host logic).
"""
import math
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .model import LayerNorm, Linear, _ZeroGradParam, _flip
from .ss3d import SS3D


class _ToTokens(torch.autograd.Function):
    """(B, C, D, H, W) -> contiguous token-major (B, L, C) and back: real transposes both ways (model._MapToTokens in 3-D)."""

    @staticmethod
    def forward(ctx, x):
        ctx.dims = tuple(x.shape[2:])
        return _flip(x.reshape(x.shape[0], x.shape[1], -1))

    @staticmethod
    def backward(ctx, g):
        return _flip(g).view(g.shape[0], g.shape[2], *ctx.dims)


class _ToVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, dims):
        return _flip(t).view(t.shape[0], t.shape[2], *dims)

    @staticmethod
    def backward(ctx, g):
        return _flip(g.reshape(g.shape[0], g.shape[1], -1)), None


class Conv3d(nn.Conv3d):
    """nn.Conv3d (same parameter names).  ``forward``: MIOpen convolution + K8 in-place channel bias; ``normed``: the convolution
    alone, for a consumer that normalises every (sample, channel) plane."""

    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        y = ops.conv_nd(x, self.weight, self.stride, self.padding)
        return y if self.bias is None else ops.channel_bias(y, self.bias)

    def normed(self, x):
        if not x.is_cuda:
            return super().forward(x)
        y = ops.conv_nd(x, self.weight, self.stride, self.padding)
        return y if self.bias is None else _ZeroGradParam.apply(y, self.bias)


def _inorm(c):
    return nn.InstanceNorm3d(c, eps=1e-5, affine=True)                       # S:913-914


def _norm_act(norm, x, act=ops.ACT_NONE, res=None):
    """act(InstanceNorm3d(x) + res) with LeakyReLU(0.01): K10 on the device."""
    if x.is_cuda:
        return ops.plane_norm(x, norm.weight, norm.bias, norm.eps, act, 0.01, res)
    y = norm(x) if res is None else norm(x) + res
    return F.leaky_relu(y, 0.01) if act == ops.ACT_LEAKY else y


class BasicResBlock(nn.Module):                                              # S:477-513
    def __init__(self, cin, cout, k, stride=1, use_1x1conv=False):
        super().__init__()
        pad = [i // 2 for i in k]
        self.conv1 = Conv3d(cin, cout, k, stride=stride, padding=pad)
        self.norm1 = _inorm(cout)
        self.conv2 = Conv3d(cout, cout, k, padding=pad)
        self.norm2 = _inorm(cout)
        self.conv3 = Conv3d(cin, cout, kernel_size=1, stride=stride) if use_1x1conv else None

    def forward(self, x):
        y = _norm_act(self.norm1, self.conv1.normed(x), ops.ACT_LEAKY)
        res = self.conv3(x) if self.conv3 is not None else x
        return _norm_act(self.norm2, self.conv2.normed(y), ops.ACT_LEAKY, res)          # act2(norm2(conv2(.)) + res)


class _ConvNorm(nn.Module):
    """dynamic_network_architectures' ConvDropoutNormReLU as the reference instantiates it (S:589-601): conv + InstanceNorm3d
    (+ LeakyReLU), reachable under ``conv`` / ``norm`` and again under ``all_modules`` (the published module's key layout)."""

    def __init__(self, c, k, nonlin):
        super().__init__()
        self.conv = Conv3d(c, c, k, stride=1, padding=[(i - 1) // 2 for i in k], bias=True)
        self.norm = _inorm(c)
        self.act = ops.ACT_LEAKY if nonlin else ops.ACT_NONE
        self.all_modules = nn.Sequential(*([self.conv, self.norm] + ([nn.LeakyReLU(0.01, inplace=True)] if nonlin else [])))

    def forward(self, x, res=None, act=None):
        return _norm_act(self.norm, self.conv.normed(x), self.act if act is None else act, res)


class BasicBlockD(nn.Module):
    """dynamic_network_architectures BasicBlockD at the reference's call sites (stride 1, equal channels: identity skip)."""

    def __init__(self, c, k):
        super().__init__()
        self.conv1 = _ConvNorm(c, k, True)
        self.conv2 = _ConvNorm(c, k, False)

    def forward(self, x):
        return self.conv2(self.conv1(x), res=x, act=ops.ACT_LEAKY)                      # nonlin2(conv2(conv1(x)) + x)


class MLPBlock(nn.Module):
    """MONAI 1.3.0 MLPBlock(hidden, mlp_dim, act="GELU", dropout 0) as used at S:421."""

    def __init__(self, hidden, mlp_dim):
        super().__init__()
        self.linear1 = Linear(hidden, mlp_dim)
        self.linear2 = Linear(mlp_dim, hidden)

    def forward(self, x):
        return self.linear2(F.gelu(self.linear1(x)))


class VSSBlock(nn.Module):                                                   # S:395-434
    def __init__(self, dim, d_state, expand, mlp_ratio):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.self_attention = SS3D(dim, d_state=d_state, expand=expand)
        self.norm2 = LayerNorm(dim)
        self.mlp = MLPBlock(dim, int(dim * mlp_ratio))

    def forward_tokens(self, t, dims):
        B, L, C = t.shape
        t = t + self.self_attention(self.norm(t).view(B, *dims, C)).view(B, L, C)       # DropPath(0) of S:426-427
        return t + self.mlp(self.norm2(t))

    def forward(self, x):                                                    # (B, C, D, H, W) -> same
        dims = tuple(x.shape[2:])
        if not x.is_cuda:
            t = x.permute(0, 2, 3, 4, 1).reshape(x.shape[0], -1, x.shape[1])
            return self.forward_tokens(t, dims).view(x.shape[0], *dims, -1).permute(0, 4, 1, 2, 3).contiguous()
        return _ToVolume.apply(self.forward_tokens(_ToTokens.apply(x), dims), dims)


class VSSLayer(nn.Module):                                                   # S:436-474
    def __init__(self, dim, depth, d_state, expand, mlp_ratio):
        super().__init__()
        self.blocks = nn.ModuleList([VSSBlock(dim, d_state, expand, mlp_ratio) for _ in range(depth)])

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


class ResidualMambaEncoder(nn.Module):                                       # S:516-705
    def __init__(self, input_channels, features, kernel_sizes, strides, n_blocks, d_state=1):
        super().__init__()
        n_stages = len(features)
        self.stem = nn.Sequential(BasicResBlock(input_channels, features[0], kernel_sizes[0], 1, True),
                                  *[BasicBlockD(features[0], kernel_sizes[0]) for _ in range(n_blocks[0] - 1)])
        mlp_ratios = [2] * 2 + [4] * (n_stages - 2)                          # S:606
        stages, mamba, cin = [], [], features[0]
        for s in range(n_stages):
            stages.append(nn.Sequential(BasicResBlock(cin, features[s], kernel_sizes[s], strides[s], True),
                                        *[BasicBlockD(features[s], kernel_sizes[s]) for _ in range(n_blocks[s] - 1)]))
            mamba.append(VSSLayer(features[s], 1, d_state, 2, mlp_ratios[s]))            # S:640-655: d_state = 1, expand = 2
            cin = features[s]
        self.mamba_layers = nn.ModuleList(mamba)
        self.stages = nn.ModuleList(stages)
        self.output_channels, self.strides, self.kernel_sizes = list(features), [list(s) for s in strides], kernel_sizes

    def forward(self, x):
        x = self.stem(x)
        ret = []
        for stage, mamba in zip(self.stages, self.mamba_layers):
            x = mamba(stage(x))
            ret.append(x)
        return ret


class UpsampleLayer(nn.Module):                                              # S:49-66
    def __init__(self, cin, cout, scale):
        super().__init__()
        self.conv = Conv3d(cin, cout, kernel_size=1)
        self.scale = tuple(int(s) for s in scale)

    def forward(self, x):
        # conv1x1(nearest(x)) == nearest(conv1x1(x)) value for value: the convolution runs on the small volume
        return F.interpolate(self.conv(x), scale_factor=self.scale, mode="nearest")


class UNetResDecoder(nn.Module):                                             # S:707-813
    def __init__(self, encoder, num_classes, n_conv, deep_supervision):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder                          # S:715 registers the encoder here too: the duplicate checkpoint keys
        n = len(encoder.output_channels)
        stages, ups, segs = [], [], []
        for s in range(1, n):
            below, skip = encoder.output_channels[-s], encoder.output_channels[-(s + 1)]
            ups.append(UpsampleLayer(below, skip, encoder.strides[-s]))
            k = encoder.kernel_sizes[-(s + 1)]
            stages.append(nn.Sequential(BasicResBlock(2 * skip, skip, k, 1, True),
                                        *[BasicBlockD(skip, k) for _ in range(n_conv[s - 1] - 1)]))
            segs.append(Conv3d(skip, num_classes, 1, 1, 0, bias=True))
        self.stages, self.upsample_layers, self.seg_layers = nn.ModuleList(stages), nn.ModuleList(ups), nn.ModuleList(segs)

    def forward(self, skips):
        lres, outs = skips[-1], []
        for s in range(len(self.stages)):
            x = self.stages[s](torch.cat((self.upsample_layers[s](lres), skips[-(s + 2)]), 1))
            if self.deep_supervision:
                outs.append(self.seg_layers[s](x))
            elif s == len(self.stages) - 1:
                outs.append(self.seg_layers[-1](x))
            lres = x
        outs = outs[::-1]
        return outs if self.deep_supervision else outs[0]


class UMambaEnc(nn.Module):                                                  # S:815-888
    def __init__(self, input_channels, features: Sequence[int], kernel_sizes, strides, n_conv_per_stage, num_classes,
                 n_conv_per_stage_decoder, deep_supervision=True):
        super().__init__()
        n_stages = len(features)
        nb = [n_conv_per_stage] * n_stages if isinstance(n_conv_per_stage, int) else list(n_conv_per_stage)
        nd = [n_conv_per_stage_decoder] * (n_stages - 1) if isinstance(n_conv_per_stage_decoder, int) else \
            list(n_conv_per_stage_decoder)
        for s in range(math.ceil(n_stages / 2), n_stages):                   # S:845-849
            nb[s] = 1
        for s in range(math.ceil((n_stages - 1) / 2 + 0.5), n_stages - 1):
            nd[s] = 1
        self.encoder = ResidualMambaEncoder(input_channels, list(features), kernel_sizes, strides, nb)
        self.decoder = UNetResDecoder(self.encoder, num_classes, nd, deep_supervision)

    @property
    def deep_supervision(self):
        return self.decoder.deep_supervision

    @deep_supervision.setter
    def deep_supervision(self, enabled):                                     # nnUNetTrainer.set_deep_supervision_enabled
        self.decoder.deep_supervision = enabled

    def forward(self, x):
        with torch.autocast(x.device.type, enabled=False):                   # fp32 whatever the caller's autocast state (B:848)
            return self.decoder(self.encoder(x.float()))


def build_network_architecture_3d(num_input_channels, num_segmentation_heads, conv_kernel_sizes, pool_op_kernel_sizes,
                                  n_conv_per_stage_encoder, n_conv_per_stage_decoder, base_num_features=32,
                                  max_num_features=320, enable_deep_supervision=True):
    """``get_umamba_enc_3d_from_plans`` (S:890-942) on the values it reads from the plans: features min(base * 2^i, max)."""
    n = len(conv_kernel_sizes)
    features = [min(base_num_features * 2 ** i, max_num_features) for i in range(n)]
    net = UMambaEnc(num_input_channels, features, [list(k) for k in conv_kernel_sizes], [list(s) for s in pool_op_kernel_sizes],
                    n_conv_per_stage_encoder, num_segmentation_heads, n_conv_per_stage_decoder, enable_deep_supervision)
    net.apply(init_weights_he)                                               # S:941 model.apply(InitWeights_He(1e-2))
    return net


def init_weights_he(module, neg_slope=1e-2):
    """``InitWeights_He`` (reference utilities/network_initialization.py:4-13): every (transposed) convolution -- the depthwise
    Conv3d of the SS3D blocks included -- gets kaiming_normal_(a = 1e-2) weights and a zero bias."""
    if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
        nn.init.kaiming_normal_(module.weight, a=neg_slope)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)


# the 3d_fullres plan shape behind BASELINE configs[3] (BTCV-shaped 96 x 160 x 160 patches): six stages, 3x3x3 kernels, five
# poolings of which the last leaves the short axis alone (96 / 16 = 6, 160 / 32 = 5)
BTCV_STRIDES = [[1, 1, 1], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2]]


def deep_supervision_scales(strides):
    """nnUNetTrainer._get_deep_supervision_scales: 1 / cumprod of the pooling strides, lowest resolution dropped."""
    scales, cur = [], [1.0, 1.0, 1.0]
    for s in strides:
        cur = [c / float(v) for c, v in zip(cur, s)]
        scales.append(list(cur))
    return scales[:-1]


def synthetic_batch_3d(batch, in_ch, size, strides, n_cls, seed=1234, device="cpu"):
    """The benchmark trainer's synthetic batch (nnUNetTrainerBenchmark_5epochs_noDataLoading.py:16-22) for a 3-D plan."""
    g = torch.Generator().manual_seed(seed)
    data = torch.rand(batch, in_ch, *size, generator=g)
    target = [torch.round(torch.rand(batch, 1, *[int(round(s * f)) for s, f in zip(size, sc)], generator=g) * (n_cls - 1))
              for sc in deep_supervision_scales(strides)]
    return data.to(device), [t.to(device) for t in target]
