"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's 3-D network ``UMambaEnc`` with SS3D blocks
(mlagg/nnunetv2/training/nnUNetTrainer/variants/mamba/UMambaEnc_SS3D.py -- cited as S below), the network behind
``nnUNetTrainerUMambaEnc_SS3D.build_network_architecture`` (variants/mamba/nnUNetTrainerUMambaEnc_SS3D.py:8-31) and the
in-tree design source for BASELINE configs[3] (the reference ships no 3-D MLAgg source: SURVEY finding 6).

Pinned by tests/golden/umamba3d_small.npz: logits, loss and gradient norms of the reference's own ``UMambaEnc`` class
(made by tests/golden/make_golden.py).  Unpinned third-party arithmetic, restated here on both sides of the fixture:
  * ``dynamic_network_architectures.building_blocks.residual.BasicBlockD`` (S:45, 589-601, 625-637, 765-777): conv-norm-nonlin,
    conv-norm, identity skip (stride 1, equal channels at every call site), add, nonlin; parameters reachable under both
    ``conv`` / ``norm`` and ``all_modules.{0,1}`` (the published module keeps an ``nn.Sequential`` of the same layers);
  * ``monai.networks.blocks.MLPBlock`` (S:18, 421): linear1 -> GELU -> linear2 (dropout 0);
  * mamba-ssm ``selective_scan_fn`` (S:27, 277-283): oracle/mlagg_oracle.selective_scan_oracle;
  * ``timm`` DropPath (S:39): rate 0 at every call site of this network (S:640-655 passes none).
Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this file.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mlagg_oracle import DropPath
from .ss3d_oracle import SS3D


class UpsampleLayer(nn.Module):                                  # S:49-66
    def __init__(self, cin, cout, scale):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=1)
        self.scale = tuple(scale)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=self.scale, mode="nearest"))


class MLPBlock(nn.Module):                                       # MONAI 1.3.0 MLPBlock(hidden, mlp_dim, act="GELU", dropout 0)
    def __init__(self, hidden, mlp_dim):
        super().__init__()
        self.linear1 = nn.Linear(hidden, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden)

    def forward(self, x):
        return self.linear2(F.gelu(self.linear1(x)))


class VSSBlock(nn.Module):                                       # S:395-434
    def __init__(self, dim, d_state, expand, mlp_ratio):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.self_attention = SS3D(dim, d_state=d_state, expand=expand)
        self.drop_path = DropPath(0.0)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLPBlock(dim, int(dim * mlp_ratio))

    def forward(self, x):                                        # (B, C, D, H, W)
        t = x.permute(0, 2, 3, 4, 1)
        t = t + self.drop_path(self.self_attention(self.norm(t)))
        t = t + self.drop_path(self.mlp(self.norm2(t)))
        return t.permute(0, 4, 1, 2, 3).contiguous()


class VSSLayer(nn.Module):                                       # S:436-474
    def __init__(self, dim, depth, d_state, expand, mlp_ratio):
        super().__init__()
        self.blocks = nn.ModuleList([VSSBlock(dim, d_state, expand, mlp_ratio) for _ in range(depth)])

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


def _norm(c):
    return nn.InstanceNorm3d(c, eps=1e-5, affine=True)           # S:913-914


class BasicResBlock(nn.Module):                                  # S:477-513
    def __init__(self, cin, cout, k, stride=1, use_1x1conv=False):
        super().__init__()
        pad = [i // 2 for i in k]
        self.conv1 = nn.Conv3d(cin, cout, k, stride=stride, padding=pad)
        self.norm1 = _norm(cout)
        self.conv2 = nn.Conv3d(cout, cout, k, padding=pad)
        self.norm2 = _norm(cout)
        self.conv3 = nn.Conv3d(cin, cout, kernel_size=1, stride=stride) if use_1x1conv else None

    def forward(self, x):
        y = F.leaky_relu(self.norm1(self.conv1(x)), 0.01)
        y = self.norm2(self.conv2(y))
        if self.conv3 is not None:
            x = self.conv3(x)
        return F.leaky_relu(y + x, 0.01)


class _ConvNorm(nn.Module):
    """ConvDropoutNormReLU of dynamic_network_architectures: the layers under their own names and again inside ``all_modules``."""

    def __init__(self, c, k, nonlin):
        super().__init__()
        self.conv = nn.Conv3d(c, c, k, stride=1, padding=[(i - 1) // 2 for i in k], bias=True)
        self.norm = _norm(c)
        mods = [self.conv, self.norm] + ([nn.LeakyReLU(0.01, inplace=True)] if nonlin else [])
        self.all_modules = nn.Sequential(*mods)

    def forward(self, x):
        return self.all_modules(x)


class BasicBlockD(nn.Module):
    def __init__(self, c, k):
        super().__init__()
        self.conv1 = _ConvNorm(c, k, True)
        self.conv2 = _ConvNorm(c, k, False)

    def forward(self, x):
        return F.leaky_relu(self.conv2(self.conv1(x)) + x, 0.01)


class ResidualMambaEncoder(nn.Module):                           # S:516-705
    def __init__(self, input_channels, features, kernel_sizes, strides, n_blocks, d_state=1):
        super().__init__()
        n_stages = len(features)
        self.stem = nn.Sequential(BasicResBlock(input_channels, features[0], kernel_sizes[0], 1, True),
                                  *[BasicBlockD(features[0], kernel_sizes[0]) for _ in range(n_blocks[0] - 1)])
        mlp_ratios = [2] * 2 + [4] * (n_stages - 2)              # S:606
        stages, mamba = [], []
        cin = features[0]
        for s in range(n_stages):
            stages.append(nn.Sequential(BasicResBlock(cin, features[s], kernel_sizes[s], strides[s], True),
                                        *[BasicBlockD(features[s], kernel_sizes[s]) for _ in range(n_blocks[s] - 1)]))
            mamba.append(VSSLayer(features[s], 1, d_state, 2, mlp_ratios[s]))      # S:640-655: d_state=1, expand=2
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


class UNetResDecoder(nn.Module):                                 # S:707-813
    def __init__(self, encoder, num_classes, n_conv, deep_supervision):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder                                   # S:715: registers the encoder a second time (duplicate keys)
        n = len(encoder.output_channels)
        stages, ups, segs = [], [], []
        for s in range(1, n):
            below, skip = encoder.output_channels[-s], encoder.output_channels[-(s + 1)]
            ups.append(UpsampleLayer(below, skip, encoder.strides[-s]))
            k = encoder.kernel_sizes[-(s + 1)]
            stages.append(nn.Sequential(BasicResBlock(2 * skip, skip, k, 1, True),
                                        *[BasicBlockD(skip, k) for _ in range(n_conv[s - 1] - 1)]))
            segs.append(nn.Conv3d(skip, num_classes, 1, 1, 0, bias=True))
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


class UMambaEnc(nn.Module):                                      # S:815-888
    def __init__(self, input_channels, features, kernel_sizes, strides, n_conv_per_stage, num_classes,
                 n_conv_per_stage_decoder, deep_supervision=True):
        super().__init__()
        n_stages = len(features)
        nb = [n_conv_per_stage] * n_stages if isinstance(n_conv_per_stage, int) else list(n_conv_per_stage)
        nd = [n_conv_per_stage_decoder] * (n_stages - 1) if isinstance(n_conv_per_stage_decoder, int) else \
            list(n_conv_per_stage_decoder)
        for s in range(math.ceil(n_stages / 2), n_stages):       # S:845-849
            nb[s] = 1
        for s in range(math.ceil((n_stages - 1) / 2 + 0.5), n_stages - 1):
            nd[s] = 1
        self.encoder = ResidualMambaEncoder(input_channels, features, kernel_sizes, strides, nb)
        self.decoder = UNetResDecoder(self.encoder, num_classes, nd, deep_supervision)

    def forward(self, x):
        return self.decoder(self.encoder(x))


def build_reference_3d_model(input_channels, num_classes, features, strides, n_conv_per_stage=2, n_conv_per_stage_decoder=2,
                             deep_supervision=True):
    """``get_umamba_enc_3d_from_plans`` (S:890-942) without the plans objects: 3x3x3 kernels at every stage."""
    net = UMambaEnc(input_channels, list(features), [[3, 3, 3]] * len(features), strides, n_conv_per_stage, num_classes,
                    n_conv_per_stage_decoder, deep_supervision)
    net.apply(init_weights_he)                                               # S:941 model.apply(InitWeights_He(1e-2))
    return net


def init_weights_he(module, neg_slope=1e-2):
    """``InitWeights_He`` (reference utilities/network_initialization.py:4-13)."""
    if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
        nn.init.kaiming_normal_(module.weight, a=neg_slope)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)


def features_for(n_stages, base=32, cap=320):
    """S:927-928: min(base * 2^i, max) with nnU-Net's 3-D defaults."""
    return [min(base * 2 ** i, cap) for i in range(n_stages)]


def deep_supervision_scales(strides):
    """nnUNetTrainer._get_deep_supervision_scales (nnUNetTrainer.py:283-... : 1 / cumprod of the pooling strides, last dropped)."""
    return [list(v) for v in 1 / np.cumprod(np.vstack(strides), axis=0)][:-1]


def synthetic_batch_3d(batch, in_ch, size, strides, n_cls, seed=1234):
    """The benchmark trainer's synthetic batch (nnUNetTrainerBenchmark_5epochs_noDataLoading.py:16-22) for a 3-D plan."""
    g = torch.Generator().manual_seed(seed)
    data = torch.rand(batch, in_ch, *size, generator=g)
    target = [torch.round(torch.rand(batch, 1, *[int(round(s * f)) for s, f in zip(size, sc)], generator=g) * (n_cls - 1))
              for sc in deep_supervision_scales(strides)]
    return data, target
