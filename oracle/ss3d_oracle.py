"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's 3-D selective-scan block ``SS3D``
(mlagg/nnunetv2/training/nnUNetTrainer/variants/mamba/UMambaEnc_SS3D.py:126-357; forward_corev0 :244-296, forward :326-352),
pinned by tests/golden/ss3d.npz (outputs of the reference class itself, made by tests/golden/make_golden.py; the
selective scan inside is the unpinned mamba-ssm restatement of oracle/mlagg_oracle.py on both sides)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .mlagg_oracle import _dt_init, selective_scan_oracle


class SS3D(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=3, expand=2):
        super().__init__()
        self.d_state, self.d_inner = d_state, int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16)
        self.in_proj = nn.Linear(d_model, self.d_inner, bias=False)
        self.conv3d = nn.Conv3d(self.d_inner, self.d_inner, d_conv, padding=(d_conv - 1) // 2, groups=self.d_inner)
        K = 12
        self.x_proj_weight = nn.Parameter(torch.zeros(K, self.dt_rank + 2 * d_state, self.d_inner))
        dts = [_dt_init(self.dt_rank, self.d_inner) for _ in range(K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], 0))
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], 0))
        self.A_logs = nn.Parameter(torch.log(torch.arange(1, d_state + 1, dtype=torch.float32).repeat(K * self.d_inner, 1)))
        self.Ds = nn.Parameter(torch.ones(K * self.d_inner))
        self.out_norm = nn.LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=False)

    def forward_core(self, x):                                  # :244-296, directions and inverse maps as the reference
        B, C, D, H, W = x.shape
        L, K = D * H * W, 12
        a = torch.stack([x.reshape(B, -1, L), x.transpose(3, 4).reshape(B, -1, L)], 1)
        b = torch.stack([x.permute(0, 1, 3, 2, 4).reshape(B, -1, L), x.permute(0, 1, 3, 4, 2).reshape(B, -1, L)], 1)
        c = torch.stack([x.permute(0, 1, 4, 2, 3).reshape(B, -1, L), x.permute(0, 1, 4, 3, 2).reshape(B, -1, L)], 1)
        xs = torch.cat([a, b, c, a.flip(-1), b.flip(-1), c.flip(-1)], 1)
        x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, self.x_proj_weight)
        dts, Bs, Cs = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=2)
        dts = torch.einsum("b k r l, k d r -> b k d l", dts, self.dt_projs_weight)
        y = selective_scan_oracle(xs.reshape(B, -1, L), dts.reshape(B, -1, L), -torch.exp(self.A_logs), Bs, Cs, self.Ds, None,
                                  self.dt_projs_bias.reshape(-1), True).view(B, K, -1, L)
        y = torch.cat([y[:, :6], y[:, 6:].flip(-1)], 1)
        shp = {1: ((D, W, H), (0, 1, 2, 4, 3)), 2: ((H, D, W), (0, 1, 3, 2, 4)), 3: ((H, W, D), (0, 1, 4, 2, 3)),
               4: ((W, D, H), (0, 1, 3, 4, 2)), 5: ((W, H, D), (0, 1, 4, 3, 2))}
        outs = []
        for k in range(K):
            yk = y[:, k]
            if k % 6 in shp:
                dims, perm = shp[k % 6]
                yk = yk.reshape(B, -1, *dims).permute(*perm).reshape(B, -1, L)
            outs.append(yk)
        return torch.stack(outs, 1)

    def forward(self, x):                                       # :326-352
        B, D, H, W, _ = x.shape
        v = F.silu(self.conv3d(self.in_proj(x).permute(0, 4, 1, 2, 3)))
        y = self.forward_core(v).sum(1).transpose(1, 2).reshape(B, D, H, W, -1)
        return self.out_proj(self.out_norm(y))
