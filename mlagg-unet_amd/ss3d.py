"""3-D selective-scan block for volumes (BASELINE configs[3]: "3D selective-scan over D*H*W tokens").

The reference ships no 3-D MLAgg source (SURVEY finding 6: only stale bytecode); the in-tree design source for a 3-D scan is
``SS3D`` of variants/mamba/UMambaEnc_SS3D.py:126-357 -- twelve scan directions (the six axis orders of (D, H, W) and their
reversals) over one volume, K = 12 groups of the same selective scan the 2-D MSMM module uses.  This module is that block
on the MI355X kernels, with the reference's parameter names (x_proj_weight (12, R + 2N, d_inner), dt_projs_weight,
dt_projs_bias, A_logs, Ds, in_proj, conv3d, out_norm, out_proj):

  * the 14 + 11 stack / permute / contiguous / flip / cat copies of ``forward_corev0`` (:251-259, :284-295) become two
    launches of K1' for volumes (csrc/index_scan.hip) driven by an int32 permutation table per volume shape;
  * x_proj of all 12 directions is ONE token-major Linear on the natural token order (K5), the rank-R dt projection lives
    inside K1 (low-rank form), the 12-way sum of :338 is the merge kernel's accumulation;
  * the scan itself is K1 with G = 12 groups (csrc/selscan.hip), forward and backward.
The depthwise 3x3x3 convolution, LayerNorm and the two Linears around the core are library / K5 / K6 calls.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .model import LayerNorm, Linear, _dt_init

_TABLES = {}


def scan_orders_3d(D, H, W, device):
    """(12, L) int32: natural (d, h, w) index of scan step l for the 12 directions of SS3D.forward_corev0
    (UMambaEnc_SS3D.py:251-259): 0 dhw, 1 dwh, 2 hdw, 3 hwd, 4 wdh, 5 whd, 6..11 their reversals."""
    key = (D, H, W, str(device))
    if key not in _TABLES:
        nat = torch.arange(D * H * W, dtype=torch.int32).view(D, H, W)
        fwd = [nat.reshape(-1), nat.permute(0, 2, 1).reshape(-1), nat.permute(1, 0, 2).reshape(-1),
               nat.permute(1, 2, 0).reshape(-1), nat.permute(2, 0, 1).reshape(-1), nat.permute(2, 1, 0).reshape(-1)]
        _TABLES[key] = torch.stack(fwd + [t.flip(0) for t in fwd]).contiguous().to(device)
    return _TABLES[key]


class SS3D(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=3, expand=2, dt_rank="auto", conv_bias=True, bias=False):
        super().__init__()
        self.d_model, self.d_state = d_model, d_state
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        if d_state == 16:
            if self.dt_rank > 4 or self.d_inner > 96:
                raise RuntimeError("SS3D on MI355X, d_state 16: dt_rank <= 4 and d_inner <= 96 (the K1 kernels' build)")
        elif d_state == 1:
            # the 3-D network's blocks (UMambaEnc_SS3D.py:640-655): K1s, csrc/selscan1.hip
            if self.d_inner % 64 or self.dt_rank not in (1, 2, 3, 4, 8, 16, 20):
                raise RuntimeError("SS3D on MI355X, d_state 1: d_inner a multiple of 64 and dt_rank in {1, 2, 3, 4, 8, 16, 20} "
                                   "(the K1s kernels' build)")
        else:
            raise RuntimeError("SS3D on MI355X: d_state 16 (K1) or 1 (K1s)")
        self.in_proj = Linear(d_model, self.d_inner, bias=bias)
        self.conv3d = nn.Conv3d(self.d_inner, self.d_inner, d_conv, padding=(d_conv - 1) // 2, groups=self.d_inner, bias=conv_bias)
        K = 12
        xp = [nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False) for _ in range(K)]
        self.x_proj_weight = nn.Parameter(torch.stack([t.weight for t in xp], 0))
        dts = [_dt_init(self.dt_rank, self.d_inner) for _ in range(K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], 0))
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], 0))
        A = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(K * self.d_inner, 1)
        self.A_logs = nn.Parameter(torch.log(A))
        self.Ds = nn.Parameter(torch.ones(K * self.d_inner))
        self.out_norm = LayerNorm(self.d_inner)
        self.out_proj = Linear(self.d_inner, d_model, bias=bias)

    def core(self, tok, dims):
        """tok (B, L, d_inner) token-major conv output in natural (d, h, w) order -> (B, L, d_inner): the sum of the 12
        re-ordered scan directions (reference forward_corev0 + torch.sum at :338)."""
        B, L, dI = tok.shape
        K, R, N = 12, self.dt_rank, self.d_state
        per = R + 2 * N
        idx = scan_orders_3d(*dims, tok.device)
        xdbl = ops.linear(tok, self.x_proj_weight.reshape(K * per, dI))                    # (B, L, 12 * per)
        if N == 1:
            # K1s: u is gathered and y scattered through `idx` inside the scan kernels; only the narrow x_proj columns
            # (R + 2 floats per token and direction) are re-ordered up front
            dtr, Bs, Cs = ops.index_scan_bc(xdbl, idx, R)
            return ops.selective_scan1(tok, idx, dtr, Bs, Cs, self.dt_projs_weight.reshape(K * dI, R),
                                       -torch.exp(self.A_logs).reshape(-1), self.Ds, self.dt_projs_bias.reshape(-1))
        dtr = ops.index_scan(xdbl, idx, R, per, 0).view(B, K, R, L)
        Bs = ops.index_scan(xdbl, idx, N, per, R).view(B, K, N, L)
        Cs = ops.index_scan(xdbl, idx, N, per, R + N).view(B, K, N, L)
        xs = ops.index_scan(tok, idx, dI, 0, 0)                                             # (B, 12 * d_inner, L)
        out = ops.selective_scan_lowrank_fn(xs, dtr, self.dt_projs_weight.reshape(K * dI, R), -torch.exp(self.A_logs), Bs, Cs,
                                            self.Ds, delta_bias=self.dt_projs_bias.reshape(-1), delta_softplus=True)
        return ops.index_merge(out, idx, dI)

    def forward(self, x):
        """x (B, D, H, W, d_model) -> same shape (reference SS3D.forward, UMambaEnc_SS3D.py:326-352)."""
        B, D, H, W, _ = x.shape
        h = self.in_proj(x.reshape(B, D * H * W, -1))
        if h.is_cuda:
            # K2v: the depthwise 3x3x3 convolution + SiLU on the token-major volume itself -- no transpose on either side (handed
            # the strided view `h.transpose(1, 2)`, PyTorch takes the volume for channels_last_3d and MIOpen runs a far slower
            # path still: the block took 19.0 ms per forward + backward at 2 x 24 x 40 x 40 tokens; with real transposes
            # around MIOpen's naive depthwise kernels 6.8 ms)
            if self.d_inner * 27 * 4 > 64 * 1024:
                # K2v stages a (C, 27) weight tile in LDS: wider maps (640 channels at the two deepest, tiny stages of the
                # 3-D network) go through it in two channel halves
                hc = self.d_inner // 2
                w, bias = self.conv3d.weight, self.conv3d.bias
                tok = torch.cat([ops.dwconv3d_nlc(h[..., :hc], w[:hc], None if bias is None else bias[:hc], (D, H, W), silu=True),
                                 ops.dwconv3d_nlc(h[..., hc:], w[hc:], None if bias is None else bias[hc:], (D, H, W), silu=True)], -1)
            else:
                tok = ops.dwconv3d_nlc(h, self.conv3d.weight, self.conv3d.bias, (D, H, W), silu=True)
        else:
            vol = F.silu(self.conv3d(h.transpose(1, 2).reshape(B, self.d_inner, D, H, W)))
            tok = vol.reshape(B, self.d_inner, -1).transpose(1, 2).contiguous()
        y = self.core(tok, (D, H, W))
        return self.out_proj(self.out_norm(y)).view(B, D, H, W, -1)
