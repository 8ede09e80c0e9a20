"""CPU oracle for the MLAgg-UNet 2D train-step hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU, fp32) restatement of the
reference algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package (``mlagg-unet_amd/``) never does.

Pinning status: the restatement is checked against the reference's own model classes
imported in the build container: ``tests/golden/make_golden.py`` runs them and stores their
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` holds this file to those fixtures.  The third-party arithmetic the reference calls but does not vendor
(mamba-ssm ``selective_scan_fn``, flash-attn, MONAI Unetr blocks, timm DropPath) is
restated from its published semantics -- **parity unpinned** at those four boundaries
(SURVEY.md section 8c).

Citations: T = mlagg/nnunetv2/training/nnUNetTrainer/nnUNetTrainer_MLAgg_2D_dt_MS.py,
M = mlagg/nnunetv2/training/nnUNetTrainer/variants/mamba/MambaSkip.py,
L/ = mlagg/nnunetv2/training/loss/, B = .../nnUNetTrainer/nnUNetTrainer.py.

Module and parameter names equal the reference's (state_dict keys are a checkpoint ABI,
B:1010-1021), so ``load_state_dict`` works in both directions.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

LAMBDA_INIT = 0.8  # T:638
SCAN_LITERAL_INDEXING = False   # tests/perf/cpu_baseline_literal.py: time the scan loop exactly as selective_scan_ref writes it


# --------------------------------------------------------------------------------------
# third-party semantics restated (unpinned)
# --------------------------------------------------------------------------------------
def selective_scan_oracle(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                          delta_softplus=False, return_last_state=False):
    """mamba-ssm ``selective_scan_ref`` semantics as called at M:445-451.

    u, delta (b, d, l); A (d, n); B, C (b, g, n, l) with channel d using group d // (d/g);
    D, delta_bias (d).  h_l = exp(delta_l A) h_{l-1} + delta_l B_l u_l ; y_l = C_l . h_l + D u_l.
    The time loop runs over ``unbind`` views (same forward arithmetic as indexing, O(L)
    backward).
    """
    dtype_in = u.dtype
    u = u.float()
    delta = delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias[..., None].float()
    if delta_softplus:
        delta = F.softplus(delta)
    b, d, l = u.shape
    n = A.shape[1]
    g = B.shape[1]
    rep = d // g
    Bf = B.float().repeat_interleave(rep, dim=1)  # "B G N L -> B (G H) N L"
    Cf = C.float().repeat_interleave(rep, dim=1)
    dA = torch.exp(delta.unsqueeze(-1) * A.float().view(1, d, 1, n))      # (b, d, l, n)
    dBu = (delta * u).unsqueeze(-1) * Bf.permute(0, 1, 3, 2)               # (b, d, l, n)
    Ct = Cf.permute(0, 1, 3, 2)                                            # (b, d, l, n)
    x = u.new_zeros(b, d, n)
    ys = []
    if SCAN_LITERAL_INDEXING:
        # selective_scan_ref's own loop shape: `deltaA[:, :, i]` slices.  Same values; the backward of every slice
        # zero-fills a (b, d, l, n) tensor, so a step is O(L^2) (figure (a)).
        for i in range(l):
            x = dA[:, :, i] * x + dBu[:, :, i]
            ys.append((x * Ct[:, :, i]).sum(-1))
    else:
        for dA_i, dBu_i, C_i in zip(dA.unbind(2), dBu.unbind(2), Ct.unbind(2)):
            x = dA_i * x + dBu_i
            ys.append((x * C_i).sum(-1))
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u * D.float().view(1, d, 1)
    if z is not None:
        out = out * F.silu(z.float())
    out = out.to(dtype_in)
    return (out, x) if return_last_state else out


def softmax_attention_oracle(q, k, v, softmax_scale=None):
    """flash-attn ``flash_attn_func(q, k, v, causal=False)`` semantics (T:745-750):
    q (b, n, h, e), k/v (b, p, h, e) -> (b, n, h, e), softmax(q k^T * e^-0.5) v."""
    e = q.shape[-1]
    scale = e ** -0.5 if softmax_scale is None else softmax_scale
    att = torch.einsum("bnhe,bphe->bhnp", q.float(), k.float()) * scale
    att = att.softmax(-1)
    return torch.einsum("bhnp,bphe->bnhe", att, v.float()).to(q.dtype)


class DropPath(nn.Module):
    """timm DropPath: per-sample Bernoulli(keep) / keep in training, identity in eval."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class _ConvOnly(nn.Module):
    """MONAI ``get_conv_layer(..., conv_only=True)`` wrapper: a bias-free conv under ``.conv``."""

    def __init__(self, cin, cout, k, stride=1, transposed=False):
        super().__init__()
        if transposed:
            self.conv = nn.ConvTranspose2d(cin, cout, k, stride=stride, bias=False)
        else:
            self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)

    def forward(self, x):
        return self.conv(x)


class UnetResBlock(nn.Module):
    """MONAI 1.3.0 ``UnetResBlock`` (structure vendored at M:581-667): conv-IN-lrelu-conv-IN,
    1x1 conv + IN on the residual when channels change, add, lrelu(0.01).  InstanceNorm
    non-affine (norm_name="instance")."""

    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.conv1 = _ConvOnly(cin, cout, k)
        self.conv2 = _ConvOnly(cout, cout, k)
        self.norm1 = nn.InstanceNorm2d(cout)
        self.norm2 = nn.InstanceNorm2d(cout)
        self.lrelu = nn.LeakyReLU(0.01)
        if cin != cout:
            self.conv3 = _ConvOnly(cin, cout, 1)
            self.norm3 = nn.InstanceNorm2d(cout)

    def forward(self, x):
        out = self.lrelu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        res = self.norm3(self.conv3(x)) if hasattr(self, "conv3") else x
        return self.lrelu(out + res)


class UnetrBasicBlock(nn.Module):
    """MONAI ``UnetrBasicBlock(res_block=True)`` (T:1339-1347): ``.layer`` = UnetResBlock."""

    def __init__(self, cin, cout):
        super().__init__()
        self.layer = UnetResBlock(cin, cout)

    def forward(self, x):
        return self.layer(x)


class UnetrUpBlock(nn.Module):
    """MONAI ``UnetrUpBlock(res_block=True, upsample_kernel_size=2)`` (T:1349-1357):
    ConvTranspose k=s=2 (no bias) -> cat skip -> UnetResBlock(2*cout -> cout)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.transp_conv = _ConvOnly(cin, cout, 2, stride=2, transposed=True)
        self.conv_block = UnetResBlock(2 * cout, cout)

    def forward(self, x, skip):
        return self.conv_block(torch.cat([self.transp_conv(x), skip], dim=1))


# --------------------------------------------------------------------------------------
# encoder (T:176-192, 230-366, 592-1179)
# --------------------------------------------------------------------------------------
class Mlp(nn.Module):  # T:176-192
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class RMSNorm(nn.Module):  # T:592-613
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x) * self.weight


def local_padding_mask(H, W, k=3):
    """(N, k*k) bool, True where the neighbour lies outside the image (T:616-622)."""
    ones = torch.ones(1, 1, H, W)
    cover = F.unfold(ones, k, padding=k // 2)          # (1, k*k, N)
    return cover[0].t() == 0


class AggregatedAttention(nn.Module):
    """T:625-784.  ``variant`` "A" = shipped flash path (logit scale 1/head_dim, finding 4),
    "B" = the commented eager path (scale head_dim^-0.5) used for fp32/CPU."""

    def __init__(self, dim, input_resolution, num_heads, local, sr_ratio, variant="B"):
        super().__init__()
        self.dim, self.num_heads, self.local, self.variant = dim, num_heads, local, variant
        self.head_dim = dim // num_heads // 2
        self.scale = self.head_dim ** -0.5
        for nm in ("lambda_q1", "lambda_k1", "lambda_q2", "lambda_k2"):
            setattr(self, nm, nn.Parameter(torch.zeros(self.head_dim).normal_(0, 0.1)))
        self.subln = RMSNorm(2 * self.head_dim, eps=1e-5)
        if local:
            self.register_buffer("padding_mask", local_padding_mask(*input_resolution), persistent=False)
        else:
            self.pool_H = input_resolution[0] // sr_ratio
            self.pool_W = input_resolution[1] // sr_ratio
            self.sr = nn.Conv2d(dim, dim, 1)
            self.norm = nn.LayerNorm(dim)
        self.q = nn.Linear(dim, dim)
        self.kv = nn.Linear(dim, 2 * dim)
        self.lepe = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)

    def lambda_full(self, like):
        l1 = torch.exp(torch.sum(self.lambda_q1 * self.lambda_k1).float()).type_as(like)
        l2 = torch.exp(torch.sum(self.lambda_q2 * self.lambda_k2).float()).type_as(like)
        return l1 - l2 + LAMBDA_INIT

    def forward(self, x, H, W):
        Bsz, N, C = x.shape
        nh, hd = self.num_heads, self.head_dim
        q = self.q(x).reshape(Bsz, N, 2 * nh, hd).permute(0, 2, 1, 3) * self.scale   # T:687-688
        k_full, v_full = self.kv(x).chunk(2, dim=-1)                                  # T:690
        lam = self.lambda_full(q)
        if self.local:
            k_img = k_full.permute(0, 2, 1).reshape(Bsz, C, H, W)
            v_img = v_full.permute(0, 2, 1).reshape(Bsz, C, H, W)
            k_unf = F.unfold(k_img, 3, padding=1).reshape(Bsz, 2 * nh, hd, 9, N).permute(0, 1, 4, 2, 3)
            v_unf = F.unfold(v_img, 3, padding=1).reshape(Bsz, nh, 2 * hd, 9, N).permute(0, 1, 4, 2, 3)
            att = (q.unsqueeze(-2) @ k_unf).squeeze(-2)                               # (B, 2nh, N, 9)
            att = att.masked_fill(self.padding_mask, float("-inf")).softmax(-1)       # T:706-707
            att = att.view(Bsz, nh, 2, N, 9)
            att = att[:, :, 0] - lam * att[:, :, 1]                                   # T:712-713
            o = (att.unsqueeze(-2) @ v_unf.transpose(-2, -1)).squeeze(-2)             # (B, nh, N, 2hd)
        else:
            x_img = x.permute(0, 2, 1).reshape(Bsz, C, H, W)
            pooled = F.adaptive_avg_pool2d(F.gelu(self.sr(x_img)), (self.pool_H, self.pool_W))
            x_ = self.norm(pooled.reshape(Bsz, C, -1).permute(0, 2, 1))               # T:721-723
            P = x_.shape[1]
            k_pool, v_pool = self.kv(x_).chunk(2, dim=-1)                              # T:728
            k_pool = k_pool.reshape(Bsz, P, 2 * nh, hd).permute(0, 2, 1, 3)
            v_pool = v_pool.reshape(Bsz, P, nh, 2 * hd).permute(0, 2, 1, 3)
            logits = q @ k_pool.transpose(-2, -1)
            if self.variant == "A":        # flash default softmax_scale applied on top (T:745-750)
                logits = logits * self.scale
            att = logits.softmax(-1).view(Bsz, nh, 2, N, P)
            att = att[:, :, 0] - lam * att[:, :, 1]
            o = att @ v_pool                                                           # (B, nh, N, 2hd)
        o = self.subln(o) * (1.0 - LAMBDA_INIT)                                        # T:716-717
        o = o.transpose(1, 2).reshape(Bsz, N, C)
        v_img = v_full.reshape(Bsz, H, W, C).permute(0, 3, 1, 2)
        return o + self.lepe(v_img).permute(0, 2, 3, 1).reshape(Bsz, N, C)             # T:781-782


class MLLABlock(nn.Module):  # T:824-915
    def __init__(self, dim, input_resolution, num_heads, mlp_ratio, drop_path, sr_ratio, variant="B"):
        super().__init__()
        self.dim, self.input_resolution = dim, tuple(input_resolution)
        self.norm1 = nn.LayerNorm(dim)
        self.in_proj = nn.Linear(dim, dim)
        self.act_proj = nn.Linear(dim, dim)
        self.dwc = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        self.attn = nn.ModuleList([
            AggregatedAttention(dim // 2, input_resolution, num_heads // 2, True, sr_ratio, variant),
            AggregatedAttention(dim // 2, input_resolution, num_heads // 2, False, sr_ratio, variant),
        ])
        self.out_proj = nn.Linear(dim, dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        H, W = self.input_resolution
        Bsz, C, h, w = x.shape
        assert (h, w) == (H, W), "input feature has wrong size"
        L = H * W
        x = x.reshape(Bsz, C, L).transpose(1, 2)
        shortcut = x
        x = self.norm1(x)
        act_res = F.silu(self.act_proj(x))
        x = self.in_proj(x).view(Bsz, H, W, C)
        x = F.silu(self.dwc(x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1).reshape(Bsz, L, C)
        xa, za = torch.chunk(x, 2, dim=-1)
        x = torch.cat([self.attn[0](xa, H, W), self.attn[1](za, H, W)], dim=-1)
        x = shortcut + self.drop_path(self.out_proj(x * act_res))
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x.reshape(Bsz, H, W, C).permute(0, 3, 1, 2)


class BasicLayer(nn.Module):  # T:918-969
    def __init__(self, dim, input_resolution, depth, num_heads, mlp_ratio, drop_path, sr_ratio, variant):
        super().__init__()
        self.blocks = nn.ModuleList([
            MLLABlock(dim, input_resolution, num_heads, mlp_ratio, drop_path[i], sr_ratio, variant)
            for i in range(depth)])

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


class Project(nn.Module):  # T:972-1001 ("project")
    def __init__(self, cin, cout, stride, last):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, stride=1, padding=1)
        self.norm1 = nn.LayerNorm(cout)
        self.last = last
        if not last:
            self.norm2 = nn.LayerNorm(cout)

    @staticmethod
    def _ln(norm, x):
        return norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2).contiguous()

    def forward(self, x):
        x = self._ln(self.norm1, F.gelu(self.conv1(x)))
        x = self.conv2(x)
        if not self.last:
            x = self._ln(self.norm2, F.gelu(x))
        return x


class PatchEmbed(nn.Module):  # T:1004-1043 (patch_norm False; all configs have even H, W)
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.proj1 = Project(in_chans, embed_dim // 2, (2, 2), last=False)
        self.proj2 = Project(embed_dim // 2, embed_dim, (patch_size // 2, patch_size // 2), last=True)

    def forward(self, x):
        return self.proj2(self.proj1(x))


class MedNeXtBlock(nn.Module):  # T:230-324 (2d, GroupNorm(groups=C), grn off)
    def __init__(self, cin, cout, exp_r, k=3, do_res=True, stride=1):
        super().__init__()
        self.do_res = do_res
        self.conv1 = nn.Conv2d(cin, cin, k, stride=stride, padding=k // 2, groups=cin)
        self.norm = nn.GroupNorm(cin, cin)
        self.conv2 = nn.Conv2d(cin, exp_r * cin, 1)
        self.conv3 = nn.Conv2d(exp_r * cin, cout, 1)

    def body(self, x):
        return self.conv3(F.gelu(self.conv2(self.norm(self.conv1(x)))))

    def forward(self, x):
        y = self.body(x)
        return x + y if self.do_res else y


class MedNeXtDownBlock(MedNeXtBlock):  # T:327-366
    def __init__(self, cin, cout, exp_r, k=3):
        super().__init__(cin, cout, exp_r, k, do_res=False, stride=2)
        self.res_conv = nn.Conv2d(cin, cout, 1, stride=2)

    def forward(self, x):
        return self.body(x) + self.res_conv(x)


class PatchExpand(nn.Module):  # T:479-546
    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.res_conv = nn.ConvTranspose2d(cin, cout, 1, stride=2)
        self.conv1 = nn.ConvTranspose2d(cin, cout, k, stride=2, padding=k // 2)
        self.norm = nn.GroupNorm(cin, cin)

    def forward(self, x):
        y = F.pad(self.conv1(self.norm(x)), (1, 0, 1, 0))
        return y + F.pad(self.res_conv(x), (1, 0, 1, 0))


class OutBlock(nn.Module):  # T:549-561
    def __init__(self, cin, n_classes):
        super().__init__()
        self.conv_out = nn.ConvTranspose2d(cin, n_classes, 1)

    def forward(self, x):
        return self.conv_out(x)


class MLLA_Enc(nn.Module):  # T:1046-1179
    def __init__(self, img_size, patch_size, in_chans, embed_dim, depths, num_heads, mlp_ratio,
                 drop_path_rate, sr_ratio, variant):
        super().__init__()
        self.num_layers = len(depths)
        res = [s // patch_size for s in img_size]
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                embed_dim * 2 ** i, (res[0] // 2 ** i, res[1] // 2 ** i), depths[i], num_heads[i],
                mlp_ratio, dpr[sum(depths[:i]):sum(depths[:i + 1])], sr_ratio[i], variant))
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


# --------------------------------------------------------------------------------------
# MSMM skip module (M:266-577, 669-804)
# --------------------------------------------------------------------------------------
def _dt_init(dt_rank, d_inner, dt_min=0.001, dt_max=0.1, floor=1e-4):  # M:348-374
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
        self.in_proj = nn.Linear(d_model, self.d_inner, bias=False)
        self.conv2d = nn.ModuleList([
            nn.Conv2d(self.d_inner, self.d_inner, 3, padding=1, groups=self.d_inner) for _ in range(stage_num)])
        xp = [nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False) for _ in range(4)]
        self.x_proj_weight = nn.Parameter(torch.stack([t.weight for t in xp], 0))        # (4, 35, 96)
        dts = [_dt_init(self.dt_rank, self.d_inner) for _ in range(4)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], 0))     # (4, 96, 3)
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], 0))         # (4, 96)
        A = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(4 * self.d_inner, 1)
        self.A_logs = nn.Parameter(torch.log(A))                                         # (384, 16)
        self.Ds = nn.Parameter(torch.ones(4 * self.d_inner))
        self.out_norm = nn.LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=False)

    def core(self, xs_img: List[torch.Tensor], scan_fn):
        """M:405-473: four-direction, multi-scale sequence; one scan; inverse re-ordering."""
        Bsz = xs_img[0].shape[0]
        K = 4
        seqs, Ls, HW = [], [], []
        for xi in xs_img:
            _, _, H, W = xi.shape
            L = H * W
            row = xi.reshape(Bsz, -1, L)
            col = xi.transpose(2, 3).reshape(Bsz, -1, L)
            both = torch.stack([row, col], dim=1)                                        # (B, 2, d, L)
            seqs.append(torch.cat([both, both.flip(-1)], dim=1))                         # (B, 4, d, L)
            Ls.append(L)
            HW.append((H, W))
        xs = torch.cat(seqs, dim=-1)
        Lc = xs.shape[-1]
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight)
        dts, Bs, Cs = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight)
        out = scan_fn(
            xs.float().reshape(Bsz, -1, Lc), dts.contiguous().float().reshape(Bsz, -1, Lc),
            -torch.exp(self.A_logs.float()), Bs.float().contiguous(), Cs.float().contiguous(),
            self.Ds.float(), z=None, delta_bias=self.dt_projs_bias.float().reshape(-1),
            delta_softplus=True, return_last_state=False).view(Bsz, K, -1, Lc)
        y = []
        for k in range(K):
            parts = list(torch.split(out[:, k], Ls, dim=-1))
            for j, p in enumerate(parts):
                H, W = HW[j]
                if k >= 2:
                    p = p.flip(-1)
                if k % 2 == 1:
                    p = p.reshape(Bsz, -1, W, H).transpose(2, 3).reshape(Bsz, -1, Ls[j])
                parts[j] = p
            y.append(torch.cat(parts, dim=-1))
        return y[0] + y[1] + y[2] + y[3]

    def forward(self, x, H, W, L_split, scan_fn=selective_scan_oracle):
        Bsz = x.shape[0]
        x = self.in_proj(x)
        imgs = []
        for i, xi in enumerate(torch.split(x, L_split, dim=1)):
            xi = xi.reshape(Bsz, H[i], W[i], -1).permute(0, 3, 1, 2).contiguous()
            imgs.append(F.silu(self.conv2d[i](xi)))
        y = self.core(imgs, scan_fn).transpose(1, 2).contiguous()
        return self.out_proj(self.out_norm(y))


class _DWConv(nn.Module):  # M:545-556
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)

    def forward(self, x, H, W):
        Bsz, N, C = x.shape
        return self.dwconv(x.transpose(1, 2).reshape(Bsz, C, H, W)).flatten(2).transpose(1, 2)


class ConvolutionalGLU(nn.Module):  # M:559-577
    def __init__(self, dim, hidden):
        super().__init__()
        hidden = int(2 * hidden / 3)
        self.fc1 = nn.Linear(dim, hidden * 2)
        self.dwconv = _DWConv(hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, H, W):
        x, v = self.fc1(x).chunk(2, dim=-1)
        return self.fc2(F.silu(self.dwconv(x, H, W)) * v)


class VSS_Conv_Block(nn.Module):  # M:669-753
    def __init__(self, feature_dims, hidden_dim, drop_path):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.conv_dims = [c - hidden_dim for c in feature_dims]
        self.ln_1 = nn.LayerNorm(hidden_dim)
        self.self_attention = SS2D_skip(len(feature_dims), hidden_dim)
        self.drop_path = DropPath(drop_path)
        self.norm2 = nn.LayerNorm(hidden_dim)
        self.mlps = nn.ModuleList([ConvolutionalGLU(hidden_dim, hidden_dim * 4) for _ in feature_dims])
        self.conv_branches = nn.ModuleList([
            nn.Sequential(nn.Conv2d(c, c, 3, padding=1), nn.InstanceNorm2d(c, affine=True), nn.SiLU())
            for c in self.conv_dims])

    def forward(self, inputs, scan_fn=selective_scan_oracle):
        Bsz = inputs[0].shape[0]
        H = [t.shape[2] for t in inputs]
        W = [t.shape[3] for t in inputs]
        Ls = [h * w for h, w in zip(H, W)]
        m_parts, c_parts = [], []
        for i, t in enumerate(inputs):
            m, c = torch.split(t, [self.hidden_dim, self.conv_dims[i]], dim=1)
            m_parts.append(m.flatten(2))
            c_parts.append(c)
        m = torch.cat(m_parts, dim=-1).permute(0, 2, 1).contiguous()
        m = m + self.drop_path(self.self_attention(self.ln_1(m), H, W, Ls, scan_fn))
        m = self.norm2(m)
        outs = []
        for i, mi in enumerate(torch.split(m, Ls, dim=1)):
            mi = mi + self.drop_path(self.mlps[i](mi, H[i], W[i]))
            mi = mi.transpose(1, 2).reshape(Bsz, -1, H[i], W[i])
            outs.append(torch.cat([mi, self.conv_branches[i](c_parts[i])], dim=1))
        return outs


class VSS_Conv_Layer(nn.Module):  # M:756-804
    def __init__(self, feature_dims, hidden_dim, depth=1, drop_path=0.0):
        super().__init__()
        self.blocks = nn.ModuleList([VSS_Conv_Block(feature_dims, hidden_dim, drop_path) for _ in range(depth)])

    def forward(self, x, scan_fn=selective_scan_oracle):
        for blk in self.blocks:
            x = blk(x, scan_fn)
        return x


# --------------------------------------------------------------------------------------
# full network (T:1183-1407)
# --------------------------------------------------------------------------------------
class MLLA_Uper(nn.Module):
    def __init__(self, img_size: Sequence[int], patch_size=2, in_channels=1, out_channels=14, embed_dim=96,
                 depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), mlp_ratio=2, dropout_path_rate=0.1,
                 sr_ratio=(16, 8, 4, 2), deep_supervision=True, variant="B"):
        super().__init__()
        self.deep_supervision = deep_supervision
        E = embed_dim
        self.mlla = MLLA_Enc(list(img_size), patch_size, in_channels, E, list(depths), list(num_heads),
                             mlp_ratio, dropout_path_rate, list(sr_ratio), variant)
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
        self.dummy_tensor = nn.Parameter(torch.tensor([1.0]))      # T:1362, never used in forward
        if deep_supervision:
            self.out_1 = OutBlock(E, out_channels)
            self.out_2 = OutBlock(2 * E, out_channels)
            self.out_3 = OutBlock(4 * E, out_channels)
            self.out_4 = OutBlock(8 * E, out_channels)

    def forward(self, x_in, scan_fn=selective_scan_oracle):
        hs = self.mlla(x_in)
        hs[1:] = self.mambaskip(hs[1:], scan_fn)
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


def build_reference_config_model(img_size, in_channels=1, num_classes=14, deep_supervision=True, variant="B"):
    """Hyper-parameters hard-coded at T:71-89."""
    return MLLA_Uper(img_size, 2, in_channels, num_classes, 96, (2, 2, 2, 2), (2, 4, 8, 16), 2, 0.1,
                     (16, 8, 4, 2), deep_supervision, variant)


# --------------------------------------------------------------------------------------
# loss (L/deep_supervision.py:17-34, L/compound_losses.py:31-57, L/dice.py:73-117,
#       L/robust_ce_loss.py:12-16) -- single-process form (no AllGatherGrad)
# --------------------------------------------------------------------------------------
def soft_dice_loss(logits, target, batch_dice=True, smooth=1e-5, loss_mask=None):
    """MemoryEfficientSoftDiceLoss.forward (L/dice.py:73-117), do_bg False; loss_mask (B, 1, ...) as in :97-107."""
    probs = logits.softmax(1)[:, 1:]
    axes = tuple(range(2, logits.ndim))
    with torch.no_grad():
        onehot = torch.zeros(logits.shape, dtype=torch.bool, device=logits.device)
        onehot.scatter_(1, target.long(), 1)
        onehot = onehot[:, 1:]
        sum_gt = onehot.sum(axes) if loss_mask is None else (onehot * loss_mask).sum(axes)
    intersect = (probs * onehot).sum(axes) if loss_mask is None else (probs * onehot * loss_mask).sum(axes)
    sum_pred = probs.sum(axes) if loss_mask is None else (probs * loss_mask).sum(axes)
    if batch_dice:
        intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
    dc = (2 * intersect + smooth) / torch.clip(sum_gt + sum_pred + smooth, 1e-8)
    return -dc.mean()


def dc_and_ce_loss(logits, target, batch_dice=True, ignore_label=None):
    """DC_and_CE_loss.forward (L/compound_losses.py:31-57): with an ignore label (:38-46) the ignored pixels are masked out of
    the dice sums (their label replaced by 0) and skipped by the cross-entropy (ignore_index: mean over the others); a batch
    without a single annotated pixel has no cross-entropy term (:50-51)."""
    if ignore_label is None:
        return F.cross_entropy(logits, target[:, 0].long()) + soft_dice_loss(logits, target, batch_dice)
    mask = (target != ignore_label).bool()
    target_dice = torch.clone(target)
    target_dice[target == ignore_label] = 0
    dc = soft_dice_loss(logits, target_dice, batch_dice, loss_mask=mask)
    ce = F.cross_entropy(logits, target[:, 0].long(), ignore_index=int(ignore_label)) if mask.sum() > 0 else 0
    return ce + dc


def deep_supervision_weights(n=5):  # T:118-126
    w = torch.tensor([1.0 / 2 ** i for i in range(n)], dtype=torch.float64)
    return (w / w.sum()).tolist()


def deep_supervision_loss(outputs, targets, batch_dice=True, ignore_label=None):
    ws = deep_supervision_weights(len(outputs))
    total = ws[0] * dc_and_ce_loss(outputs[0], targets[0], batch_dice, ignore_label)
    for w, o, t in zip(ws[1:], outputs[1:], targets[1:]):
        total = total + w * dc_and_ce_loss(o, t, batch_dice, ignore_label)
    return total


# --------------------------------------------------------------------------------------
# synthetic batches + train step (B:833-863; benchmarking/...noDataLoading.py:16-22)
# --------------------------------------------------------------------------------------
def synthetic_batch(batch, in_ch, H, W, n_cls, seed=1234, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    data = torch.rand(batch, in_ch, H, W, generator=g)
    target = [torch.round(torch.rand(batch, 1, H >> s, W >> s, generator=g) * (n_cls - 1)) for s in range(5)]
    return data.to(device), [t.to(device) for t in target]


def make_optimizer(model):  # T:137-147
    return torch.optim.AdamW(model.parameters(), 5e-4, weight_decay=3e-5, eps=1e-4)


def train_step(model, optim, data, target, batch_dice=True, scan_fn=selective_scan_oracle):
    optim.zero_grad()
    out = model(data, scan_fn)
    loss = deep_supervision_loss(out, target, batch_dice)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 12)
    optim.step()
    return loss.detach()


# --------------------------------------------------------------------------------------
# deterministic, construction-order-independent weights (lets the reference model, this
# oracle and the product model share identical parameters without shipping 108 MB)
# --------------------------------------------------------------------------------------
def deterministic_fill_(state_dict_like, seed=0):
    """Overwrite every tensor of a state_dict (keyed by the ABI names) in place from a
    per-key seeded generator.  Ranges keep the dynamics realistic (norm scales near 1,
    A_logs near log(1..16), dt bias in the softplus^-1([1e-3, 1e-1]) band)."""
    import zlib
    for key in sorted(state_dict_like.keys()):
        t = state_dict_like[key]
        if not torch.is_floating_point(t):
            continue
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + seed) & 0x7FFFFFFF)
        u = torch.rand(t.shape, generator=g, dtype=torch.float32) * 2 - 1          # U(-1, 1)
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("A_logs"):
            base = torch.log(torch.arange(1, t.shape[1] + 1, dtype=torch.float32)).expand_as(t)
            v = base + 0.1 * u
        elif key.endswith("Ds"):
            v = 1.0 + 0.1 * u
        elif key.endswith("dt_projs_bias"):
            v = -4.5 + 2.3 * u
        elif key.endswith("dummy_tensor"):
            v = torch.ones_like(u)
        elif leaf.startswith("lambda_"):
            v = 0.1 * u
        elif t.ndim == 1 and leaf == "weight":          # LayerNorm / GroupNorm / InstanceNorm / RMSNorm scales
            v = 1.0 + 0.1 * u
        elif t.ndim == 1:                               # biases
            v = 0.05 * u
        else:
            if key.endswith("x_proj_weight") or key.endswith("dt_projs_weight"):
                fan_in = t.shape[-1]                    # stacked per-direction Linear weights
            else:
                fan_in = t[0].numel()                   # only a scale; exact fan-in is irrelevant
            v = u * (1.5 / max(fan_in, 1)) ** 0.5
        t.copy_(v.to(t.dtype))
    return state_dict_like
