"""GPU: the 3-D selective-scan block on the MI355X kernels (mlagg_unet_amd.ss3d.SS3D: K1' by index table + K1 with 12 groups)
against the REFERENCE's own SS3D outputs and gradients (tests/golden/ss3d.npz), plus the index kernels alone."""
import os

import numpy as np
import pytest
import torch

from oracle import mlagg_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ss3d.npz"))


def test_index_scan_and_merge_are_transposes():
    from mlagg_unet_amd import ops, ss3d
    D, H, W, C, B = 5, 7, 6, 40, 2
    L = D * H * W
    idx = ss3d.scan_orders_3d(D, H, W, DEV)
    g = torch.Generator().manual_seed(1)
    tok = torch.randn(B, L, C, generator=g)
    seq = ops.index_scan(tok.to(DEV), idx, C)
    want = torch.stack([tok[:, idx[k].cpu().long(), :].transpose(1, 2) for k in range(12)], 1).reshape(B, 12 * C, L)
    assert torch.equal(seq.cpu(), want)                                  # pure data movement: bit-exact
    y = torch.randn(B, 12 * C, L, generator=g)
    merged = ops.index_merge(y.to(DEV), idx, C)
    ref = torch.zeros(B, L, C, dtype=torch.float64)
    for k in range(12):
        ref[:, idx[k].cpu().long(), :] += y[:, k * C:(k + 1) * C].transpose(1, 2).double()
    assert float((merged.cpu().double() - ref).abs().max()) < 1e-5     # 12-way float sum
    assert torch.equal(merged, ops.index_merge(y.to(DEV), idx, C))      # ... in a fixed order: bit-reproducible
    # column-block form: direction k reads its own columns of a wide row (the x_proj output)
    wide = torch.randn(B, L, 12 * 9, generator=g)
    rows = ops.index_scan(wide.to(DEV), idx, 4, 9, 3)
    want = torch.stack([wide[:, idx[k].cpu().long(), 9 * k + 3:9 * k + 7].transpose(1, 2) for k in range(12)], 1).reshape(B, 48, L)
    assert torch.equal(rows.cpu(), want)


def test_ss3d_block_matches_reference_golden():
    from mlagg_unet_amd import ss3d
    blk = ss3d.SS3D(16)
    O.deterministic_fill_(blk.state_dict(), seed=12)
    blk = blk.to(DEV).eval()
    x = torch.from_numpy(G["x"]).to(DEV).requires_grad_(True)
    y = blk(x)
    assert float((y.detach().cpu() - torch.from_numpy(G["y"])).abs().max()) < 1e-3          # north_star tolerance
    y.backward(torch.from_numpy(G["gy"]).to(DEV))
    gx = torch.from_numpy(G["gx"])
    assert float((x.grad.cpu() - gx).abs().max()) < 1e-3 * max(1.0, float(gx.abs().max()))
    for n, p in blk.named_parameters():
        ref = torch.from_numpy(G["grad/" + n])
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 2e-3 * max(1.0, float(ref.abs().max())), (n, err)


def test_ss3d_at_a_btcv_like_stage_shape():
    """A stage of the BASELINE configs[3] patch (96 x 160 x 160 / 4 per axis): L = 24 * 40 * 40 = 38400 tokens, 12 directions --
    runs, is finite, and the 12-way merge of an all-ones scan output counts 12 everywhere."""
    from mlagg_unet_amd import ops, ss3d
    torch.manual_seed(0)
    blk = ss3d.SS3D(48).to(DEV)
    x = torch.randn(1, 24, 40, 40, 48, device=DEV, requires_grad=True)
    y = blk(x)
    y.square().mean().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    idx = ss3d.scan_orders_3d(24, 40, 40, DEV)
    ones = torch.ones(1, 12 * 8, 38400, device=DEV)
    assert torch.equal(ops.index_merge(ones, idx, 8), torch.full((1, 38400, 8), 12.0, device=DEV))


@pytest.mark.parametrize("B,D,H,W,C,silu,bias", [(2, 5, 7, 6, 40, True, True), (1, 1, 3, 9, 96, False, True), (2, 4, 4, 4, 8, True, False),
                                                 (1, 9, 10, 11, 96, True, True)])
def test_dwconv3d_matches_torch_conv3d(B, D, H, W, C, silu, bias):
    """K2v (token-major depthwise 3x3x3 + SiLU) against F.conv3d(groups=C) in double precision: forward and all gradients;
    volumes thinner than the kernel (D = 1), odd extents, channel counts that leave a partial 64-channel block."""
    import torch.nn.functional as F
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(B * 100 + D + C)
    L = D * H * W
    x = torch.randn(B, L, C, generator=g)
    w = torch.randn(C, 1, 3, 3, 3, generator=g) * 0.3
    b = torch.randn(C, generator=g) if bias else None
    gy = torch.randn(B, L, C, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    vol = F.conv3d(xr.transpose(1, 2).reshape(B, C, D, H, W), wr, br, padding=1, groups=C)
    yr = (F.silu(vol) if silu else vol).reshape(B, C, L).transpose(1, 2)
    yr.backward(gy.double())
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True) if bias else None
    y = ops.dwconv3d_nlc(xg, wg, bg, (D, H, W), silu=silu)
    y.backward(gy.to(DEV))
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 2e-5
    s = max(1.0, float(wr.grad.abs().max()))
    assert float((wg.grad.cpu().double() - wr.grad).abs().max()) < 2e-5 * s * 5
    if bias:
        assert float((bg.grad.cpu().double() - br.grad).abs().max()) < 1e-4 * max(1.0, float(br.grad.abs().max()))
    # a strided input (channel block of a wider row, as after a stacked projection) is taken without a copy
    wide = torch.randn(B, L, C + 8, generator=g).to(DEV)
    y2 = ops.dwconv3d_nlc(wide[..., 4:4 + C], wg.detach(), bg.detach() if bias else None, (D, H, W), silu=silu)
    y3 = ops.dwconv3d_nlc(wide[..., 4:4 + C].contiguous(), wg.detach(), bg.detach() if bias else None, (D, H, W), silu=silu)
    assert torch.equal(y2, y3)
