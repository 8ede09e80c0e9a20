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
    assert float((merged.cpu().double() - ref).abs().max()) < 1e-5     # 12-way float sum, order not fixed
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
