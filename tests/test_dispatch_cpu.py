"""CPU tier: host-side dispatch logic added in round 3 (no kernel runs here): which tensors the split-bf16 convolution kernels and the
gradient arena accept, and that everything else keeps the plain torch semantics."""
import torch

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import model as PM, ops


def test_split_cols_is_a_plain_split_off_the_device():
    t = torch.randn(2, 6, 20, requires_grad=True)
    a, b, c = ops.split_cols(t * 1.0, (8, 4, 8))
    ra, rb, rc = (t * 1.0).split([8, 4, 8], dim=-1)
    assert torch.equal(a, ra) and torch.equal(b, rb) and torch.equal(c, rc)
    assert not hasattr(a, "_mlagg_slot")
    (a.sum() + 2 * c.sum()).backward()
    want = torch.cat([torch.ones(2, 6, 8), torch.zeros(2, 6, 4), 2 * torch.ones(2, 6, 8)], dim=-1)
    assert torch.equal(t.grad, want)


def test_convolution_kernels_decline_host_tensors_and_unsupported_geometries():
    x = torch.randn(2, 96, 64, 64)
    assert not ops.conv1x1_supported(x, torch.randn(192, 96, 1, 1), (1, 1), (0, 0), (1, 1), 1)
    assert not ops.conv3x3_supported(x, torch.randn(96, 96, 3, 3), (1, 1), (1, 1), (1, 1), 1)
    assert not ops.conv3x3x3_supported(torch.randn(1, 32, 8, 16, 16), torch.randn(32, 32, 3, 3, 3), (1, 1, 1), (1, 1, 1))
    # the shape rules themselves (they only read sizes): strides, paddings, groups, kernel sizes
    lib = mlagg_unet_amd._lib.lib()
    assert lib.mlagg_conv1x1_supported(192, 96, 64 * 64) and not lib.mlagg_conv1x1_supported(192, 24, 64 * 64)        # contraction % 16
    assert not lib.mlagg_conv1x1_supported(192, 96, 50)                                                                # pixels % 16, >= 96
    assert lib.mlagg_conv3x3_supported(48, 48, 16, 16) and not lib.mlagg_conv3x3_supported(48, 1, 256, 256)            # the one-channel stem
    assert lib.mlagg_conv3x3_wgrad_supported(96, 48, 32, 32) and not lib.mlagg_conv3x3_wgrad_supported(96, 48, 30, 30)  # W % 8
    assert lib.mlagg_conv3x3x3_supported(32, 32, 8, 16, 16) and not lib.mlagg_conv3x3x3_supported(32, 1, 8, 16, 16)
    assert lib.mlagg_conv3x3x3_wgrad_supported(32, 32, 8, 16, 40) and not lib.mlagg_conv3x3x3_wgrad_supported(32, 32, 8, 16, 20)
    assert lib.mlagg_conv3x3_workspace_bytes(48, 96) == 3 * 9 * 48 * 96 * 2 and lib.mlagg_conv3x3x3_workspace_bytes(32, 32) == 3 * 27 * 32 * 32 * 2
    assert lib.mlagg_column_sum_workspace_floats(100, 64) == 0 and lib.mlagg_column_sum_workspace_floats(7840, 384) > 0


def test_weight_gradient_policy_reads_only_shapes():
    assert ops._k19_wgrad(96, 48, 256, 256) and ops._k19_wgrad(48, 96, 128, 128)
    assert not ops._k19_wgrad(48, 48, 256, 256)          # 48 x 48 channels: 56 % tile fill, MIOpen ties or wins
    assert not ops._k19_wgrad(720, 720, 16, 16)          # 256 pixels
    assert ops._k18_product(192, 96, 128 * 128) and not ops._k18_product(192, 96, 64 * 64) and not ops._k18_product(96, 48, 256 * 256)


def test_residual_norm_off_the_device_is_residual_then_norm():
    torch.manual_seed(0)
    dp = PM.DropPath(0.0)
    norm = torch.nn.LayerNorm(96)          # (the product LayerNorm is a device kernel and refuses host tensors)
    skip, branch = torch.randn(2, 5, 96), torch.randn(2, 5, 96)
    x, n = dp.residual_norm(skip, branch, norm)
    assert torch.equal(x, skip + branch)
    assert torch.allclose(n, torch.nn.functional.layer_norm(skip + branch, (96,), norm.weight, norm.bias, norm.eps))
