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
    # a sample of the input is one buffer resource with 32-bit byte offsets: I planes under 2 GB (and planes under 2^32 / 36 elements);
    # larger ones are refused, not wrapped
    pmax = ((1 << 29) - 17) // 16
    assert lib.mlagg_conv3x3_supported(16, 16, 1, pmax) and not lib.mlagg_conv3x3_supported(16, 16, 1, pmax + 1)
    assert lib.mlagg_conv3x3_supported(48, 16, 1, pmax) and not lib.mlagg_conv3x3_supported(16, 32, 1, pmax)
    assert lib.mlagg_conv3x3x3_supported(16, 16, 320, 320, 320) and not lib.mlagg_conv3x3x3_supported(16, 16, 324, 324, 324)
    assert lib.mlagg_conv3x3x3_wgrad_supported(32, 32, 8, 16, 40) and not lib.mlagg_conv3x3x3_wgrad_supported(32, 32, 8, 16, 20)
    assert lib.mlagg_conv3x3_workspace_bytes(48, 96) == 3 * 9 * 48 * 96 * 2 and lib.mlagg_conv3x3x3_workspace_bytes(32, 32) == 3 * 27 * 32 * 32 * 2
    assert lib.mlagg_column_sum_workspace_floats(100, 64) == 0 and lib.mlagg_column_sum_workspace_floats(7840, 384) > 0


def test_weight_gradient_policy_reads_only_shapes():
    assert ops._k19_wgrad(96, 48, 256, 256) and ops._k19_wgrad(48, 96, 128, 128)
    assert ops._k19_wgrad(48, 48, 256, 256)              # round 4: 16-wide tiles fill 48 x 48 channels (236 vs MIOpen's 486 us)
    assert ops._k19_wgrad(48, 1, 256, 256)               # ... the one-channel stem included (a 48 x 16 tile)
    assert not ops._k19_wgrad(32, 32, 256, 256)          # below 48 channels: the library in fp32 ...
    assert ops._k19_wgrad(32, 32, 256, 256, 1)           # ... K19's one-product form in the 16-bit modes
    assert not ops._k19_wgrad(720, 720, 16, 16)          # 256 pixels
    assert ops._k18_product(192, 96, 128 * 128) and ops._k18_product(192, 96, 64 * 64) and not ops._k18_product(192, 96, 32 * 32)
    assert not ops._k18_product(96, 48, 256 * 256)        # contraction 48 between two wide sides: the library's GEMM kernels


def test_residual_norm_off_the_device_is_residual_then_norm():
    torch.manual_seed(0)
    dp = PM.DropPath(0.0)
    norm = torch.nn.LayerNorm(96)          # (the product LayerNorm is a device kernel and refuses host tensors)
    skip, branch = torch.randn(2, 5, 96), torch.randn(2, 5, 96)
    x, n = dp.residual_norm(skip, branch, norm)
    assert torch.equal(x, skip + branch)
    assert torch.allclose(n, torch.nn.functional.layer_norm(skip + branch, (96,), norm.weight, norm.bias, norm.eps))


def test_stacked_projection_weights_refresh_route_gradients_and_survive_deepcopy():
    """model._Stack: values = torch.cat of the sources (a row slice of kv.weight and the 1 x 1 convolution's weight among them),
    gradients land in the sources, a refreshed stack is not copied again while its sources are unchanged, a copy of the module stacks
    ITS parameters, and the state_dict has the reference's keys only."""
    import copy
    att = PM.AggregatedAttention(96, (8, 8), 2, False, 2, "B")
    d = 96
    w, b = att._w_stack.get(), att._b_stack.get()
    assert torch.equal(w, torch.cat([att.q.weight, att.kv.weight[d:], att.sr.weight.view(d, d)]))
    assert torch.equal(b, torch.cat([att.q.bias, att.kv.bias[d:], att.sr.bias]))
    (w.sum() * 2 + b.sum()).backward()
    assert torch.equal(att.q.weight.grad, torch.full_like(att.q.weight, 2.0))
    assert float(att.kv.weight.grad[:d].abs().max()) == 0 and torch.equal(att.kv.weight.grad[d:], torch.full((d, d), 2.0))
    assert torch.equal(att.sr.weight.grad, torch.full_like(att.sr.weight, 2.0)) and torch.equal(att.sr.bias.grad, torch.ones(d))
    with torch.no_grad():
        att.q.weight.add_(1.0)                                  # an optimizer step
    assert torch.equal(att._w_stack.get()[:d].detach(), att.q.weight.detach())      # a stack nobody refreshed copies on use
    PM.refresh_stacks([att._w_stack])
    v = att._w_stack.buf._version
    att._w_stack.get()
    PM.refresh_stacks([att._w_stack])
    assert att._w_stack.buf._version == v                      # sources unchanged: no second copy
    c = copy.deepcopy(att)
    with torch.no_grad():
        c.q.weight.zero_()
    assert float(c._w_stack.get()[:d].abs().max()) == 0 and float(att._w_stack.get()[:d].abs().max()) > 0
    assert sorted(att.state_dict()) == sorted(k for k in att.state_dict() if "stack" not in k) and len(att.state_dict()) == 15


def test_weight_stacks_survive_two_forwards_and_never_serve_stale_rows():
    """model._Stack rewrites its buffer only when a source changed: two forward passes between optimizer steps share one unmodified
    buffer (both backward passes run: no 'modified by an inplace operation'), an in-place parameter change or ClipAdamW's epoch bump
    makes the next use copy again -- also for a stack the network refreshed but did not consume."""
    from mlagg_unet_amd import model
    a, b = torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(2, 4))
    st = model._Stack(lambda: [a, b])
    x = torch.randn(5, 4)
    y1 = (x @ st.get().t()).sum()
    v = st.buf._version
    y2 = (2 * x @ st.get().t()).sum()                     # second forward before the first backward
    assert st.buf._version == v                           # ... did not touch the buffer
    y1.backward()
    y2.backward()
    assert torch.allclose(a.grad, 3 * x.sum(0).expand(3, 4)) and torch.allclose(b.grad, 3 * x.sum(0).expand(2, 4))
    with torch.no_grad():
        a.mul_(2.0)                                       # an optimizer step of torch's: version counter moves
    assert torch.equal(st.get()[:3], a.detach())
    model.refresh_stacks([st])                            # refreshed by "the network", not consumed ...
    a.data.add_(1.0)                                      # ... then rewritten behind autograd's back (what ClipAdamW's kernels do)
    ops.invalidate_weight_images()                        # ... which is why ClipAdamW bumps the epoch
    assert torch.equal(st.get()[:3], a.detach())
