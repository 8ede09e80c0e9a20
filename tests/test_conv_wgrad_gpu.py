"""GPU parity: K15, the tap-GEMM weight gradient of the full convolutions (csrc/conv_wgrad.hip through the C ABI), against
torch's convolution weight gradient in float64 on the host; and ops.conv_nd (MIOpen forward / data gradient + K15) end to end."""
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # (B, I, O, dims, k, stride)
    (2, 3, 5, (5, 6, 7), 3, 1),            # 3-D 3x3x3, odd extents, partial 32-channel tiles
    (1, 40, 70, (4, 9, 10), 3, 1),         # more than one tile of input and output channels (27 taps: three tap groups)
    (2, 4, 6, (6, 5, 8), 3, 2),            # stride 2: the 8 parity phases, odd and even extents
    (1, 33, 8, (7, 7, 3), 3, 2),
    (2, 6, 4, (5, 6, 7), 1, 1),            # 1x1x1
    (2, 6, 4, (6, 5, 8), 1, 2),            # 1x1x1 stride 2 (the residual convolution of a down-sampling BasicResBlock)
    (2, 1, 32, (8, 16, 16), 3, 1),         # the stem: one input channel
    (2, 5, 7, (9, 13), 3, 1),              # 2-D 3x3
    (1, 48, 96, (12, 20), 1, 1),           # 2-D 1x1
    (1, 32, 32, (24, 40, 40), 3, 1),       # a stage of BASELINE configs[3] at 1/64 of its voxels: several slabs per sample
]


@gpu
@pytest.mark.parametrize("B,I,O,dims,k,stride", CASES)
def test_conv_weight_grad_matches_torch(B, I, O, dims, k, stride):
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(B * 100 + I + O + k + stride)
    nd = len(dims)
    x = torch.randn(B, I, *dims, generator=g)
    w = torch.randn(O, I, *([k] * nd), generator=g).double().requires_grad_(True)
    conv = F.conv3d if nd == 3 else F.conv2d
    y = conv(x.double(), w, None, stride, k // 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dev = torch.device("cuda:0")
    assert ops.conv_wgrad_supported(x.to(dev), w.float().to(dev), (stride,) * nd, (k // 2,) * nd)
    got = ops.conv_weight_grad(x.to(dev), dy.to(dev), k, stride).view(w.shape).cpu().double()
    ref = w.grad
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-5, float((got - ref).abs().max())


@gpu
def test_conv_nd_forward_and_both_gradients():
    from mlagg_unet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 6, 10, 12, generator=g)
    w = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.2
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, 2, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = ops.conv_nd(xg, wg, (2, 2, 2), (1, 1, 1))
    y.backward(dy.to(dev))
    for name, a, b in (("y", y.detach(), yr.detach()), ("dx", xg.grad, xr.grad), ("dW", wg.grad, wr.grad)):
        assert float((a.cpu().double() - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-5, name
    # anisotropic stride (the last pooling of the BTCV plan): not a K15 shape, plain library path, same numbers
    y2 = ops.conv_nd(xg, wg, (1, 2, 2), (1, 1, 1))
    assert float((y2.detach().cpu().double() - F.conv3d(xr, wr, None, (1, 2, 2), 1).detach()).abs().max()) < 1e-4


K16_CASES = [
    # (B, I, O, dims, k)              stride 1, last extent a multiple of 4, >= 8 input channels
    (2, 8, 40, (5, 6, 8), 3),          # two tiles of output channels, partial chunk of input channels
    (1, 40, 8, (4, 9, 12), 3),         # two chunks of input channels (32 + 8)
    (2, 16, 6, (3, 5, 8), 1),          # 1x1x1
    (1, 32, 32, (24, 40, 40), 3),      # a stage of BASELINE configs[3] at 1/64 of its voxels
    (1, 9, 33, (2, 3, 4), 3),          # odd channel counts, a volume smaller than one workgroup's 512 voxels
    # "last workgroup past Q" (the geometry behind commit 3cf46f4): the padded box holds Q = 5 * 13 * 16 = 1040 = 2 * 512 + 16 voxels,
    # so 496 of the last workgroup's 512 positions lie past it -- their loads must be clamped into the guarded rows (before the fix
    # they ran up to a plane + a row beyond the LAST sample's guard, i.e. off the end of the buffer)
    (2, 16, 16, (3, 11, 8), 3),
]


@gpu
@pytest.mark.parametrize("B,I,O,dims,k", K16_CASES)
def test_conv_taps_forward_and_gradients_match_torch(B, I, O, dims, k):
    """K16 (forward, data gradient) + K15 (weight gradient) behind ops.conv_nd against torch's conv3d in float64."""
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(I * 7 + O)
    x = torch.randn(B, I, *dims, generator=g)
    w = torch.randn(O, I, k, k, k, generator=g) * (I * k ** 3) ** -0.5
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, 1, k // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    dev = torch.device("cuda:0")
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    assert ops.conv_taps_supported(xg, wg, (1, 1, 1), (k // 2,) * 3)
    y = ops.conv_nd(xg, wg, (1, 1, 1), (k // 2,) * 3)
    y.backward(dy.to(dev))
    for name, a, b in (("y", y.detach(), yr.detach()), ("dx", xg.grad, xr.grad), ("dW", wg.grad, wr.grad)):
        err = float((a.cpu().double() - b).abs().max())
        assert err <= 2e-5 * float(b.abs().max()) + 1e-5, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,O,H,W,sliced", [(2, 96, 192, 32, 32, False), (3, 192, 96, 16, 24, True), (2, 48, 96, 24, 20, False),
                                              (1, 384, 768, 8, 12, False), (2, 96, 14, 16, 16, False), (2, 768, 384, 16, 7 * 16, True),
                                              (2, 48, 40, 10, 16, False), (2, 1, 48, 32, 32, False), (2, 24, 48, 16, 16, True),
                                              (1, 14, 48, 12, 16, False)])
def test_conv1x1_matches_float64(monkeypatch, B, I, O, H, W, sliced):
    """K18 (1 x 1 convolution on the 16-bit matrix instructions, fp32 operands as three bf16 pieces): output, data gradient and
    weight gradient against float64 conv2d, at the error of an fp32 GEMM.  Cases: channel counts off the 96-wide groups, pixel
    counts that are not multiples of 96 (clamped lanes), a channel-slice input (sample stride > C * P), and an output width (14,
    40) whose data gradient has a contraction that is not a multiple of 16, and inputs of 1 / 24 / 14 channels (the stem, a head's
    data gradient): ragged contractions run on zero-padded weight columns (mlagg_conv1x1_fwd_ragged)."""
    from mlagg_unet_amd import ops
    for name, v in (("K18_FWD_MIN_PIXELS", 0), ("K18_FWD_MIN_K", 16), ("K18_WGRAD_MIN_PIXELS", 0)):      # every product on K18
        monkeypatch.setattr(ops, name, v)
    g = torch.Generator().manual_seed(B * I + O)
    wide = torch.randn(B, I + 16, H, W, generator=g).to(DEV)
    x = (wide[:, 8:8 + I] if sliced else wide[:, :I].contiguous()).detach()
    w = (torch.randn(O, I, 1, 1, generator=g) * I ** -0.5).to(DEV)
    gy = torch.randn(B, O, H, W, generator=g).to(DEV)
    assert ops.conv1x1_supported(x, w, (1, 1), (0, 0), (1, 1), 1) and ops._k18_product(O, I, H * W)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr)
    yr.backward(gy.double())
    xp, wp = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yp = ops.conv1x1(xp, wp)
    yp.backward(gy)
    for name, got, want in (("y", yp, yr), ("dx", xp.grad, xr.grad), ("dW", wp.grad, wr.grad)):
        err = float((got.detach().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert err < 2e-6, (name, err)


@pytest.mark.gpu
def test_conv1x1_dispatch_rules_and_large_map():
    """ops.conv1x1_supported / Conv1x1Fn: which convolutions take K18, and a 128 x 128 map where all three products run on it."""
    from mlagg_unet_amd import ops
    x = torch.randn(2, 96, 128, 128, device=DEV)
    w = torch.randn(192, 96, 1, 1, device=DEV) * 0.1
    assert ops.conv1x1_supported(x, w, (1, 1), (0, 0), (1, 1), 1) and ops._k18_product(192, 96, 128 * 128) and ops._k18_product(96, 192, 128 * 128)
    assert not ops.conv1x1_supported(x, w, (2, 2), (0, 0), (1, 1), 1)                   # strided
    assert not ops.conv1x1_supported(x[:, :, :32, :32], w, (1, 1), (0, 0), (1, 1), 1)    # 1024 pixels: the library is faster
    assert ops._k18_product(48, 24, 128 * 128) and ops._k18_product(14, 48, 256 * 256) and ops._k18_product(48, 14, 256 * 256)   # thin sides
    assert not ops._k18_product(48, 48, 128 * 128) and not ops._k18_product(96, 72, 128 * 128)     # short contraction / not a multiple of 16
    gy = torch.randn(2, 192, 128, 128, device=DEV)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    yr.backward(gy.double())
    xp, wp = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yp = ops.conv1x1(xp, wp)
    yp.backward(gy)
    for name, got, want in (("y", yp, yr), ("dx", xp.grad, xr.grad), ("dW", wp.grad, wr.grad)):
        err = float((got.detach().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert err < 2e-6, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,O,H,W,sliced", [(2, 48, 48, 16, 16, False), (2, 96, 48, 12, 20, True), (1, 144, 144, 10, 10, False),
                                              (2, 16, 40, 9, 13, False), (1, 48, 96, 3, 32, False), (2, 32, 14, 24, 5, False),
                                              (1, 48, 48, 128, 128, False), (3, 48, 33, 6, 16, True), (2, 16, 16, 2, 48, False),
                                              (2, 80, 48, 7, 64, False), (1, 48, 48, 256, 256, False), (1, 96, 48, 256, 256, False)])
def test_conv3x3_matches_float64(monkeypatch, B, I, O, H, W, sliced):
    """K19 (3 x 3 convolution as nine shifted split-bf16 GEMMs): output and data gradient against float64 conv2d at the error of an
    fp32 convolution.  Cases: widths that are not multiples of the lanes' pixel runs (runs straddle image rows), three-row images,
    a 5-pixel-wide image, pixel counts that are not multiples of the 64 / 96-pixel tiles (clamped lanes), a channel-slice input,
    output widths off the 32-channel tiles, and one full-size map."""
    from mlagg_unet_amd import ops
    monkeypatch.setattr(ops, "K19_MIN_PIXELS", 0)
    monkeypatch.setattr(ops, "_k19_wgrad", lambda O, I, H, W, form=3: W % 16 == 0)      # the weight gradient too where the kernel supports it
    g = torch.Generator().manual_seed(B * I + O + H)
    wide = torch.randn(B, I + 16, H, W, generator=g).to(DEV)
    x = (wide[:, 8:8 + I] if sliced else wide[:, :I].contiguous()).detach()
    w = (torch.randn(O, I, 3, 3, generator=g) * (9 * I) ** -0.5).to(DEV)
    gy = torch.randn(B, O, H, W, generator=g).to(DEV)
    assert ops.conv3x3_supported(x, w, (1, 1), (1, 1), (1, 1), 1) and ops._k19_product(O, I, H, W)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(gy.double())
    xp, wp = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yp = ops.conv3x3(xp, wp)
    yp.backward(gy)
    for name, got, want, tol in (("y", yp, yr, 2e-6), ("dx", xp.grad, xr.grad, 2e-6), ("dW", wp.grad, wr.grad, 4e-6)):
        err = float((got.detach().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert err < tol, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,O,H,W", [(2, 96, 48, 32, 32), (1, 48, 20, 16, 24), (3, 32, 8, 8, 16)])
def test_transposed_2x2_stride2_convolution_on_k18(monkeypatch, B, I, O, H, W):
    """ops.conv_t2x2 (K18 on the (4 O, I) tap matrix + pixel shuffle; UnetrUpBlock's transposed convolution, T:1340-1368) against
    float64 conv_transpose2d: output, data gradient, weight gradient."""
    from mlagg_unet_amd import ops
    for name, v in (("K18_FWD_MIN_PIXELS", 0), ("K18_FWD_MIN_K", 16)):
        monkeypatch.setattr(ops, name, v)
    g = torch.Generator().manual_seed(B * I + O)
    x = torch.randn(B, I, H, W, generator=g).to(DEV)
    w = (torch.randn(I, O, 2, 2, generator=g) * I ** -0.5).to(DEV)
    gy = torch.randn(B, O, 2 * H, 2 * W, generator=g).to(DEV)
    assert ops.conv_t2x2_supported(x, w, (2, 2), (0, 0), (0, 0), (1, 1), 1)
    assert not ops.conv_t2x2_supported(x, w, (2, 2), (1, 1), (0, 0), (1, 1), 1) and not ops.conv_t2x2_supported(x, w[..., :1], (2, 2), (0, 0), (0, 0), (1, 1), 1)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2)
    yr.backward(gy.double())
    xp, wp = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yp = ops.conv_t2x2(xp, wp)
    yp.backward(gy)
    for name, got, want in (("y", yp, yr), ("dx", xp.grad, xr.grad), ("dW", wp.grad, wr.grad)):
        assert got.shape == want.shape, name
        err = float((got.detach().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert err < 2e-6, (name, err)


@pytest.mark.gpu
def test_channel_slice_gradients_are_read_in_place():
    """The halves of `torch.cat([up, skip], 1)`'s gradient (T:1360-1366) are channel slices of one map: K10's backward
    (mlagg_plane_norm_bwd_strided) and the inverse pixel shuffle of the kernel-2 transposed convolution read them where they are;
    results equal those from contiguous copies, bit for bit."""
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 48, 16, 24
    wide = torch.randn(B, 2 * C, 2 * H, 2 * W, generator=g).to(DEV)
    x = torch.randn(B, C, 2 * H, 2 * W, generator=g).to(DEV)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    outs = []
    for sl in (wide[:, C:], wide[:, C:].contiguous()):
        xs, gs, bs = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        y = ops.plane_norm(xs, gs, bs, 1e-5, ops.ACT_LEAKY, 0.01)
        y.backward(sl)
        outs.append((xs.grad, gs.grad, bs.grad))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    xi = torch.randn(B, 96, H, W, generator=g).to(DEV)
    wt = (torch.randn(96, C, 2, 2, generator=g) * 0.1).to(DEV)
    outs = []
    for sl in (wide[:, :C], wide[:, :C].contiguous()):
        xs, ws = xi.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        ops.conv_t2x2(xs, ws).backward(sl)
        outs.append((xs.grad, ws.grad))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert ops._map_slice(wide[:, C:], "dy")[1] == wide.stride(0) and ops._map_slice(wide[:, :, ::2], "dy")[1] == 0     # rows skipped: copied


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(32, 32), (8, 12)])
def test_split_planes_hands_on_one_gradient_buffer(monkeypatch, H, W):
    """ops.split_planes (the (mamba | conv) halves of the MSMM inputs, MambaSkip.py:727-733): the token transpose's and K19's backward
    write into ONE (B, C, H, W) buffer, which is the split's gradient -- equal, bit for bit, to the concatenation autograd would build;
    a piece whose consumer is a plain torch op is copied into place."""
    from mlagg_unet_amd import model, ops
    monkeypatch.setattr(ops, "K19_MIN_PIXELS", 0)
    g = torch.Generator().manual_seed(H)
    B, C, hd = 2, 96, 48
    x0 = torch.randn(B, C, H, W, generator=g).to(DEV)
    w = (torch.randn(C - hd, C - hd, 3, 3, generator=g) * 0.05).to(DEV)
    gt, gy = torch.randn(B, H * W, hd, generator=g).to(DEV), torch.randn(B, C - hd, H, W, generator=g).to(DEV)
    grads = []
    for mode in ("arena", "plain", "torch consumer"):
        xs, ws = x0.clone().requires_grad_(True), w.clone().requires_grad_(True)
        t = xs * 1.0
        if mode == "plain":
            a, b = t.split([hd, C - hd], dim=1)
            tok, y = model._MapToTokens.apply(a), ops.Conv3x3Fn.apply(b, ws)
        else:
            a, b = ops.split_planes(t, (hd, C - hd))
            assert a._mlagg_slot.dim == 1 and not a._mlagg_slot.claimed
            tok = model._map_to_tokens(a)
            y = ops.conv3x3(b, ws) if mode == "arena" else torch.nn.functional.conv2d(b, ws, None, 1, 1)
            assert a._mlagg_slot.claimed and b._mlagg_slot.claimed == (mode == "arena")
        ((tok * gt).sum() + (y * gy).sum()).backward()
        grads.append((xs.grad, ws.grad))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    assert torch.equal(grads[2][0][:, :hd], grads[1][0][:, :hd])
    assert float((grads[2][0][:, hd:] - grads[1][0][:, hd:]).abs().max()) < 1e-4 * float(grads[1][0].abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,O,H,W", [(2, 96, 48, 32, 32), (1, 192, 96, 16, 16), (2, 48, 40, 8, 16)])
def test_convolution_pair_sums_the_input_gradient_in_the_kernel(monkeypatch, B, I, O, H, W):
    """ops.conv_pair = (conv3x3(x, W3), conv1x1(x, W1)) of one map (UnetResBlock conv1 / conv3, M:581-667): outputs and weight gradients
    equal the separate Functions bit for bit; the input gradient -- K18 adding its product to K19's -- equals float64's sum."""
    from mlagg_unet_amd import ops
    for name, v in (("K19_MIN_PIXELS", 0), ("K18_FWD_MIN_PIXELS", 0), ("K18_FWD_MIN_K", 16), ("K18_WGRAD_MIN_PIXELS", 0), ("K19_WGRAD_MIN_PIXELS", 0)):
        monkeypatch.setattr(ops, name, v)
    g = torch.Generator().manual_seed(I + O)
    x = torch.randn(B, I, H, W, generator=g).to(DEV)
    w3 = (torch.randn(O, I, 3, 3, generator=g) * (9 * I) ** -0.5).to(DEV)
    w1 = (torch.randn(O, I, 1, 1, generator=g) * I ** -0.5).to(DEV)
    g3, g1 = torch.randn(B, O, H, W, generator=g).to(DEV), torch.randn(B, O, H, W, generator=g).to(DEV)
    res = []
    for paired in (True, False):
        xs, a3, a1 = x.clone().requires_grad_(True), w3.clone().requires_grad_(True), w1.clone().requires_grad_(True)
        c3, c1 = ops.conv_pair(xs, a3, a1) if paired else (ops.conv3x3(xs, a3), ops.conv1x1(xs, a1))
        torch.autograd.backward([c3, c1], [g3, g1])
        res.append((c3.detach(), c1.detach(), a3.grad, a1.grad, xs.grad))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    xr, r3, r1 = x.double().requires_grad_(True), w3.double(), w1.double()
    torch.autograd.backward([F.conv2d(xr, r3, None, 1, 1), F.conv2d(xr, r1)], [g3.double(), g1.double()])
    err = float((res[0][4].double() - xr.grad).abs().max() / xr.grad.abs().max())
    assert err < 2e-6, err


def _rounded_products(x, w, gy, pad, t):
    """The three products of a stride-1 convolution as the reference's autocast step computes them (nnUNetTrainer.py:848): operands
    rounded to the 16-bit type `t`, exact sums (float64 here; the kernels sum in fp32)."""
    r = lambda v: v.to(t).double()                                                          # noqa: E731
    xr, wr, gr = r(x), r(w), r(gy)
    y = F.conv2d(xr, wr, None, 1, pad)
    dx = torch.nn.grad.conv2d_input(xr.shape, wr, gr, 1, pad)
    dW = torch.nn.grad.conv2d_weight(xr, wr.shape, gr, 1, pad)
    return y, dx, dW


@pytest.mark.gpu
@pytest.mark.parametrize("form,t", [(1, torch.bfloat16), (2, torch.float16)])
@pytest.mark.parametrize("k,B,I,O,H,W,sliced", [(3, 2, 48, 48, 32, 32, False), (3, 2, 96, 48, 12, 32, True), (3, 1, 48, 1, 16, 64, False),
                                                (3, 2, 1, 48, 32, 32, False), (3, 1, 144, 33, 10, 16, False),
                                                (1, 2, 96, 192, 32, 32, False), (1, 3, 192, 96, 16, 24, True), (1, 2, 48, 14, 16, 16, False)])
def test_convolutions_in_the_16bit_operand_forms(monkeypatch, form, t, k, B, I, O, H, W, sliced):
    """K18 / K19 with ONE rounded product (csrc/opmode.h; the 16-bit modes of the step, BASELINE configs[2] / [4]): forward, data gradient
    and weight gradient equal the exact products of the operands rounded to bf16 / fp16, up to fp32 summation.  Cases include the
    one-channel stem (forward on the library in that type, weight gradient on K19), a one-channel output (data gradient of the stem
    shape), a 14-channel head (library data gradient) and a channel-slice input."""
    from mlagg_unet_amd import ops
    monkeypatch.setattr(ops, "LP_K_MIN_PIXELS", 0)
    g = torch.Generator().manual_seed(B * I + O + H + k)
    wide = torch.randn(B, I + 16, H, W, generator=g).to(DEV)
    x = (wide[:, 8:8 + I] if sliced else wide[:, :I].contiguous()).detach()
    w = (torch.randn(O, I, k, k, generator=g) * (k * k * I) ** -0.5).to(DEV)
    gy = torch.randn(B, O, H, W, generator=g).to(DEV)
    sup = ops.conv3x3_supported if k == 3 else ops.conv1x1_supported
    assert sup(x, w, (1, 1), (k // 2,) * 2, (1, 1), 1, form)
    yr, dxr, dWr = _rounded_products(x, w, gy, k // 2, t)
    xp, wp = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yp = (ops.conv3x3 if k == 3 else ops.conv1x1)(xp, wp, form)
    assert yp.dtype == torch.float32
    yp.backward(gy)
    # products that fall to the library in this form return ITS 16-bit result: a 16-bit rounding of the output (and of whatever the
    # library's solver keeps in that type on the way) on top
    lib_tol = 2.0 ** -7 if t == torch.bfloat16 else 2.0 ** -8
    on_k = {"y": (ops._k19_product(O, I, H, W, form) if k == 3 else ops._k18_product(O, I, H * W, form)),
            "dx": (ops._k19_product(I, O, H, W, form) if k == 3 else ops._k18_product(I, O, H * W, form)),
            "dW": (ops._k19_wgrad(O, I, H, W, form) if k == 3 else True)}
    for name, got, want in (("y", yp, yr), ("dx", xp.grad, dxr), ("dW", wp.grad, dWr)):
        err = float((got.detach().double() - want).abs().max() / want.abs().max())
        assert err < (4e-6 if on_k[name] else lib_tol), (name, err, on_k[name])
    assert on_k["dW"] and (on_k["y"] or I == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,O,dims", [(1, 32, 32, (6, 10, 16)), (2, 16, 48, (5, 7, 9)), (1, 64, 32, (3, 8, 12)), (1, 32, 40, (2, 4, 12)),
                                        (1, 32, 32, (24, 40, 40)), (2, 48, 33, (4, 6, 8)), (1, 16, 16, (2, 2, 24))])
def test_conv3x3x3_matches_float64(B, I, O, dims):
    """K19 with nine kernel rows: 3 x 3 x 3 convolution straight on the unpadded NCDHW volume -- output, data gradient (the same
    kernel on the transposed, tap-flipped weight) and weight gradient (K19's with one wave per kernel slice where W % 8 == 0 -- row
    halves of a block in different image rows at W = 8, 24, 40 -- else K15 on padded copies made in backward) against float64 conv3d.
    Odd widths (one-pixel runs), two-slice volumes, output widths off the 32-channel tiles, a stage of BASELINE configs[3] at 1/64."""
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(I + O + dims[0])
    x = torch.randn(B, I, *dims, generator=g).to(DEV)
    w = (torch.randn(O, I, 3, 3, 3, generator=g) * (27 * I) ** -0.5).to(DEV)
    gy = torch.randn(B, O, *dims, generator=g).to(DEV)
    assert ops.conv3x3x3_supported(x, w, (1, 1, 1), (1, 1, 1))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, 1, 1)
    yr.backward(gy.double())
    xp, wp = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yp = ops.conv_nd(xp, wp, (1, 1, 1), (1, 1, 1))
    yp.backward(gy)
    for name, got, want, tol in (("y", yp, yr, 2e-6), ("dx", xp.grad, xr.grad, 2e-6), ("dW", wp.grad, wr.grad, 1e-5)):
        err = float((got.detach().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert err < tol, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,I,O,H", [("3x3", 48, 48, 256), ("3x3", 96, 48, 256), ("3x3", 96, 96, 128), ("1x1", 96, 192, 128), ("1x1", 96, 48, 256)])
def test_split_bf16_convolutions_at_the_headline_sizes(kind, I, O, H):
    """K18 / K19 at the full batch-10 shapes of BASELINE configs[1], where float64 on the host is too slow to be the checker: against
    the library's fp32 convolution on the same inputs (two independent fp32 implementations: 2e-5 of the maximum) and through
    linearity in the input (conv(a x1 + x2) = a conv(x1) + conv(x2)), forward, data gradient and weight gradient."""
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(H + I + O)
    k = 3 if kind == "3x3" else 1
    x1 = torch.randn(10, I, H, H, generator=g).to(DEV)
    x2 = torch.randn(10, I, H, H, generator=g).to(DEV)
    w = (torch.randn(O, I, k, k, generator=g) * (k * k * I) ** -0.5).to(DEV)
    gy = torch.randn(10, O, H, H, generator=g).to(DEV)
    fn = ops.conv3x3 if kind == "3x3" else ops.conv1x1
    xs, ws = x1.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = fn(xs, ws)
    y.backward(gy)
    xr, wr = x1.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, k // 2)
    yr.backward(gy)
    for name, got, want in (("y", y, yr), ("dx", xs.grad, xr.grad), ("dW", ws.grad, wr.grad)):
        err = float((got.detach() - want.detach()).abs().max() / want.detach().abs().max())
        assert err < 2e-5, (name, err)
    with torch.no_grad():
        lin = fn(1.5 * x1 + x2, w) - (1.5 * y.detach() + fn(x2, w))
        assert float(lin.abs().max()) < 2e-5 * float(y.detach().abs().max())
