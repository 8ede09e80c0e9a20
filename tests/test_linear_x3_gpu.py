"""GPU parity of K5's round-4 form (csrc/linear_x3.hip): projections on weight images, forward / data gradient / GELU epilogues
against float64, ragged shapes, every tile variant; the fused Mlp (reference T:176-192); the per-network image set."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _images(lib, w):
    N, K = w.shape
    st = torch.cuda.current_stream().cuda_stream
    img = torch.empty(lib.mlagg_weight_image_bytes(N, K), dtype=torch.uint8, device=DEV)
    imgT = torch.empty(lib.mlagg_weight_image_bytes(K, N), dtype=torch.uint8, device=DEV)
    from mlagg_unet_amd import _lib
    _lib.check(lib.mlagg_weight_image(w.data_ptr(), w.stride(0), img.data_ptr(), imgT.data_ptr(), N, K, st), "image")
    return img, imgT


def test_weight_image_is_the_exact_three_piece_split():
    """img[piece][n][k]: hi + mid + lo == w exactly (bf16 pieces read back as fp32), pad columns zero; imgT is the image of W^T."""
    from mlagg_unet_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(70, 48, generator=g) * torch.exp(2 * torch.randn(70, 1, generator=g))).to(DEV)
    img, imgT = _images(lib, w)

    def decode(buf, rows, cols):
        cp = (cols + 31) // 32 * 32
        u = buf.view(torch.int16).view(3, rows, cp).to(torch.int32) << 16
        return u.view(torch.float32)
    p = decode(img, 70, 48)
    assert torch.equal(p.sum(0)[:, :48], w) and float(p[:, :, 48:].abs().max()) == 0.0
    pt = decode(imgT, 48, 70)
    assert torch.equal(pt.sum(0)[:, :70], w.t()) and float(pt[:, :, 70:].abs().max()) == 0.0


SHAPES = [(16384, 192, 96), (4100, 140, 96), (2560, 768, 1536), (640, 384, 768), (1030, 100, 48), (9000, 96, 144), (513, 64, 40)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_x3_matches_float64(M, N, K):
    """Forward (bias) and the data gradient on the W^T image against float64: strided rows, ragged M / N, K % 32 != 0; the shapes
    reach the 128 x 96, 128 x 64 and 64 x 64 tile variants of the default dispatch."""
    from mlagg_unet_amd import _lib
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(M + N + K)
    xw = (torch.randn(M, K + 4, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(DEV)       # rows of very different scale
    x = xw[:, :K]
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    img, imgT = _images(lib, w)
    y = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(lib.mlagg_linear_x3(x.data_ptr(), xw.stride(0), img.data_ptr(), b.data_ptr(), y.data_ptr(), N, None, None, 0, M, N, K, 0, st), "fwd")
    ref = torch.addmm(b.double(), x.double(), w.double().t())
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    assert float(((y.double() - ref).abs() / scale).max()) < 3e-6
    if N % 8 == 0:
        dy = torch.randn(M, N, generator=g).to(DEV)
        dx = torch.full((M, K), float("nan"), device=DEV)
        _lib.check(lib.mlagg_linear_x3(dy.data_ptr(), N, imgT.data_ptr(), None, dx.data_ptr(), K, None, None, 0, M, K, N, 0, st), "dgrad")
        dref = dy.double() @ w.double()
        assert float((dx.double() - dref).abs().max() / dref.abs().max()) < 3e-6


@pytest.mark.parametrize("M,C", [(10240, 384), (2560, 768), (700, 96)])
def test_mlp_fused_matches_float64(M, C, monkeypatch):
    """ops.mlp == fc2(GELU(fc1(x))) (exact erf GELU), outputs and all five gradients against float64 -- also at token counts the
    model leaves to the tuned library GEMM (X3_MIN_ROWS lowered for the test)."""
    from mlagg_unet_amd import ops
    monkeypatch.setattr(ops, "X3_MIN_ROWS", 512)
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g)
    w1, b1 = torch.randn(2 * C, C, generator=g) * C ** -0.5, torch.randn(2 * C, generator=g) * 0.1
    w2, b2 = torch.randn(C, 2 * C, generator=g) * (2 * C) ** -0.5, torch.randn(C, generator=g) * 0.1
    gy = torch.randn(M, C, generator=g)
    la = [t.to(DEV).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    assert ops.mlp_supported(la[0], la[1], la[3])
    ya = ops.mlp(*la)
    ya.backward(gy.to(DEV))
    lb = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    yb = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(lb[0], lb[1], lb[2])), lb[3], lb[4])
    yb.backward(gy.double())
    assert float((ya.detach().cpu().double() - yb.detach()).abs().max()) < 1e-5 * float(yb.abs().max())
    for name, a_, b_ in zip(("dx", "dW1", "db1", "dW2", "db2"), la, lb):
        assert float((a_.grad.cpu().double() - b_.grad).abs().max()) < 2e-5 * float(b_.grad.abs().max()), name


def test_image_set_serves_current_images_only(monkeypatch):
    """Registered images are used only while the matrix is unchanged: an in-place change (version counter) or ClipAdamW's raw-pointer
    step (epoch) after the build falls back to an image built on the fly -- the result always equals the product with the CURRENT
    weight."""
    from mlagg_unet_amd import ops, trainer
    monkeypatch.setattr(ops, "X3_MIN_ROWS", 512)
    g = torch.Generator().manual_seed(4)
    lin = torch.nn.Linear(96, 192).to(DEV)
    x = torch.randn(2048, 96, generator=g).to(DEV)
    images = ops.WeightImageSet()

    def run():
        with images:
            return ops.linear(x, lin.weight, lin.bias)

    def want():
        return torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())

    y0 = run()                                                         # records the parameter, image built on the fly
    assert len(images.tensors) == 1
    y1 = run()                                                         # served by the one-launch build
    assert torch.equal(y0, y1) and float((y1.double() - want()).abs().max()) < 1e-4
    with images:
        with torch.no_grad():
            lin.weight.mul_(2.0)                                       # in-place change AFTER the build of this forward
        y2 = ops.linear(x, lin.weight, lin.bias)
    assert float((y2.double() - want()).abs().max()) < 1e-4
    opt = trainer.ClipAdamW(lin.parameters(), 0.1)
    with images:
        lin.weight.grad, lin.bias.grad = torch.ones_like(lin.weight), torch.ones_like(lin.bias)
        opt.step()                                                     # rewrites the weight without touching its version counter
        y3 = ops.linear(x, lin.weight, lin.bias)
    torch.cuda.synchronize()
    assert float((y3.double() - want()).abs().max()) < 1e-4
    assert float((y3 - y2).abs().max()) > 1e-2
