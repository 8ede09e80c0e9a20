"""GPU: boundary #3 -- the `flash_attn.flash_attn_func` drop-in (mlagg_unet_amd.shims) on the HIP kernels of
csrc/flash_attn.hip, against exact softmax attention in double precision on the SAME 16-bit inputs, at the shapes the
reference calls it with (T:745-750: head_dim 24, P = (H/sr)(W/sr) pooled keys, nh heads per call).

Tolerance: inputs are exact on both sides; the kernel rounds its OUTPUT to the 16-bit type (relative 2^-9 bf16, 2^-11
fp16) and gradients are products of 16-bit-rounded tensors: 3 output ulps."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, k, v, scale):
    qd, kd, vd = q.double(), k.double(), v.double()
    att = torch.einsum("bnhe,bphe->bhnp", qd, kd) * scale
    return torch.einsum("bhnp,bphe->bnhe", att.softmax(-1), vd)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,N,P,nh", [(2, 4096, 64, 1), (1, 1000, 49, 4), (2, 300, 320, 2), (1, 70, 16, 8)])
def test_flash_attn_func_matches_exact_attention(dtype, B, N, P, nh):
    import mlagg_unet_amd.shims as shims
    shims.install()
    from flash_attn import flash_attn_func
    g = torch.Generator().manual_seed(N + P)
    e = 24
    q = (torch.randn(B, N, nh, e, generator=g)).to(dtype)
    k = (torch.randn(B, P, nh, e, generator=g)).to(dtype)
    v = (torch.randn(B, P, nh, e, generator=g)).to(dtype)
    go = torch.randn(B, N, nh, e, generator=g).to(dtype)
    qg, kg, vg = [t.to(DEV).requires_grad_(True) for t in (q, k, v)]
    out = flash_attn_func(qg, kg, vg, causal=False)             # default softmax_scale = 24^-0.5, as the reference relies on
    assert out.dtype == dtype and out.shape == q.shape
    out.backward(go.to(DEV))
    qr, kr, vr = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = _ref(qr, kr, vr, e ** -0.5)
    ref.backward(go.double())
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    for name, got, want in (("out", out, ref), ("dq", qg.grad, qr.grad), ("dk", kg.grad, kr.grad), ("dv", vg.grad, vr.grad)):
        err = float((got.detach().cpu().double() - want.detach()).abs().max())
        assert err <= 3 * ulp * float(want.detach().abs().max()) + 1e-6, (name, err, float(want.abs().max()))


def test_flash_attn_func_contract():
    """flash-attn only takes fp16 / bf16 CUDA tensors (SURVEY finding 5); the shim keeps that contract and refuses the
    arguments the reference never passes."""
    import mlagg_unet_amd.shims as shims
    q = torch.zeros(1, 8, 1, 24, device=DEV)
    with pytest.raises(RuntimeError):
        shims.flash_attn_func(q, q, q)                          # fp32
    h = q.half()
    with pytest.raises(RuntimeError):
        shims.flash_attn_func(h, h, h, causal=True)
    with pytest.raises(RuntimeError):
        shims.flash_attn_func(h, h, h, dropout_p=0.1)
    with pytest.raises(RuntimeError):
        shims.flash_attn_func(h.cpu(), h.cpu(), h.cpu())
    with pytest.raises(RuntimeError):
        shims.flash_attn_func(torch.zeros(1, 8, 1, 32, device=DEV).half(), torch.zeros(1, 8, 1, 32, device=DEV).half(),
                              torch.zeros(1, 8, 1, 32, device=DEV).half())       # head_dim 32: not an MLAgg-UNet shape
