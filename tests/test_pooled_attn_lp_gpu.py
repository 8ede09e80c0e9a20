"""GPU parity: K4lp, the 16-bit MFMA form of the pooled differential attention (csrc/pooled_attn_lp.hip, through the C ABI) --
what replaces the four flash_attn_func calls, the lambda subtraction, RMSNorm and 0.2 gain of nnUNetTrainer_MLAgg_2D_dt_MS.py:733-760
in the bf16 / fp16 modes -- against the same arithmetic in float64 on the host (autograd for the gradients) and against the fp32
kernel K4.  Tolerances are those of 16-bit operands: 2^-11 (fp16) / 2^-8 (bf16) per rounded factor, fp32 sums."""
import pytest
import torch

gpu = pytest.mark.gpu
DEV = "cuda:0"


def _reference(q, k, v, lam, w, nh, scale):
    B, N, d = q.shape
    P = k.shape[1]
    q4 = q.view(B, N, nh, 2, 24) * scale
    k4 = k.view(B, P, nh, 2, 24)
    v4 = v.view(B, P, nh, 48)
    a = torch.softmax(torch.einsum("bnhre,bphre->bhrnp", q4, k4), dim=-1)
    att = a[:, :, 0] - lam * a[:, :, 1]
    o = torch.einsum("bhnp,bphe->bnhe", att, v4)
    o = o * torch.rsqrt(o.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    return (0.2 * o).reshape(B, N, d)


CASES = [
    # (B, N, P, nh, scale)
    (2, 300, 49, 2, 24 ** -0.5),        # ragged token tile and ragged key tile (BASELINE configs[2]: 49 pooled keys)
    (1, 1024, 320, 1, 1.0 / 24),        # configs[4]: 320 keys (10 key tiles), the shipped flash path's double scaling (variant A)
    (1, 64, 16, 4, 24 ** -0.5),         # half a key tile, four heads
    (3, 70, 64, 1, 24 ** -0.5),         # the headline's 64 keys, fewer tokens than a workgroup covers
]


@gpu
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,N,P,nh,scale", CASES)
def test_pooled_attn_lp_matches_float64(B, N, P, nh, scale, dtype):
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(N + P)
    d = 48 * nh
    wide = torch.randn(B, N, d + 8, generator=g)               # q arrives as a column block of a wider projection
    k, v = torch.randn(B, P, d, generator=g), torch.randn(B, P, d, generator=g)
    lam, w = torch.tensor(0.37), torch.rand(48, generator=g) + 0.5
    gy = torch.randn(B, N, d, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (wide[:, :, :d], k, v, lam, w)]
    ref = _reference(ref_in[0], ref_in[1], ref_in[2], ref_in[3], ref_in[4], nh, scale)
    ref.backward(gy.double())
    wd = wide.to(DEV).requires_grad_(True)
    dev_in = [k.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True), lam.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)]
    with ops.compute_precision(dtype):
        out = ops.pooled_diff_attn(wd[:, :, :d], dev_in[0], dev_in[1], dev_in[2], dev_in[3], nh, scale)
    out.backward(gy.to(DEV))
    tol = 4e-3 if dtype == "fp16" else 3e-2
    pairs = [("out", out.detach(), ref.detach()), ("dq", wd.grad[:, :, :d], ref_in[0].grad), ("dk", dev_in[0].grad, ref_in[1].grad),
             ("dv", dev_in[1].grad, ref_in[2].grad), ("dlam", dev_in[2].grad, ref_in[3].grad), ("dsubln", dev_in[3].grad, ref_in[4].grad)]
    for name, a, b in pairs:
        err = float((a.cpu().double() - b).abs().max())
        assert err <= tol * float(b.abs().max()) + 1e-6, (name, err, float(b.abs().max()))
    assert float(wd.grad[:, :, d:].abs().max()) == 0.0          # nothing leaks into the neighbouring columns


@gpu
def test_pooled_attn_lp_is_close_to_the_fp32_kernel_and_used_only_in_16_bit_modes():
    from mlagg_unet_amd import ops
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(2, 256, 96, generator=g).to(DEV), torch.randn(2, 64, 96, generator=g).to(DEV),
               torch.randn(2, 64, 96, generator=g).to(DEV))
    lam, w = torch.tensor(0.4, device=DEV), (torch.rand(48, generator=g) + 0.5).to(DEV)
    o32 = ops.pooled_diff_attn(q, k, v, lam, w, 2, 24 ** -0.5)
    with ops.compute_precision("fp16"):
        o16 = ops.pooled_diff_attn(q, k, v, lam, w, 2, 24 ** -0.5)
    assert 0.0 < float((o32 - o16).abs().max()) < 4e-3 * float(o32.abs().max())
    # invariant to the order of the pooled keys (different key tiles, same sums)
    perm = torch.randperm(64, generator=g).to(DEV)
    with ops.compute_precision("fp16"):
        o16p = ops.pooled_diff_attn(q, k[:, perm].contiguous(), v[:, perm].contiguous(), lam, w, 2, 24 ** -0.5)
    assert float((o16p - o16).abs().max()) < 2e-3 * float(o16.abs().max())
