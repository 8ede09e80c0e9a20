"""GPU parity: HIP selective scan (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO

gpu = pytest.mark.gpu


def _case(b, G, Hc, L, seed=0, with_D=True, with_bias=True):
    g = torch.Generator().manual_seed(seed)
    d, N = G * Hc, 16
    u = torch.randn(b, d, L, generator=g)
    delta = torch.randn(b, d, L, generator=g) * 0.5
    A = -torch.exp(torch.randn(d, N, generator=g) * 0.3 + 1.0)
    B = torch.randn(b, G, N, L, generator=g)
    C = torch.randn(b, G, N, L, generator=g)
    D = torch.randn(d, generator=g) if with_D else None
    bias = (torch.randn(d, generator=g) - 3.0) if with_bias else None
    dout = torch.randn(b, d, L, generator=g)
    return u, delta, A, B, C, D, bias, dout


def _np(t):
    return None if t is None else t.numpy()


CASES = [
    # (b, G, Hc, L)            what it covers
    (2, 4, 96, 5440),          # BASELINE config 1 shape: 128x128 -> L_cat 5440, D = 384
    (1, 4, 96, 200),           # L not a multiple of the 64-step chunk (ragged last chunk, L % 4 == 0)
    (2, 2, 20, 130),           # partial wave (Hc*4 = 80 lanes), L % 4 != 0 (scalar loads)
    (1, 1, 160, 77),           # group wider than one workgroup (2 channel blocks -> atomic dB/dC)
    (1, 3, 8, 1),              # single step
]


@gpu
@pytest.mark.parametrize("b,G,Hc,L", CASES)
def test_selscan_fwd_bwd_matches_oracle(b, G, Hc, L):
    from mlagg_unet_amd.ops import selective_scan_fn
    u, delta, A, B, C, D, bias, dout = _case(b, G, Hc, L)
    dev = torch.device("cuda:0")
    args = [t.to(dev).requires_grad_(True) if t is not None else None for t in (u, delta, A, B, C, D, bias)]
    y = selective_scan_fn(args[0], args[1], args[2], args[3], args[4], args[5], None, args[6], True)
    y_ref = CO.selscan_fwd(_np(u), _np(delta), _np(A), _np(B), _np(C), _np(D), _np(bias), True)
    scale = np.abs(y_ref).max()
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, atol=1e-4 * scale, rtol=1e-4)
    y.backward(dout.to(dev))
    ref = CO.selscan_bwd(_np(u), _np(delta), _np(A), _np(B), _np(C), _np(D), _np(bias), _np(dout), True)
    for name, t, r in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dbias"), args, ref):
        got = t.grad.cpu().numpy()
        s = max(np.abs(r).max(), 1e-6)
        np.testing.assert_allclose(got, r, atol=2e-4 * s, rtol=1e-3, err_msg=name)


@gpu
def test_selscan_no_D_no_bias_no_softplus():
    from mlagg_unet_amd.ops import selective_scan_fn
    u, delta, A, B, C, _, _, dout = _case(1, 2, 32, 96, seed=3, with_D=False, with_bias=False)
    delta = delta.abs() * 0.2
    dev = torch.device("cuda:0")
    args = [t.to(dev).requires_grad_(True) for t in (u, delta, A, B, C)]
    y = selective_scan_fn(*args, None, None, None, False)
    y_ref = CO.selscan_fwd(_np(u), _np(delta), _np(A), _np(B), _np(C), None, None, False)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, atol=1e-4 * np.abs(y_ref).max(), rtol=1e-4)
    y.backward(dout.to(dev))
    ref = CO.selscan_bwd(_np(u), _np(delta), _np(A), _np(B), _np(C), None, None, _np(dout), False)
    for name, t, r in zip(("du", "ddelta", "dA", "dB", "dC"), args, ref[:5]):
        s = max(np.abs(r).max(), 1e-6)
        np.testing.assert_allclose(t.grad.cpu().numpy(), r, atol=2e-4 * s, rtol=1e-3, err_msg=name)


@gpu
def test_selscan_rejects_bad_arguments():
    from mlagg_unet_amd.ops import selective_scan_fn
    dev = torch.device("cuda:0")
    u = torch.zeros(1, 8, 16, device=dev)
    A = torch.zeros(8, 8, device=dev)               # N = 8 is not built
    Bm = torch.zeros(1, 1, 8, 16, device=dev)
    with pytest.raises(RuntimeError):
        selective_scan_fn(u, u, A, Bm, Bm, None, None, None, True)
    with pytest.raises(RuntimeError):               # CPU tensors: no fallback
        selective_scan_fn(u.cpu(), u.cpu(), torch.zeros(8, 16), torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16))


LR_CASES = [
    # (b, G, Hc, L, R)
    (2, 4, 96, 5440, 3),       # BASELINE config 1 shape, the model's rank (dt_rank = ceil(48 / 16))
    (1, 4, 96, 200, 3),        # ragged last chunk
    (2, 2, 20, 130, 2),        # partial wave, L % 4 != 0, rank 2
    (1, 1, 160, 77, 4),        # group wider than one workgroup (atomic dB / dC / d(dtr)), rank 4
    (1, 3, 8, 1, 1),           # single step, rank 1
]


@gpu
@pytest.mark.parametrize("b,G,Hc,L,R", LR_CASES)
def test_selscan_lowrank_matches_oracle(b, G, Hc, L, R):
    """K1 with the dt projection folded in == einsum("b k r l, k d r -> b k d l") + selective scan of the oracle
    (reference MambaSkip.py:430-451), forward and every gradient (d(dtr) and dWdt by the chain rule from the oracle's
    d(delta))."""
    from mlagg_unet_amd.ops import selective_scan_lowrank_fn
    u, _, A, B, C, D, bias, dout = _case(b, G, Hc, L, seed=5)
    g = torch.Generator().manual_seed(50 + L)
    d = G * Hc
    dtr = torch.randn(b, G, R, L, generator=g)
    Wdt = torch.randn(d, R, generator=g) * 0.4
    delta = torch.einsum("bgrl,gdr->bgdl", dtr, Wdt.view(G, Hc, R)).reshape(b, d, L).contiguous()
    dev = torch.device("cuda:0")
    args = [t.to(dev).requires_grad_(True) if t is not None else None for t in (u, dtr, Wdt, A, B, C, D, bias)]
    y = selective_scan_lowrank_fn(*args[:6], args[6], args[7], True)
    y_ref = CO.selscan_fwd(_np(u), _np(delta), _np(A), _np(B), _np(C), _np(D), _np(bias), True)
    scale = np.abs(y_ref).max()
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, atol=1e-4 * scale, rtol=1e-4)
    y.backward(dout.to(dev))
    du, ddelta, dA, dB, dC, dD, dbias = CO.selscan_bwd(_np(u), _np(delta), _np(A), _np(B), _np(C), _np(D), _np(bias),
                                                       _np(dout), True)
    dd = np.asarray(ddelta, dtype=np.float64).reshape(b, G, Hc, L)
    ddtr = np.einsum("bgdl,gdr->bgrl", dd, Wdt.double().numpy().reshape(G, Hc, R))
    dW = np.einsum("bgdl,bgrl->gdr", dd, dtr.double().numpy()).reshape(d, R)
    for name, t, r in zip(("du", "ddtr", "dWdt", "dA", "dB", "dC", "dD", "dbias"), args,
                          (du, ddtr, dW, dA, dB, dC, dD, dbias)):
        got = t.grad.cpu().numpy()
        s = max(np.abs(r).max(), 1e-6)
        np.testing.assert_allclose(got, r, atol=3e-4 * s, rtol=1e-3, err_msg=name)


@gpu
def test_selscan_lowrank_rejects_rank_above_4():
    from mlagg_unet_amd.ops import selective_scan_lowrank_fn
    dev = torch.device("cuda:0")
    u = torch.zeros(1, 8, 16, device=dev)
    with pytest.raises(RuntimeError):
        selective_scan_lowrank_fn(u, torch.zeros(1, 1, 5, 16, device=dev), torch.zeros(8, 5, device=dev),
                                  -torch.ones(8, 16, device=dev), torch.zeros(1, 1, 16, 16, device=dev),
                                  torch.zeros(1, 1, 16, 16, device=dev))


@gpu
def test_selscan_full_size_properties():
    """BASELINE config-2 scan shape (D = 384, G = 4, R = 3, L_cat = 21760; batch 2 of 10), where the oracle is too slow to
    be the checker: properties that hold at any size.  (i) the scan is LINEAR in u for fixed delta / A / B / C / D;
    (ii) it is CAUSAL: y[..., :t] does not depend on u[..., t:], bit for bit; (iii) every gradient is linear in dout."""
    from mlagg_unet_amd.ops import selective_scan_lowrank_fn
    b, G, Hc, L, R, N = 2, 4, 96, 21760, 3, 16
    d = G * Hc
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)                      # noqa: E731
    u1, u2 = rn(b, d, L), rn(b, d, L)
    dtr, Wdt = rn(b, G, R, L), rn(d, R) * 0.4
    A = -torch.exp(rn(d, N) * 0.3 + 1.0)
    Bm, Cm, Dv, bias = rn(b, G, N, L), rn(b, G, N, L), rn(d), rn(d) - 3.0

    def run(u, dout=None):
        leaves = [t.clone().requires_grad_(True) for t in (u, dtr, Wdt, A, Bm, Cm, Dv, bias)]
        y = selective_scan_lowrank_fn(*leaves[:6], leaves[6], leaves[7], True)
        if dout is None:
            return y.detach()
        y.backward(dout)
        return y.detach(), [t.grad for t in leaves]

    y1, y2 = run(u1), run(u2)
    y12 = run(0.7 * u1 - 1.3 * u2)
    scale = float(y1.abs().max())
    assert float((y12 - (0.7 * y1 - 1.3 * y2)).abs().max()) < 2e-5 * scale * 10          # (i): fp32 sums of ~1e2 terms
    t0 = 12345                                                                            # inside a chunk, inside a tile
    u3 = u1.clone()
    u3[..., t0:] = rn(b, d, L - t0)
    y3 = run(u3)
    assert torch.equal(y3[..., :t0], y1[..., :t0])                                        # (ii)
    assert float((y3[..., t0:] - y1[..., t0:]).abs().max()) > 1e-3 * scale
    d1, d2 = rn(b, d, L), rn(b, d, L)
    _, ga = run(u1, d1)
    _, gb = run(u1, d2)
    _, gab = run(u1, d1 + 0.5 * d2)
    for name, a_, b_, ab in zip(("du", "ddtr", "dWdt", "dA", "dB", "dC", "dD", "dbias"), ga, gb, gab):
        want = a_ + 0.5 * b_
        s = float(want.abs().max())
        assert float((ab - want).abs().max()) < 2e-4 * s, name                            # (iii)
