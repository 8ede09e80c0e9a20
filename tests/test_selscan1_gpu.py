"""GPU parity: K1s, the d_state = 1 selective scan on token-major volumes with in-kernel scan orders (csrc/selscan1.hip, through the
C ABI) vs the double-precision C oracle of the scan (oracle/selscan_ref.c) fed with explicitly re-ordered sequences, i.e. what
SS3D.forward_corev0 builds with stack / permute / flip / cat (UMambaEnc_SS3D.py:251-296)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO

gpu = pytest.mark.gpu


def _case(B, L, C, K, R, seed=0):
    g = torch.Generator().manual_seed(seed)
    tok = torch.randn(B, L, C, generator=g)
    half = (K + 1) // 2
    perms = [torch.randperm(L, generator=g) for _ in range(half)]
    idx = torch.stack((perms + [p.flip(0) for p in perms])[:K]).to(torch.int32)          # reversed twins, like directions 6..11
    dtr = torch.randn(B, K, R, L, generator=g)
    Bs = torch.randn(B, K, L, generator=g)
    Cs = torch.randn(B, K, L, generator=g)
    Wdt = torch.randn(K * C, R, generator=g) * R ** -0.5
    A = -torch.exp(torch.randn(K * C, generator=g) * 0.3)
    D = torch.randn(K * C, generator=g)
    bias = torch.randn(K * C, generator=g) * 1.5 - 3.0
    dout = torch.randn(B, L, C, generator=g)
    return tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, dout


def _oracle(tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, dout):
    """Explicit scan-order tensors -> C oracle -> gradients folded back to the kernel's operands (float64 host algebra)."""
    B, L, C = tok.shape
    K, R = idx.shape[0], dtr.shape[2]
    ix = idx.long()
    xs = torch.stack([tok[:, ix[k], :].transpose(1, 2) for k in range(K)], 1).reshape(B, K * C, L)       # (B, K*C, L)
    delta = torch.einsum("bkrl,kcr->bkcl", dtr.double(), Wdt.view(K, C, R).double()).reshape(B, K * C, L).float()
    A2, B4, C4 = A.view(-1, 1), Bs.view(B, K, 1, L), Cs.view(B, K, 1, L)
    n = lambda t: t.contiguous().numpy()
    out = torch.from_numpy(CO.selscan_fwd(n(xs), n(delta), n(A2), n(B4), n(C4), n(D), n(bias), True)).view(B, K, C, L)
    y = torch.zeros(B, L, C, dtype=torch.float64)
    for k in range(K):
        y[:, ix[k], :] += out[:, k].transpose(1, 2).double()
    douts = torch.stack([dout[:, ix[k], :].transpose(1, 2) for k in range(K)], 1).reshape(B, K * C, L)
    du, ddelta, dA, dB, dC, dD, dbias = (torch.from_numpy(v) for v in
                                         CO.selscan_bwd(n(xs), n(delta), n(A2), n(B4), n(C4), n(D), n(bias), n(douts), True))
    dtok = torch.zeros(B, L, C, dtype=torch.float64)
    du = du.view(B, K, C, L)
    for k in range(K):
        dtok[:, ix[k], :] += du[:, k].transpose(1, 2).double()
    dd = ddelta.view(B, K, C, L).double()
    ddtr = torch.einsum("bkcl,kcr->bkrl", dd, Wdt.view(K, C, R).double())
    dW = torch.einsum("bkcl,bkrl->kcr", dd, dtr.double()).reshape(K * C, R)
    return y, dict(tok=dtok, dtr=ddtr, Bs=dB.view(B, K, L), Cs=dC.view(B, K, L), Wdt=dW, A=dA.view(-1), D=dD, bias=dbias)


CASES = [
    # (B, L, C, K, R)            what it covers
    (2, 300, 64, 3, 2),          # ragged everywhere: L % 64 and L % 16 != 0, odd direction count, several 64-step chunks
    (1, 4133, 128, 2, 4),        # two channel blocks (per-block partials of dB / dC / ddtr + their reduction), R = 4: two LDS rounds
    (2, 150, 64, 12, 20),        # the deepest stage of configs[3]: 6x5x5 tokens, rank 20, 12 directions
    (1, 64, 192, 1, 1),          # one full tile, three channel blocks, rank 1
    (1, 1, 64, 2, 3),            # a single step
    (2, 174100, 64, 12, 2),      # 2048-step chunks (prefix over 86 chunks) at the widest stage's channel / rank shape
]


@gpu
@pytest.mark.parametrize("B,L,C,K,R", CASES)
def test_selscan1_fwd_bwd_matches_oracle(B, L, C, K, R):
    from mlagg_unet_amd import _lib
    from mlagg_unet_amd.ops import selective_scan1
    case = _case(B, L, C, K, R, seed=L)
    tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, dout = case
    if L == 174100:
        assert _lib.lib().mlagg_selscan1_chunk(B, L, K) == 2048
    dev = torch.device("cuda:0")
    names = ("tok", "dtr", "Bs", "Cs", "Wdt", "A", "D", "bias")
    dv = {nm: t.to(dev).requires_grad_(True) for nm, t in zip(names, (tok, dtr, Bs, Cs, Wdt, A, D, bias))}
    y = selective_scan1(dv["tok"], idx.to(dev), dv["dtr"], dv["Bs"], dv["Cs"], dv["Wdt"], dv["A"], dv["D"], dv["bias"])
    y_ref, grads = _oracle(*case)
    scale = float(y_ref.abs().max())
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.numpy(), atol=1e-4 * scale, rtol=1e-4)
    y.backward(dout.to(dev))
    for nm in names:
        r = grads[nm].double().numpy()
        s = max(float(np.abs(r).max()), 1e-6)
        np.testing.assert_allclose(dv[nm].grad.cpu().numpy(), r, atol=2e-4 * s, rtol=1e-3, err_msg=nm)


@gpu
def test_selscan1_rejects_what_it_is_not_built_for():
    from mlagg_unet_amd.ops import selective_scan1
    dev = torch.device("cuda:0")
    tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, _ = (t.to(dev) for t in _case(1, 32, 64, 2, 2))
    with pytest.raises(RuntimeError):                                      # rank 5 has no instantiation
        selective_scan1(tok, idx, torch.zeros(1, 2, 5, 32, device=dev), Bs, Cs, torch.zeros(128, 5, device=dev), A, D, bias)
    with pytest.raises(RuntimeError):                                      # 48 channels: not a whole wave of channels
        selective_scan1(tok[:, :, :48].contiguous(), idx, dtr, Bs, Cs, Wdt[:96], A[:96], D[:96], bias[:96])
    with pytest.raises(RuntimeError):
        selective_scan1(tok, idx.long(), dtr, Bs, Cs, Wdt, A, D, bias)
