"""GPU parity of K1f (csrc/selscan_tok.hip, `ops.msmm_scan`): SS2D_skip.forward_corev0 behind x_proj + the four-way sum (reference
MambaSkip.py:405-473, 534) on token-major tensors, against (i) the double-precision C oracle of the selective scan fed with the
scan sequences the reference builds (M:414-422) and (ii) the round-3 chain of this package (K1' cross_scan / cross_merge around the
(B, D, L) scan), forward and every gradient."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K, HC, N, R, XB = 4, 96, 16, 3, 36


def _inputs(b, HW, seed):
    g = torch.Generator().manual_seed(seed)
    L = sum(h * w for h, w in HW)
    xc = torch.randn(b, L, HC, generator=g)
    xdbl = torch.randn(b, L, K * XB, generator=g)
    xdbl.view(b, L, K, XB)[..., 3] = 0.0                         # the pad column of every direction is a product with a zero row
    Wdt = torch.randn(K * HC, R, generator=g) * 0.4
    A = -torch.exp(torch.randn(K * HC, N, generator=g) * 0.3 + 1.0)
    D = torch.randn(K * HC, generator=g)
    bias = torch.randn(K * HC, generator=g) - 3.0
    dy = torch.randn(b, L, HC, generator=g)
    return xc, xdbl, Wdt, A, D, bias, dy, L


def _reference_orders(HW):
    """The four scan orders as the reference builds them (M:414-422): per scale x (C, H, W) -> stack([x.flatten(), x.transpose(H, W)
    .flatten()]) and their flips, scales concatenated per direction.  Returned as token indices (4, L)."""
    rows, off = [[], [], [], []], 0
    for H, W in HW:
        tok = torch.arange(H * W).view(H, W)
        hw, wh = tok.flatten(), tok.t().contiguous().flatten()
        for k, t in enumerate((hw, wh, hw.flip(0), wh.flip(0))):
            rows[k].append(t + off)
        off += H * W
    return torch.stack([torch.cat(r) for r in rows])


CASES = [
    # (batch, maps)                                     what it covers
    (2, [(64, 64), (32, 32), (16, 16), (8, 8)]),        # BASELINE config 1 (128 x 128 input): L_cat 5440 = 85 whole chunks
    (1, [(16, 16), (8, 8), (4, 4), (2, 2)]),            # L = 340: ragged last chunk, scales that end inside a chunk
    (2, [(12, 20), (6, 10), (4, 4)]),                   # non-square maps, three scales, L = 316
    (1, [(2, 2)]),                                      # a single 4-step sequence
]


@pytest.mark.parametrize("b,HW", CASES)
def test_index_table_is_the_references_scan_order(b, HW):
    from mlagg_unet_amd import ops
    assert torch.equal(ops.msmm_scan_index(HW, "cpu").long(), _reference_orders(HW))


@pytest.mark.parametrize("b,HW", CASES)
def test_msmm_scan_matches_oracle(b, HW):
    from mlagg_unet_amd import ops
    xc, xdbl, Wdt, A, D, bias, dy, L = _inputs(b, HW, seed=11 + len(HW))
    idx = _reference_orders(HW)                                   # (4, L) int64
    leaves = [t.to(DEV).requires_grad_(True) for t in (xc, xdbl, Wdt, A, D, bias)]
    y = ops.msmm_scan(leaves[0], leaves[1], ops.msmm_scan_index(HW, DEV), *leaves[2:])
    # the oracle's inputs: scan sequences gathered the way forward_corev0 lays them out
    xv = xdbl.view(b, L, K, XB)
    xs = torch.stack([xc[:, idx[k]] for k in range(K)], 1).permute(0, 1, 3, 2).reshape(b, K * HC, L).contiguous()
    dtr = torch.stack([xv[:, idx[k], k, :R] for k in range(K)], 1).permute(0, 1, 3, 2).contiguous()            # (b, K, R, L)
    Bs = torch.stack([xv[:, idx[k], k, 4:4 + N] for k in range(K)], 1).permute(0, 1, 3, 2).contiguous()        # (b, K, N, L)
    Cs = torch.stack([xv[:, idx[k], k, 4 + N:] for k in range(K)], 1).permute(0, 1, 3, 2).contiguous()
    delta = torch.einsum("bkrl,kdr->bkdl", dtr, Wdt.view(K, HC, R)).reshape(b, K * HC, L).contiguous()
    npf = lambda t: t.numpy()                                                                                   # noqa: E731
    y_seq = CO.selscan_fwd(npf(xs), npf(delta), npf(A), npf(Bs), npf(Cs), npf(D), npf(bias), True)              # (b, 384, L)
    y_ref = np.zeros((b, L, HC))
    for k in range(K):
        y_ref[:, idx[k].numpy()] += np.asarray(y_seq, dtype=np.float64).reshape(b, K, HC, L)[:, k].transpose(0, 2, 1)
    scale = np.abs(y_ref).max()
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, atol=1e-4 * scale, rtol=1e-4)
    y.backward(dy.to(DEV))
    dout = torch.stack([dy[:, idx[k]] for k in range(K)], 1).permute(0, 1, 3, 2).reshape(b, K * HC, L).contiguous()
    du, ddelta, dA, dB, dC, dD, dbias = CO.selscan_bwd(npf(xs), npf(delta), npf(A), npf(Bs), npf(Cs), npf(D), npf(bias), npf(dout), True)
    dd = np.asarray(ddelta, dtype=np.float64).reshape(b, K, HC, L)
    ddtr = np.einsum("bkdl,kdr->bkrl", dd, Wdt.double().numpy().reshape(K, HC, R))
    dW = np.einsum("bkdl,bkrl->kdr", dd, dtr.double().numpy()).reshape(K * HC, R)
    dxc = np.zeros((b, L, HC))
    dxd = np.zeros((b, L, K, XB))
    du4 = np.asarray(du, dtype=np.float64).reshape(b, K, HC, L)
    for k in range(K):
        ik = idx[k].numpy()
        dxc[:, ik] += du4[:, k].transpose(0, 2, 1)
        dxd[:, ik, k, :R] = ddtr[:, k].transpose(0, 2, 1)
        dxd[:, ik, k, 4:4 + N] = np.asarray(dB, dtype=np.float64)[:, k].transpose(0, 2, 1)
        dxd[:, ik, k, 4 + N:] = np.asarray(dC, dtype=np.float64)[:, k].transpose(0, 2, 1)
    for name, t, r in zip(("dxc", "dxdbl", "dWdt", "dA", "dD", "dbias"), leaves, (dxc, dxd.reshape(b, L, K * XB), dW, dA, dD, dbias)):
        got = t.grad.cpu().numpy()
        s = max(np.abs(r).max(), 1e-6)
        np.testing.assert_allclose(got, r, atol=3e-4 * s, rtol=1e-3, err_msg=name)
    assert float(leaves[1].grad.view(b, L, K, XB)[..., 3].abs().max()) == 0.0        # pad columns: written, with zeros


@pytest.mark.parametrize("b,HW", CASES[:3])
def test_msmm_scan_matches_the_cross_scan_chain(b, HW):
    """Same arithmetic, two data paths: K1f against cross_scan_bc / cross_scan / selective_scan_lowrank_fn / cross_merge."""
    from mlagg_unet_amd import ops
    xc, xdbl, Wdt, A, D, bias, dy, L = _inputs(b, HW, seed=3)
    la = [t.to(DEV).requires_grad_(True) for t in (xc, xdbl, Wdt, A, D, bias)]
    ya = ops.msmm_scan(la[0], la[1], ops.msmm_scan_index(HW, DEV), *la[2:])
    ya.backward(dy.to(DEV))
    lb = [t.to(DEV).requires_grad_(True) for t in (xc, xdbl, Wdt, A, D, bias)]
    x35 = lb[1].view(b, L, K, XB)[..., [0, 1, 2] + list(range(4, XB))].reshape(b, L, K * 35)
    dtr, Bs, Cs = ops.cross_scan_bc(x35, HW, R, N)
    xs = ops.cross_scan(lb[0], HW, HC, 1)
    out = ops.selective_scan_lowrank_fn(xs, dtr, lb[2], lb[3], Bs, Cs, lb[4], delta_bias=lb[5], delta_softplus=True)
    yb = ops.cross_merge(out, HW, HC)
    yb.backward(dy.to(DEV))
    s = float(yb.abs().max())
    assert float((ya - yb).abs().max()) < 2e-5 * s
    for name, a_, b_ in zip(("dxc", "dxdbl", "dWdt", "dA", "dD", "dbias"), la, lb):
        want = b_.grad
        assert float((a_.grad - want).abs().max()) < 1e-4 * max(float(want.abs().max()), 1e-6), name


def test_msmm_scan_full_size_properties():
    """BASELINE config-2 shape (L_cat = 21760 = 340 whole chunks, batch 2 of 10), where the oracle is too slow to be the checker:
    (i) linear in u for fixed projections; (ii) gradients linear in dy; (iii) two runs are bit-identical (no atomics)."""
    from mlagg_unet_amd import ops
    HW = [(128, 128), (64, 64), (32, 32), (16, 16)]
    xc, xdbl, Wdt, A, D, bias, dy, L = _inputs(2, HW, seed=5)
    idx = ops.msmm_scan_index(HW, DEV)
    xd, Wd, Ad, Dd, bd = (t.to(DEV) for t in (xdbl, Wdt, A, D, bias))
    g = torch.Generator().manual_seed(8)
    u1, u2 = xc.to(DEV), torch.randn(xc.shape, generator=g).to(DEV)

    def run(u, d_out=None):
        leaves = [t.clone().requires_grad_(True) for t in (u, xd, Wd, Ad, Dd, bd)]
        y = ops.msmm_scan(leaves[0], leaves[1], idx, *leaves[2:])
        if d_out is None:
            return y.detach()
        y.backward(d_out)
        return y.detach(), [t.grad for t in leaves]

    y1, y2, y12 = run(u1), run(u2), run(0.7 * u1 - 1.3 * u2)
    scale = float(y1.abs().max())
    assert float((y12 - (0.7 * y1 - 1.3 * y2)).abs().max()) < 2e-4 * scale
    d1, d2 = dy.to(DEV), torch.randn(dy.shape, generator=g).to(DEV)
    ya, ga = run(u1, d1)
    yb, gb = run(u1, d2)
    _, gab = run(u1, d1 + 0.5 * d2)
    for name, a_, b_, ab in zip(("dxc", "dxdbl", "dWdt", "dA", "dD", "dbias"), ga, gb, gab):
        want = a_ + 0.5 * b_
        assert float((ab - want).abs().max()) < 2e-4 * float(want.abs().max()), name
    yc, gc = run(u1, d1)
    assert torch.equal(ya, yc) and all(torch.equal(p, q) for p, q in zip(ga, gc))


def test_msmm_scan_rejects_bad_arguments():
    from mlagg_unet_amd import ops
    HW = [(4, 4)]
    xc, xdbl, Wdt, A, D, bias, _, L = _inputs(1, HW, seed=1)
    idx = ops.msmm_scan_index(HW, DEV)
    args = [t.to(DEV) for t in (xc, xdbl, Wdt, A, D, bias)]
    with pytest.raises(RuntimeError):                              # odd sequence length
        ops.msmm_scan(args[0][:, :15], args[1][:, :15], idx[:, :15].contiguous(), *args[2:])
    with pytest.raises(RuntimeError):                              # index table of the wrong type
        ops.msmm_scan(args[0], args[1], idx.long(), *args[2:])
    with pytest.raises(RuntimeError):                              # host tensors: no fallback
        ops.msmm_scan(xc, xdbl, idx.cpu(), Wdt, A, D, bias)
