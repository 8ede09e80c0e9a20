"""CPU: the 3-D selective-scan block.  Oracle (oracle/ss3d_oracle.py) against the REFERENCE's own SS3D
class (tests/golden/ss3d.npz, variants/mamba/UMambaEnc_SS3D.py:126-357), and the product's direction tables."""
import os

import numpy as np
import torch

from oracle import mlagg_oracle as O
from oracle import ss3d_oracle as S3

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ss3d.npz"))


def test_oracle_ss3d_matches_reference_golden():
    blk = S3.SS3D(16).eval()
    O.deterministic_fill_(blk.state_dict(), seed=12)
    x = torch.from_numpy(G["x"]).requires_grad_(True)
    y = blk(x)
    assert float((y.detach() - torch.from_numpy(G["y"])).abs().max()) < 2e-5
    y.backward(torch.from_numpy(G["gy"]))
    assert float((x.grad - torch.from_numpy(G["gx"])).abs().max()) < 2e-5 * max(1.0, float(np.abs(G["gx"]).max()))
    for n, p in blk.named_parameters():
        ref = torch.from_numpy(G["grad/" + n])
        assert float((p.grad - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), n


def test_scan_order_tables_are_the_reference_permutations():
    """ss3d.scan_orders_3d against the reference's stack / permute / flip construction (forward_corev0 :251-259) applied to
    an index volume; every row is a permutation and rows 6..11 are the reversals."""
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import ss3d
    D, H, W = 3, 4, 5
    L = D * H * W
    idx = ss3d.scan_orders_3d(D, H, W, "cpu")
    assert idx.shape == (12, L) and idx.dtype == torch.int32
    x = torch.arange(L, dtype=torch.float32).view(1, 1, D, H, W)
    a = torch.stack([x.reshape(1, -1, L), x.transpose(3, 4).reshape(1, -1, L)], 1)
    b = torch.stack([x.permute(0, 1, 3, 2, 4).reshape(1, -1, L), x.permute(0, 1, 3, 4, 2).reshape(1, -1, L)], 1)
    c = torch.stack([x.permute(0, 1, 4, 2, 3).reshape(1, -1, L), x.permute(0, 1, 4, 3, 2).reshape(1, -1, L)], 1)
    xs = torch.cat([a, b, c, a.flip(-1), b.flip(-1), c.flip(-1)], 1).view(12, L)
    assert torch.equal(xs.int(), idx)
    for k in range(12):
        assert sorted(idx[k].tolist()) == list(range(L))
