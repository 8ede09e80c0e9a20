"""GPU: gradient TENSORS (not only norms) and a TRAIN-mode step against the reference network's own outputs.

tests/golden/full_model_<tag>_grads.npz (make_golden.golden_gradient_tensors): 44 parameter gradients spread over the encoder, the
downs, the MSMM skip module and the decoder, sub-sampled, at the headline size, the ACDC-like and the Endovis-like shapes -- a gradient
of the right norm and the wrong direction cannot pass.  tests/golden/full_model_64_train_mode.npz (make_golden.golden_train_mode): the
reference network in train mode with the keep masks of its 19 DropPath calls preset (four dropped branches, two of them in the MSMM
module) and recorded; the product network replays exactly these factors (model._DropPathPool.inject), so the stochastic-depth path (K6 residual + LayerNorm with per-sample scale, K8 scaled residual,
reference T:903-907, M:741-745) is compared value by value."""
import os

import numpy as np
import pytest
import torch

from oracle import mlagg_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _compare_gradient_tensors(net, g, rtol):
    params = dict(net.named_parameters())
    worst = ("", 0.0)
    for n in (str(v) for v in g["names"]):
        want = g["grad/" + n]
        stride = int(g["stride/" + n])
        got = params[n].grad.detach().reshape(-1)[::stride].cpu().numpy()
        assert got.shape == want.shape, n
        scale = float(np.abs(want).max())
        err = float(np.abs(got - want).max())
        if scale > 0:
            worst = max(worst, (n, err / scale), key=lambda t: t[1])
        # the cosine pins the direction, the max-norm error the values (floor: gradients that are analytically ~0)
        assert err <= rtol * scale + 1e-7, (n, err, scale)
        if scale > 1e-6:
            cos = float((got.astype(np.float64) * want).sum() / (np.linalg.norm(got.astype(np.float64)) * np.linalg.norm(want) + 1e-300))
            assert cos > 1 - 1e-4, (n, cos)
    return worst


@pytest.mark.parametrize("tag", ["256_variantB", "224_variantB", "512x640_variantA"])
def test_gradient_tensors_match_the_reference(tag):
    from mlagg_unet_amd import model as PM, trainer as TR
    g = np.load(os.path.join(GOLD, f"full_model_{tag}_grads.npz"))
    img = tuple(int(v) for v in g["img"])
    in_ch, n_cls, batch, variant = int(g["in_ch"]), int(g["n_cls"]), int(g["batch"]), str(g["variant"])
    m = PM.build_network_architecture(img, in_ch, n_cls, True, variant)
    O.deterministic_fill_(m.state_dict())
    m = m.to(DEV).eval()
    data, target = O.synthetic_batch(batch, in_ch, *img, n_cls, seed=int(g["data_seed"]))
    loss = TR.deep_supervision_loss(m(data.to(DEV)), [t.to(DEV) for t in target], batch_dice=True)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    loss.backward()
    assert len(g["names"]) >= 40
    worst = _compare_gradient_tensors(m, g, rtol=5e-3)
    print(tag, "worst gradient-tensor error (relative to the tensor's max)", worst)


def test_train_mode_step_matches_the_reference_with_its_droppath_draws():
    from mlagg_unet_amd import model as PM, trainer as TR
    g = np.load(os.path.join(GOLD, "full_model_64_train_mode.npz"))
    img = tuple(int(v) for v in g["img"])
    m = PM.build_network_architecture(img, 1, int(g["n_cls"]), True, "B")
    O.deterministic_fill_(m.state_dict())
    m = m.to(DEV).train()
    data, target = O.synthetic_batch(int(g["batch"]), 1, *img, int(g["n_cls"]), seed=int(g["data_seed"]))
    masks = torch.from_numpy(g["masks"])
    assert tuple(masks.shape) == (19, 2) and int((masks == 0).sum()) == 4       # four dropped branches (two in the MSMM module)
    m._dp_pool.inject(masks)
    out = m(data.to(DEV))
    for i, o in enumerate(out):
        assert float((o.detach().cpu() - torch.from_numpy(g[f"out{i}"])).abs().max()) < 1e-3, i
    loss = TR.deep_supervision_loss(out, [t.to(DEV) for t in target], batch_dice=True)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    loss.backward()
    norms = {n: float(p.grad.double().norm()) for n, p in m.named_parameters() if p.grad is not None}
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        assert abs(norms[str(n)] - ref) <= 5e-3 * max(ref, 1e-3), (n, norms[str(n)], ref)
    _compare_gradient_tensors(m, g, rtol=5e-3)
    # the eval-mode output of the same network differs: the injected factors really reached the kernels
    m.eval()
    with torch.no_grad():
        ev = m(data.to(DEV))
    assert float((ev[0] - out[0].detach()).abs().max()) > 1e-2
    # a forward that makes fewer DropPath calls than factors were injected is an error, not a silent partial use
    m.train()
    m._dp_pool.inject(torch.ones(25, 2))
    with pytest.raises(RuntimeError):
        m(data.to(DEV))
