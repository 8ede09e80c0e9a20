"""CPU: the 3-D network oracle (oracle/umamba3d_oracle.py) against the reference's own UMambaEnc
(variants/mamba/UMambaEnc_SS3D.py:815-888; fixture tests/golden/umamba3d_small.npz made by tests/golden/make_golden.py with the
reference classes), and the host logic of the product's 3-D model (state_dict keys = the reference's, duplicates included)."""
import os

import numpy as np
import pytest
import torch

from oracle import mlagg_oracle as O
from oracle import umamba3d_oracle as U

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "umamba3d_small.npz")
CFG = dict(size=(8, 64, 64), in_ch=1, n_cls=5, batch=2,
           strides=[[1, 1, 1], [2, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2], [1, 2, 2]])          # make_golden.UMAMBA3D_SMALL


def _oracle_net():
    n = len(CFG["strides"])
    net = U.build_reference_3d_model(CFG["in_ch"], CFG["n_cls"], U.features_for(n), CFG["strides"]).eval()
    O.deterministic_fill_(net.state_dict(), seed=21)
    return net


def test_oracle_network_matches_the_reference_golden():
    gold = np.load(GOLD)
    net = _oracle_net()
    assert sorted(net.state_dict().keys()) == [str(k) for k in gold["state_keys"]]          # 614 keys incl. decoder.encoder.*
    data, target = U.synthetic_batch_3d(CFG["batch"], CFG["in_ch"], CFG["size"], CFG["strides"], CFG["n_cls"], seed=77)
    out = net(data)
    for i, o in enumerate(out):
        assert float((o.detach() - torch.from_numpy(gold[f"out{i}"])).abs().max()) < 1e-4, i
    loss = O.deep_supervision_loss(out, target, batch_dice=False)
    assert abs(float(loss.detach()) - float(gold["loss"])) < 1e-5
    loss.backward()
    grads = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert sorted(grads) == [str(n) for n in gold["grad_names"]]
    for n, want in zip(gold["grad_names"], gold["grad_norms"]):
        got = float(grads[str(n)].double().norm())
        assert abs(got - want) <= 1e-3 * want + 2e-6, (str(n), got, want)     # conv biases in front of an InstanceNorm: exact gradient 0, rounding noise ~4e-7
    for k in gold.files:
        if k.startswith("grad/"):
            g = grads[k[5:]]
            assert float((g - torch.from_numpy(gold[k])).abs().max()) <= 1e-3 * float(np.abs(gold[k]).max()) + 2e-6, k


def test_deep_supervision_scales_and_targets_follow_the_pooling_strides():
    sc = U.deep_supervision_scales(CFG["strides"])
    assert len(sc) == 5 and sc[0] == [1.0, 1.0, 1.0] and sc[4] == [1 / 8, 1 / 16, 1 / 16]
    _, target = U.synthetic_batch_3d(1, 1, CFG["size"], CFG["strides"], 5)
    assert [tuple(t.shape[2:]) for t in target] == [(8, 64, 64), (4, 32, 32), (2, 16, 16), (1, 8, 8), (1, 4, 4)]


def test_product_3d_network_has_the_reference_checkpoint_keys():
    """Module / parameter names are the checkpoint ABI: the product's UMambaEnc must expose exactly the reference's 614 keys (the
    ``decoder.encoder.*`` and ``all_modules.*`` duplicates included) with the reference's shapes."""
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import model3d
    gold = np.load(GOLD)
    n = len(CFG["strides"])
    net = model3d.build_network_architecture_3d(CFG["in_ch"], CFG["n_cls"], [[3, 3, 3]] * n, CFG["strides"], [2] * n, [2] * (n - 1))
    sd = net.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in gold["state_keys"]]
    ref = _oracle_net().state_dict()
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in _oracle_net().parameters())
    assert model3d.deep_supervision_scales(CFG["strides"]) == U.deep_supervision_scales(CFG["strides"])
    net.deep_supervision = False
    assert net.decoder.deep_supervision is False
    with pytest.raises(RuntimeError):                        # the device ops have no host path
        net(torch.zeros(1, 1, 8, 64, 64))


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_3d_builders_apply_the_references_he_initialisation(which):
    """get_umamba_enc_3d_from_plans ends with model.apply(InitWeights_He(1e-2)) (UMambaEnc_SS3D.py:941; utilities/
    network_initialization.py:4-13): every (transposed) convolution -- SS3D's depthwise Conv3d included -- has a zero bias and
    kaiming_normal_(a = 1e-2) weights, i.e. std = sqrt(2 / (1 + a^2) / fan_in)."""
    import math
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import model3d
    n = len(CFG["strides"])
    torch.manual_seed(3)
    if which == "product":
        net = model3d.build_network_architecture_3d(CFG["in_ch"], CFG["n_cls"], [[3, 3, 3]] * n, CFG["strides"], [2] * n, [2] * (n - 1))
    else:
        net = U.build_reference_3d_model(CFG["in_ch"], CFG["n_cls"], U.features_for(n), CFG["strides"])
    convs = [m for m in net.modules() if isinstance(m, (torch.nn.Conv3d, torch.nn.ConvTranspose3d))]
    assert len(convs) > 30 and any(m.groups > 1 for m in convs)                       # the depthwise convolutions are among them
    for m in convs:
        if m.bias is not None:
            assert float(m.bias.detach().abs().max()) == 0.0
        fan_in = m.weight.shape[1] * m.weight[0, 0].numel()
        want = math.sqrt(2.0 / (1 + 1e-2 ** 2) / fan_in)
        if m.weight.numel() >= 20000:                                                  # enough samples for a 5 % test of the std
            assert abs(float(m.weight.detach().std()) - want) < 0.05 * want, (tuple(m.weight.shape), float(m.weight.std()), want)
