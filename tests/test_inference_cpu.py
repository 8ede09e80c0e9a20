"""Sliding-window inference: oracle vs the reference's own outputs
(tests/golden/sliding_window.npz), product host logic vs the oracle on a tiny CPU network."""
import os

import numpy as np
import pytest
import torch

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import inference as PI
from oracle import inference_oracle as IO

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sliding_window.npz"))
CASES = (("mirror", 0, (0, 1)), ("plain", 0, None), ("padded", 1, (1,)))


def test_gaussian_and_steps_match_reference():
    assert np.array_equal(IO.compute_gaussian((32, 32)).astype(np.float32), GOLD["gaussian_32"])
    assert np.array_equal(IO.compute_gaussian((256, 256)).astype(np.float32)[::16, ::16], GOLD["gaussian_256"])
    assert np.array_equal(PI.compute_gaussian((32, 32)).numpy(), GOLD["gaussian_32"])
    flat = []
    for a, b, c in (((40, 50), (32, 32), 0.5), ((512, 640), (256, 256), 0.5), ((256, 256), (256, 256), 0.5),
                    ((300, 257), (256, 256), 0.25)):
        o, p = IO.compute_steps(a, b, c), PI.compute_steps_for_sliding_window(a, b, c)
        assert o == p
        flat += [v for ax in o for v in ax + [-1]]
    assert np.array_equal(np.asarray(flat), GOLD["steps"])
    with pytest.raises(RuntimeError):
        PI.compute_steps_for_sliding_window((20, 20), (32, 32), 0.5)


@pytest.mark.parametrize("tag,which,mirror", CASES)
def test_oracle_reproduces_reference_bitwise(tag, which, mirror):
    net, img, small = IO.sliding_window_case()
    r = IO.predict_sliding_window(net, (img, small)[which], 3, (32, 32), mirror_axes=mirror)
    assert r.dtype == torch.half
    assert np.array_equal(r.float().numpy(), GOLD[tag])


@pytest.mark.parametrize("tag,which,mirror", CASES)
@pytest.mark.parametrize("tile_batch", [1, 5])
def test_product_host_logic(tag, which, mirror, tile_batch):
    net, img, small = IO.sliding_window_case()
    image = (img, small)[which]
    got = PI.predict_sliding_window_return_logits(net, image, 3, (32, 32), mirror_axes=mirror, tile_batch=tile_batch,
                                                  device="cpu")
    exact = IO.predict_sliding_window(net, image, 3, (32, 32), mirror_axes=mirror, accum_dtype=torch.float32)
    assert got.dtype == torch.float32 and got.shape == exact.shape
    assert float((got - exact).abs().max()) < 2e-6
    # against the reference itself: its half accumulators lose the image corners, where the summed Gaussian
    # weight is a half subnormal (~1e-7); compare where the weight is representable
    ref = torch.from_numpy(GOLD[tag])
    g = PI.compute_gaussian((32, 32))
    w = torch.zeros(exact.shape[1:])
    data, rev = IO.pad_nd_image(image, (32, 32))
    wp = torch.zeros(data.shape[1:])
    for sl in IO.slicers_2d_tiles(data.shape[1:], (32, 32), 0.5):
        wp[sl[1:]] += g
    w = wp[rev[1:]]
    ok = (w > 1e-4).expand_as(ref)
    assert ok.float().mean() > 0.9
    assert float((got - ref).abs()[ok].max()) < 1e-2          # half accumulators: ~3 ulp of a logit of size ~2


def test_rejects_deep_supervision_outputs_and_bad_axes():
    net, img, _ = IO.sliding_window_case()

    class DS(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.n = net

        def forward(self, x):
            return [self.n(x), self.n(x)]
    with pytest.raises(RuntimeError):
        PI.predict_sliding_window_return_logits(DS(), img, 3, (32, 32), device="cpu")
    with pytest.raises(RuntimeError):
        PI.predict_sliding_window_return_logits(net, img, 3, (32, 32), mirror_axes=(2,), device="cpu")


def test_training_checkpoint_loads_into_inference_network():
    """SURVEY finding 7d: the reference's strict load of a deep-supervision checkpoint into the
    no-deep-supervision network fails on out_1..out_4; the product drops exactly those heads."""
    from oracle import mlagg_oracle as O
    train = O.build_reference_config_model((64, 64), 1, 3, deep_supervision=True)
    infer = O.build_reference_config_model((64, 64), 1, 3, deep_supervision=False)
    sd = train.state_dict()
    with pytest.raises(RuntimeError):
        infer.load_state_dict(sd, strict=True)
    dropped = PI.load_inference_weights(infer, sd)
    assert dropped and all(k.split(".")[0] in ("out_1", "out_2", "out_3", "out_4") for k in dropped)
    for k, v in infer.state_dict().items():
        assert torch.equal(v, sd[k])
    sd["bogus.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        PI.load_inference_weights(infer, sd)
