"""GPU: the augmentation chain on HBM-resident batches against the scipy oracle (same cases as the CPU test), and the
prefetching loader running it on its side stream in front of a train step."""
import numpy as np
import pytest
import torch

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import augmentation as AUG
from mlagg_unet_amd import dataloading as DL
from oracle import augmentation_oracle as AO
from oracle import dataloading_oracle as DO
from tests import _augmentation_cases as K

pytestmark = pytest.mark.gpu
TOL = 3e-5                   # fp32 prefilter matmul + device pow / exp against float64 scipy, images of amplitude ~5


@pytest.mark.parametrize("keys", list(K.STAGES) + [None], ids=lambda k: "chain" if k is None else k[0])
def test_device_transforms_match_the_oracle(keys):
    shape = K.OUT if keys is not None and keys[0] != "do_rot" else K.IN
    data, seg = K.images(shape=shape)
    p = K.forced_params() if keys is None else K.only(K.forced_params(), keys)
    noise = np.random.RandomState(5).randn(K.B, K.C, *K.OUT).astype(np.float32)
    got_d, got_s = AUG.GpuAugmenter(K.OUT, "cuda:0").apply(torch.from_numpy(data).cuda(), torch.from_numpy(seg).cuda(), p,
                                                           torch.from_numpy(noise).cuda())
    want_d, want_s = AO.apply(data.copy(), seg.copy(), K.OUT, p, noise)
    assert got_d.is_cuda and got_s.is_cuda
    assert np.abs(got_d.cpu().numpy() - want_d).max() < TOL
    assert (got_s.cpu().numpy() != want_s).mean() < 1e-4          # a linear weight of exactly 0.5 may round either way in fp32


def test_augmented_prefetch_drives_a_train_step(tmp_path):
    from mlagg_unet_amd import model, trainer
    DO.write_synthetic_dataset(str(tmp_path), unpack=True, small=False)
    aug = AUG.GpuAugmenter((64, 64), "cuda:0", seed=3, labels=[0, 1, 2, 3])
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 2, aug.initial_patch_size(), (64, 64), [0, 1, 2, 3], 0.33)
    feed = DL.PrefetchLoader(dl, "cuda:0", num_workers=2, depth=3, augmenter=aug)
    try:
        torch.manual_seed(0)
        net = model.build_network_architecture((64, 64), 1, 4, True, "B").cuda().train()
        opt, _ = trainer.configure_optimizers(net)
        for _ in range(3):
            data, target = feed.next()
            assert data.shape == (2, 1, 64, 64) and [t.shape[-1] for t in target] == [64, 32, 16, 8, 4]
            loss = trainer.train_step(net, opt, data, target)
        assert torch.isfinite(loss).item()
    finally:
        feed.close()
