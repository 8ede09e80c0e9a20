"""GPU: the device half of the real-data input path: pinned host batch -> HBM on a side stream, label
-1 -> 0 and the five deep-supervision targets built on the device, against the oracle's restatement of the reference's
RemoveLabelTransform + DownsampleSegForDSTransform2 chain (nnUNetTrainer.py:713, 746-748), and a train step fed by it."""
import numpy as np
import pytest
import torch

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import dataloading as DL
from oracle import dataloading_oracle as DO

pytestmark = pytest.mark.gpu
SCALES = [[1.0 / 2 ** i] * 2 for i in range(5)]          # reference T:101-104


@pytest.mark.parametrize("patch", [(64, 64), (96, 160)])
def test_device_target_pyramid_matches_oracle(tmp_path, patch):
    DO.write_synthetic_dataset(str(tmp_path), unpack=True, small=False)
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 3, patch, patch, [0, 1, 2, 3], 0.5, rng=np.random.RandomState(4))
    b = dl.generate_train_batch()
    assert b["data"].is_pinned() and b["seg"].is_pinned()
    side = torch.cuda.Stream()
    data, targets = DL.to_device(b, "cuda:0", stream=side)
    torch.cuda.current_stream().wait_stream(side)
    want = DO.downsample_seg_for_ds(b["seg"].numpy(), SCALES)
    assert torch.equal(data.cpu(), b["data"])
    for got, w in zip(targets, want):
        assert got.dtype == torch.float32 and got.is_cuda
        assert np.array_equal(got.cpu().numpy(), w)          # labels: bit-exact


def test_prefetched_batches_drive_a_train_step(tmp_path):
    from mlagg_unet_amd import model, trainer
    DO.write_synthetic_dataset(str(tmp_path), unpack=True, small=False)
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 2, (64, 64), (64, 64), [0, 1, 2, 3], 0.33)
    feed = DL.PrefetchLoader(dl, "cuda:0", num_workers=2, depth=3)
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


def test_prefetch_loader_reports_a_dead_worker(tmp_path):
    DO.write_synthetic_dataset(str(tmp_path), unpack=True)
    ds = DL.Dataset(str(tmp_path))
    dl = DL.DataLoader2D(ds, 2, (32, 32), (32, 32), [0, 1, 2, 3], 0.33)
    import os
    for f in os.listdir(str(tmp_path)):
        if f.endswith(".pkl"):
            os.remove(os.path.join(str(tmp_path), f))      # every worker's first batch now fails
    feed = DL.PrefetchLoader(dl, "cuda:0", num_workers=2, depth=2)
    try:
        with pytest.raises(RuntimeError):
            feed.next()
    finally:
        feed.close()
