"""Real-data input path: oracle vs the reference loader's own batches
(tests/golden/dataloader.npz, numpy seeded), product loader vs the oracle draw for draw, device-side target pyramid."""
import os

import numpy as np
import pytest
import torch

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import dataloading as DL
from oracle import dataloading_oracle as DO

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataloader.npz"))
CASES = {"npz": (False, (40, 48), (32, 32), 4, 0.33), "npy": (True, (32, 32), (32, 32), 5, 0.5),
         "ign": (True, (36, 36), (32, 32), 6, 0.33)}          # "ign": dataset with an ignore label (partially annotated cases)


@pytest.mark.parametrize("tag", ["npz", "npy", "ign"])
def test_oracle_and_product_reproduce_reference_batches(tag, tmp_path):
    unpack, patch, final, bs, fg = CASES[tag]
    ign = tag == "ign"
    DO.write_synthetic_dataset(str(tmp_path), unpack=unpack, ignore_label=4 if ign else None)
    labels = [0, 1, 2, 3]
    ora = DO.DataLoader2D(DO.Dataset(str(tmp_path)), bs, patch, final, labels, fg, has_ignore=ign)
    np.random.seed(7)
    ob = [ora.generate_train_batch() for _ in range(3)]
    prod = DL.DataLoader2D(DL.Dataset(str(tmp_path)), bs, patch, final, labels, fg, pin_memory=False, has_ignore=ign)
    np.random.seed(7)
    pb = [prod.generate_train_batch() for _ in range(3)]
    for it in range(3):
        assert [str(k) for k in ob[it]["keys"]] == list(GOLD[f"{tag}_keys_{it}"])
        assert np.array_equal(ob[it]["data"], GOLD[f"{tag}_data_{it}"]) and np.array_equal(ob[it]["seg"], GOLD[f"{tag}_seg_{it}"])
        assert [str(k) for k in pb[it]["keys"]] == list(GOLD[f"{tag}_keys_{it}"])
        assert pb[it]["data"].dtype == torch.float32 and pb[it]["seg"].dtype == torch.int16
        assert np.array_equal(pb[it]["data"].numpy(), GOLD[f"{tag}_data_{it}"])
        assert np.array_equal(pb[it]["seg"].numpy(), GOLD[f"{tag}_seg_{it}"])
    assert (GOLD[f"{tag}_seg_0"] == -1).any()                    # padding / outside-body label is present in the fixture
    if ign:
        assert any((GOLD[f"ign_seg_{it}"] == 4).any() for it in range(3))      # the ignore label reaches the batches


def test_oversampled_samples_contain_foreground(tmp_path):
    DO.write_synthetic_dataset(str(tmp_path), n_cases=3, unpack=True)
    ds = DL.Dataset(str(tmp_path), ["case_000", "case_002"])       # both have foreground
    dl = DL.DataLoader2D(ds, 6, (24, 24), (24, 24), [0, 1, 2, 3], 1.0, rng=np.random.RandomState(0), pin_memory=False)
    for _ in range(5):
        b = dl.generate_train_batch()
        assert all((b["seg"][j] > 0).any() for j in range(6))


def test_device_targets_pyramid(tmp_path):
    DO.write_synthetic_dataset(str(tmp_path), unpack=True)
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 2, (32, 32), (32, 32), [0, 1, 2, 3], 0.5,
                         rng=np.random.RandomState(1), pin_memory=False)
    b = dl.generate_train_batch()
    data, targets = DL.to_device(b, "cpu")
    assert data.shape == (2, 1, 32, 32) and [t.shape[-1] for t in targets] == [32, 16, 8, 4, 2]
    seg = b["seg"].float().clamp_min(0)
    assert torch.equal(targets[0], seg) and float(targets[0].min()) == 0.0
    assert torch.equal(targets[1], seg[:, :, 1::2, 1::2])          # nearest-exact: the odd pixel of each 2x2 cell
    assert torch.equal(targets[2], seg[:, :, 2::4, 2::4])


def test_target_pyramid_equals_the_reference_transform_chain(tmp_path):
    """to_device's label clean-up + nearest down-sampling vs the oracle's restatement of RemoveLabelTransform(-1, 0) +
    DownsampleSegForDSTransform2(order 0) (nnUNetTrainer.py:713, 746-748) on a non-square patch."""
    DO.write_synthetic_dataset(str(tmp_path), unpack=True, small=False)
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 3, (96, 160), (96, 160), [0, 1, 2, 3], 0.5,
                         rng=np.random.RandomState(4), pin_memory=False)
    b = dl.generate_train_batch()
    _, targets = DL.to_device(b, "cpu")
    want = DO.downsample_seg_for_ds(b["seg"].numpy(), [[1.0 / 2 ** i] * 2 for i in range(5)])
    for got, w in zip(targets, want):
        assert np.array_equal(got.numpy(), w)


def test_prefetch_loader_raises_when_its_workers_fail(tmp_path):
    DO.write_synthetic_dataset(str(tmp_path), unpack=True)
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 2, (32, 32), (32, 32), [0, 1, 2, 3], 0.33, pin_memory=False)
    for f in os.listdir(str(tmp_path)):
        if f.endswith(".pkl"):
            os.remove(os.path.join(str(tmp_path), f))
    pf = DL.PrefetchLoader(dl, "cpu", num_workers=2, depth=2)
    try:
        with pytest.raises(RuntimeError, match="worker failed"):
            pf.next()
    finally:
        pf.close()


def test_prefetch_loader_on_cpu(tmp_path):
    DO.write_synthetic_dataset(str(tmp_path), unpack=True)
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 3, (32, 32), (32, 32), [0, 1, 2, 3], 0.33, pin_memory=False)
    pf = DL.PrefetchLoader(dl, "cpu", num_workers=2, depth=3)
    try:
        for _ in range(6):
            data, targets = pf.next()
            assert data.shape == (3, 1, 32, 32) and len(targets) == 5 and torch.isfinite(data).all()
    finally:
        pf.close()


def test_out_of_range_label_is_refused(tmp_path):
    """A label above the dataset's largest would make torch's nll_loss fail in the reference; the fused loss kernel has no
    such check, so the loader refuses the batch."""
    DO.write_synthetic_dataset(str(tmp_path), unpack=True)
    dl = DL.DataLoader2D(DL.Dataset(str(tmp_path)), 4, (32, 32), (32, 32), [0, 1, 2], 0.0, rng=np.random.RandomState(0),
                         pin_memory=False)                      # the files hold label 3
    with pytest.raises(RuntimeError, match="segmentation label 3 > 2"):
        for _ in range(20):
            dl.generate_train_batch()
