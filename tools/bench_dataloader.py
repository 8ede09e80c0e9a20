#!/usr/bin/env python
"""Does the train step hold its synthetic-batch throughput with the real-data input path attached?
Writes a synthetic preprocessed 2-D dataset (unpacked .npy, AbdomenMR-like slice sizes) under --dir, then measures
(a) the loader alone (batches/s per thread count), (b) train steps fed by PrefetchLoader vs resident synthetic batches.
    python tools/bench_dataloader.py [--dir /tmp/mlagg_ds] [--cases 12] [--steps 20] [--workers 6]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import dataloading as DL  # noqa: E402
from mlagg_unet_amd import miopen_tuning, model, trainer  # noqa: E402


def write_synthetic_dataset(folder, n_cases, labels, seed=3):
    """nnUNet_preprocessed-style 2-D case folder with AbdomenMR-like slice sizes: <case>.npz, unpacked .npy / _seg.npy, and
    <case>.pkl with `class_locations` (up to 10 000 sampled voxels per class, as the reference preprocessor stores them)."""
    import pickle
    os.makedirs(folder, exist_ok=True)
    rng = np.random.RandomState(seed)
    for i in range(n_cases):
        D, H, W = 5 + i, 300 + 8 * i, 280 + 12 * i
        data = rng.standard_normal((1, D, H, W)).astype(np.float32)
        seg = np.zeros((1, D, H, W), dtype=np.int16)
        for lab in labels:
            d, y, x = rng.randint(0, D), rng.randint(2, H - 40), rng.randint(2, W - 40)
            seg[0, d:d + 2, y:y + 30, x:x + 30] = lab
        name = f"case_{i:03d}"
        np.savez_compressed(os.path.join(folder, name + ".npz"), data=data, seg=seg)
        np.save(os.path.join(folder, name + ".npy"), data)
        np.save(os.path.join(folder, name + "_seg.npy"), seg)
        locs = {}
        for lab in labels:
            a = np.argwhere(seg == lab)
            locs[lab] = a[rng.choice(len(a), min(10000, len(a)), replace=False)] if len(a) else []
        with open(os.path.join(folder, name + ".pkl"), "wb") as fh:
            pickle.dump({"class_locations": locs}, fh)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/tmp/mlagg_ds")
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workers", type=int, default=6)
    a = ap.parse_args()
    if not os.path.isdir(a.dir) or not os.listdir(a.dir):
        write_synthetic_dataset(a.dir, a.cases, tuple(range(1, 14)))
    ds = DL.Dataset(a.dir)
    B, patch = 10, (256, 256)
    labels = list(range(14))
    for nw in (1, a.workers):
        dl = DL.DataLoader2D(ds, B, patch, patch, labels, 0.33, rng=np.random.RandomState(0))
        t0 = time.perf_counter()
        if nw == 1:
            for _ in range(20):
                dl.generate_train_batch()
            n = 20
        else:
            pf = DL.PrefetchLoader(dl, "cpu", num_workers=nw, depth=8)
            for _ in range(60):
                pf.next()
            pf.close()
            n = 60
        dt = time.perf_counter() - t0
        print(f"loader alone, {nw} thread(s): {n / dt:.1f} batches/s = {n * B / dt:.0f} images/s", flush=True)

    dev = torch.device("cuda:0")
    miopen_tuning.use_tuned_convolutions()
    torch.manual_seed(0)
    net = model.build_network_architecture(patch, 1, 14, True, "B").to(dev).train()
    opt, _ = trainer.configure_optimizers(net)
    data, target = trainer.synthetic_batch(B, 1, *patch, 14, device=dev)
    for _ in range(4):
        trainer.train_step(net, opt, data, target)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.train_step(net, opt, data, target)
    torch.cuda.synchronize()
    syn = (time.perf_counter() - t0) / a.steps
    dl = DL.DataLoader2D(ds, B, patch, patch, labels, 0.33)
    pf = DL.PrefetchLoader(dl, dev, num_workers=a.workers, depth=6)
    for _ in range(3):
        trainer.train_step(net, opt, *pf.next())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.train_step(net, opt, *pf.next())
    torch.cuda.synchronize()
    real = (time.perf_counter() - t0) / a.steps
    pf.close()
    # the full training input path: loader at the initial patch size + the augmentation chain on the device (B:666-701)
    from mlagg_unet_amd import augmentation  # noqa: E402
    aug = augmentation.GpuAugmenter(patch, dev, seed=0, labels=labels)
    dl = DL.DataLoader2D(ds, B, aug.initial_patch_size(), patch, labels, 0.33)
    pf = DL.PrefetchLoader(dl, dev, num_workers=a.workers, depth=6, augmenter=aug)
    for _ in range(3):
        trainer.train_step(net, opt, *pf.next())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.train_step(net, opt, *pf.next())
    torch.cuda.synchronize()
    augd = (time.perf_counter() - t0) / a.steps
    pf.close()
    print(f"train step, resident synthetic batch : {syn * 1e3:.2f} ms  ({B / syn:.1f} images/s)")
    print(f"train step, PrefetchLoader ({a.workers} threads): {real * 1e3:.2f} ms  ({B / real:.1f} images/s)")
    print(f"train step, PrefetchLoader + device augmentation: {augd * 1e3:.2f} ms  ({B / augd:.1f} images/s)")


if __name__ == "__main__":
    main()
