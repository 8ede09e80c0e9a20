"""Throughput of the real-data input path at the headline shape (batch 10, loader patch 301x301 -> 256x256):
the device augmentation chain (augmentation.GpuAugmenter).  The CPU figure beside it (the scipy restatement of the reference's
batchgenerators chain, one core) is tests/perf/augmentation_cpu_baseline.py: oracle code is only run from tests/.
    python tools/bench_input_path.py [--batches 50]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import augmentation as AUG  # noqa: E402
from mlagg_unet_amd import dataloading as DL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    aug = AUG.GpuAugmenter((256, 256), dev, seed=0, labels=[0, 1, 2, 3])
    init = aug.initial_patch_size()
    rng = np.random.RandomState(0)
    data = torch.from_numpy(rng.randn(10, 1, *init).astype(np.float32)).pin_memory()
    seg = torch.from_numpy(rng.randint(-1, 4, (10, 1, *init)).astype(np.float32)).pin_memory()
    for _ in range(5):
        DL.to_device({"data": data, "seg": seg}, dev, augmenter=aug)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.batches):
        DL.to_device({"data": data, "seg": seg}, dev, augmenter=aug)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / a.batches
    print(json.dumps({"workload": "augmentation chain B:666-701, batch 10, 301x301 -> 256x256, H2D included",
                      "gpu_ms_per_batch": round(gpu * 1e3, 3), "gpu_img_per_s": round(10 / gpu, 1)}))


if __name__ == "__main__":
    main()
