"""CPU baseline of the training augmentation chain: the scipy / numpy restatement of the reference's batchgenerators transforms
(oracle/augmentation_oracle.py) on ONE host core at the headline shape (batch 10, 301 x 301 -> 256 x 256), the figure quoted
beside tools/bench_input_path.py's device number.  Lives under tests/ because oracle code is test infrastructure.
    python tests/perf/augmentation_cpu_baseline.py [--batches 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import augmentation as AUG  # noqa: E402
from oracle import augmentation_oracle as AO  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=3)
    a = ap.parse_args()
    aug = AUG.GpuAugmenter((256, 256), "cpu", seed=0, labels=[0, 1, 2, 3])
    init = aug.initial_patch_size()
    rng = np.random.RandomState(0)
    data = rng.randn(10, 1, *init).astype(np.float32)
    seg = rng.randint(-1, 4, (10, 1, *init)).astype(np.float32)
    t0 = time.perf_counter()
    for i in range(a.batches):
        p = AUG.draw_params(np.random.RandomState(i), 10, 1, aug.rotation)
        AO.apply(data.copy(), seg.copy(), (256, 256), p, rng.randn(10, 1, 256, 256).astype(np.float32))
    cpu = (time.perf_counter() - t0) / a.batches
    print(json.dumps({"workload": "augmentation chain B:666-701 (scipy restatement), batch 10, 301x301 -> 256x256, one core",
                      "cpu_oracle_ms_per_batch": round(cpu * 1e3, 1), "cpu_img_per_s_one_core": round(10 / cpu, 1)}))


if __name__ == "__main__":
    main()
