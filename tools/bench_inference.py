"""Sliding-window inference throughput on one MI355X: tiles/s of the product network at the
reference's 256x256 patch, batched (this repo) vs the reference's execution pattern (one tile, one flip per forward).
    python tools/bench_inference.py [--slices 4] [--size 512 640] [--tile-batch 8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import inference as PI  # noqa: E402
from mlagg_unet_amd import model as PM  # noqa: E402


class OneAtATime(torch.nn.Module):
    """Feeds the network one sample per forward: the launch pattern of the reference loop."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x):
        return torch.cat([self.net(x[i:i + 1]) for i in range(x.shape[0])])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=4)
    ap.add_argument("--size", type=int, nargs=2, default=(512, 640))
    ap.add_argument("--tile-batch", type=int, default=8)
    ap.add_argument("--classes", type=int, default=14)
    a = ap.parse_args()
    tile = (256, 256)
    net = PM.build_network_architecture(tile, 1, a.classes, False, "B").to("cuda:0").eval()
    image = torch.randn(1, a.slices, *a.size)
    steps = PI.compute_steps_for_sliding_window(a.size, tile, 0.5)
    ntiles = a.slices * len(steps[0]) * len(steps[1])
    for name, n, tb in (("batched", net, a.tile_batch), ("one-at-a-time", OneAtATime(net), 1)):
        PI.predict_sliding_window_return_logits(n, image[:, :1], a.classes, tile, (0, 1), tile_batch=tb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = PI.predict_sliding_window_return_logits(n, image, a.classes, tile, (0, 1), tile_batch=tb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:14s} tile_batch={tb}: {ntiles} tiles x 4 mirror variants in {dt * 1e3:.1f} ms "
              f"= {ntiles / dt:.1f} tiles/s ({4 * ntiles / dt:.1f} forwards/s)  checksum {float(out.sum()):.4f}", flush=True)


if __name__ == "__main__":
    main()
