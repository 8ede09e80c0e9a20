#!/usr/bin/env python
"""cProfile of the host side of one train step (which Python frames the enqueue time goes to):  python tools/host_profile.py --config 3"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--precision", default=None)
    a = ap.parse_args()
    import bench
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import model, trainer
    cfg = dict(bench.CONFIGS[a.config])
    if a.precision:
        cfg["precision"] = a.precision
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = model.build_network_architecture(cfg["img"], cfg["in_ch"], cfg["classes"], True, cfg["variant"], cfg["precision"]).to(dev).train()
    opt, sched = trainer.configure_optimizers(net)
    sched.step(0)
    data, target = trainer.synthetic_batch(cfg["batch"], cfg["in_ch"], *cfg["img"], cfg["classes"], seed=1234, device=dev)
    scaler = torch.amp.GradScaler("cuda") if cfg["precision"] == "fp16" else None
    step = lambda: trainer.train_step(net, opt, data, target, batch_dice=True, ddp=False, grad_scaler=scaler)      # noqa: E731
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
