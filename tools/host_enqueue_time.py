#!/usr/bin/env python
"""Host time to ENQUEUE one train step (no synchronisation inside the measured region) next to the synchronised step time: says
whether a configuration is bound by the GPU or by the Python / launch path.   python tools/host_enqueue_time.py --config 3"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--precision", default=None)
    a = ap.parse_args()
    import bench
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import gemm_tuning, miopen_tuning, model, trainer
    cfg = dict(bench.CONFIGS[a.config])
    if a.precision:
        cfg["precision"] = a.precision
    dev = torch.device("cuda", 0)
    miopen_tuning.use_tuned_convolutions(enabled=a.config == 2 and cfg["precision"] == "fp32")
    gemm_tuning.use_tuned_gemms(enabled=a.config == 2 and a.precision is None)
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
    host, total = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    print(f"config {a.config} {cfg['precision']}: host enqueue {sorted(host)[len(host) // 2]:.2f} ms, step incl. GPU {sorted(total)[len(total) // 2]:.2f} ms "
          f"(flags: arena={os.environ.get('MLAGG_GRAD_ARENA', '1')} resln={os.environ.get('MLAGG_FUSED_RESIDUAL_NORM', '1')})")


if __name__ == "__main__":
    main()
