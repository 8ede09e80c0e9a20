#!/usr/bin/env python
"""Which model statement launches each library (ATen) kernel of the headline train step: one step under torch.profiler with input
shapes and Python stacks, grouped by (operator, input shapes, innermost frame inside this package).

    python tools/aten_tail.py [--config 2] > gpurun_out/aten_tail.md
"""
import argparse
import re
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--top", type=int, default=400)
    ap.add_argument("--all", action="store_true", help="include convolutions and GEMMs")
    a = ap.parse_args()
    import bench
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import gemm_tuning, miopen_tuning, model, trainer
    cfg = bench.CONFIGS[a.config]
    dev = torch.device("cuda", 0)
    miopen_tuning.use_tuned_convolutions(enabled=a.config == 2)
    gemm_tuning.use_tuned_gemms(enabled=a.config == 2)
    torch.manual_seed(0)
    net = model.build_network_architecture(cfg["img"], cfg["in_ch"], cfg["classes"], True, cfg["variant"], cfg["precision"]).to(dev).train()
    opt, sched = trainer.configure_optimizers(net)
    sched.step(0)
    data, target = trainer.synthetic_batch(cfg["batch"], cfg["in_ch"], *cfg["img"], cfg["classes"], seed=1234, device=dev)
    scaler = torch.amp.GradScaler("cuda") if cfg["precision"] == "fp16" else None
    for _ in range(4):
        trainer.train_step(net, opt, data, target, batch_dice=True, ddp=False, grad_scaler=scaler)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        trainer.train_step(net, opt, data, target, batch_dice=True, ddp=False, grad_scaler=scaler)
        torch.cuda.synchronize()
    groups = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if not dt or not ev.name.startswith("aten::"):
            continue
        if not a.all and re.search(r"convolution|aten::mm|aten::addmm|aten::bmm", ev.name):
            continue
        frame = ""
        for fr in (ev.stack or []):
            if "unet_amd" in fr:
                frame = fr.split("unet_amd")[-1][-70:]
                break
        if not frame:                               # backward nodes carry no Python stack: name the autograd node instead
            par = ev.cpu_parent
            while par is not None and not frame:
                if "Backward" in par.name or "autograd::engine" in par.name:
                    frame = par.name[:70]
                par = par.cpu_parent
        shapes = str([s for s in (ev.input_shapes or []) if s])[:70]
        g = groups[(ev.name, shapes, frame)]
        g[0] += dt
        g[1] += 1
    rows = sorted(groups.items(), key=lambda kv: -kv[1][0])
    total = sum(v[0] for _, v in rows)
    print(f"# ATen operators with device time, one train step of config {a.config} (torch.profiler): {total / 1e3:.2f} ms\n")
    print("| operator | input shapes | frame | calls | us |\n|---|---|---|---|---|")
    for (name, shapes, frame), (us, n) in rows[:a.top]:
        print(f"| {name} | {shapes} | {frame} | {n} | {us:.0f} |")


if __name__ == "__main__":
    main()
