#!/usr/bin/env python
"""Chronological list of the device-time-carrying aten / custom ops of one train step whose inputs have a given
leading shape (default stage 0 of the encoder: 10 x 16384 tokens).  Diagnostics only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import model, trainer  # noqa: E402

KEY = sys.argv[1] if len(sys.argv) > 1 else "16384"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = model.build_network_architecture((256, 256), 1, 14, True, "B").to(dev).train()
opt, _ = trainer.configure_optimizers(net)
data, target = trainer.synthetic_batch(10, 1, 256, 256, 14, device=dev)
for _ in range(3):
    trainer.train_step(net, opt, data, target)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.train_step(net, opt, data, target)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.self_device_time_total > 0 and e.device_type == torch.autograd.DeviceType.CPU]
evs.sort(key=lambda e: e.time_range.start)
for e in evs:
    shp = str([s for s in (e.input_shapes or []) if s])
    if KEY in shp:
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        print(f"{e.self_device_time_total:8.1f} us  {e.name:40s} <- {par:45s} {shp[:110]}")
