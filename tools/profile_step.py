#!/usr/bin/env python
"""torch.profiler view of one train step at config 2: which Python lines launch the copy / add / fill /
sum kernels (diagnostics for fusion work; not part of the product)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import model, trainer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = model.build_network_architecture((256, 256), 1, 14, True, "B").to(dev).train()
opt, _ = trainer.configure_optimizers(net)
data, target = trainer.synthetic_batch(10, 1, 256, 256, 14, device=dev)
for _ in range(3):
    trainer.train_step(net, opt, data, target)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_step(net, opt, data, target)
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul",
         "aten::cat", "aten::clone", "aten::contiguous")
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.name in WATCH and ev.device_time > 0:
        frames = [f for f in (ev.stack or []) if "mlagg" in f or "trainer" in f or "ops.py" in f]
        where = frames[0].split("/")[-1] if frames else "(autograd/optimizer)"
        key = (ev.name, where[:70])
        agg[key][0] += ev.device_time / 1e3
        agg[key][1] += 1
tot = collections.defaultdict(float)
for (n, w), (ms, c) in agg.items():
    tot[n] += ms
print({k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
for (n, w), (ms, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{ms:7.3f} ms  x{c:4d}  {n:16s} {w}")
