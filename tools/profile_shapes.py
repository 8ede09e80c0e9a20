#!/usr/bin/env python
"""torch.profiler view of one train step at config 2, aggregated by (aten op, input shapes): which tensors the
copy / add / mul / fill / reduce kernels of the step work on.  Diagnostics for fusion work only."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.train_step(net, opt, data, target)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
tot = collections.defaultdict(float)
for ev in prof.events():
    t = ev.self_device_time_total
    if t <= 0 or not ev.name.startswith("aten::"):
        continue
    shp = str([s for s in (ev.input_shapes or []) if s])[:100]
    agg[(ev.name, shp)][0] += t / 1e3
    agg[(ev.name, shp)][1] += 1
    tot[ev.name] += t / 1e3
print({k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:30]})
SKIP = ("convolution", "mm", "addmm", "bmm")
for (n, w), (ms, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if ms < 0.04:
        break
    if any(k in n for k in SKIP):
        continue
    print(f"{ms:7.3f} ms  x{c:4d}  {n:34s} {w}")
