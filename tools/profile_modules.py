#!/usr/bin/env python
"""Per-module forward / backward GPU time of one train step at config 2 (HIP events from module hooks):
where the step's milliseconds sit, by reference module name.  Diagnostics only."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import model, trainer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = model.build_network_architecture((256, 256), 1, 14, True, "B").to(dev).train()
opt, _ = trainer.configure_optimizers(net)
data, target = trainer.synthetic_batch(10, 1, 256, 256, 14, device=dev)
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 3
PREFIX = tuple(sys.argv[2:])          # only modules under these prefixes (default: all)
ev = collections.defaultdict(list)


def mk(name, kind):
    def hook(*_):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev[(name, kind)].append(e)
    return hook


for name, mod in net.named_modules():
    if not name or name.count(".") >= DEPTH or (PREFIX and not name.startswith(PREFIX)):
        continue
    mod.register_forward_pre_hook(mk(name, "f0"))
    mod.register_forward_hook(mk(name, "f1"))
    mod.register_full_backward_pre_hook(mk(name, "b0"))
    mod.register_full_backward_hook(mk(name, "b1"))
for i in range(4):
    if i == 3:
        ev.clear()
        t0 = torch.cuda.Event(enable_timing=True); t0.record()
    trainer.train_step(net, opt, data, target)
t1 = torch.cuda.Event(enable_timing=True); t1.record()
torch.cuda.synchronize()
print(f"step (with hooks): {t0.elapsed_time(t1):.2f} ms")
rows = []
for name, _ in net.named_modules():
    if (name, "f0") not in ev:
        continue
    f = sum(a.elapsed_time(b) for a, b in zip(ev[(name, "f0")], ev[(name, "f1")]))
    b = sum(a.elapsed_time(b) for a, b in zip(ev[(name, "b0")], ev[(name, "b1")])) if (name, "b1") in ev and len(ev[(name, "b0")]) == len(ev[(name, "b1")]) else float("nan")
    rows.append((name, f, b))
for name, f, b in rows:
    print(f"{'  ' * name.count('.')}{name:40s} fwd {f:7.3f}  bwd {b:7.3f}")
