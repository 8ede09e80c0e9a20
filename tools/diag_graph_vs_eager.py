#!/usr/bin/env python
"""Diagnostic: hipGraph replay vs eager steps in deterministic mode -- which parameters differ and by how much."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import model, trainer  # noqa: E402

trainer.set_deterministic(True)
torch.manual_seed(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = model.build_network_architecture((size, size), 1, 14, True, "B").cuda().eval()
twin, twin2 = copy.deepcopy(net), copy.deepcopy(net)
opt, _ = trainer.configure_optimizers(net, capturable=True)
opt_t, _ = trainer.configure_optimizers(twin)
opt_t2, _ = trainer.configure_optimizers(twin2)
batches = [trainer.synthetic_batch(2, 1, size, size, 14, seed=40 + i, device="cuda") for i in range(3)]
graphed = trainer.GraphedTrainStep(net, opt, *batches[0], batch_dice=True, warmup=3)
for _ in range(3):
    trainer.train_step(twin, opt_t, *batches[0])
    trainer.train_step(twin2, opt_t2, *batches[0])
for data, target in batches:
    a = float(graphed(data, target)); b = float(trainer.train_step(twin, opt_t, data, target)); c = float(trainer.train_step(twin2, opt_t2, data, target))
    print("loss graph %.7f eager %.7f eager2 %.7f" % (a, b, c))
for name, other in (("graph vs eager", twin), ("eager vs eager2", twin2)):
    ref = net if name.startswith("graph") else twin
    diffs = []
    for (k, p), q in zip(ref.state_dict().items(), other.state_dict().values()):
        d = float((p.float() - q.float()).abs().max())
        if d > 0:
            diffs.append((d, k, int(((p.float() - q.float()).abs() > 0).sum()), p.numel()))
    diffs.sort(reverse=True)
    print(name, ":", len(diffs), "tensors differ of", len(ref.state_dict()))
    for d in diffs[:12]:
        print("   %.3e %s (%d / %d elements)" % d)
print("checksum eager  %.12e" % sum(float(p.double().sum()) for p in twin.state_dict().values()))
print("checksum graph  %.12e" % sum(float(p.double().sum()) for p in net.state_dict().values()))
