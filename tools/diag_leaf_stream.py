#!/usr/bin/env python
"""Diagnostic: the gradients of one backward on the backward stream vs the same backward with the weight gradients on the
leaf-gradient stream (ops.leaf_grad_overlap), deterministic mode: which parameters differ, by how much, and twice in a row."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import model, ops, trainer  # noqa: E402

trainer.set_deterministic(True)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
net = model.build_network_architecture((size, size), 1, 14, True, "B").cuda().eval()
data, target = trainer.synthetic_batch(2, 1, size, size, 14, seed=40, device="cuda")


def grads(overlap):
    net.zero_grad(set_to_none=True)
    loss = trainer.deep_supervision_loss(net(data), target)
    with ops.leaf_grad_overlap(overlap):
        loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}


a, b, c, d = grads(False), grads(True), grads(True), grads(False)
for name, u, v in (("plain vs plain", a, d), ("overlap vs overlap", b, c), ("plain vs overlap", a, b)):
    diffs = sorted(((float((u[k] - v[k]).abs().max() / (u[k].abs().max() + 1e-30)), k) for k in u if not torch.equal(u[k], v[k])), reverse=True)
    print(name, ":", len(diffs), "of", len(u), "gradients differ")
    for r, k in diffs[:25]:
        print("   %.2e  %s" % (r, k))

# five optimisation steps with and without the overlap, on twins
import copy  # noqa: E402
na, nb = copy.deepcopy(net), copy.deepcopy(net)
oa, _ = trainer.configure_optimizers(na)
ob, _ = trainer.configure_optimizers(nb)
batches = [trainer.synthetic_batch(2, 1, size, size, 14, seed=50 + i, device="cuda") for i in range(5)]
for i, (dt, tg) in enumerate(batches):
    ops.LEAF_STREAM = True
    la = float(trainer.train_step(na, oa, dt, tg))
    ga = {n: p.grad.clone() for n, p in na.named_parameters() if p.grad is not None}
    ops.LEAF_STREAM = False
    lb = float(trainer.train_step(nb, ob, dt, tg))
    gb = {n: p.grad.clone() for n, p in nb.named_parameters() if p.grad is not None}
    gd = sorted(((float((ga[k] - gb[k]).abs().max() / (gb[k].abs().max() + 1e-30)), k) for k in ga if not torch.equal(ga[k], gb[k])), reverse=True)
    wd = [k for (k, p), q in zip(na.state_dict().items(), nb.state_dict().values()) if not torch.equal(p, q)]
    print(f"step {i}: loss overlap {la:.7f} plain {lb:.7f}; {len(gd)} gradients differ, {len(wd)} parameters differ after the step")
    for r, k in gd[:8]:
        print("   %.2e  %s" % (r, k))
    if gd:
        break
