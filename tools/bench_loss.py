#!/usr/bin/env python
"""Time of the deep-supervision Dice+CE loss (forward + backward to the 5 logit maps) at config 2 shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import trainer  # noqa: E402

dev = torch.device("cuda:0")
B, C = 10, 14
outs = [torch.randn(B, C, 256 >> s, 256 >> s, device=dev, requires_grad=True) for s in range(5)]
_, tg = trainer.synthetic_batch(B, 1, 256, 256, C, device=dev)


def step():
    loss = trainer.deep_supervision_loss(outs, tg, batch_dice=True)
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20):
    step()
t1.record()
torch.cuda.synchronize()
print(f"deep-supervision loss fwd+bwd: {t0.elapsed_time(t1) / 20:.3f} ms")
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step()
    torch.cuda.synchronize()
n = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print("device kernels per loss evaluation:", n)
