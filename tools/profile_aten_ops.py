#!/usr/bin/env python
"""Where do the remaining ATen elementwise / cat / copy kernels of a train step come from?  One profiled step at config 2;
prints (op, input shapes, nearest repo frame) -> calls and device time, forward and backward (autograd ops carry the frame of
the forward op that created the node only through their name, so backward ops are grouped by name + shapes).
    python tools/profile_aten_ops.py [--ops add,cat,copy_,clone,contiguous,fill_,zero_,mul]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import miopen_tuning, model, trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="add,add_,cat,copy_,clone,contiguous,fill_,zero_,mul,sum,index_select,slice_backward,select_backward")
    a = ap.parse_args()
    want = {"aten::" + o for o in a.ops.split(",")}
    dev = torch.device("cuda:0")
    miopen_tuning.use_tuned_convolutions()
    torch.manual_seed(0)
    net = model.build_network_architecture((256, 256), 1, 14, True, "B").to(dev).train()
    opt, _ = trainer.configure_optimizers(net)
    data, target = trainer.synthetic_batch(10, 1, 256, 256, 14, device=dev)
    for _ in range(3):
        trainer.train_step(net, opt, data, target)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        trainer.train_step(net, opt, data, target)
        torch.cuda.synchronize()
    acc = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name not in want:
            continue
        dt = getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0)
        frame = ""
        for fr in (ev.stack or []):
            if "mlagg" in fr and "site-packages" not in fr:
                frame = fr.split("/")[-1][:70]
                break
        shapes = str([tuple(s) for s in (ev.input_shapes or []) if s])[:80]
        k = (ev.name, shapes, frame)
        acc[k][0] += 1
        acc[k][1] += dt
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"total device time of the selected ops: {tot / 1e3:.2f} ms in {sum(v[0] for _, v in rows)} calls")
    for (name, shapes, frame), (n, t) in rows[:70]:
        print(f"{t:8.1f} us {n:4d}x  {name:22s} {shapes:82s} {frame}")


if __name__ == "__main__":
    main()
