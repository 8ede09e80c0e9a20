#!/usr/bin/env python
"""Which tensors of one train-step's forward / backward differ between two runs on identical inputs and weights?
(the plugin-vs-trainer trajectory test was loosened because runs differ; by which kernel?)

    python tools/find_nondeterminism.py [--size 256] [--batch 10] [--deterministic]

Prints, per mode, whether logits and loss are bit-identical and every parameter whose gradient is not, in module order."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: F401,E402
from mlagg_unet_amd import model, trainer  # noqa: E402


def one_run(net, data, target):
    net.zero_grad(set_to_none=True)
    out = net(data)
    loss = trainer.deep_supervision_loss(out, target, True, False)
    loss.backward()
    return [o.detach().clone() for o in out], loss.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()
                                                                      if p.grad is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--deterministic", action="store_true", help="torch.backends.cudnn.deterministic = True")
    a = ap.parse_args()
    torch.backends.cudnn.deterministic = a.deterministic
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = model.build_network_architecture((a.size, a.size), 1, 14, True, "B").to(dev).eval()      # eval: no DropPath draws
    data, target = trainer.synthetic_batch(a.batch, 1, a.size, a.size, 14, seed=1234, device=dev)
    ref = one_run(net, data, target)
    bad_out, bad_loss, bad = False, False, {}
    for _ in range(a.repeats):
        out, loss, grads = one_run(net, data, target)
        bad_out |= any(not torch.equal(x, y) for x, y in zip(out, ref[0]))
        bad_loss |= not torch.equal(loss, ref[1])
        for n, g in grads.items():
            if not torch.equal(g, ref[2][n]):
                rel = float((g - ref[2][n]).abs().max() / (ref[2][n].abs().max() + 1e-30))
                bad[n] = max(bad.get(n, 0.0), rel)
    print(f"mode: K15_2D={os.environ.get('MLAGG_K15_2D', '0')} cudnn.deterministic={a.deterministic} det_loss={os.environ.get('MLAGG_DETERMINISTIC', '0')}")
    print(f"logits bit-identical: {not bad_out}; loss bit-identical: {not bad_loss}; gradients that differ: {len(bad)} of {len(ref[2])}")
    for n, _ in net.named_parameters():
        if n in bad:
            print(f"  {n:70s} max rel diff {bad[n]:.2e}")


if __name__ == "__main__":
    main()
