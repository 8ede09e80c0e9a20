#!/usr/bin/env python
"""SS3D (the 3-D 12-direction selective-scan block) at a BTCV-like stage shape: forward + backward of one
block on the MI355X, per-kernel times from the library's own timers.
    python tools/bench_ss3d.py [--dims 24 40 40] [--batch 2] [--d-model 48] [--iters 10]
BASELINE configs[3]: 96 x 160 x 160 patches, batch 2 per GPU; after the stem's /4 the first scan stage sees 24 x 40 x 40 =
38400 tokens per volume (L per scan direction), 12 x 96 = 1152 scan channels."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import profiling, ss3d  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[24, 40, 40])
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--d-model", type=int, default=48)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blk = ss3d.SS3D(a.d_model).to(dev)
    D, H, W = a.dims
    x = torch.randn(a.batch, D, H, W, a.d_model, device=dev, requires_grad=True)
    gy = torch.randn(a.batch, D, H, W, a.d_model, device=dev)

    def step():
        y = blk(x)
        y.backward(gy)
        return y
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.iters):
        step()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.iters
    profiling.select_all()
    step()
    table = profiling.collect()
    L = D * H * W
    # op-boundary bytes of the scan in its low-rank form: forward 4 L (2 D' + G R + 2 G N), backward
    # 4 L (3 D' + 2 G R + 4 G N) per volume with D' = 12 * d_inner channels, G = 12 groups
    dI, G, R, N = 2 * a.d_model, 12, blk.dt_rank, 16
    fwd_b = 4 * L * (2 * G * dI + G * R + 2 * G * N) * a.batch
    bwd_b = 4 * L * (3 * G * dI + 2 * G * R + 4 * G * N) * a.batch
    k = {n: round(v["ms"], 4) for n, v in table.items() if v["ms"] > 0}
    scan_bwd = table.get("selscan_bwd_group_kernel", {"ms": 0})["ms"]
    print(json.dumps({"workload": f"SS3D block fwd+bwd, batch {a.batch}, volume {D}x{H}x{W} = {L} tokens, d_model {a.d_model} "
                                  f"(12 directions x {dI} channels, dt_rank {R})",
                      "ms_per_fwd_bwd": round(ms, 3), "volumes_per_s": round(a.batch / ms * 1e3, 1),
                      "tokens_per_s": round(a.batch * L / ms * 1e3),
                      "scan_algorithmic_bytes": {"fwd": fwd_b, "bwd": bwd_b},
                      "scan_bwd_main_kernel": {"ms": round(scan_bwd, 4),
                                               "achieved_GBs": round(bwd_b / scan_bwd / 1e6, 1) if scan_bwd else None,
                                               "frac_of_8TBs": round(bwd_b / scan_bwd / 1e6 / 8000, 4) if scan_bwd else None},
                      "kernels_ms": k}))


if __name__ == "__main__":
    main()
