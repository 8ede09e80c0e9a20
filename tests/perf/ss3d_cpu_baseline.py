"""CPU baseline of the SS3D block (oracle/ss3d_oracle.py: the reference module's arithmetic in plain PyTorch, scan loop over
`unbind` views) on a bounded sample, scaled per token, beside tools/bench_ss3d.py's device number.
    python tests/perf/ss3d_cpu_baseline.py [--dims 12 20 20] [--batch 1] [--threads 16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ss3d_oracle as SO  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[12, 20, 20])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--d-model", type=int, default=48)
    ap.add_argument("--threads", type=int, default=min(16, len(os.sched_getaffinity(0))))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    torch.manual_seed(0)
    blk = SO.SS3D(a.d_model)
    D, H, W = a.dims
    x = torch.randn(a.batch, D, H, W, a.d_model, requires_grad=True)
    gy = torch.randn(a.batch, D, H, W, a.d_model)
    blk(x).backward(gy)                                     # warm-up
    t0 = time.perf_counter()
    blk(x).backward(gy)
    dt = time.perf_counter() - t0
    L = D * H * W
    print(json.dumps({"workload": f"SS3D oracle fwd+bwd, batch {a.batch}, volume {D}x{H}x{W} = {L} tokens, d_model {a.d_model}",
                      "threads": a.threads, "s_per_fwd_bwd": round(dt, 2), "tokens_per_s": round(a.batch * L / dt)}))


if __name__ == "__main__":
    main()
