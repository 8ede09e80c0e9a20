#!/usr/bin/env python
"""Micro-benchmark of the hand-written ops at BASELINE config-2 shapes (B = 10, 256x256), used under
rocprofv3 (--kernel-trace / --pmc) to study one kernel family at a time.

    python tools/bench_ops.py scan|scanlr|msmm|local|pooled|dwconv|wgrad [--iters 10]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import ops  # noqa: E402

DEV = "cuda:0"


def scan(B=10, L=21760):
    g = torch.Generator(device=DEV).manual_seed(0)
    D, G, N = 384, 4, 16
    u = torch.randn(B, D, L, device=DEV, generator=g).requires_grad_(True)
    delta = (torch.randn(B, D, L, device=DEV, generator=g) * 0.5).requires_grad_(True)
    A = (-torch.exp(torch.randn(D, N, device=DEV, generator=g) * 0.3 + 1)).requires_grad_(True)
    Bm = torch.randn(B, G, N, L, device=DEV, generator=g).requires_grad_(True)
    Cm = torch.randn(B, G, N, L, device=DEV, generator=g).requires_grad_(True)
    Dv = torch.randn(D, device=DEV, generator=g).requires_grad_(True)
    bias = (torch.randn(D, device=DEV, generator=g) - 3).requires_grad_(True)
    gy = torch.randn(B, D, L, device=DEV, generator=g)

    def step():
        y = ops.selective_scan_fn(u, delta, A, Bm, Cm, Dv, None, bias, True)
        y.backward(gy)
    return step


def scanlr(B=10, L=21760, R=3):
    """K1 as the model runs it: rank-3 delta projection inside the scan (BASELINE config 2 MSMM shapes)."""
    g = torch.Generator(device=DEV).manual_seed(0)
    D, G, N = 384, 4, 16
    u = torch.randn(B, D, L, device=DEV, generator=g).requires_grad_(True)
    dtr = torch.randn(B, G, R, L, device=DEV, generator=g).requires_grad_(True)
    Wdt = (torch.randn(D, R, device=DEV, generator=g) * 0.3).requires_grad_(True)
    A = (-torch.exp(torch.randn(D, N, device=DEV, generator=g) * 0.3 + 1)).requires_grad_(True)
    Bm = torch.randn(B, G, N, L, device=DEV, generator=g).requires_grad_(True)
    Cm = torch.randn(B, G, N, L, device=DEV, generator=g).requires_grad_(True)
    Dv = torch.randn(D, device=DEV, generator=g).requires_grad_(True)
    bias = (torch.randn(D, device=DEV, generator=g) - 3).requires_grad_(True)
    gy = torch.randn(B, D, L, device=DEV, generator=g)

    def step():
        ops.selective_scan_lowrank_fn(u, dtr, Wdt, A, Bm, Cm, Dv, bias, True).backward(gy)
    return step


def msmm(B=10, HW=((128, 128), (64, 64), (32, 32), (16, 16))):
    """K1f as the model runs it since round 4: the token-major scan of all four directions (BASELINE config 2 MSMM shapes)."""
    g = torch.Generator(device=DEV).manual_seed(0)
    L = sum(h * w for h, w in HW)
    xc = torch.randn(B, L, 96, device=DEV, generator=g).requires_grad_(True)
    xdbl = torch.randn(B, L, 144, device=DEV, generator=g).requires_grad_(True)
    Wdt = (torch.randn(384, 3, device=DEV, generator=g) * 0.3).requires_grad_(True)
    A = (-torch.exp(torch.randn(384, 16, device=DEV, generator=g) * 0.3 + 1)).requires_grad_(True)
    Dv = torch.randn(384, device=DEV, generator=g).requires_grad_(True)
    bias = (torch.randn(384, device=DEV, generator=g) - 3).requires_grad_(True)
    gy = torch.randn(B, L, 96, device=DEV, generator=g)
    idx = ops.msmm_scan_index(HW, DEV)

    def step():
        ops.msmm_scan(xc, xdbl, idx, Wdt, A, Dv, bias).backward(gy)
    return step


def local(B=10, H=128, W=128, nh=1):
    g = torch.Generator(device=DEV).manual_seed(0)
    d = 48 * nh
    q = torch.randn(B, H * W, d, device=DEV, generator=g).requires_grad_(True)
    kv = torch.randn(B, H * W, 2 * d, device=DEV, generator=g).requires_grad_(True)
    lam = torch.tensor(0.8, device=DEV).requires_grad_(True)
    sw = torch.ones(48, device=DEV).requires_grad_(True)
    lw = (torch.randn(d, 1, 3, 3, device=DEV, generator=g) * 0.2).requires_grad_(True)
    lb = torch.zeros(d, device=DEV).requires_grad_(True)
    gy = torch.randn(B, H * W, d, device=DEV, generator=g)

    def step():
        ops.local_diff_attn(q, kv, lam, sw, lw, lb, H, W, nh, 24 ** -0.5).backward(gy)
    return step


def pooled(B=10, N=16384, P=64, nh=1):
    g = torch.Generator(device=DEV).manual_seed(0)
    d = 48 * nh
    q = torch.randn(B, N, d, device=DEV, generator=g).requires_grad_(True)
    kp = torch.randn(B, P, d, device=DEV, generator=g).requires_grad_(True)
    vp = torch.randn(B, P, d, device=DEV, generator=g).requires_grad_(True)
    lam = torch.tensor(0.8, device=DEV).requires_grad_(True)
    sw = torch.ones(48, device=DEV).requires_grad_(True)
    gy = torch.randn(B, N, d, device=DEV, generator=g)

    def step():
        ops.pooled_diff_attn(q, kp, vp, lam, sw, nh, 24 ** -0.5).backward(gy)
    return step


def dwconv(B=10, H=128, W=128, C=96):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(B, H * W, C, device=DEV, generator=g).requires_grad_(True)
    w = (torch.randn(C, 1, 3, 3, device=DEV, generator=g) * 0.2).requires_grad_(True)
    b = torch.zeros(C, device=DEV).requires_grad_(True)
    gy = torch.randn(B, H * W, C, device=DEV, generator=g)

    def step():
        ops.dwconv3x3_nlc(x, w, b, H, W, silu=True).backward(gy)
    return step


def wgrad(M=163840, O=96, I=96):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(M, I, device=DEV, generator=g)
    w = torch.randn(O, I, device=DEV, generator=g).requires_grad_(True)
    b = torch.zeros(O, device=DEV).requires_grad_(True)
    gy = torch.randn(M, O, device=DEV, generator=g)

    def step():
        ops.linear(x, w, b).backward(gy)
    return step


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["scan", "scanlr", "msmm", "local", "pooled", "dwconv", "wgrad"])
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    fn = globals()[a.what]()
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    print(f"{a.what}: {t0.elapsed_time(t1) / a.iters:.3f} ms per fwd+bwd")
