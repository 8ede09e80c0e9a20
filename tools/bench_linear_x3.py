#!/usr/bin/env python
"""K5 round-4 form (csrc/linear_x3.hip: weight images, activations split in registers) against the round-3 split-bf16 kernel
(linear_lp.hip MODE 2) and the library GEMM on the projection shapes of config 2 -- forward and data gradient, time and error
against float64.  TFLOP/s are real flop of the product (2 M N K) per second."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [  # (M, K = in features, N = out features, where)
    (163840, 96, 192, "s0 in|act, fc1"), (163840, 48, 144, "s0 q|kv"), (163840, 96, 96, "s0 out_proj"),
    (163840, 192, 96, "s0 fc2"), (40960, 192, 384, "s1 in|act, fc1"), (40960, 96, 288, "s1 q|kv"), (40960, 192, 192, "s1 out_proj"),
    (40960, 384, 192, "s1 fc2"), (10240, 384, 768, "s2 in|act, fc1"), (10240, 192, 576, "s2 q|kv"), (10240, 384, 384, "s2 out_proj"),
    (10240, 768, 384, "s2 fc2"), (2560, 768, 1536, "s3 in|act, fc1"), (2560, 384, 1152, "s3 q|kv"), (2560, 768, 768, "s3 out_proj"),
    (2560, 1536, 768, "s3 fc2"), (640, 192, 384, "s2 pooled kv"), (640, 384, 768, "s3 pooled kv"),
    (217600, 48, 96, "msmm in_proj"), (217600, 96, 144, "msmm x_proj"), (217600, 96, 48, "msmm out_proj"),
    (163840, 48, 256, "msmm glu fc1"), (163840, 128, 48, "msmm glu fc2"),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


def main():
    if os.environ.get("BENCH_TUNED_GEMM", "0") == "1":         # the step's library side: the committed TunableOp table
        from mlagg_unet_amd import gemm_tuning
        gemm_tuning.use_tuned_gemms(enabled=True)
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()                                                                      # noqa: E731
    print(f"{'shape':38s} | fwd: {'new us':>7s} {'TF/s':>6s} {'r3 x3':>7s} {'blas':>7s} | dgrad: {'new us':>7s} {'TF/s':>6s} {'r3 x3':>7s} {'blas':>7s} | "
          "rel err vs float64: fwd new, r3, blas; dgrad new | image us")
    tot = [0.0] * 6
    for M, K, N, where in SHAPES:
        x = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV) * K ** -0.5
        b = torch.randn(N, device=DEV)
        dy = torch.randn(M, N, device=DEV)
        img = torch.empty(lib.mlagg_weight_image_bytes(N, K), dtype=torch.uint8, device=DEV)
        imgT = torch.empty(lib.mlagg_weight_image_bytes(K, N), dtype=torch.uint8, device=DEV)
        ti = timeit(lambda: _lib.check(lib.mlagg_weight_image(p(w), K, p(img), p(imgT), N, K, st), "image"))
        y, y3 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        dx, dx3 = torch.empty(M, K, device=DEV), torch.empty(M, K, device=DEV)
        fl = 2.0 * M * K * N
        f = timeit(lambda: _lib.check(lib.mlagg_linear_x3(p(x), K, p(img), p(b), p(y), N, None, None, 0, M, N, K, 0, st), "x3 fwd"))
        f3 = timeit(lambda: _lib.check(lib.mlagg_linear_lp_fwd(p(x), K, p(w), p(b), p(y3), N, M, N, K, 3, st), "r3 fwd"))
        fb = timeit(lambda: torch.addmm(b, x, w.t()))
        d = timeit(lambda: _lib.check(lib.mlagg_linear_x3(p(dy), N, p(imgT), None, p(dx), K, None, None, 0, M, K, N, 0, st), "x3 dgrad"))
        wt = w.t().contiguous()
        d3 = timeit(lambda: _lib.check(lib.mlagg_linear_lp_fwd(p(dy), N, p(wt), None, p(dx3), K, M, K, N, 3, st), "r3 dgrad"))
        db_ = timeit(lambda: torch.mm(dy, w))
        rows = slice(0, min(M, 4096))
        ref = torch.addmm(b.double(), x[rows].double(), w.double().t())
        e = [float((t[rows].double() - ref).abs().max() / ref.abs().max()) for t in (y, y3, torch.addmm(b, x, w.t()))]
        dref = dy[rows].double() @ w.double()
        ed = float((dx[rows].double() - dref).abs().max() / dref.abs().max())
        for i, v in enumerate((f, f3, fb, d, d3, db_)):
            tot[i] += v
        print(f"{str((M, K, N)) + ' ' + where:38s} | {f:12.1f} {fl / f / 1e6:6.1f} {f3:7.1f} {fb:7.1f} | {d:14.1f} {fl / d / 1e6:6.1f} {d3:7.1f} {db_:7.1f} | "
              f"{e[0]:.1e} {e[1]:.1e} {e[2]:.1e}; {ed:.1e} | {ti:.1f}", flush=True)
    print("totals us: fwd new %.0f r3 %.0f blas %.0f | dgrad new %.0f r3 %.0f blas %.0f" % tuple(tot))
    # epilogues: GELU pair and GELU' against torch
    M, K, N = 10240, 384, 768
    x, w, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) * K ** -0.5, torch.randn(N, device=DEV)
    img = torch.empty(lib.mlagg_weight_image_bytes(N, K), dtype=torch.uint8, device=DEV)
    _lib.check(lib.mlagg_weight_image(p(w), K, p(img), None, N, K, st), "image")
    pre, act = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    tg = timeit(lambda: _lib.check(lib.mlagg_linear_x3(p(x), K, p(img), p(b), p(pre), N, p(act), None, 0, M, N, K, 1, st), "gelu"))
    ref = torch.addmm(b, x, w.t())
    print(f"GELU epilogue {tg:.1f} us: pre err {float((pre - ref).abs().max()):.2e}, act err {float((act - torch.nn.functional.gelu(ref)).abs().max()):.2e}")
    dy = torch.randn(M, K, device=DEV)           # gradient arriving at fc2's input ... use W^T of a (K_out = K) layer: dgelu of (M, N)
    w2 = torch.randn(K, N, device=DEV) * N ** -0.5          # fc2: N -> K
    imgT = torch.empty(lib.mlagg_weight_image_bytes(N, K), dtype=torch.uint8, device=DEV)
    _lib.check(lib.mlagg_weight_image(p(w2), N, None, p(imgT), K, N, st), "imageT")
    dpre = torch.empty(M, N, device=DEV)
    td = timeit(lambda: _lib.check(lib.mlagg_linear_x3(p(dy), K, p(imgT), None, p(dpre), N, None, p(pre), N, M, N, K, 2, st), "dgelu"))
    pr = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(pr).backward(dy @ w2)
    print(f"GELU' epilogue {td:.1f} us: err {float((dpre - pr.grad).abs().max()):.2e} (max |ref| {float(pr.grad.abs().max()):.2f})")


if __name__ == "__main__":
    main()
