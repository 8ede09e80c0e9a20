#!/usr/bin/env python
"""K5 / K5w microbenchmark at the projection shapes of config 2: forward, data gradient and weight gradient through the
C ABI, with the rocBLAS (torch.mm) time of the same product beside it.  TFLOP/s against the fp32 MFMA peak (157.3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [  # (M, K = in features, N = out features, where)
    (163840, 96, 192, "s0 in|act, fc1"), (163840, 48, 144, "s0 q|kv"), (163840, 96, 96, "s0 out_proj"),
    (163840, 192, 96, "s0 fc2"), (40960, 192, 384, "s1 in|act, fc1"), (40960, 96, 288, "s1 q|kv"),
    (40960, 384, 192, "s1 fc2"), (10240, 384, 768, "s2 fc1"), (10240, 768, 384, "s2 fc2"), (2560, 768, 1536, "s3 fc1"),
    (2560, 1536, 768, "s3 fc2"), (217600, 48, 96, "msmm in_proj"), (217600, 96, 140, "msmm x_proj"),
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
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    print(f"{'shape':34s} {'fwd us':>8s} {'TF/s':>6s} {'blas':>7s} {'x3':>7s} {'TF/s':>6s} | {'dgrad':>7s} {'TF/s':>6s} {'blas':>7s} {'x3':>7s} | {'wgrad':>7s} {'TF/s':>6s} {'blas':>7s} {'x3':>7s} {'TF/s':>6s}"
          "   max |err| vs float64 / max: fwd K5, x3, blas; wgrad K5w, x3")
    tot = [0.0] * 9
    only = os.environ.get("MLAGG_BENCH_ONLY")
    for M, K, N, where in SHAPES:
        if only and only not in where:
            continue
        x = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV)
        b = torch.randn(N, device=DEV)
        y = torch.empty(M, N, device=DEV)
        dy = torch.randn(M, N, device=DEV)
        dx = torch.empty(M, K, device=DEV)
        dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        ws = torch.empty(lib.mlagg_linear_wgrad_workspace_floats(M, N, K), device=DEV)
        fl = 2.0 * M * K * N
        f = timeit(lambda: _lib.check(lib.mlagg_linear_fwd(p(x), K, p(w), p(b), p(y), N, M, N, K, st), "fwd"))
        if not os.environ.get("MLAGG_K5_DEBUG"):
            assert float((y - torch.addmm(b, x, w.t())).abs().max()) < 1e-2 * K ** 0.5
        fb = timeit(lambda: torch.addmm(b, x, w.t()))
        # the same product with the fp32 operands as three bf16 pieces each (MLAGG_DTYPE_BF16X3 = 3)
        y3 = torch.empty_like(y)
        f3 = timeit(lambda: _lib.check(lib.mlagg_linear_lp_fwd(p(x), K, p(w), p(b), p(y3), N, M, N, K, 3, st), "fwd x3"))
        rows = slice(0, 4096)
        ref = torch.addmm(b.double(), x[rows].double(), w.double().t())
        errs = [float((t[rows].double() - ref).abs().max() / ref.abs().max()) for t in (y, y3, torch.addmm(b, x, w.t()))]
        d = timeit(lambda: _lib.check(lib.mlagg_linear_dgrad(p(dy), N, p(w), p(dx), K, M, N, K, st), "dgrad"))
        if not os.environ.get("MLAGG_K5_DEBUG"):
            assert float((dx - dy @ w).abs().max()) < 1e-2 * N ** 0.5
        dbl = timeit(lambda: torch.mm(dy, w))
        dx3 = torch.empty_like(dx)
        d3 = timeit(lambda: _lib.check(lib.mlagg_linear_lp_dgrad(p(dy), N, p(w), p(dx3), K, M, N, K, 3, st), "dgrad x3"))
        assert float((dx3 - dx).abs().max()) < 1e-4 * N ** 0.5
        g = timeit(lambda: _lib.check(lib.mlagg_linear_wgrad(p(dy), N, p(x), K, p(dW), p(db), p(ws), M, N, K, st), "wgrad"))
        gb = timeit(lambda: (torch.mm(dy.t(), x), dy.sum(0)))
        dW3, db3 = torch.empty_like(dW), torch.empty_like(db)
        g3 = timeit(lambda: _lib.check(lib.mlagg_linear_wgrad_x3(p(dy), N, p(x), K, p(dW3), p(db3), p(ws), M, N, K, st), "wgrad x3"))
        wref = dy[:32768].double().t() @ x[:32768].double()
        lib.mlagg_linear_wgrad(p(dy), N, p(x), K, p(dW), p(db), p(ws), min(M, 32768), N, K, st)
        e1 = float((dW.double() - wref).abs().max() / wref.abs().max())
        lib.mlagg_linear_wgrad_x3(p(dy), N, p(x), K, p(dW3), p(db3), p(ws), min(M, 32768), N, K, st)
        e3 = float((dW3.double() - wref).abs().max() / wref.abs().max())
        assert float((db3 - db).abs().max()) < 1e-3 * float(db.abs().max() + 1)
        for i, v in enumerate((f, fb, d, dbl, g, gb, f3, d3, g3)):
            tot[i] += v
        print(f"{str((M, K, N)) + ' ' + where:34s} {f:8.1f} {fl / f / 1e6:6.1f} {fb:7.1f} {f3:7.1f} {fl / f3 / 1e6:6.1f} | {d:7.1f} {fl / d / 1e6:6.1f} {dbl:7.1f} {d3:7.1f} | "
              f"{g:7.1f} {fl / g / 1e6:6.1f} {gb:7.1f} {g3:7.1f} {fl / g3 / 1e6:6.1f}   {errs[0]:.1e} {errs[1]:.1e} {errs[2]:.1e}; {e1:.1e} {e3:.1e}",
              flush=True)
    print("totals us: fwd %.0f (blas %.0f)  dgrad %.0f (blas %.0f)  wgrad %.0f (blas %.0f)  x3: fwd %.0f dgrad %.0f wgrad %.0f" % tuple(tot))


if __name__ == "__main__":
    main()
