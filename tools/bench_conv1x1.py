#!/usr/bin/env python
"""K18 against MIOpen on the 1 x 1 convolution shapes of the 256 x 256 step (batch 10): forward, data gradient (= forward on the
transposed weight) and weight gradient, timed through the C ABI (no autograd overhead)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import _lib, miopen_tuning  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [(96, 192, 128), (192, 96, 128), (192, 384, 64), (384, 192, 64), (384, 768, 32), (768, 384, 32), (96, 48, 256), (192, 192, 64),
          (384, 384, 32), (768, 768, 16)]          # (I, O, H = W)


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
    miopen_tuning.use_tuned_convolutions(enabled=True)
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    B = 10
    print(f"{'(I, O, H)':20s} {'fwd K18':>8s} {'TF/s':>6s} {'MIOpen':>8s} | {'dgrad':>8s} {'MIOpen':>8s} | {'wgrad':>8s} {'TF/s':>6s} {'MIOpen':>8s}")
    tot = [0.0] * 6
    for I, O, H in SHAPES:
        P = H * H
        x = torch.randn(B, I, H, H, device=DEV)
        w = torch.randn(O, I, 1, 1, device=DEV) * I ** -0.5
        wt = w.view(O, I).t().contiguous()
        gy = torch.randn(B, O, H, H, device=DEV)
        y, dx, dW = torch.empty_like(gy), torch.empty_like(x), torch.empty(O, I, device=DEV)
        ws = torch.empty(lib.mlagg_conv1x1_wgrad_workspace_floats(B, O, I, P), device=DEV)
        fl = 2.0 * B * P * I * O
        f = timeit(lambda: _lib.check(lib.mlagg_conv1x1_fwd(x.data_ptr(), I * P, w.data_ptr(), None, y.data_ptr(), O * P, B, O, I, P, st), "f"))
        assert float((y - F.conv2d(x, w)).abs().max()) < 1e-3
        fm = timeit(lambda: F.conv2d(x, w))
        d = timeit(lambda: _lib.check(lib.mlagg_conv1x1_fwd(gy.data_ptr(), O * P, wt.data_ptr(), None, dx.data_ptr(), I * P, B, I, O, P, st), "d"))
        dm = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False)))
        g = timeit(lambda: _lib.check(lib.mlagg_conv1x1_wgrad(gy.data_ptr(), O * P, x.data_ptr(), I * P, dW.data_ptr(), ws.data_ptr(), B, O, I, P, st), "g"))
        gm = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False)))
        for i, v in enumerate((f, fm, d, dm, g, gm)):
            tot[i] += v
        print(f"{str((I, O, H)):20s} {f:8.1f} {fl / f / 1e6:6.1f} {fm:8.1f} | {d:8.1f} {dm:8.1f} | {g:8.1f} {fl / g / 1e6:6.1f} {gm:8.1f}", flush=True)
    print("totals us: fwd %.0f (MIOpen %.0f)  dgrad %.0f (MIOpen %.0f)  wgrad %.0f (MIOpen %.0f)" % tuple(tot))


if __name__ == "__main__":
    main()
