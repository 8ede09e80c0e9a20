#!/usr/bin/env python
"""K19 against MIOpen on the 3 x 3 convolution shapes of the 256 x 256 step (batch 10): forward and data gradient (= the forward
kernel on the transposed, tap-flipped weight), timed through the C ABI."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import _lib, miopen_tuning  # noqa: E402

DEV = torch.device("cuda:0")
FORM = int(os.environ.get("BENCH_FORM", "3"))          # operand form of the K19 products: 3 = three bf16 pieces (fp32), 1 = bf16, 2 = fp16
TOL = 1e-3 if FORM == 3 else 5e-2
SHAPES = [(48, 48, 256), (96, 48, 256), (48, 48, 128), (48, 96, 128), (96, 96, 128), (144, 144, 64), (336, 336, 32), (720, 720, 16)]


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
    print(f"{'(I, O, H)':20s} {'fwd K19':>8s} {'TF/s':>6s} {'MIOpen':>8s} | {'dgrad':>8s} {'TF/s':>6s} {'MIOpen':>8s} | {'wgrad':>8s} {'TF/s':>6s} {'MIOpen':>8s}")
    tot = [0.0] * 4
    for I, O, H in SHAPES:
        P = H * H
        x = torch.randn(B, I, H, H, device=DEV)
        w = torch.randn(O, I, 3, 3, device=DEV) * (9 * I) ** -0.5
        gy = torch.randn(B, O, H, H, device=DEV)
        y, dx = torch.empty_like(gy), torch.empty_like(x)
        ws = torch.empty(lib.mlagg_conv3x3_workspace_bytes(max(O, I), max(O, I)), device=DEV, dtype=torch.uint8)
        fl = 2.0 * B * P * I * O * 9
        f = timeit(lambda: _lib.check(lib.mlagg_conv3x3_fwd_lp(x.data_ptr(), I * P, w.data_ptr(), 0, None, y.data_ptr(), O * P, ws.data_ptr(), B, O, I, H, H, FORM, st), "f"))
        ref = F.conv2d(x, w, None, 1, 1)
        assert float((y - ref).abs().max()) < TOL, float((y - ref).abs().max())
        fm = timeit(lambda: F.conv2d(x, w, None, 1, 1))
        d = timeit(lambda: _lib.check(lib.mlagg_conv3x3_fwd_lp(gy.data_ptr(), O * P, w.data_ptr(), 1, None, dx.data_ptr(), I * P, ws.data_ptr(), B, I, O, H, H, FORM, st), "d"))
        dref = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False))[0]
        assert float((dx - dref).abs().max()) < TOL
        dm = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
        gm = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
        dW = torch.empty(O, I, 3, 3, device=DEV)
        wws = torch.empty(lib.mlagg_conv3x3_wgrad_workspace_floats(B, O, I, H, H), device=DEV)
        gk = timeit(lambda: _lib.check(lib.mlagg_conv3x3_wgrad_lp(gy.data_ptr(), O * P, x.data_ptr(), I * P, dW.data_ptr(), wws.data_ptr(), B, O, I, H, H, FORM, st), "g"))
        wref = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
        assert float((dW - wref).abs().max()) < 2 * TOL * float(wref.abs().max()), float((dW - wref).abs().max())
        for i, v in enumerate((f, fm, d, dm)):
            tot[i] += v
        print(f"{str((I, O, H)):20s} {f:8.1f} {fl / f / 1e6:6.1f} {fm:8.1f} | {d:8.1f} {fl / d / 1e6:6.1f} {dm:8.1f} | {gk:8.1f} {fl / gk / 1e6:6.1f} {gm:8.1f}", flush=True)
    print("totals us: fwd %.0f (MIOpen %.0f)  dgrad %.0f (MIOpen %.0f)" % tuple(tot))


if __name__ == "__main__":
    main()
