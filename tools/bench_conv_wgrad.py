#!/usr/bin/env python
"""Weight-gradient time of the path's dense convolutions: MIOpen (torch.ops.aten.convolution_backward, weight only) vs K15
(ops.conv_weight_grad: pad copies + tap GEMM + reduce).  python tools/bench_conv_wgrad.py [2d|3d]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: F401,E402
from mlagg_unet_amd import ops  # noqa: E402

SHAPES_2D = [  # (I, O, H, W, k) at batch 10: the 256 x 256 train step
    (48, 48, 128, 128, 3), (48, 96, 128, 128, 3), (96, 96, 128, 128, 3),
    (48, 48, 128, 128, 3), (144, 144, 64, 64, 3), (336, 336, 32, 32, 3), (720, 720, 16, 16, 3),
    (1, 48, 256, 256, 3), (48, 48, 256, 256, 3), (96, 48, 256, 256, 3), (96, 48, 256, 256, 1), (1, 48, 256, 256, 1),
    (96, 192, 128, 128, 1), (192, 96, 128, 128, 1), (192, 384, 64, 64, 1), (384, 192, 64, 64, 1), (384, 768, 32, 32, 1),
    (768, 384, 32, 32, 1), (192, 192, 64, 64, 1), (384, 384, 32, 32, 1), (768, 768, 16, 16, 1),
]
SHAPES_3D = [  # (I, O, D, H, W, k, stride) at batch 2: the 96 x 160 x 160 train step
    (32, 32, 96, 160, 160, 3, 1), (64, 32, 96, 160, 160, 3, 1), (32, 64, 96, 160, 160, 3, 2), (64, 64, 48, 80, 80, 3, 1),
    (128, 128, 24, 40, 40, 3, 1), (256, 256, 12, 20, 20, 3, 1), (320, 320, 6, 10, 10, 3, 1), (64, 32, 96, 160, 160, 1, 1),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "2d"
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    for sh in (SHAPES_2D if which == "2d" else SHAPES_3D):
        if which == "2d":
            I, O, H, W, k = sh
            B, dims, stride = 10, (H, W), 1
        else:
            I, O, D, H, W, k, stride = sh
            B, dims = 2, (D, H, W)
        nd = len(dims)
        x = torch.randn(B, I, *dims, device=dev)
        w = torch.randn(O, I, *([k] * nd), device=dev)
        y = (torch.nn.functional.conv3d if nd == 3 else torch.nn.functional.conv2d)(x, w, None, stride, k // 2)
        dy = torch.randn_like(y)
        lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (stride,) * nd, (k // 2,) * nd, (1,) * nd, False, (0,) * nd, 1,
                                                          (False, True, False))
        mine = lambda: ops.conv_weight_grad(x, dy, k, stride)
        t_lib, t_mine = timeit(lib), timeit(mine)
        flop = 2.0 * B * y[0, 0].numel() * I * O * k ** nd
        print(f"{str(sh):40s} MIOpen {t_lib:8.3f} ms   K15 {t_mine:8.3f} ms  ({flop / t_mine / 1e9:6.1f} TF/s)   ratio {t_lib / t_mine:5.2f}", flush=True)


if __name__ == "__main__":
    main()
