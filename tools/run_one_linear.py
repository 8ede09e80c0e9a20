#!/usr/bin/env python
"""One projection shape through one kernel, for rocprofv3 --pmc:  python tools/run_one_linear.py M K N new|r3|blas [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import _lib  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4])
which = sys.argv[4]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
img = torch.empty(lib.mlagg_weight_image_bytes(N, K), dtype=torch.uint8, device=dev)
_lib.check(lib.mlagg_weight_image(x.data_ptr() * 0 + w.data_ptr(), K, img.data_ptr(), None, N, K, st), "image")
for _ in range(iters):
    if which == "new":
        _lib.check(lib.mlagg_linear_x3(x.data_ptr(), K, img.data_ptr(), b.data_ptr(), y.data_ptr(), N, None, None, 0, M, N, K, 0, st), "new")
    elif which == "r3":
        _lib.check(lib.mlagg_linear_lp_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, N, K, 3, st), "r3")
    else:
        torch.addmm(b, x, w.t(), out=y)
torch.cuda.synchronize()
print("done", float(y[0, 0]))
