#!/usr/bin/env python
"""A few launches of one K19 product on one shape (for rocprofv3 --pmc):  python tools/run_one_conv_wgrad.py I O H [form] [iters] [batch] [wgrad|fwd]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import _lib  # noqa: E402

I, O, H = (int(v) for v in sys.argv[1:4])
form = int(sys.argv[4]) if len(sys.argv) > 4 else 3
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
B = int(sys.argv[6]) if len(sys.argv) > 6 else 10
kind = sys.argv[7] if len(sys.argv) > 7 else "wgrad"
dev = torch.device("cuda:0")
lib = _lib.lib()
P = H * H
x, gy = torch.randn(B, I, H, H, device=dev), torch.randn(B, O, H, H, device=dev)
dW = torch.empty(O, I, 3, 3, device=dev)
ws = torch.empty(lib.mlagg_conv3x3_wgrad_workspace_floats(B, O, I, H, H), device=dev)
st = torch.cuda.current_stream().cuda_stream
w = torch.randn(O, I, 3, 3, device=dev) * (9 * I) ** -0.5
wimg = torch.empty(lib.mlagg_conv3x3_workspace_bytes(O, I), device=dev, dtype=torch.uint8)
for _ in range(iters):
    if kind == "fwd":
        _lib.check(lib.mlagg_conv3x3_fwd_lp(x.data_ptr(), I * P, w.data_ptr(), 0, None, gy.data_ptr(), O * P, wimg.data_ptr(), B, O, I, H, H, form, st), "fwd")
        continue
    _lib.check(lib.mlagg_conv3x3_wgrad_lp(gy.data_ptr(), O * P, x.data_ptr(), I * P, dW.data_ptr(), ws.data_ptr(), B, O, I, H, H, form, st), "wgrad")
torch.cuda.synchronize()
print("ok", float(dW.abs().sum()))
